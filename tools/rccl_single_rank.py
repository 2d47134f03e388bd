"""Sanity of the RCCL path bench.py takes at N > 1, on the one GPU available: a world-size-1 nccl group bound to cuda:0 with the same
calls (barrier, all_reduce MAX on a float64 CUDA scalar, all_gather of per-image means)."""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
dist.barrier()
t = torch.tensor([1.25], dtype=torch.float64, device="cuda"); dist.all_reduce(t, op=dist.ReduceOp.MAX)
bufs = [torch.empty(8, device="cuda")]; dist.all_gather(bufs, torch.arange(8, dtype=torch.float32, device="cuda"))
print("rccl ok:", float(t.item()), bufs[0].tolist())
dist.destroy_process_group()
