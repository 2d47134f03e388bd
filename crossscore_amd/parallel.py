"""Batch sharding of the CrossScore forward over the GPUs of one node.

Every batch item (query, N refs) is independent -- no op mixes the batch dimension (task/core.py:134-161,
per-sample attention, no BatchNorm) -- and the reference scales the same way (Lightning DDP + distributed
sampler, zero collectives in the forward: task/predict.py:119-135).  So the multi-GPU design is "replicas +
contiguous batch shard": one process per GPU, a full weight replica each, rank r owns items
[r*B/G, (r+1)*B/G).  RCCL (torch.distributed backend "nccl") carries only the per-image mean scores
(B floats, all_gather) and timing scalars -- latency-bound messages; there is no data-path collective.
"""
from __future__ import annotations

import os
from typing import List, Tuple

import torch
import torch.distributed as dist


def shard_bounds(global_batch: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous split; the first (global_batch % world_size) ranks take one extra item."""
    if global_batch < 0 or world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError("bad shard arguments")
    base, rem = divmod(global_batch, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


# Collectives are skipped when the group has one rank (nothing to exchange).  Tests set this to True to push the very same calls through
# the backend on a world-size-1 group (tests/test_rccl_single_rank.py: RCCL on the one GPU a test box has).
COLLECTIVES_AT_WORLD_1 = False


def _single() -> bool:
    return not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not COLLECTIVES_AT_WORLD_1)


def init_from_env(backend: str | None = None, single_rank_group: bool = False) -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment; initialises the process group when
    WORLD_SIZE > 1 (or, with single_rank_group, also for WORLD_SIZE = 1: the RCCL rehearsal on one GPU).
    backend defaults to nccl (= RCCL on ROCm) when a GPU is present, else gloo."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if (world > 1 or single_rank_group) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            # bind this process to its GPU before the first collective (RCCL picks the current device for barrier / all_reduce)
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend=backend, rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def gather_means(local_means: torch.Tensor, global_batch: int) -> torch.Tensor:
    """All ranks receive the (global_batch,) per-image mean scores in item order.  Shards may be ragged, so each
    rank pads to the largest shard before the all_gather."""
    if _single():
        return local_means
    world = dist.get_world_size()
    sizes = [shard_bounds(global_batch, world, r) for r in range(world)]
    mx = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros(mx, dtype=local_means.dtype, device=local_means.device)
    pad[: local_means.numel()] = local_means
    bufs: List[torch.Tensor] = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    return torch.cat([b[: hi - lo] for b, (lo, hi) in zip(bufs, sizes)])


def max_over_ranks(value: float, device: torch.device | str = "cpu") -> float:
    if _single():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def rank_census(device: torch.device | str = "cpu", **extra) -> List[dict]:
    """One record per rank, gathered on every rank in rank order: who took part in this run.  Each record carries the rank, its
    LOCAL_RANK, host, pid, the device it is bound to (index, name, PCI bus id and UUID for a GPU) and whatever `extra` the caller adds
    (bench.py: that rank's own ms per step).  With it "did the collective see N ranks on N distinct GPUs" is answerable from rank 0's
    JSON line alone (task/predict.py:119-135 scales the same way: one process per device)."""
    import socket

    rec = {"rank": int(os.environ.get("RANK", "0")), "local_rank": int(os.environ.get("LOCAL_RANK", "0")),
           "host": socket.gethostname(), "pid": os.getpid(), "device": str(device)}
    dev = torch.device(device)
    if dev.type == "cuda":
        pr = torch.cuda.get_device_properties(dev)
        rec.update(device_name=pr.name, pci_bus_id=f"{getattr(pr, 'pci_domain_id', 0):04x}:{getattr(pr, 'pci_bus_id', 0):02x}:{getattr(pr, 'pci_device_id', 0):02x}",
                   uuid=str(getattr(pr, "uuid", "")))
    rec.update(extra)
    if _single():
        return [rec]
    out: List[dict] = [None] * dist.get_world_size()  # type: ignore[list-item]
    dist.all_gather_object(out, rec)
    return out


def backend_info() -> dict:
    """Backend of the process group and the collective library's version (torch.cuda.nccl.version() is RCCL's on ROCm)."""
    info = {"backend": None, "world_size": 1, "collective_library": None}
    if dist.is_available() and dist.is_initialized():
        info["backend"] = str(dist.get_backend())
        info["world_size"] = dist.get_world_size()
    if info["backend"] == "nccl":
        try:
            info["collective_library"] = "RCCL " + ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception as e:  # noqa: BLE001
            info["collective_library"] = f"nccl backend (version query failed: {e})"
    elif info["backend"] == "gloo":
        info["collective_library"] = "gloo (CPU rehearsal)"
    return info


def barrier() -> None:
    if not _single():
        dist.barrier()


def shutdown() -> None:
    """Tears the process group down (quiet exit of multi-rank runs)."""
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()
