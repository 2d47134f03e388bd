"""Child process of tests/test_rccl_single_rank.py (started before anything touches the GPU in THIS process): the process-group calls
bench.py makes at N > 1 -- parallel.init_from_env("nccl"), barrier, max_over_ranks, gather_means, rank_census, backend_info, shutdown -- on a
world-size-1 RCCL group bound to cuda:0, with device tensors.  Prints one JSON line."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29544")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

from crossscore_amd import parallel  # noqa: E402

parallel.COLLECTIVES_AT_WORLD_1 = True
rank, local_rank, world = parallel.init_from_env("nccl", single_rank_group=True)
dev = torch.device("cuda", local_rank)
parallel.barrier()
mx = parallel.max_over_ranks(1.25, dev)
means = parallel.gather_means(torch.arange(8, dtype=torch.float32, device=dev) / 8, 8)
census = parallel.rank_census(dev, ms_per_step=6.0)
info = parallel.backend_info()
parallel.barrier()
parallel.shutdown()
print(json.dumps({"rank": rank, "world": world, "max": mx, "means": means.cpu().tolist(), "means_device": str(means.device), "census": census, "info": info}), flush=True)
