"""The RCCL branch of the rank plumbing under pytest: bench.py's own parallel.init_from_env("nccl") and every collective it
issues at N > 1, on a world-size-1 group bound to the test box's one GPU, in a child process started before that process touches the GPU.
It proves nothing about scaling; it proves the nccl (= RCCL on ROCm) code path runs, on device tensors, and reports itself as RCCL.
The reference scales the same way -- one process per device, no forward collective (task/predict.py:119-135)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def test_rccl_single_rank_group_runs_the_bench_collectives():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_PORT=str(port))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, os.path.join(HERE, "rccl_single_rank_child.py")], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-3000:]
    line = [ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["world"] == 1 and out["rank"] == 0
    assert out["max"] == 1.25
    assert out["means"] == [i / 8 for i in range(8)] and out["means_device"].startswith("cuda")
    assert len(out["census"]) == 1 and out["census"][0]["device"] == "cuda:0" and out["census"][0]["ms_per_step"] == 6.0
    assert out["info"]["backend"] == "nccl" and out["info"]["world_size"] == 1 and out["info"]["collective_library"].startswith("RCCL "), out["info"]
