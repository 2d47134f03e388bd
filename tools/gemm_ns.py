"""Experiment: does a deeper LDS-DMA ring (more bytes in flight per CU) raise the staging rate?  128x128 tiles (NSUB=2)."""
import ctypes as C, os, subprocess, sys, importlib
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
src = os.path.join(REPO, "crossscore_amd", "csrc")
from crossscore_amd import _lib
import hip_helpers as hh
M, N, K = 65760, 1280, 384
A = torch.randn(M, K, device="cuda").bfloat16(); W = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16(); b = torch.randn(N, device="cuda")
o = torch.empty(M, N, device="cuda", dtype=torch.float16)
for ns in (2, 3, 4, 5):
    out = f"/tmp/libcs_ns{ns}.so"
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DCS_ABLATE", f"-DCS_NS_OVERRIDE={ns}",
                           "-Wno-unused-value", "-o", out] + [os.path.join(src, f) for f in ("api.hip", "gemm.hip", "attention.hip", "elementwise.hip", "preprocess.hip")])
    _lib._lib = None; _lib.LIB_PATH = out
    line = [f"NS={ns} (128x128 tiles, {2*(ns-1)*16} KB in flight per CU):"]
    for ab, nm in ((0, "full"), (3, "dma-only"), (1, "no-epilogue")):
        os.environ["CS_GEMM_ABLATE"] = str(ab)
        for _ in range(3): hh.gemm(A, W, b, _lib.EPI_BIAS_F16, out=o)
        torch.cuda.synchronize(); e0 = torch.cuda.Event(True); e1 = torch.cuda.Event(True); e0.record()
        for _ in range(10): hh.gemm(A, W, b, _lib.EPI_BIAS_F16, out=o)
        e1.record(); torch.cuda.synchronize()
        line.append(f"{nm}={e0.elapsed_time(e1)*100:.1f}us")
    print(" ".join(line), flush=True)
