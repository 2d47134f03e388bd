"""Timing-only ablation of the GEMM kernel (run on the GPU box): which phase sets the time of one launch?
Builds a debug copy of the library with -DCS_ABLATE into /tmp and times the encoder shapes with phases removed."""
import ctypes as C, os, subprocess, sys, time
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
src = os.path.join(REPO, "crossscore_amd", "csrc")
out = "/tmp/libcs_ablate.so"
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DCS_ABLATE", "-Wno-unused-value", "-I" + os.path.join(REPO, "include"),
                       "-o", out] + [os.path.join(src, f) for f in ("api.hip", "gemm.hip", "attention.hip", "elementwise.hip", "preprocess.hip", "panel.hip")])
from crossscore_amd import _lib
_lib.LIB_PATH = out

sys.path.insert(0, os.path.join(REPO, "tests"))
import hip_helpers as hh
dev = "cuda"
MM = int(os.environ.get("CS_ABL_M", "65760"))
shapes = {"qkv": (MM, 1152, 384, _lib.EPI_BIAS_F16), "outproj": (MM, 384, 384, _lib.EPI_RESID_F32),
          "fc1": (MM, 1536, 384, _lib.EPI_BIAS_GELU_F16), "fc2": (MM, 384, 1536, _lib.EPI_RESID_F32),
          "qkvB": (MM, 2304, 768, _lib.EPI_BIAS_F16), "fc1B": (MM, 3072, 768, _lib.EPI_BIAS_GELU_F16), "fc2B": (MM, 768, 3072, _lib.EPI_RESID_F32)}
names = {0: "full", 1: "no-epilogue", 2: "no-mfma", 4: "no-dma", 3: "dma-only", 6: "epilogue-only", 5: "mfma-only", 7: "empty"}
only = os.environ.get("CS_ABL_SHAPES")
if only: shapes = {k: v for k, v in shapes.items() if k in only.split(",")}
modes = [int(x) for x in os.environ.get("CS_ABL_MODES", "0,1,2,4,3,5,6,7").split(",")]
# the residual-prefetch variant (RESID_F32 with K >= 9 slices) has no ablation hooks for its prefetched rows: timing it with phases removed
# faulted on the GPU once -- only the full kernel (mode 0) is run for those shapes
for sn, (M, N, K, epi) in shapes.items():
    A = torch.randn(M, K, device=dev).half(); W = (torch.randn(N, K, device=dev) / K ** 0.5).half()
    b = torch.randn(N, device=dev)
    resid = torch.randn(M, N, device=dev) if epi == _lib.EPI_RESID_F32 else None
    o = torch.empty(M, N, device=dev, dtype=torch.float32 if epi == _lib.EPI_RESID_F32 else torch.float16)
    line = [f"{sn:8s} M={M} N={N} K={K}:"]
    for tall in [int(x) for x in os.environ.get("CS_ABL_NSUB", "3").split(",")]:
        os.environ["CS_GEMM_NSUB"] = str(tall)
        for ab, nm in [(m, names.get(m & 7, "?") + ("+line128" if m & 8 else "")) for m in modes if m == 0 or epi != _lib.EPI_RESID_F32]:
            os.environ["CS_GEMM_ABLATE"] = str(ab)
            for _ in range(3):
                hh.gemm(A, W, b, epi, resid=resid, out=o)
            torch.cuda.synchronize(); e0 = torch.cuda.Event(True); e1 = torch.cuda.Event(True); e0.record()
            for _ in range(10):
                hh.gemm(A, W, b, epi, resid=resid, out=o)
            e1.record(); torch.cuda.synchronize()
            line.append(f"t{tall}/{nm}={e0.elapsed_time(e1) * 100:.1f}")
    print(" ".join(line), flush=True)
