"""Per-wave phase clocks of the GEMM kernel (debug build, -DCS_ABLATE): where does a wave's time go inside a K slice?"""
import ctypes as C, os, subprocess, sys
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
lib = _lib.load()
lib.cs_gemm_dbg_set.argtypes = [C.c_void_p]; lib.cs_gemm_dbg_set.restype = C.c_int
dev = "cuda"
MM = int(os.environ.get("CS_ABL_M", "65536"))
shapes = {"qkv": (MM, 1152, 384, _lib.EPI_BIAS_F16), "outproj": (MM, 384, 384, _lib.EPI_RESID_F32),
          "fc1": (MM, 1536, 384, _lib.EPI_BIAS_GELU_F16), "fc2": (MM, 384, 1536, _lib.EPI_RESID_F32),
          "qkvB": (MM, 2304, 768, _lib.EPI_BIAS_F16), "fc1B": (MM, 3072, 768, _lib.EPI_BIAS_GELU_F16), "fc2B": (MM, 768, 3072, _lib.EPI_RESID_F32)}
only = os.environ.get("CS_ABL_SHAPES")
if only: shapes = {k: v for k, v in shapes.items() if k in only.split(",")}
dbg = torch.zeros(512 * 4 * 8, dtype=torch.int64, device=dev)
assert lib.cs_gemm_dbg_set(C.c_void_p(dbg.data_ptr())) == 0
names = ["vmcnt wait", "barrier", "prefetch+DMA issue", "LDS reads+MFMA", "epilogue step", "between slices", "slices", "total"]
for sn, (M, N, K, epi) in shapes.items():
    A = torch.randn(M, K, device=dev).half(); W = (torch.randn(N, K, device=dev) / K ** 0.5).half()
    b = torch.randn(N, device=dev)
    resid = torch.randn(M, N, device=dev) if epi == _lib.EPI_RESID_F32 else None
    o = torch.empty(M, N, device=dev, dtype=torch.float32 if epi == _lib.EPI_RESID_F32 else torch.float16)
    for ab in [int(x) for x in os.environ.get("CS_ABL_MODES", "0").split(",")]:
        os.environ["CS_GEMM_ABLATE"] = str(ab)
        for _ in range(3):
            dbg.zero_(); hh.gemm(A, W, b, epi, resid=resid, out=o)
        torch.cuda.synchronize()
        d = dbg.view(512, 4, 8).double().cpu()
        act = d[:, :, 6] > 0
        per = d[act]                      # (waves, 8)
        sl = per[:, 6].mean().item()
        tot = per[:, 7].mean().item()
        line = f"{sn:8s} mode {ab}: slices/wave {sl:.0f}, total {tot:.0f} clk ({tot / sl:.0f}/slice) |"
        for k in range(6):
            line += f" {names[k]} {per[:, k].mean().item() / sl:.0f}"
        print(line, flush=True)
