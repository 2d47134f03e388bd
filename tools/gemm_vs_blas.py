"""Per-shape timing of the encoder GEMMs: this library's kernel (with its fused epilogue) vs torch.matmul (hipBLASLt, fp16
out, no epilogue) on the same operands. Calibrates how far the fused kernels are from a tuned library GEMM."""
import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from crossscore_amd import _lib
import hip_helpers as hh
dev = "cuda"
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
for Mimgs in [int(v) for v in os.environ.get("CS_GVB_IMGS", "48,12,96,8,22").split(",")]:
    M = Mimgs * 1370
    shapes = {"qkv": (M, 1152, 384, _lib.EPI_BIAS_F16), "outproj": (M, 384, 384, _lib.EPI_RESID_F32),
              "fc1": (M, 1536, 384, _lib.EPI_BIAS_GELU_F16), "fc2": (M, 384, 1536, _lib.EPI_RESID_F32)}
    if Mimgs in (96, 8, 22, 44, 88):  # ViT-B (cfg-4: 96 images per step, 8 per encoder chunk)
        shapes = {"qkvB": (M, 2304, 768, _lib.EPI_BIAS_F16), "outprojB": (M, 768, 768, _lib.EPI_RESID_F32),
                  "fc1B": (M, 3072, 768, _lib.EPI_BIAS_GELU_F16), "fc2B": (M, 768, 3072, _lib.EPI_RESID_F32)}
    for sn, (M, N, K, epi) in shapes.items():
        A = torch.randn(M, K, device=dev).half(); W = (torch.randn(N, K, device=dev) / K ** 0.5).half()
        b = torch.randn(N, device=dev)
        resid = torch.randn(M, N, device=dev) if epi == _lib.EPI_RESID_F32 else None
        o = torch.empty(M, N, device=dev, dtype=torch.float32 if epi == _lib.EPI_RESID_F32 else torch.float16)
        lib = _lib.load()
        lib.cs_debug_gemm256_enable(0)
        t_old = timeit(lambda: hh.gemm(A, W, b, epi, resid=resid, out=o))
        lib.cs_debug_gemm256_enable(1)
        t_us = timeit(lambda: hh.gemm(A, W, b, epi, resid=resid, out=o))
        t_plain = timeit(lambda: hh.gemm(A, W, b, _lib.EPI_BIAS_F16, out=o if o.dtype == torch.float16 else None)) if False else 0
        Wt = W.t().contiguous(); ob = torch.empty(M, N, device=dev, dtype=torch.float16)
        A = A.contiguous()
        t_blas = timeit(lambda: torch.matmul(A, W.t(), out=ob))
        t_blas2 = timeit(lambda: torch.matmul(A, Wt, out=ob))
        fl = 2.0 * M * N * K
        print(f"M={M} {sn:8s} N={N} K={K}: ours {t_us:7.1f} us ({fl/t_us/1e6:6.0f} TF/s) [128-row kernel {t_old:7.1f} us] | hipBLASLt NT {t_blas:7.1f} us ({fl/t_blas/1e6:6.0f}) NN {t_blas2:7.1f} us", flush=True)
