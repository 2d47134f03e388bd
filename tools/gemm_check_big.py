"""Correctness of the residual GEMM at full encoder sizes against torch (fp32 accumulate of the same fp16 operands)."""
import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from crossscore_amd import _lib
import hip_helpers as hh
dev = "cuda"
torch.manual_seed(0)
for M in (65536, 65760, 16440, 1370 * 6, 128 * 7 + 5):
    for (N, K) in ((384, 384), (384, 1536), (128, 320)):
        A = torch.randn(M, K, device=dev).bfloat16(); W = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
        b = torch.randn(N, device=dev); resid = torch.randn(M, N, device=dev)
        o = torch.empty(M, N, device=dev)
        hh.gemm(A, W, b, _lib.EPI_RESID_F32, resid=resid, out=o)
        torch.cuda.synchronize()
        ref = resid + (A.float() @ W.float().t() + b)
        err = (o - ref).abs().max().item()
        o2 = torch.empty_like(o); hh.gemm(A, W, b, _lib.EPI_RESID_F32, resid=resid, out=o2); torch.cuda.synchronize()
        print(f"M={M} N={N} K={K}: max err {err:.3e} deterministic={bool((o == o2).all())}", flush=True)
        assert err < 2e-3
print("ok")
