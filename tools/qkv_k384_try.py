"""Would the 256 x 256 x 64-tile GEMM beat the 128-row kernel on the ViT-S QKV shape (K = 384)?  N must be a multiple of 256 for it: the
1152 QKV columns padded to 1280.  The large-tile kernel takes K >= 384 since round 4 (cs_debug_gemm256_kmin moves the gate)."""
import os, sys, math
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
import hip_helpers as hh
from crossscore_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(1)
def bench(M, N, K, on):
    A = torch.randn(M, K, generator=g).to(dev).half(); W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev).half(); b = torch.randn(N, generator=g).to(dev)
    lib.cs_debug_gemm256_enable(1 if on else 0)
    out = torch.empty((M, N), dtype=torch.float16, device=dev)
    for _ in range(3): hh.gemm(A, W, b, _lib.EPI_BIAS_F16, out=out)
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): hh.gemm(A, W, b, _lib.EPI_BIAS_F16, out=out)
    e.record(); torch.cuda.synchronize()
    us = 1e3 * a.elapsed_time(e) / 20
    lib.cs_debug_gemm256_enable(1)
    ref = A[:4096].float() @ W.float().t() + b
    err = (out[:4096].float() - ref).abs()
    print(f"   (256-tile kernel {'on' if on else 'off'}: max |out - fp32 ref| over 4096 rows {float(err.max()):.2e}, max relative to 6e-4*|ref|+5e-5: {float((err / (6e-4 * ref.abs() + 5e-5)).max()):.2f})")
    return us, out
for M in (65760, 32880):
    for N in (1152, 1280):
        u128, o1 = bench(M, N, 384, False)
        line = f"M={M} N={N} K=384: 128-row kernel {u128:6.1f} us ({2.0 * M * N * 384 / u128 / 1e6:5.0f} TF/s)"
        if N % 256 == 0:
            u256, o2 = bench(M, N, 384, True)
            line += f" | 256-tile kernel {u256:6.1f} us ({2.0 * M * N * 384 / u256 / 1e6:5.0f} TF/s), results equal: {bool(torch.equal(o1, o2))}, differing elements {int((o1 != o2).sum())} of {o1.numel()}, max ulp distance {int((o1.view(torch.int16).int() - o2.view(torch.int16).int()).abs().max())}"
        print(line, flush=True)
