"""Time of the decoder's linear + residual + LayerNorm launch (csrc/rowln.hip) against the two launches it replaces, M = 10 952 rows (cfg-2)."""
import os, sys, math
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
import hip_helpers as hh
from crossscore_amd import _lib
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(1)
M, C = 10952, 384
A = torch.randn(M, C, generator=g).to(dev).half(); W = (torch.randn(C, C, generator=g) / math.sqrt(C)).to(dev).half()
b = torch.randn(C, generator=g).to(dev); res = torch.randn(M, C, generator=g).to(dev); gam = torch.ones(C, device=dev); bet = torch.zeros(C, device=dev)
def t(fn, n=50):
    for _ in range(5): fn()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(e) / n
of = torch.zeros((M, C), device=dev); oh = torch.zeros((M, C), dtype=torch.float16, device=dev)
lib = _lib.load()
def fused(r=res, o32=of, o16=oh):
    _lib.check(lib.cs_op_linear_layernorm(hh._p(A), hh._p(W), hh._p(b), hh._p(r), hh._p(gam), hh._p(bet), 1e-5, hh._p(o32), hh._p(o16), M, C, hh._stream()))
y = torch.zeros((M, C), device=dev)
def two():
    hh.gemm(A, W, b, _lib.EPI_RESID_F32, resid=res, out=y)
    _lib.check(lib.cs_op_layernorm(hh._p(y), M, C, hh._p(gam), hh._p(bet), 1e-5, hh._p(of), hh._p(oh), hh._stream()))
print(f"fused: {t(fused):.1f} us; without residual {t(lambda: fused(None)):.1f}; fp32 out only {t(lambda: fused(res, of, None)):.1f}; fp16 out only {t(lambda: fused(res, None, oh)):.1f}")
print(f"GEMM (fp32 residual epilogue) + LayerNorm: {t(two):.1f} us")
# second stage: the sub-block's next linear in the same launch (n2 = C: Q projection / linear1 / head; n2 = 3 C: the next layer's QKV)
for n2, act in ((C, 1), (3 * C, 0)):
    W2 = (torch.randn(n2, C, generator=g) / math.sqrt(C)).to(dev).half(); b2 = torch.randn(n2, generator=g).to(dev)
    o2 = torch.zeros((M, n2), dtype=torch.float16, device=dev)
    epi = _lib.EPI_BIAS_RELU_F16 if act == 1 else _lib.EPI_BIAS_F16
    def three():
        _lib.check(lib.cs_op_linear_layernorm_linear(hh._p(A), hh._p(W), hh._p(b), hh._p(res), hh._p(gam), hh._p(bet), 1e-5, hh._p(of), None,
                                                     hh._p(W2), hh._p(b2), n2, act, hh._p(o2), M, C, hh._stream()))
    def two_b():
        fused(res, of, oh)
        hh.gemm(oh, W2, b2, epi, out=o2)
    print(f"n2 = {n2}: linear + LayerNorm + linear in one launch {t(three):.1f} us; linear + LayerNorm, then the GEMM {t(two_b):.1f} us")
