"""Packed-half GELU of the panel kernel (csrc/panel.hip, fp16 operand mode): fits the polynomial, emulates the kernel's instruction
sequence in half arithmetic (every v_pk_fma_f16 is one rounding of an exact product-sum) and prints the coefficients and the error figures
quoted in panel.hip / DESIGN.md.  `--budget` also runs the oracle's fp16-operand emulation of a cfg-2 item (ViT-S, 518x518, 5 references)
with this GELU in place of the exact one (CPU, ~2 minutes).

    GELU(x) = relu(x) - |x| Phi(-|x|);   d = clamp(1 - |x|/4, 0, 1);   z = d*d - 1/2;   -|x| Phi(-|x|) ~ P6(z)

Why not Phi(x) = 0.5 + x Q(x^2) as in the fp32 epilogues: in half precision its Horner terms reach 6 and cancel to 0.1 at |x| = 4 (1e-2 of
error; `--naive` prints it).  In z the correction term is a bump of height 0.17 with sum |c_k| 2^-k = 0.76: next to nothing cancels."""
import math, sys
import numpy as np
from scipy.special import erf

f16 = np.float16
R = 4.0


def r16(x):
    return np.asarray(x, dtype=np.float64).astype(f16).astype(np.float64)


def gelu(x):
    return 0.5 * x * (1 + erf(x / math.sqrt(2)))


def psi(a):
    return 0.5 * (1 - erf(a / math.sqrt(2)))


def minimax(A, target, iters=60):
    w = np.ones(len(target))
    for _ in range(iters):  # Lawson's iteration
        c, *_ = np.linalg.lstsq(A * w[:, None], target * w, rcond=None)
        e = np.abs(A @ c - target)
        w = w * (0.5 + e / e.max())
        w /= w.mean()
    return c, e.max()


def fit(D=6, n=20001):
    ap = np.linspace(0, 1, n)[1:]
    z = (1 - ap) ** 2 - 0.5
    return minimax(np.stack([z ** k for k in range(D + 1)], 1), -(R * ap) * psi(R * ap))


def kernel_gelu(x32, c):
    """The eleven instructions of pk_gelu_block<0..5>, per value."""
    xh = r16(x32)
    t = np.maximum(xh, -xh)                      # v_pk_max_f16 x, -x
    t = np.clip(r16(t * r16(-1 / R) + 1.0), 0, 1)  # v_pk_fma_f16 t, -1/4, 1.0 clamp
    t = r16(t * t - 0.5)                         # v_pk_fma_f16 t, t, -0.5
    ch = [r16(v) for v in c]
    q = r16(t * ch[6] + ch[5])
    for k in (4, 3, 2, 1, 0):
        q = r16(q * t + ch[k])
    return r16(q + np.maximum(xh, 0))            # v_pk_max_f16 x, 0; v_pk_add_f16


def naive_gelu(x32, D=7, Rn=4.2):
    x = np.linspace(1e-3, Rn, 4001)
    c, _ = minimax(np.stack([x * (x * x) ** k for k in range(D + 1)], 1), 0.5 - psi(x))
    xh = r16(x32)
    cc = np.clip(xh, -r16(Rn), r16(Rn))
    t = r16(cc * cc)
    q = r16(c[-1]) * np.ones_like(t)
    for k in range(D - 1, -1, -1):
        q = r16(q * t + r16(c[k]))
    return r16(xh * np.clip(r16(cc * q + 0.5), 0, 1))


if __name__ == "__main__":
    c, em = fit()
    ch = np.float16(c)
    print("fit error (minimax on |x| <= 4):", f"{em:.2e}", " sum |c_k| 2^-k =", f"{sum(abs(v) * 0.5 ** k for k, v in enumerate(c)):.3f}")
    for k, v in enumerate(ch):
        print(f"  c{k} = {float(v):+.12g}   0x{int(v.view(np.uint16)):04x}")
    rng = np.random.default_rng(0)
    for sigma in (0.5, 1.0, 2.0, 4.0):
        x = (rng.standard_normal(400000) * sigma).astype(np.float32).astype(np.float64)
        ref = gelu(x)
        rms = lambda y: float(np.sqrt(((y - ref) ** 2).mean()))
        line = f"x ~ N(0, {sigma}^2): kernel rms {rms(kernel_gelu(x, c)):.2e}   GELU(half(x)) rounded (the reference's 16-mixed) {rms(r16(gelu(r16(x)))):.2e}   output rounding alone {rms(r16(ref)):.2e}"
        if "--naive" in sys.argv:
            line += f"   Horner form of Phi in half {rms(naive_gelu(x)):.2e}"
        print(line)
    xl = np.linspace(-8, 8, 1600001)
    e = np.abs(kernel_gelu(xl, c) - gelu(xl))
    print(f"max error on [-8, 8]: {e.max():.2e} at x = {xl[e.argmax()]:.3f}")
    if "--budget" in sys.argv:
        import os, torch
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from crossscore_amd import synth
        from crossscore_amd.config import model_config
        from crossscore_amd.model import CrossScoreNet
        from oracle import crossscore_oracle as orc
        net = CrossScoreNet(model_config())
        Wt = orc.to_torch(synth.make_state_dict(net.arch, 1))
        q, r = (torch.from_numpy(a) for a in synth.make_inputs(1, 5, 518, 518, 1))
        cfg = dict(enc_heads=net.arch.enc_heads)
        ref = orc.forward(Wt, cfg, q, r)["score_map_ref_cross"]
        exact = orc.forward(Wt, cfg, q, r, emulate_bf16="f16")["score_map_ref_cross"]
        h = lambda t: t.half().float()
        cf = [float(v) for v in ch]

        def g(x):
            xh = h(x)
            t = h(xh.abs() * (-1 / R) + 1.0).clamp(0, 1)
            t = h(t * t - 0.5)
            qq = h(t * cf[6] + cf[5])
            for k in (4, 3, 2, 1, 0):
                qq = h(qq * t + cf[k])
            return h(qq + xh.clamp(min=0))
        orc.gelu_erf = g
        pk = orc.forward(Wt, cfg, q, r, emulate_bf16="f16")["score_map_ref_cross"]
        print(f"score-map MAE vs the fp32 oracle, fp16 operands everywhere: exact GELU {float((exact - ref).abs().mean()):.3e}, packed-half GELU {float((pk - ref).abs().mean()):.3e}")
