"""Parity of each HIP kernel (through the C ABI) against the fp32 oracle / a plain fp32 torch restatement on the
same seeded inputs.  Floating point: tolerances are stated per test (fp16 operands, fp32 accumulation)."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from crossscore_amd import _lib  # noqa: E402
from oracle import crossscore_oracle as orc  # noqa: E402
import hip_helpers as hh  # noqa: E402

DEV = "cuda"


def _rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


def _t(a, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV).to(dtype)


def _bf(x):
    return x.to(torch.float16)


def _same_f16(a, b):
    """a == b as fp16, except that a product rounded ONCE (the kernel: v_fma_mixlo_f16 / a fused multiply-convert) may sit one
    ulp from the same product rounded to fp32 first and to fp16 second (torch): rare, never more than one ulp."""
    d = (a.view(torch.int16).int() - b.view(torch.int16).int()).abs()
    return int(d.max()) <= 1 and float((d > 0).float().mean()) < 2e-3


# ------------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 384, 384), (1370, 1152, 384), (257, 196 + 60, 128), (65, 1536, 384),
                                   (1000, 384, 1536), (2740, 2304, 768), (65760, 1152, 384), (5000, 136, 256), (16440, 384, 128)])
def test_gemm_bias_f16(M, N, K):
    g = _rng(M + N + K)
    A = _bf(_t(g.standard_normal((M, K), dtype=np.float32)))
    W = _bf(_t(g.standard_normal((N, K), dtype=np.float32) / math.sqrt(K)))
    b = _t(g.standard_normal((N,), dtype=np.float32))
    out = hh.gemm(A, W, b, _lib.EPI_BIAS_F16)
    ref = A.float() @ W.float().t() + b
    torch.cuda.synchronize()
    # out is fp16-rounded: half an ulp = rel 2^-11 (4.9e-4) of |ref|, plus fp32 accumulation-order noise
    err = (out.float() - ref).abs()
    assert (err <= 6e-4 * ref.abs() + 5e-5).all(), float((err - 6e-4 * ref.abs()).max())


def test_gemm_exact_integer_layout():
    """A = I-like / asymmetric integer operands: any row<->col swap or fragment mis-mapping is an exact mismatch."""
    M, N, K = 256, 256, 128
    A = torch.zeros((M, K), device=DEV)
    A[torch.arange(M), torch.arange(M) % K] = 1.0  # row m selects k = m % K
    W = (torch.arange(N, device=DEV)[:, None] * 3 + torch.arange(K, device=DEV)[None, :] * 7) % 61 - 30.0  # asymmetric ints
    out = hh.gemm(_bf(A), _bf(W.float()), None, _lib.EPI_RESID_F32)
    ref = A @ W.float().t()
    torch.cuda.synchronize()
    assert torch.equal(out, ref)


@pytest.mark.parametrize("epi", [_lib.EPI_BIAS_GELU_F16, _lib.EPI_BIAS_RELU_F16, _lib.EPI_BIAS_LEAKY_F16])
def test_gemm_activations(epi):
    M, N, K = 513, 384, 384
    g = _rng(epi)
    A = _bf(_t(g.standard_normal((M, K), dtype=np.float32)))
    W = _bf(_t(g.standard_normal((N, K), dtype=np.float32) / math.sqrt(K)))
    b = _t(g.standard_normal((N,), dtype=np.float32))
    out = hh.gemm(A, W, b, epi)
    y = A.float() @ W.float().t() + b
    ref = {_lib.EPI_BIAS_GELU_F16: orc.gelu_erf(y.cpu()).to(DEV), _lib.EPI_BIAS_RELU_F16: torch.relu(y),
           _lib.EPI_BIAS_LEAKY_F16: torch.where(y >= 0, y, 0.01 * y)}[epi]
    torch.cuda.synchronize()
    err = (out.float() - ref).abs()
    # fp16 output rounding (rel 4.9e-4) + the GELU fit (<= 2.1e-4 absolute)
    assert (err <= 6e-4 * ref.abs() + 3e-4).all(), float((err - 6e-4 * ref.abs()).max())


@pytest.mark.parametrize("M,with_scale,with_resid", [(700, True, True), (1370, False, True), (700, False, False), (128, True, True)])
def test_gemm_resid_f32(M, with_scale, with_resid):
    """x += lambda * (A Wo^T + b): LayerScale is folded into the packed rows of W and into the bias (as cs_finalize does),
    the epilogue is a plain in-place residual add (full-tile fast path and ragged last row panel)."""
    N, K = 384, 1536
    g = _rng(11 + M)
    A = _bf(_t(g.standard_normal((M, K), dtype=np.float32)))
    Wf = _t(g.standard_normal((N, K), dtype=np.float32) / math.sqrt(K))
    b = _t(g.standard_normal((N,), dtype=np.float32))
    s = _t(0.5 + 0.2 * g.standard_normal((N,), dtype=np.float32)) if with_scale else None
    r = _t(g.standard_normal((M, N), dtype=np.float32)) if with_resid else None
    W = hh.pack_f16(Wf, row_scale=s)
    assert _same_f16(W, _bf(Wf * s[:, None] if with_scale else Wf))
    out = r.clone() if with_resid else None  # in-place residual update, as the encoder uses it
    out = hh.gemm(A, W, b * s if with_scale else b, _lib.EPI_RESID_F32, resid=out, out=out)
    ref = A.float() @ W.float().t() + (b * s if with_scale else b)
    if with_resid:
        ref = ref + r
    torch.cuda.synchronize()
    assert (out - ref).abs().max() < 2e-4  # fp32 out; only accumulation order differs


# ---- the 256 x 256 x 64-tile kernel (csrc/gemm256.hip): shapes with K >= 384 (six K tiles: ViT-S QKV / decoder K/V), N % 256 == 0, M >= 256 ----
def _gemm256(on):
    _lib.load().cs_debug_gemm256_enable(1 if on else 0)


def test_gemm256_exact_integer_layout():
    """Asymmetric integer operands over several K tiles, ragged M: any row<->column swap, fragment mis-mapping, swizzle error or a K tile
    read before it landed is an exact mismatch (all products and sums are small integers: exact in fp16 x fp16 -> fp32)."""
    _gemm256_integer_layout(700, 512, 1024)
    _gemm256_integer_layout(1370, 1280, 384)   # six K tiles: the shortest K loop the kernel takes (prologue + seam with nothing in between)


def _gemm256_integer_layout(M, N, K):
    A = ((torch.arange(M, device=DEV)[:, None] * 5 + torch.arange(K, device=DEV)[None, :] * 3) % 7 - 3.0)
    W = ((torch.arange(N, device=DEV)[:, None] * 3 + torch.arange(K, device=DEV)[None, :] * 7) % 5 - 2.0)
    b = (torch.arange(N, device=DEV) % 11 - 5.0)
    ref = A @ W.t() + b
    out = hh.gemm(_bf(A), _bf(W), b, _lib.EPI_RESID_F32)
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    # one-hot rows: row m selects k = (37 m) % K of W
    A1 = torch.zeros((M, K), device=DEV)
    A1[torch.arange(M), (torch.arange(M) * 37) % K] = 1.0
    out = hh.gemm(_bf(A1), _bf(W), None, _lib.EPI_BIAS_F16)
    torch.cuda.synchronize()
    assert torch.equal(out.float(), A1 @ W.t())


@pytest.mark.parametrize("M,N,K", [(256, 256, 512), (300, 768, 768), (1370, 2304, 768), (10960, 768, 3072), (10960, 3072, 768),
                                   (2740, 768, 1536), (513, 256, 640), (700, 4096, 1024),  # (N > 3072: bias read from memory per tile)
                                   (1370, 1280, 384), (10952, 1536, 384), (32880, 1280, 384)])  # K = 384: ViT-S QKV (padded) and decoder K/V
def test_gemm256_matches_reference_and_small_tile_kernel(M, N, K):
    g = _rng(M + N + K + 1)
    A = _bf(_t(g.standard_normal((M, K), dtype=np.float32)))
    W = _bf(_t(g.standard_normal((N, K), dtype=np.float32) / math.sqrt(K)))
    b = _t(g.standard_normal((N,), dtype=np.float32))
    r = _t(g.standard_normal((M, N), dtype=np.float32))
    try:
        outs = {}
        for on in (True, False):
            _gemm256(on)
            o16 = hh.gemm(A, W, b, _lib.EPI_BIAS_F16)
            og = hh.gemm(A, W, b, _lib.EPI_BIAS_GELU_F16)
            x = r.clone()
            o32 = hh.gemm(A, W, b, _lib.EPI_RESID_F32, resid=x, out=x)
            torch.cuda.synchronize()
            outs[on] = (o16, og, o32)
    finally:
        _gemm256(True)
    y = A.float() @ W.float().t() + b
    for on in (True, False):
        o16, og, o32 = outs[on]
        err = (o16.float() - y).abs()
        assert (err <= 6e-4 * y.abs() + 5e-5).all(), (on, float((err - 6e-4 * y.abs()).max()))
        refg = orc.gelu_erf(y.cpu()).to(DEV)
        err = (og.float() - refg).abs()
        assert (err <= 6e-4 * refg.abs() + 3e-4).all(), (on, float((err - 6e-4 * refg.abs()).max()))
        assert (o32 - (y + r)).abs().max() < 3e-4 * max(1.0, math.sqrt(K / 768)), on
    # both kernels start the accumulators at the bias and add the 32-deep k-steps of v_mfma_f32_16x16x32 in ascending k: the same fp32 sums
    assert torch.equal(outs[True][2], outs[False][2])
    assert torch.equal(outs[True][0], outs[False][0])


def test_ops_in_bf16_operand_mode():
    """The single-op entry points with bfloat16 operands (cs_debug_set_op_operand_dtype(1)): both GEMM kernels exact on small integers (which
    bf16 holds exactly), bias + GELU / residual epilogues and attention against fp32 references at bf16 tolerances."""
    lib = _lib.load()
    assert lib.cs_debug_set_op_operand_dtype(1) == 0
    try:
        for (M, N, K) in ((300, 384, 256), (700, 512, 1024)):  # 128-row kernel / 256 x 256 x 64-tile kernel
            A = ((torch.arange(M, device=DEV)[:, None] * 5 + torch.arange(K, device=DEV)[None, :] * 3) % 7 - 3.0)
            W = ((torch.arange(N, device=DEV)[:, None] * 3 + torch.arange(K, device=DEV)[None, :] * 7) % 5 - 2.0)
            b = (torch.arange(N, device=DEV) % 11 - 5.0)
            out = hh.gemm(A.to(torch.bfloat16).view(torch.float16), W.to(torch.bfloat16).view(torch.float16), b, _lib.EPI_RESID_F32)
            torch.cuda.synchronize()
            assert torch.equal(out, A @ W.t() + b), (M, N, K)
        g = _rng(77)
        M, N, K = 1370, 768, 768
        A = _t(g.standard_normal((M, K), dtype=np.float32)).to(torch.bfloat16)
        W = _t(g.standard_normal((N, K), dtype=np.float32) / math.sqrt(K)).to(torch.bfloat16)
        b = _t(g.standard_normal((N,), dtype=np.float32))
        y = A.float() @ W.float().t() + b
        og = hh.gemm(A.view(torch.float16), W.view(torch.float16), b, _lib.EPI_BIAS_GELU_F16).view(torch.bfloat16)
        refg = orc.gelu_erf(y.cpu()).to(DEV)
        torch.cuda.synchronize()
        err = (og.float() - refg).abs()
        assert (err <= 4.2e-3 * refg.abs() + 3e-4).all(), float((err - 4.2e-3 * refg.abs()).max())  # bf16 output: half an ulp = rel 2^-8
        # attention, dh = 64 and 48
        # the 256 x 256 tile kernel's fp32 residual epilogue and the 128-row kernel's, on random data (ragged M)
        for (M, N, K) in ((1370, 768, 768), (700, 768, 3072), (300, 384, 384)):
            A = _t(g.standard_normal((M, K), dtype=np.float32)).to(torch.bfloat16)
            W = _t(g.standard_normal((N, K), dtype=np.float32) / math.sqrt(K)).to(torch.bfloat16)
            b = _t(g.standard_normal((N,), dtype=np.float32))
            res = _t(g.standard_normal((M, N), dtype=np.float32))
            out = hh.gemm(A.view(torch.float16), W.view(torch.float16), b, _lib.EPI_RESID_F32, resid=res)
            ref = res + A.float() @ W.float().t() + b
            torch.cuda.synchronize()
            assert (out - ref).abs().max() < 2e-4 * math.sqrt(K / 768), (M, N, K, float((out - ref).abs().max()))  # exact products, fp32 accumulation order only
        for dh, heads in ((64, 6), (48, 8), (96, 8), (128, 8)):
            B, Lq, Lk = 2, 300, 500
            Q = _t(g.standard_normal((B, Lq, heads * dh), dtype=np.float32))
            Kt = _t(g.standard_normal((B, Lk, heads * dh), dtype=np.float32))
            V = _t(g.standard_normal((B, Lk, heads * dh), dtype=np.float32))
            Qb, Kb, Vb = (t.to(torch.bfloat16) for t in (Q, Kt, V))
            O = hh.attention(Qb.view(torch.float16), Kb.view(torch.float16), Vb.view(torch.float16), heads, dh).view(torch.bfloat16)
            qh, kh, vh = (t.float().view(B, -1, heads, dh).transpose(1, 2) for t in (Qb, Kb, Vb))
            ref = torch.softmax(qh @ kh.transpose(-1, -2) / math.sqrt(dh), -1) @ vh
            ref = ref.transpose(1, 2).reshape(B, Lq, heads * dh)
            torch.cuda.synchronize()
            d = (O.float() - ref).abs()
            assert float(d.max()) < 3e-2 and float(d.mean()) < 3e-3, (dh, float(d.max()), float(d.mean()))
    finally:
        lib.cs_debug_set_op_operand_dtype(0)


def _row_partials(x, sp):
    """(sum, sumsq) of each row of x over the column ranges the producing epilogue's (column tile, wave) pairs own."""
    M, Cc = x.shape
    tiles = sp // 4
    bn = Cc // tiles if Cc % 192 == 0 else 128
    wn = bn // 4
    out = torch.zeros((M, sp, 2), device=x.device)
    for t in range(tiles):
        for w in range(4):
            c0 = t * bn + w * wn
            seg = x[:, c0:min(c0 + wn, Cc)]
            out[:, t * 4 + w, 0] = seg.sum(1)
            out[:, t * 4 + w, 1] = (seg * seg).sum(1)
    return out


@pytest.mark.parametrize("M,Cc,N,epi", [(700, 384, 1152, _lib.EPI_LN_F16), (1370, 384, 1536, _lib.EPI_LN_GELU_F16),
                                        (300, 128, 384, _lib.EPI_LN_F16), (257, 768, 768, _lib.EPI_LN_GELU_F16)])
def test_gemm_layernorm_folded_consumer(M, Cc, N, epi):
    """LN(x) W^T + b computed as rstd*(fp16(x) W'^T - mu*s) + c from the producer's per-row partial sums
    (HF modeling_dinov2.py:365,373 LayerNorm -> query/key/value / fc1)."""
    g = _rng(M + N)
    x = _t(2.0 * g.standard_normal((M, Cc), dtype=np.float32) + 0.7)           # non-zero row means
    gam = _t(1 + 0.2 * g.standard_normal((Cc,), dtype=np.float32))
    bet = _t(0.1 * g.standard_normal((Cc,), dtype=np.float32))
    Wf = _t(g.standard_normal((N, Cc), dtype=np.float32) / math.sqrt(Cc))
    b = _t(0.1 * g.standard_normal((N,), dtype=np.float32))
    sp = 4 * hh.column_tiles(Cc)
    Wp = hh.pack_f16(Wf, col_scale=gam)
    assert _same_f16(Wp, _bf(Wf * gam[None, :]))
    s, c = hh.ln_fold_consts(Wp, Wf, bet, b)
    assert (s - Wp.float().sum(1)).abs().max() < 1e-4 and (c - (b + Wf @ bet)).abs().max() < 1e-4
    part = _row_partials(x, sp)
    out = hh.gemm(_bf(x), Wp, c, epi, ln_part=part, col_s=s, ln_eps=1e-6)
    ref = orc.layer_norm(x.cpu(), gam.cpu(), bet.cpu(), 1e-6).to(DEV) @ Wf.t() + b
    if epi == _lib.EPI_LN_GELU_F16:
        ref = orc.gelu_erf(ref.cpu()).to(DEV)
    torch.cuda.synchronize()
    err = (out.float() - ref).abs()
    # fp16 rounding of x and of gamma*W (operands, rel 4.9e-4 each over K terms) + fp16 output of |ref| up to ~8
    assert err.max() < 1e-2 and err.mean() < 8e-4, (float(err.max()), float(err.mean()))


@pytest.mark.parametrize("M,Cc", [(700, 384), (128, 384), (300, 128), (1370, 768)])
def test_gemm_resid_producer_emits_bf16_rows_and_partials(M, Cc):
    """RESID_F32_LN: x += A Wo^T + b, plus the fp16 copy of the new rows and their per-(tile, wave) partial sums."""
    K = 256
    g = _rng(M * 3 + Cc)
    A = _bf(_t(g.standard_normal((M, K), dtype=np.float32)))
    W = _bf(_t(g.standard_normal((Cc, K), dtype=np.float32) / math.sqrt(K)))
    b = _t(g.standard_normal((Cc,), dtype=np.float32))
    r = _t(g.standard_normal((M, Cc), dtype=np.float32))
    sp = 4 * hh.column_tiles(Cc)
    xb = torch.zeros((M, Cc), dtype=torch.float16, device=DEV)
    st = torch.full((M, sp, 2), 777.0, device=DEV)
    out = r.clone()
    hh.gemm(A, W, b, _lib.EPI_RESID_F32_LN, resid=out, out=out, out_f16=xb, stats_out=st)
    ref = A.float() @ W.float().t() + b + r
    torch.cuda.synchronize()
    assert (out - ref).abs().max() < 2e-4
    assert torch.equal(xb, _bf(out))
    want = _row_partials(out, sp)
    assert (st - want).abs().max() < 2e-3 * (1 + want.abs().max())  # fp32 sums in a different order
    mu = st[:, :, 0].sum(1) / Cc
    assert (mu - out.mean(1)).abs().max() < 1e-5


@pytest.mark.parametrize("bf16", [False, True])
@pytest.mark.parametrize("M,Cc,K", [(256, 768, 768), (1370, 768, 3072), (300, 1024, 1024), (10960, 768, 768)])
def test_gemm256_layernorm_folded_chain(M, Cc, K, bf16):
    """The ViT-B encoder's LayerNorm fold on the 256-tile GEMM (gemm256.hip LN = 2 / 1; HF modeling_dinov2.py:361-380): the residual epilogue
    (x += A Wo^T + b) also emits 16-bit(x) and per-row partial sums over its 64-column wave slices, cs_op_ln_finalize turns them into
    (mean, rstd), and the consuming projection computes LN(x) W^T + b as rstd * (16-bit(x) W'^T - mean * s) + c [+ GELU]."""
    lib = _lib.load()
    assert lib.cs_debug_set_op_operand_dtype(1 if bf16 else 0) == 0
    try:
        rd = (lambda t: t.to(torch.bfloat16).view(torch.float16)) if bf16 else _bf       # fp32 -> the operand type's bits (carried as uint16 / half)
        fl = (lambda t: t.view(torch.bfloat16).float()) if bf16 else (lambda t: t.float())
        g = _rng(M + Cc + K + (7 if bf16 else 0))
        A = rd(_t(g.standard_normal((M, K), dtype=np.float32)))
        W = rd(_t(g.standard_normal((Cc, K), dtype=np.float32) / math.sqrt(K)))
        b = _t(g.standard_normal((Cc,), dtype=np.float32))
        r = _t(3.0 * g.standard_normal((M, Cc), dtype=np.float32) + 0.5)
        r[:, 5] += 300.0   # DINOv2-style outlier channels in the residual stream
        r[:, 200] -= 250.0
        sp = Cc // 64
        x16 = torch.zeros((M, Cc), dtype=torch.float16, device=DEV)
        st = torch.full((M, sp, 2), 777.0, device=DEV)
        x = r.clone()
        hh.gemm(A, W, b, _lib.EPI_RESID_F32_LN, resid=x, out=x, out_f16=x16, stats_out=st)
        ref = fl(A) @ fl(W).t() + b + r
        torch.cuda.synchronize()
        assert (x - ref).abs().max() < 3e-4 * max(1.0, math.sqrt(K / 768)) * 4   # (|x| up to 300: fp32 half-ulp 1.5e-5, sums of K products)
        assert torch.equal(x16, rd(x))
        want = torch.stack([x.view(M, sp, 64).sum(2), (x * x).view(M, sp, 64).sum(2)], dim=2)
        assert (st - want).abs().max() < 2e-5 * float(want.abs().max())
        stat = hh.ln_finalize(st, Cc)
        torch.cuda.synchronize()
        mu, var = x.double().mean(1), x.double().var(1, unbiased=False)
        assert (stat[:M, 0, 0].double() - mu).abs().max() < 1e-4
        assert ((stat[:M, 0, 1].double() * torch.sqrt(var + 1e-6)) - 1).abs().max() < 2e-4   # rstd, relative (E[x^2] - mu^2 in fp32)
        assert (stat[M:] == 0).all()
        # consumer: a QKV-like (N = 3 Cc) and an fc1-like (N = 4 Cc, GELU) projection of LN(x)
        gam = _t(1 + 0.2 * g.standard_normal((Cc,), dtype=np.float32))
        bet = _t(0.1 * g.standard_normal((Cc,), dtype=np.float32))
        for N, epi in ((3 * Cc, _lib.EPI_LN_F16), (4 * Cc, _lib.EPI_LN_GELU_F16)):
            Wf = _t(g.standard_normal((N, Cc), dtype=np.float32) / math.sqrt(Cc))
            bb = _t(0.1 * g.standard_normal((N,), dtype=np.float32))
            Wp = rd(Wf * gam[None, :])
            s = fl(Wp).sum(1)
            c = bb + Wf @ bet
            out = hh.gemm(x16, Wp, c, epi, ln_part=stat, col_s=s, ln_eps=1e-6)
            torch.cuda.synchronize()
            # (a) the same arithmetic in fp32 from the same rounded operands: summation order only
            xn = (fl(x16) - stat[:M, 0, 0:1]) * stat[:M, 0, 1:2]
            same = xn @ fl(Wp).t() + c
            # (b) the layer's fp32 definition
            true = orc.layer_norm(x.cpu(), gam.cpu(), bet.cpu(), 1e-6).to(DEV) @ Wf.t() + bb
            if epi == _lib.EPI_LN_GELU_F16:
                same, true = orc.gelu_erf(same.cpu()).to(DEV), orc.gelu_erf(true.cpu()).to(DEV)
            o = fl(out)
            rel = 4.2e-3 if bf16 else 6e-4   # half an ulp of the 16-bit output
            err = (o - same).abs()
            assert (err <= rel * same.abs() + (4e-3 if bf16 else 1.5e-3)).all(), float((err - rel * same.abs()).max())
            err = (o - true).abs()
            assert err.mean() < (8e-3 if bf16 else 1.2e-3) and err.max() < (0.2 if bf16 else 3e-2), (float(err.mean()), float(err.max()))
    finally:
        lib.cs_debug_set_op_operand_dtype(0)


@pytest.mark.parametrize("offset", [8.0, 30.0])
def test_gemm256_layernorm_fold_with_a_row_offset(offset):
    """the folded LayerNorm takes its variance as E[x^2] - mu^2 from fp32 sums and multiplies 16-bit(x), the UN-normalised rows.
    Both lose accuracy when a row's mean is large against its spread.  Rows x = offset * sigma + N(0, sigma), sigma = 1, C = 768, fp16 operands:
      * statistics: the fp32 sums carry ~6e-8 * C * offset^2 of absolute error -> rstd relative error <= 1e-4 at offset 30;
      * consumer: the half-precision step of x is 2^-7 at |x| in [8, 16) and 2^-6 in [16, 32): rounding noise 2.3e-3 / 4.5e-3 rms per element
        against sigma = 1, which a projection with unit-norm rows passes on 1:1 -- the supported range is |mu| <~ 30 sigma at about 5e-3 of LN output
        error; beyond it ln_fold = 2 (LayerNorm launches on the fp32 stream) is the setting.  A DINOv2 residual row has |mu| << sigma (sigma is set
        by its outlier channels), which the +-300 channels of test_gemm256_layernorm_folded_chain model."""
    M, Cc, K = 512, 768, 768
    g = _rng(int(offset) + 901)
    A = _bf(_t(g.standard_normal((M, K), dtype=np.float32)))
    W = _bf(_t(g.standard_normal((Cc, K), dtype=np.float32) / math.sqrt(K) * 0.5))
    b = _t(0.1 * g.standard_normal((Cc,), dtype=np.float32))
    r = _t(g.standard_normal((M, Cc), dtype=np.float32) * 0.85 + offset)
    sp = Cc // 64
    x16 = torch.zeros((M, Cc), dtype=torch.float16, device=DEV)
    st = torch.zeros((M, sp, 2), device=DEV)
    x = r.clone()
    hh.gemm(A, W, b, _lib.EPI_RESID_F32_LN, resid=x, out=x, out_f16=x16, stats_out=st)
    stat = hh.ln_finalize(st, Cc)
    torch.cuda.synchronize()
    mu, var = x.double().mean(1), x.double().var(1, unbiased=False)
    e_mu = float((stat[:M, 0, 0].double() - mu).abs().max())
    e_rs = float(((stat[:M, 0, 1].double() * torch.sqrt(var + 1e-6)) - 1).abs().max())
    gam = _t(1 + 0.2 * g.standard_normal((Cc,), dtype=np.float32))
    bet = _t(0.1 * g.standard_normal((Cc,), dtype=np.float32))
    N = 3 * Cc
    Wf = _t(g.standard_normal((N, Cc), dtype=np.float32) / math.sqrt(Cc))
    bb = _t(0.1 * g.standard_normal((N,), dtype=np.float32))
    Wp = _bf(Wf * gam[None, :])
    out = hh.gemm(x16, Wp, bb + Wf @ bet, _lib.EPI_LN_F16, ln_part=stat, col_s=Wp.float().sum(1), ln_eps=1e-6)
    torch.cuda.synchronize()
    true = orc.layer_norm(x.cpu(), gam.cpu(), bet.cpu(), 1e-6).to(DEV) @ Wf.t() + bb
    err = (out.float() - true).abs()
    print(f"row offset {offset} sigma: mean err {e_mu:.2e}, rstd rel err {e_rs:.2e}, consumer err mean {float(err.mean()):.2e} max {float(err.max()):.2e}")
    assert e_mu < 2e-4 and e_rs < 3e-4, (e_mu, e_rs)
    step = 2.0 ** -7 if offset < 16 else 2.0 ** -6
    assert float(err.mean()) < 3 * step / math.sqrt(12) + 1e-3 and float(err.max()) < 12 * step + 1e-2, (float(err.mean()), float(err.max()))


def test_gemm_patch_epilogue_and_im2col():
    """im2col + patch GEMM == conv patchify + cls/pos placement (HF:97-149), rows m -> img*T + 1 + p."""
    I, H, W, P, Cc = 3, 75, 90, 14, 128
    gh, gw = H // P, W // P
    Np, T = gh * gw, gh * gw + 1
    g = _rng(5)
    x = _t(g.standard_normal((I, 3, H, W), dtype=np.float32))
    wconv = _t(g.standard_normal((Cc, 3, P, P), dtype=np.float32) / math.sqrt(588))
    b = _t(g.standard_normal((Cc,), dtype=np.float32))
    pos = _t(g.standard_normal((T, Cc), dtype=np.float32))
    Kp = 640
    A = hh.im2col(x, P, Kp)
    # im2col vs unfold restatement
    xr = x[:, :, : gh * P, : gw * P].reshape(I, 3, gh, P, gw, P).permute(0, 2, 4, 1, 3, 5).reshape(I * Np, 588)
    assert torch.equal(A[:, :588], _bf(xr)) and (A[:, 588:] == 0).all()
    Wp = hh.pack_f16(wconv.reshape(Cc, 588), Kp)
    assert torch.equal(Wp[:, :588], _bf(wconv.reshape(Cc, 588))) and (Wp[:, 588:] == 0).all()
    out = torch.full((I * T, Cc), 7.0, device=DEV)
    hh.gemm(A, Wp, b, _lib.EPI_PATCH_F32, out=out, pos=pos, Np=Np, K=Kp)
    ref = (A[:, :588].float() @ Wp[:, :588].float().t() + b).reshape(I, Np, Cc) + pos[None, 1:]
    torch.cuda.synchronize()
    o3 = out.reshape(I, T, Cc)
    assert (o3[:, 1:] - ref).abs().max() < 2e-4
    assert (o3[:, 0] == 7.0).all()  # CLS rows untouched by the GEMM


def test_patch_embed_mean_centred_vs_fp32_conv():
    """Mean-centred patch embedding (im2col_rows_kernel's per-patch channel means + the PATCH epilogue's mean * sum(W) term) against
    an fp32 convolution (HF modeling_dinov2.py:141-149) on smooth, natural-image-like input, where a patch is mostly its mean: the
    centred form must beat the plain one clearly, and noise input (patch means ~ 0) must be unaffected."""
    I, H, W, P, Cc = 2, 98, 112, 14, 128
    gh, gw = H // P, W // P
    Np = gh * gw
    g = _rng(17)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    smooth = np.stack([[1.8 * np.sin(xx / (19.0 + 3 * c) + i) + 1.2 * np.cos(yy / (23.0 + c)) + 0.7 * (c - 1) for c in range(3)] for i in range(I)]).astype(np.float32)
    smooth += 0.05 * g.standard_normal(smooth.shape, dtype=np.float32)
    noise = g.standard_normal((I, 3, H, W), dtype=np.float32)
    wconv = _t(g.standard_normal((Cc, 3, P, P), dtype=np.float32) / math.sqrt(588))
    b = _t(g.standard_normal((Cc,), dtype=np.float32))
    pos = _t(g.standard_normal((1 + Np, Cc), dtype=np.float32))
    errs = {}
    for name, img in (("smooth", smooth), ("noise", noise)):
        x = _t(img)
        ref = torch.nn.functional.conv2d(x.double(), wconv.double(), b.double(), stride=P).flatten(2).transpose(1, 2) + pos[None, 1:].double()
        for centred in (0, 1):
            out = hh.patch_embed(x, wconv, b, pos, P, centred).reshape(I, 1 + Np, Cc)
            torch.cuda.synchronize()
            assert (out[:, 0] == 7.0).all()
            errs[(name, centred)] = float((out[:, 1:].double() - ref).abs().mean())
    print("patch embedding mean abs error vs fp64 conv:", {k: f"{v:.2e}" for k, v in errs.items()})
    assert errs[("smooth", 1)] < 0.5 * errs[("smooth", 0)]           # the coherent weight-rounding term is gone
    assert errs[("smooth", 1)] < 1.5 * errs[("noise", 1)] + 1e-4      # and what is left is ordinary operand rounding
    assert abs(errs[("noise", 1)] - errs[("noise", 0)]) < 0.2 * errs[("noise", 0)] + 1e-5


@pytest.mark.parametrize("I,H,W,Cc", [(3, 518, 518, 384), (2, 518, 686, 384), (1, 1036, 1036, 384), (2, 224, 238, 768), (1, 14, 28, 384),
                                      (2, 75, 90, 384)])  # (trailing pixels past the last whole patch are ignored, HF:141-149)
def test_patch_embed_one_launch_matches_two_kernel_path_and_conv(I, H, W, Cc):
    """csrc/patch.hip (strip -> centred 16-bit tile in LDS -> MFMA -> token rows) against (a) the im2col + GEMM pair it replaces -- the A
    operand has the same bits (same mean, same rounding), only the fp32 summation order of the products differs -- and (b) an fp64
    convolution (HF modeling_dinov2.py:141-149).  Shapes: the benchmark's 37-patch rows, rows cut into two runs (49 and 74 patches), a
    two-pass width (768), a single-patch-row image; smooth images, where the mean term carries most of the value."""
    P = 14
    gh, gw = H // P, W // P
    Np = gh * gw
    g = _rng(1000 + H + W + Cc)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    img = np.stack([[1.8 * np.sin(xx / (19.0 + 3 * c) + i) + 1.2 * np.cos(yy / (23.0 + c)) + 0.7 * (c - 1) for c in range(3)] for i in range(I)]).astype(np.float32)
    img += 0.3 * g.standard_normal(img.shape, dtype=np.float32)
    x = _t(img)
    wconv = _t(g.standard_normal((Cc, 3, P, P), dtype=np.float32) / math.sqrt(588))
    b = _t(g.standard_normal((Cc,), dtype=np.float32))
    pos = _t(g.standard_normal((1 + Np, Cc), dtype=np.float32))
    one = hh.patch_embed_fused(x, wconv, b, pos, P).reshape(I, 1 + Np, Cc)
    two = hh.patch_embed(x, wconv, b, pos, P, 1).reshape(I, 1 + Np, Cc)
    torch.cuda.synchronize()
    assert (one[:, 0] == 7.0).all()  # CLS rows are not this kernel's
    assert torch.isfinite(one).all()
    d = (one[:, 1:] - two[:, 1:]).abs()
    assert d.max() < 2e-5, float(d.max())  # fp32 summation order only (measured 4e-6)
    ref = torch.nn.functional.conv2d(x.double(), wconv.double(), b.double(), stride=P).flatten(2).transpose(1, 2) + pos[None, 1:].double()
    e1 = (one[:, 1:].double() - ref).abs().mean()
    e2 = (two[:, 1:].double() - ref).abs().mean()
    assert e1 < 1.05 * e2 + 1e-7 and e1 < 3e-4, (float(e1), float(e2))


def test_patch_embed_one_launch_refuses_what_it_does_not_take():
    """odd row pitch (8-byte loads), C not a multiple of 384, other patch sizes: CS_ERR_BAD_ARG from the op; cs_forward falls back to im2col + GEMM"""
    g = _rng(5)
    for (H, W, Cc, P) in ((70, 91, 384, 14), (70, 84, 256, 14)):
        x = _t(g.standard_normal((1, 3, H, W), dtype=np.float32))
        w = _t(g.standard_normal((Cc, 3, P, P), dtype=np.float32))
        b = _t(g.standard_normal((Cc,), dtype=np.float32))
        pos = _t(g.standard_normal((1 + (H // P) * (W // P), Cc), dtype=np.float32))
        with pytest.raises(ValueError):
            hh.patch_embed_fused(x, w, b, pos, P)


def test_patch_embed_one_launch_bf16_operands():
    I, H, W, Cc, P = 2, 518, 518, 384, 14
    Np = (H // P) * (W // P)
    g = _rng(77)
    x = _t(g.standard_normal((I, 3, H, W), dtype=np.float32))
    wconv = _t(g.standard_normal((Cc, 3, P, P), dtype=np.float32) / math.sqrt(588))
    b = _t(g.standard_normal((Cc,), dtype=np.float32))
    pos = _t(g.standard_normal((1 + Np, Cc), dtype=np.float32))
    lib = _lib.load()
    lib.cs_debug_set_op_operand_dtype(1)
    try:
        one = hh.patch_embed_fused(x, wconv, b, pos, P).reshape(I, 1 + Np, Cc)
        two = hh.patch_embed(x, wconv, b, pos, P, 1).reshape(I, 1 + Np, Cc)
        torch.cuda.synchronize()
    finally:
        lib.cs_debug_set_op_operand_dtype(0)
    assert (one[:, 1:] - two[:, 1:]).abs().max() < 2e-5
    ref = torch.nn.functional.conv2d(x.double(), wconv.double(), b.double(), stride=P).flatten(2).transpose(1, 2) + pos[None, 1:].double()
    assert (one[:, 1:].double() - ref).abs().mean() < 4e-3  # bfloat16 operands: 8 significant bits


def test_attention_dh96_long_keys():
    """Decoder cross-attention at ViT-B size: dh = 96, Lk = 13 690 (cfg-3: 10 references of 1369 patches), ragged in both directions."""
    dh, heads, Lq, Lk = 96, 8, 300, 13690
    g = _rng(96)
    Q = _bf(_t(1.5 * g.standard_normal((1, Lq, heads * dh), dtype=np.float32)))
    K = _bf(_t(1.5 * g.standard_normal((1, Lk, heads * dh), dtype=np.float32)))
    V = _bf(_t(g.standard_normal((1, Lk, heads * dh), dtype=np.float32)))
    Q = hh.prescale_q(Q, dh)
    O, lse = hh.attention(Q, K, V, heads, dh, lse=True, q_scale=1.0)
    ref, _, lse_ref = _attn_ref(Q, K, V, heads, dh)
    torch.cuda.synchronize()
    err = (O.float() - ref).abs()
    assert err.max() < 4e-3 and err.mean() < 4e-4, (float(err.max()), float(err.mean()))
    assert (lse * math.log(2.0) - lse_ref).abs().max() < 5e-4


@pytest.mark.parametrize("gh,gw,B", [(5, 6, 3), (37, 37, 8), (1, 1, 5), (16, 9, 2)])
def test_head_score_mean_from_the_same_launch(gh, gw, B):
    """score_summariser.py:180-192: score_map.mean(dim=[-1, -2]).  The head launch leaves it (CsGemmParams::mean_*): the score map is the one the
    launch writes without the mean, the mean is the fp64 mean of that map to fp32 rounding, an image's mean has the same bits wherever the image
    sits in the batch (row tiles of 128 straddle images: 30 / 1369 / 1 / 144 patch rows per image), alone or not, launch after launch (the
    finisher waves put the arrival counters back to zero), and a launch that finds a counter not at zero is what the contract says it is: wrong."""
    P, Cc = 14, 384
    Np = gh * gw
    g = _rng(90 + gh)
    one = g.standard_normal((Np, Cc), dtype=np.float32)
    rows = np.concatenate([one if b in (0, B - 1) else g.standard_normal((Np, Cc), dtype=np.float32) for b in range(B)])
    A = _bf(_t(rows))
    W = _bf(_t(g.standard_normal((P * P, Cc), dtype=np.float32) / math.sqrt(Cc)))
    b = _t(g.standard_normal((P * P,), dtype=np.float32))
    plain, _, _ = hh.head_score(A, W, b, B, gh, gw, P, want_mean=False)
    score, mean, cnt = hh.head_score(A, W, b, B, gh, gw, P)
    torch.cuda.synchronize()
    assert torch.equal(score, plain)
    ref = score.double().mean(dim=(-1, -2))
    err = float((mean.double() - ref).abs().max())
    print(f"head mean {gh}x{gw} B={B}: max |mean - fp64 mean of the map| = {err:.2e}")
    assert err < 2e-7 * max(1.0, math.sqrt(Np * P * P) / 64)
    assert int(cnt.abs().sum()) == 0
    assert mean[0].item() == mean[B - 1].item()           # the same image first and last in the batch
    solo, smean, _ = hh.head_score(A[:Np].contiguous(), W, b, 1, gh, gw, P)
    assert torch.equal(solo[0], score[0]) and smean[0].item() == mean[0].item()
    for _ in range(3):                                     # the same counters again
        _, again, _ = hh.head_score(A, W, b, B, gh, gw, P, cnt=cnt)
        assert torch.equal(again, mean)
    assert int(cnt.abs().sum()) == 0


@pytest.mark.parametrize("act,powp", [(0, 1.0), (0, 2.0), (0, 0.5), (1, 1.0)])
def test_gemm_head_score_jigsaw(act, powp):
    """sigmoid/tanh (+pow) + jigsaw store (regression_layer.py:26-62, utils/misc/image.py:8-21)."""
    B, gh, gw, P, Cc = 2, 5, 6, 14, 128
    Np = gh * gw
    g = _rng(9)
    A = _bf(_t(g.standard_normal((B * Np, Cc), dtype=np.float32)))
    W = _bf(_t(g.standard_normal((P * P, Cc), dtype=np.float32) / math.sqrt(Cc)))
    b = _t(g.standard_normal((P * P,), dtype=np.float32))
    out = torch.zeros((B, gh * P, gw * P), device=DEV)
    hh.gemm(A, W, b, _lib.EPI_HEAD_SCORE, out=out, Np=Np, gw=gw, P=P, act=act, powp=powp, ldc=4)
    y = A.float() @ W.float().t() + b
    y = torch.sigmoid(y) if act == 0 else torch.tanh(y)
    if powp != 1.0:
        y = y ** powp
    ref = orc.jigsaw_to_image(y.cpu().view(B, Np, P, P), gh, gw).to(DEV)
    torch.cuda.synchronize()
    assert (out - ref).abs().max() < 1e-5


def test_gemm_rejects_bad_shapes():
    A = torch.zeros((64, 96), dtype=torch.float16, device=DEV)
    W = torch.zeros((64, 96), dtype=torch.float16, device=DEV)
    with pytest.raises(ValueError):
        hh.gemm(A, W)  # K=96 not a multiple of 64


# ------------------------------------------------------------------------------------------- attention
def _attn_ref(Q, K, V, heads, dh, prescaled=True):
    """fp32 reference.  prescaled: Q is hip_helpers.prescale_q(..) (carries log2(e)/sqrt(dh), as the forward's Q projections emit it),
    so the natural-log logits are q.k * ln 2; otherwise raw Q and q.k / sqrt(dh)."""
    B, Lq, _ = Q.shape
    Lk = K.shape[1]
    q = Q.float().view(B, Lq, heads, dh).transpose(1, 2)
    k = K.float().view(B, Lk, heads, dh).transpose(1, 2)
    v = V.float().view(B, Lk, heads, dh).transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) * (math.log(2.0) if prescaled else 1.0 / math.sqrt(dh))
    p = torch.softmax(s, dim=-1)
    o = (p @ v).transpose(1, 2).reshape(B, Lq, heads * dh)
    return o, p, torch.logsumexp(s, dim=-1)


@pytest.mark.parametrize("dh,heads,Lq,Lk,B", [
    (64, 2, 128, 64, 1), (64, 6, 1370, 1370, 2), (64, 2, 31, 31, 3), (64, 1, 129, 65, 1),
    (48, 8, 1369, 1369, 1), (48, 8, 300, 1500, 2), (96, 8, 257, 513, 1), (16, 8, 30, 60, 2), (16, 8, 200, 333, 1),
    (128, 8, 257, 700, 1), (128, 2, 64, 64, 2),  # dinov2-large's decoder heads (C = 1024 / 8)
    (192, 8, 257, 700, 1), (192, 2, 64, 129, 2),  # dinov2-giant's decoder heads (C = 1536 / 8)
])
def test_attention_matches_fp32(dh, heads, Lq, Lk, B):
    g = _rng(dh * 1000 + Lq + Lk)
    Cc = heads * dh
    Q = _bf(_t(1.5 * g.standard_normal((B, Lq, Cc), dtype=np.float32)))
    K = _bf(_t(1.5 * g.standard_normal((B, Lk, Cc), dtype=np.float32)))
    V = _bf(_t(g.standard_normal((B, Lk, Cc), dtype=np.float32)))
    Q = hh.prescale_q(Q, dh)
    O, lse = hh.attention(Q, K, V, heads, dh, lse=True, q_scale=1.0)
    ref, _, lse_ref = _attn_ref(Q, K, V, heads, dh)
    torch.cuda.synchronize()
    # P is rounded to fp16 before PV and O to fp16 on store (half an ulp = 2e-3 at |O| ~ 4), fp32 softmax statistics
    err = (O.float() - ref).abs()
    assert err.max() < 4e-3 and err.mean() < 4e-4, (float(err.max()), float(err.mean()))
    assert (lse * math.log(2.0) - lse_ref).abs().max() < 5e-4


def test_attention_raw_q_scaled_in_kernel():
    """q_scale = 0: raw Q, multiplied by log2(e)/sqrt(dh) and rounded to fp16 once more inside the kernel (rel 4.9e-4 of the logits);
    the weights entry rounds the same way, so its rows still sum to one."""
    dh, heads, Lq, Lk, B = 64, 3, 200, 700, 2
    g = _rng(123)
    Cc = heads * dh
    Q = _bf(_t(1.5 * g.standard_normal((B, Lq, Cc), dtype=np.float32)))
    K = _bf(_t(1.5 * g.standard_normal((B, Lk, Cc), dtype=np.float32)))
    V = _bf(_t(g.standard_normal((B, Lk, Cc), dtype=np.float32)))
    O, lse = hh.attention(Q, K, V, heads, dh, lse=True)
    Pw = hh.attention_weights(Q, K, heads, dh, lse, head=1)
    ref, p_ref, lse_ref = _attn_ref(Q, K, V, heads, dh, prescaled=False)
    torch.cuda.synchronize()
    err = (O.float() - ref).abs()
    assert err.max() < 8e-3 and err.mean() < 6e-4, (float(err.max()), float(err.mean()))
    assert (lse * math.log(2.0) - lse_ref).abs().max() < 3e-3
    assert (Pw.sum(-1) - 1).abs().max() < 1e-4
    assert (Pw - p_ref[:, 1]).abs().max() < 2e-3


def test_attention_exact_layout():
    """One-hot softmax (huge logit on a chosen key per query) makes O[q] == V[key(q)] exactly: catches any key /
    d permutation error in the transposed-read / accumulator-as-operand path."""
    dh, heads, Lq, Lk = 64, 1, 160, 200
    Q = torch.zeros((1, Lq, dh), device=DEV)
    K = torch.zeros((1, Lk, dh), device=DEV)
    sel = (torch.arange(Lq, device=DEV) * 37 + 11) % Lk
    # 8-bit codes in +-16 on dims 0..7 : q.k is maximal (8*256) only for the matching key
    def code(i):
        bits = ((i[:, None] >> torch.arange(8, device=DEV)[None]) & 1).float() * 2 - 1
        return bits * 16
    K[0, :, :8] = code(torch.arange(Lk, device=DEV))
    Q[0, :, :8] = code(sel)
    V = torch.arange(Lk * dh, device=DEV).float().view(1, Lk, dh) % 251 - 125  # exact in fp16
    O = hh.attention(_bf(Q), _bf(K), _bf(V), heads, dh)  # raw Q: the kernel applies log2(e)/sqrt(dh) itself
    torch.cuda.synchronize()
    assert (O.float()[0] - V[0][sel]).abs().max() < 1e-20  # exact up to the e^-64 tails of the other keys


def test_attention_forces_rescale_branch():
    """Spike in a LATE key tile: the running max jumps there, so the online-softmax rescale of O and l is exercised
    against a full fp32 reference (cdna guide rule 26)."""
    dh, heads, Lq, Lk = 64, 2, 96, 640
    g = _rng(77)
    Q = _t(g.standard_normal((1, Lq, heads * dh), dtype=np.float32))
    K = _t(g.standard_normal((1, Lk, heads * dh), dtype=np.float32))
    V = _t(g.standard_normal((1, Lk, heads * dh), dtype=np.float32))
    K[0, 600] = 6.0 * Q[0, 5]  # key 600 (tile 9) dominates query 5
    K[0, 3] = 3.0 * Q[0, 40]   # early spike for query 40
    Qb, Kb, Vb = hh.prescale_q(Q, dh), _bf(K), _bf(V)
    O = hh.attention(Qb, Kb, Vb, heads, dh, q_scale=1.0)
    ref, _, _ = _attn_ref(Qb, Kb, Vb, heads, dh)
    torch.cuda.synchronize()
    assert (O.float() - ref).abs().max() < 4e-3


@pytest.mark.parametrize("step,tiles", [(0.9, 8), (1.0, 24), (-1.5, 12)])
def test_attention_lazy_reference_point(step, tiles):
    """The softmax reference point is only moved when a row's tile maximum exceeds it by more than 2^8 (attention.hip kTau): logits that
    climb by `step` (base-2 units) per 64-key tile stay below the threshold for several tiles (P grows up to 256 with no rescale), then
    cross it; falling logits never move it.  All three must match the fp32 softmax."""
    dh, heads, Lq = 64, 1, 64
    Lk = 64 * tiles
    g = _rng(int(10 * abs(step)) + tiles)
    u = g.standard_normal(dh).astype(np.float32)
    u /= np.linalg.norm(u)
    Q = _t(np.tile(u[None, None, :], (1, Lq, 1)) * (1 + 0.05 * g.standard_normal((1, Lq, 1)).astype(np.float32)))  # q.u ~ 1 in base-2 units
    ramp = (np.arange(Lk) // 64).astype(np.float32) * step
    K = _t(ramp[None, :, None] * u[None, None, :] + 0.3 * g.standard_normal((1, Lk, dh)).astype(np.float32))
    V = _t(g.standard_normal((1, Lk, dh), dtype=np.float32))
    Qs, Kb, Vb = _bf(Q), _bf(K), _bf(V)  # Q taken as already scaled: q_scale = 1
    O, lse = hh.attention(Qs, Kb, Vb, heads, dh, lse=True, q_scale=1.0)
    ref, _, lse_ref = _attn_ref(Qs, Kb, Vb, heads, dh)
    torch.cuda.synchronize()
    assert (O.float() - ref).abs().max() < 4e-3
    assert (lse * math.log(2.0) - lse_ref).abs().max() < 5e-4


def test_attention_strided_packed_qkv():
    """Reads Q/K/V in place from a packed [tokens][3C] projection, as the encoder does."""
    dh, heads, T, B = 64, 2, 150, 2
    Cc = heads * dh
    g = _rng(3)
    qkv = _bf(_t(g.standard_normal((B, T, 3 * Cc), dtype=np.float32)))
    qkv[:, :, :Cc] = hh.prescale_q(qkv[:, :, :Cc], dh)
    O = hh.attention(qkv[:, :, :Cc], qkv[:, :, Cc:2 * Cc], qkv[:, :, 2 * Cc:], heads, dh, q_scale=1.0)
    ref, _, _ = _attn_ref(qkv[:, :, :Cc], qkv[:, :, Cc:2 * Cc], qkv[:, :, 2 * Cc:], heads, dh)
    torch.cuda.synchronize()
    assert (O.float() - ref).abs().max() < 4e-3


@pytest.mark.parametrize("dh,heads,Lq,Lk,packed", [(64, 6, 1370, 1370, True), (48, 8, 300, 1369, False), (96, 8, 257, 6845 % 2000 + 13, False),
                                                   (128, 8, 130, 65, False), (16, 8, 70, 30, True)])
def test_attention_ignores_poison_behind_the_last_key(dh, heads, Lq, Lk, packed):
    """K / V are views into a larger buffer whose rows behind key Lk-1 hold NaN / Inf bit patterns: the ragged last tile must not
    read them -- a masked key has p = 0, and 0 * NaN would reach O.  Encoder shape (packed [T][3C] projection, 1370 keys) and decoder shapes."""
    g = _rng(dh + Lk)
    Cc = heads * dh
    pad = 130  # more than one 64-key tile of poison rows
    poison = torch.tensor([float("nan"), float("inf"), -float("inf"), 65504.0], device=DEV).to(torch.float16)
    if packed:
        buf = poison[torch.arange((Lk + pad) * 3 * Cc, device=DEV) % 4].view(1, Lk + pad, 3 * Cc).clone()
        buf[:, :Lk] = _bf(_t(g.standard_normal((1, Lk, 3 * Cc), dtype=np.float32)))
        Q = hh.prescale_q(_bf(_t(1.5 * g.standard_normal((1, Lq, Cc), dtype=np.float32))), dh)
        K, V = buf[:, :Lk, Cc:2 * Cc], buf[:, :Lk, 2 * Cc:]
    else:
        kb = poison[torch.arange((Lk + pad) * Cc, device=DEV) % 4].view(1, Lk + pad, Cc).clone()
        vb = kb.clone()
        kb[:, :Lk] = _bf(_t(1.5 * g.standard_normal((1, Lk, Cc), dtype=np.float32)))
        vb[:, :Lk] = _bf(_t(g.standard_normal((1, Lk, Cc), dtype=np.float32)))
        Q = hh.prescale_q(_bf(_t(1.5 * g.standard_normal((1, Lq, Cc), dtype=np.float32))), dh)
        K, V = kb[:, :Lk], vb[:, :Lk]
    O, lse = hh.attention(Q, K, V, heads, dh, lse=True, q_scale=1.0)
    ref, _, lse_ref = _attn_ref(Q, K.contiguous(), V.contiguous(), heads, dh)
    torch.cuda.synchronize()
    assert torch.isfinite(O.float()).all() and torch.isfinite(lse).all()
    err = (O.float() - ref).abs()
    assert err.max() < 4e-3 and err.mean() < 4e-4, (float(err.max()), float(err.mean()))
    assert (lse * math.log(2.0) - lse_ref).abs().max() < 5e-4


def test_attention_weights_one_head():
    dh, heads, Lq, Lk, B = 48, 8, 70, 210, 2
    g = _rng(8)
    Cc = heads * dh
    Q = hh.prescale_q(_t(1.5 * g.standard_normal((B, Lq, Cc), dtype=np.float32)), dh)
    K = _bf(_t(1.5 * g.standard_normal((B, Lk, Cc), dtype=np.float32)))
    V = _bf(_t(g.standard_normal((B, Lk, Cc), dtype=np.float32)))
    _, lse = hh.attention(Q, K, V, heads, dh, lse=True, q_scale=1.0)
    Pw = hh.attention_weights(Q, K, heads, dh, lse, head=3, q_scale=1.0)
    _, p_ref, _ = _attn_ref(Q, K, V, heads, dh)
    torch.cuda.synchronize()
    assert (Pw - p_ref[:, 3]).abs().max() < 1e-4
    assert (Pw.sum(-1) - 1).abs().max() < 1e-4


def test_attention_rejects_unsupported_head_dim():
    Q = torch.zeros((1, 8, 32), dtype=torch.float16, device=DEV)
    with pytest.raises(ValueError):
        hh.attention(Q, Q, Q, 1, 32)


# ------------------------------------------------------------------------------- linear + residual + LayerNorm in one launch
@pytest.mark.parametrize("M,with_resid", [(1369, True), (64, True), (65, False), (10952, True), (7, True)])
def test_linear_layernorm_one_launch(M, with_resid):
    """csrc/rowln.hip: LN(x + y W^T + b) as the post-norm decoder layer closes its sub-blocks (transformer.py:157-173; without x when
    decoder_do_short_cut is off), against fp32 torch on the same fp16 operands, and against the two-launch form of the forward (GEMM with the
    fp32 residual epilogue, then the LayerNorm kernel): same operands, fp32 accumulation in another order."""
    g = _rng(M + 3)
    Cc = 384
    A = _bf(_t(g.standard_normal((M, Cc), dtype=np.float32)))
    W = _bf(_t(g.standard_normal((Cc, Cc), dtype=np.float32) / math.sqrt(Cc)))
    b = _t(g.standard_normal((Cc,), dtype=np.float32))
    res = _t(2.0 * g.standard_normal((M, Cc), dtype=np.float32)) if with_resid else None
    if res is not None:
        res[:, 5] += 30.0  # an outlier channel
    gam = _t(1.0 + 0.3 * g.standard_normal((Cc,), dtype=np.float32))
    bet = _t(0.2 * g.standard_normal((Cc,), dtype=np.float32))
    of, oh = hh.linear_layernorm(A, W, b, res, gam, bet, 1e-5)
    pre = A.float() @ W.float().t() + b + (res if res is not None else 0.0)
    ref = torch.nn.functional.layer_norm(pre.double(), (Cc,), gam.double(), bet.double(), 1e-5).float()
    torch.cuda.synchronize()
    assert (of - ref).abs().max() < 2e-4, float((of - ref).abs().max())  # fp32 accumulation and statistics: order-of-summation noise only
    e16 = (oh.float() - ref).abs()
    assert (e16 <= 6e-4 * ref.abs() + 2.5e-4).all(), float((e16 - 6e-4 * ref.abs()).max())  # + one fp16 rounding (half an ulp = 4.9e-4 relative; the outlier channel reaches |12|)
    y = hh.gemm(A, W, b, _lib.EPI_RESID_F32, resid=res)
    of2, oh2 = hh.layernorm(y, gam, bet, 1e-5)
    torch.cuda.synchronize()
    assert (of - of2).abs().max() < 2e-4
    # in place on the residual stream, as the forward calls it
    if res is not None:
        r2 = res.clone()
        lib = _lib.load()
        _lib.check(lib.cs_op_linear_layernorm(hh._p(A), hh._p(W), hh._p(b), hh._p(r2), hh._p(gam), hh._p(bet), 1e-5, hh._p(r2), None, M, Cc, hh._stream()))
        torch.cuda.synchronize()
        assert torch.equal(r2, of)
    with pytest.raises(ValueError):
        hh.linear_layernorm(A[:, :256].contiguous(), W[:256, :256].contiguous(), b[:256], None, gam[:256], bet[:256], 1e-5)


@pytest.mark.parametrize("M,n2,act2", [(1369, 384, 0), (1369, 384, 1), (200, 384, 2), (1369, 1152, 0), (65, 1152, 0), (10952, 384, 1)])
def test_linear_layernorm_linear_one_launch(M, n2, act2):
    """The second stage of csrc/rowln.hip: the sub-block's NEXT linear in the same launch (the cross-attention's Q projection behind norm1,
    linear1 + ReLU behind norm2, the next layer's packed QKV projection / the head's first linear + LeakyReLU behind norm3;
    transformer.py:157-210, cross_reference.py:45-50).  Its input is the normalised rows rounded to fp16 -- exactly what the separate GEMM
    reads -- so the reference is built from the one-launch LayerNorm's own fp16 rows: fp32 torch on those and, as the forward's other form,
    the GEMM kernel with the matching epilogue."""
    g = _rng(M + n2 + act2)
    Cc = 384
    A = _bf(_t(g.standard_normal((M, Cc), dtype=np.float32)))
    W = _bf(_t(g.standard_normal((Cc, Cc), dtype=np.float32) / math.sqrt(Cc)))
    b = _t(g.standard_normal((Cc,), dtype=np.float32))
    res = _t(2.0 * g.standard_normal((M, Cc), dtype=np.float32))
    gam = _t(1.0 + 0.3 * g.standard_normal((Cc,), dtype=np.float32))
    bet = _t(0.2 * g.standard_normal((Cc,), dtype=np.float32))
    W2 = _bf(_t(g.standard_normal((n2, Cc), dtype=np.float32) / math.sqrt(Cc)))
    b2 = _t(0.5 * g.standard_normal((n2,), dtype=np.float32))
    of1, oh1 = hh.linear_layernorm(A, W, b, res, gam, bet, 1e-5)
    of, oh, o2 = hh.linear_layernorm_linear(A, W, b, res, gam, bet, 1e-5, W2, b2, act2, want_f32=True, want_f16=True)
    torch.cuda.synchronize()
    assert torch.equal(of, of1) and torch.equal(oh, oh1)  # the first stage is the same code
    pre = oh.float() @ W2.float().t() + b2
    ref = {0: pre, 1: torch.relu(pre), 2: torch.nn.functional.leaky_relu(pre, 0.01)}[act2]
    e = (o2.float() - ref).abs()
    assert (e <= 6e-4 * ref.abs() + 1e-4).all(), float((e - 6e-4 * ref.abs()).max())  # fp32 accumulation order + one fp16 rounding
    epi = {0: _lib.EPI_BIAS_F16, 1: _lib.EPI_BIAS_RELU_F16, 2: _lib.EPI_BIAS_LEAKY_F16}[act2]
    two = hh.gemm(oh, W2, b2, epi)
    torch.cuda.synchronize()
    d = (o2.float() - two.float()).abs()
    assert (d <= 1.1e-3 * ref.abs() + 1e-4).all(), float(d.max())  # at most one fp16 ulp apart
    assert float((d > 0).float().mean()) < 0.05                      # and almost everywhere identical
    # only the second output, written over the first stage's input rows (the forward's last sub-block: the head's hidden rows replace linear1's)
    if n2 == Cc:
        A2 = A.clone()
        _, _, o3 = hh.linear_layernorm_linear(A2, W, b, res, gam, bet, 1e-5, W2, b2, act2, want_f32=False, want_f16=False, out2=A2)
        torch.cuda.synchronize()
        assert o3 is A2 and torch.equal(A2, o2)
    with pytest.raises(ValueError):
        hh.linear_layernorm_linear(A, W, b, res, gam, bet, 1e-5, W2[:200].contiguous(), b2[:200], 0)
    with pytest.raises(ValueError):
        hh.linear_layernorm_linear(A, W, b, res, gam, bet, 1e-5, W2, b2, 3)


# ------------------------------------------------------------------------------- LayerNorm & position tables
@pytest.mark.parametrize("M,Cc,eps", [(1370, 384, 1e-6), (77, 768, 1e-5), (5, 128, 1e-6)])
def test_layernorm(M, Cc, eps):
    g = _rng(M)
    x = _t(3.0 * g.standard_normal((M, Cc), dtype=np.float32) + 1.0)
    gam = _t(1 + 0.2 * g.standard_normal((Cc,), dtype=np.float32))
    bet = _t(0.1 * g.standard_normal((Cc,), dtype=np.float32))
    of, ob = hh.layernorm(x, gam, bet, eps)
    ref = orc.layer_norm(x.cpu(), gam.cpu(), bet.cpu(), eps).to(DEV)
    torch.cuda.synchronize()
    assert (of - ref).abs().max() < 2e-5
    assert torch.equal(ob, _bf(of))


@pytest.mark.parametrize("G,gh,gw", [(5, 5, 6), (37, 74, 74), (37, 37, 49), (37, 20, 11)])
def test_pos_bicubic(G, gh, gw):
    Cc = 64
    g = _rng(G + gh)
    pos = _t(g.standard_normal((1 + G * G, Cc), dtype=np.float32))
    out = hh.pos_bicubic(pos, G, gh, gw)
    ref = torch.cat([pos[:1].cpu(), orc.bicubic_resize_grid(pos[1:].cpu().reshape(G, G, Cc), gh, gw).reshape(gh * gw, Cc)])
    torch.cuda.synchronize()
    assert (out.cpu() - ref).abs().max() < 3e-5  # fp32, different summation order of the 16 taps


@pytest.mark.parametrize("tag,backbone,h,w", [("tiny", "synthetic/dinov2-tiny", 5, 6), ("tiny", "synthetic/dinov2-tiny", 7, 4),
                                              ("small", "facebook/dinov2-small", 37, 49), ("small", "facebook/dinov2-small", 74, 74),
                                              ("small", "facebook/dinov2-small", 20, 31), ("base", "facebook/dinov2-base", 37, 49)])
def test_pos_bicubic_legacy_matches_torch_scale_factor_golden(tag, backbone, h, w):
    """The scale_factor convention of the reference's pinned transformers 4.33.3 (cs_config.pos_interp_legacy): the kernel's table against
    tests/golden/g6_pos_legacy.npz = torch's F.interpolate(scale_factor=((h + 0.1) / G, (w + 0.1) / G), bicubic) on the synthetic tables."""
    import os
    from crossscore_amd import synth
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g6_pos_legacy.npz"))
    arch = synth.BACKBONES[backbone]
    pos = _t(synth.make_state_dict(arch, int(g[f"table_{tag}_{h}x{w}_seed"]))["backbone.embeddings.position_embeddings"][0])
    out = hh.pos_bicubic(pos, arch.pos_grid, h, w, legacy=True).cpu().numpy()
    assert np.abs(out[g[f"table_{tag}_{h}x{w}_rows_idx"]] - g[f"table_{tag}_{h}x{w}_rows"]).max() < 3e-5
    assert np.abs(out.mean(axis=1, dtype=np.float64) - g[f"table_{tag}_{h}x{w}_chmean"]).max() < 3e-6
    out_size = hh.pos_bicubic(pos, arch.pos_grid, h, w, legacy=False).cpu().numpy()
    assert np.abs(out_size - out).max() > 1e-3  # the two conventions differ: the switch is live


@pytest.mark.parametrize("gh,gw", [(37, 37), (5, 6), (74, 74), (37, 49)])
def test_pe_bilinear(gh, gw):
    Cc = 64
    g = _rng(gh * gw)
    pe = _t(g.standard_normal((40, 40, Cc), dtype=np.float32))
    out = hh.pe_bilinear(pe, gh, gw)
    ref = orc.bilinear_resize_grid_align_corners(pe.cpu(), gh, gw).reshape(gh * gw, Cc)
    torch.cuda.synchronize()
    assert (out.cpu() - ref).abs().max() < 3e-5


@pytest.mark.parametrize("gh,gw", [(37, 37), (5, 6), (74, 74), (37, 49), (1, 7)])
def test_pe_bicubic_mode(gh, gw):
    """model.pos_enc.multi_view.interpolate_mode=bicubic (positional_encoding.py:61-69 with align_corners=True): kernel vs the oracle's
    restatement (itself pinned against torch's F.interpolate in tests/test_oracle_golden.py); mode 0 is the bilinear kernel."""
    g = _rng(gh * 50 + gw)
    pe = _t(g.standard_normal((40, 40, 96), dtype=np.float32))
    out = hh.pe_interp(pe, gh, gw, 1)
    ref = orc.multiview_pe({"pos_enc_fn.PE": pe.cpu()[None]}, gh, gw, "bicubic")
    torch.cuda.synchronize()
    assert (out.cpu() - ref).abs().max() < 3e-5  # fp32 source coordinates (dst * (39 / (g - 1))) and 16-tap order: measured 1.5e-5 on N(0,1) tables
    assert torch.equal(hh.pe_interp(pe, gh, gw, 0), hh.pe_bilinear(pe, gh, gw))
    with pytest.raises(ValueError):
        hh.pe_interp(pe, gh, gw, 2)


def test_streams_overlap_probe():
    """cs_op_streams_overlap: the probe behind the choice of lane / pipeline streams.  A stream never overlaps with work that must wait for
    it, so probing a stream against itself is refused; among a handful of fresh streams at least one pair overlaps (8 hardware queues)."""
    import ctypes as C

    lib = _lib.load()
    s = [torch.cuda.Stream() for _ in range(4)]
    flag = C.c_int(-1)
    assert lib.cs_op_streams_overlap(C.c_void_p(s[0].cuda_stream), C.c_void_p(s[0].cuda_stream), C.byref(flag)) != 0
    seen = []
    for i in range(1, 4):
        _lib.check(lib.cs_op_streams_overlap(C.c_void_p(s[0].cuda_stream), C.c_void_p(s[i].cuda_stream), C.byref(flag)))
        assert flag.value in (0, 1)
        seen.append(flag.value)
    assert any(seen), seen
