"""TEST INFRASTRUCTURE ONLY -- CPU fp32 restatement of the CrossScore inference hot path.

This file is the parity oracle for the HIP path.  Only tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg may import it; the product package `crossscore_amd` never does.

It restates, in plain tensor arithmetic (matmul / exp / erf / elementwise, no nn.Module, no
nn.MultiheadAttention, no transformers), what the reference executes for
`CrossScoreNet.forward` (/root/reference/task/core.py:58-117).  The arithmetic of the reference lives in
two un-vendored dependencies: `transformers` (pinned 4.33.3 in environment.yaml:340; the installed,
runnable one is 5.15.0 -> "HF" below = transformers/models/dinov2/modeling_dinov2.py of 5.15.0) and
`torch.nn` (pinned 2.1.2; installed 2.10.0 -> "TORCH" = torch/nn/functional.py).  Known drift: the
encoder pos-embed interpolation API (scale_factor in 4.33 vs size= in 5.x); identical at 518x518.

Parity pinning: the reference ships no tests or golden vectors, so this oracle is
pinned against outputs of the reference itself imported in the build container
(tests/golden/make_golden.py -> tests/golden/*.npz, checked by tests/test_oracle_golden.py).

Weights are a flat dict keyed exactly like the checkpoint state_dict (without the "model." prefix).
"""
from __future__ import annotations

import math
from typing import Callable, Dict, Optional

import torch

Tensor = torch.Tensor


def bicubic_resize_grid_align_corners(grid: Tensor, oh: int, ow: int) -> Tensor:
    """F.interpolate(mode='bicubic', align_corners=True) of a (gh,gw,C) grid: what model/positional_encoding.py:61-69 executes with
    interpolate_mode=bicubic.  src = dst*(in-1)/(out-1) per axis (aten area_pixel_compute_source_index, align_corners branch), four taps
    with border-clamped indices, A = -0.75, x first then y (aten upsample_bicubic2d).  Pinned against torch in tests/test_oracle_golden.py."""
    gh, gw, C = grid.shape

    def axis(n_in: int, n_out: int):
        scale = (n_in - 1) / (n_out - 1) if n_out > 1 else 0.0
        dst = torch.arange(n_out, dtype=torch.float32)
        src = dst * torch.tensor(scale, dtype=torch.float32)
        i0 = torch.floor(src)
        t = src - i0
        i0 = i0.to(torch.int64)
        w = _cubic_coeffs(t)
        idx = [torch.clamp(i0 + d, 0, n_in - 1).to(grid.device) for d in (-1, 0, 1, 2)]
        return idx, [c.to(grid.device) for c in w]

    iy, wy = axis(gh, oh)
    ix, wx = axis(gw, ow)
    out = torch.zeros(oh, ow, C, dtype=grid.dtype, device=grid.device)
    for a in range(4):
        rows = grid[iy[a]]
        acc = torch.zeros(oh, ow, C, dtype=grid.dtype, device=grid.device)
        for b in range(4):
            acc = acc + rows[:, ix[b]] * wx[b][None, :, None]
        out = out + acc * wy[a][:, None, None]
    return out


# --------------------------------------------------------------------------------------------------
# primitive restatements
# --------------------------------------------------------------------------------------------------
def layer_norm(x: Tensor, g: Tensor, b: Tensor, eps: float) -> Tensor:
    """Biased-variance LayerNorm over the last dim (TORCH nn.LayerNorm; eps 1e-6 in DINOv2
    HF:344,348,447, eps 1e-5 in the decoder transformer.py:60)."""
    mu = x.mean(dim=-1, keepdim=True)
    xc = x - mu
    var = (xc * xc).mean(dim=-1, keepdim=True)
    return xc / torch.sqrt(var + eps) * g + b


def linear(x: Tensor, w: Tensor, b: Optional[Tensor], rnd: Callable[[Tensor], Tensor]) -> Tensor:
    """y = x W^T + b with nn.Linear layout W:(out,in)."""
    y = rnd(x) @ rnd(w).t()
    return y if b is None else y + b


def gelu_erf(x: Tensor) -> Tensor:
    """Exact GELU (HF ACT2FN['gelu'], used at HF:293-297)."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def softmax_lastdim(s: Tensor) -> Tensor:
    m = s.max(dim=-1, keepdim=True).values
    e = torch.exp(s - m)
    return e / e.sum(dim=-1, keepdim=True)


def attention(q: Tensor, k: Tensor, v: Tensor, heads: int, rnd, need_weights: bool = False):
    """Multi-head softmax(QK^T/sqrt(dh))V; head i owns channels [i*dh,(i+1)*dh).
    q:(B,Lq,C) k,v:(B,Lk,C).  HF:153-173 (encoder) / TORCH functional.py:6576-6642 (decoder)."""
    B, Lq, C = q.shape
    Lk = k.shape[1]
    dh = C // heads
    qh = q.view(B, Lq, heads, dh).transpose(1, 2)  # (B,h,Lq,dh)
    kh = k.view(B, Lk, heads, dh).transpose(1, 2)
    vh = v.view(B, Lk, heads, dh).transpose(1, 2)
    s = (rnd(qh) @ rnd(kh).transpose(-1, -2)) * (1.0 / math.sqrt(dh))
    p = softmax_lastdim(s)
    o = rnd(p) @ rnd(vh)  # (B,h,Lq,dh)
    o = o.transpose(1, 2).reshape(B, Lq, C)
    return o, (p if need_weights else None)


def _cubic_coeffs(t: Tensor, A: float = -0.75):
    """PyTorch cubic convolution coefficients (aten UpSample.h get_cubic_upsample_coefficients)."""
    def c1(x):  # |x| <= 1
        return ((A + 2.0) * x - (A + 3.0)) * x * x + 1.0

    def c2(x):  # 1 < |x| < 2
        return ((A * x - 5.0 * A) * x + 8.0 * A) * x - 4.0 * A

    return c2(t + 1.0), c1(t), c1(1.0 - t), c2(2.0 - t)


def bicubic_resize_grid(grid: Tensor, oh: int, ow: int, grow: float = 0.0) -> Tensor:
    """F.interpolate(size=(oh,ow), mode='bicubic', align_corners=False) of a (gh,gw,C) grid, as used by
    Dinov2Embeddings.interpolate_pos_encoding (HF:57-95).  src = (dst+0.5)*in/out-0.5 (not clamped for
    cubic), 4 taps with border-clamped indices, separable."""
    gh, gw, C = grid.shape

    def axis(n_in: int, n_out: int):
        scale = n_in / (n_out + grow)  # grow = 0.1: the scale_factor form of transformers 4.33.3 (see encoder_pos_embed)
        dst = torch.arange(n_out, dtype=torch.float32)
        src = (dst + 0.5) * scale - 0.5
        i0 = torch.floor(src)
        t = src - i0
        i0 = i0.to(torch.int64)
        w = _cubic_coeffs(t)
        idx = [torch.clamp(i0 + d, 0, n_in - 1).to(grid.device) for d in (-1, 0, 1, 2)]
        return idx, [c.to(grid.device) for c in w]

    iy, wy = axis(gh, oh)
    ix, wx = axis(gw, ow)
    # interpolate along x for each needed source row, then along y (same order as aten: for each y tap,
    # x-interpolate, then combine)
    out = torch.zeros(oh, ow, C, dtype=grid.dtype, device=grid.device)
    for a in range(4):
        rows = grid[iy[a]]  # (oh,gw,C)
        acc = torch.zeros(oh, ow, C, dtype=grid.dtype, device=grid.device)
        for b in range(4):
            acc = acc + rows[:, ix[b]] * wx[b][None, :, None]
        out = out + acc * wy[a][:, None, None]
    return out


def bilinear_resize_grid_align_corners(grid: Tensor, oh: int, ow: int) -> Tensor:
    """F.interpolate(mode='bilinear', align_corners=True) of a (gh,gw,C) grid
    (model/positional_encoding.py:61-69): src = dst*(in-1)/(out-1) per axis."""
    gh, gw, C = grid.shape

    def axis(n_in: int, n_out: int):
        scale = (n_in - 1) / (n_out - 1) if n_out > 1 else 0.0
        dst = torch.arange(n_out, dtype=torch.float32)
        src = dst * torch.tensor(scale, dtype=torch.float32)
        i0 = src.to(torch.int64)
        i1 = torch.clamp(i0 + 1, max=n_in - 1)
        l1 = src - i0.to(torch.float32)
        d = grid.device
        return i0.to(d), i1.to(d), (1.0 - l1).to(d), l1.to(d)

    y0, y1, wy0, wy1 = axis(gh, oh)
    x0, x1, wx0, wx1 = axis(gw, ow)
    top = grid[y0][:, x0] * wx0[None, :, None] + grid[y0][:, x1] * wx1[None, :, None]
    bot = grid[y1][:, x0] * wx0[None, :, None] + grid[y1][:, x1] * wx1[None, :, None]
    return top * wy0[:, None, None] + bot * wy1[:, None, None]


# --------------------------------------------------------------------------------------------------
# model stages
# --------------------------------------------------------------------------------------------------
def patch_embed(imgs: Tensor, Wt: Dict[str, Tensor], P: int, rnd) -> Tensor:
    """Conv2d(3,C,k=P,s=P) as an im2col GEMM with K order (channel,dy,dx); trailing H%P / W%P pixels are
    dropped (HF:139-149).  imgs:(I,3,H,W) -> (I,h*w,C), token p = i*w + j."""
    I, ch, H, W = imgs.shape
    h, w = H // P, W // P
    x = imgs[:, :, : h * P, : w * P].reshape(I, ch, h, P, w, P)
    x = x.permute(0, 2, 4, 1, 3, 5).reshape(I, h * w, ch * P * P)
    wmat = Wt["backbone.embeddings.patch_embeddings.projection.weight"].reshape(-1, ch * P * P)
    return linear(x, wmat, Wt["backbone.embeddings.patch_embeddings.projection.bias"], rnd)


def encoder_pos_embed(Wt: Dict[str, Tensor], h: int, w: int, H: int, W: int, legacy: bool = False) -> Tensor:
    """(1+h*w, C) position table: parameter as is when grid matches and H==W, else bicubic (HF:57-95).
    legacy=True: the interpolation call of the reference's PINNED transformers 4.33.3 (environment.yaml:340), which passed
    scale_factor=((h + 0.1) / G, (w + 0.1) / G) -- torch then maps dst -> (dst + 0.5) * G / (h + 0.1) - 0.5 -- instead of size=(h, w).
    That release is not installable offline: the branch restates its published call, and is pinned by tests/golden/g6_pos_legacy.npz -- the
    same F.interpolate(scale_factor=...) call executed by the installed torch on the synthetic tables, and the imported reference model run end
    to end with its interpolate_pos_encoding replaced by that call (tests/golden/make_golden.py --only g6; tests/test_oracle_golden.py)."""
    pos = Wt["backbone.embeddings.position_embeddings"][0]  # (1+G*G, C)
    G2 = pos.shape[0] - 1
    if h * w == G2 and H == W:
        return pos
    G = int(round(math.sqrt(G2)))
    grid = pos[1:].reshape(G, G, -1)
    res = bicubic_resize_grid(grid, h, w, 0.1 if legacy else 0.0).reshape(h * w, -1)
    return torch.cat([pos[:1], res], dim=0)


def dinov2_encoder(imgs: Tensor, Wt: Dict[str, Tensor], enc_heads: int, P: int, rnd, taps=None, pos_legacy: bool = False) -> Tensor:
    """Dinov2Model.forward -> last_hidden_state (I,1+h*w,C).  HF:97-116 embeddings, HF:361-380 layers
    (pre-LN, LayerScale, exact GELU), HF:465-470 final LayerNorm."""
    I, _, H, W = imgs.shape
    h, w = H // P, W // P
    x = patch_embed(imgs, Wt, P, rnd)
    if taps is not None:
        taps["patch_embed"] = x.clone()
    cls = Wt["backbone.embeddings.cls_token"].expand(I, -1, -1)
    x = torch.cat([cls, x], dim=1) + encoder_pos_embed(Wt, h, w, H, W, pos_legacy)[None]
    if taps is not None:
        taps["embeddings"] = x.clone()
    l = 0
    while f"backbone.encoder.layer.{l}.norm1.weight" in Wt:
        p = f"backbone.encoder.layer.{l}."
        u = layer_norm(x, Wt[p + "norm1.weight"], Wt[p + "norm1.bias"], 1e-6)
        q = linear(u, Wt[p + "attention.attention.query.weight"], Wt[p + "attention.attention.query.bias"], rnd)
        k = linear(u, Wt[p + "attention.attention.key.weight"], Wt[p + "attention.attention.key.bias"], rnd)
        v = linear(u, Wt[p + "attention.attention.value.weight"], Wt[p + "attention.attention.value.bias"], rnd)
        a, _ = attention(q, k, v, enc_heads, rnd)
        a = linear(a, Wt[p + "attention.output.dense.weight"], Wt[p + "attention.output.dense.bias"], rnd)
        x = x + a * Wt[p + "layer_scale1.lambda1"]
        u = layer_norm(x, Wt[p + "norm2.weight"], Wt[p + "norm2.bias"], 1e-6)
        if p + "mlp.weights_in.weight" in Wt:
            # Dinov2SwiGLUFFN (HF modeling_dinov2.py:300-316, use_swiglu_ffn: facebook/dinov2-giant): x1, x2 = weights_in(u).chunk(2, -1);
            # weights_out(silu(x1) * x2)
            x12 = linear(u, Wt[p + "mlp.weights_in.weight"], Wt[p + "mlp.weights_in.bias"], rnd)
            x1, x2 = x12.chunk(2, dim=-1)
            m = x1 * torch.sigmoid(x1) * x2
            if taps is not None:
                taps[f"enc_mlp_hidden_absmax_{l}"] = m.abs().amax(dim=(0, 1))
            m = linear(m, Wt[p + "mlp.weights_out.weight"], Wt[p + "mlp.weights_out.bias"], rnd)
        else:
            m = gelu_erf(linear(u, Wt[p + "mlp.fc1.weight"], Wt[p + "mlp.fc1.bias"], rnd))
            if taps is not None:
                taps[f"enc_mlp_hidden_absmax_{l}"] = m.abs().amax(dim=(0, 1))  # per hidden unit (range tests of the 16-bit operand types)
            m = linear(m, Wt[p + "mlp.fc2.weight"], Wt[p + "mlp.fc2.bias"], rnd)
        x = x + m * Wt[p + "layer_scale2.lambda1"]
        if taps is not None:
            taps[f"enc_layer_{l}"] = x.clone()
        l += 1
    return layer_norm(x, Wt["backbone.layernorm.weight"], Wt["backbone.layernorm.bias"], 1e-6)


def multiview_pe(Wt: Dict[str, Tensor], h: int, w: int, mode: str = "bilinear") -> Tensor:
    """(h*w, C) grid added to every view (model/positional_encoding.py:42-75); mode = cfg.model.pos_enc.multi_view.interpolate_mode."""
    PE = Wt["pos_enc_fn.PE"][0]  # (pe_h,pe_w,C)
    if PE.shape[0] == h and PE.shape[1] == w:
        return PE.reshape(h * w, -1)
    if mode == "bicubic":
        return bicubic_resize_grid_align_corners(PE, h, w).reshape(h * w, -1)
    if mode != "bilinear":
        raise ValueError("align_corners option can only be set with the interpolating modes: linear | bilinear | bicubic | trilinear")
    return bilinear_resize_grid_align_corners(PE, h, w).reshape(h * w, -1)


def mha_block(x: Tensor, mem: Tensor, Wt, prefix: str, heads: int, rnd, need_weights: bool):
    """nn.MultiheadAttention(x, mem, mem) slow path: packed in-projection rows [0:C)=Wq,[C:2C)=Wk,[2C:3C)=Wv
    (TORCH functional.py:5785-5860), per-head softmax attention, out_proj."""
    C = x.shape[-1]
    w_in, b_in = Wt[prefix + ".in_proj_weight"], Wt[prefix + ".in_proj_bias"]
    q = linear(x, w_in[:C], b_in[:C], rnd)
    k = linear(mem, w_in[C : 2 * C], b_in[C : 2 * C], rnd)
    v = linear(mem, w_in[2 * C :], b_in[2 * C :], rnd)
    o, p = attention(q, k, v, heads, rnd, need_weights)
    return linear(o, Wt[prefix + ".out_proj.weight"], Wt[prefix + ".out_proj.bias"], rnd), p


def decoder(tgt: Tensor, mem: Tensor, Wt, cfg: dict, rnd, need_weights: bool, head_id: int, taps=None):
    """TransformerDecoderCustomised (2 layers, same memory, no final norm; transformer.py:213-268) of
    post-norm TransformerDecoderLayerCustomised (transformer.py:157-173), ReLU FFN with dff = C."""
    x = tgt
    heads = cfg.get("dec_heads", 8)
    w_last = None
    l = 0
    while f"ref_cross.attn.layers.{l}.norm1.weight" in Wt:
        p = f"ref_cross.attn.layers.{l}."
        if cfg.get("do_self_attn", True):
            sa, _ = mha_block(x, x, Wt, p + "self_attn", heads, rnd, False)
            x = layer_norm(x + sa if cfg.get("do_short_cut", True) else sa,
                           Wt[p + "norm1.weight"], Wt[p + "norm1.bias"], 1e-5)
        if taps is not None:
            taps[f"dec{l}_after_sa"] = x.clone()
        ca, pw = mha_block(x, mem, Wt, p + "multihead_attn", heads, rnd, need_weights)
        x = layer_norm(x + ca if cfg.get("do_short_cut", True) else ca,
                       Wt[p + "norm2.weight"], Wt[p + "norm2.bias"], 1e-5)
        if taps is not None:
            taps[f"dec{l}_after_ca"] = x.clone()
        ff = linear(torch.relu(linear(x, Wt[p + "linear1.weight"], Wt[p + "linear1.bias"], rnd)),
                    Wt[p + "linear2.weight"], Wt[p + "linear2.bias"], rnd)
        x = layer_norm(x + ff, Wt[p + "norm3.weight"], Wt[p + "norm3.bias"], 1e-5)
        if taps is not None:
            taps[f"dec{l}_out"] = x.clone()
        if pw is not None:
            w_last = pw[:, head_id]
        l += 1
    return x, w_last


def check_metric_prediction_config(metric_type, metric_min, metric_max):
    """utils/check_config.py:1-28."""
    if metric_type not in ("ssim", "mse", "mae"):
        raise ValueError(f"Invalid metric type {metric_type}")
    valid_max = metric_max == 1
    valid_min = (metric_min in (-1, 0)) if metric_type == "ssim" else (metric_min == 0)
    if not (valid_min and valid_max):
        raise ValueError(f"Invalid metric range {metric_min} to {metric_max} for {metric_type}")


def regression_power(metric_type: str, metric_min, power_factor) -> float:
    """model/regression_layer.py:40-62."""
    if metric_min == 0:
        p = {"ssim": 1, "mae": 2, "mse": 4}[metric_type] if power_factor == "default" else power_factor
    else:
        p = 1
    return float(p)


def regression_layer(x: Tensor, metric_type: str, metric_min, metric_max, power_factor) -> Tensor:
    """model/regression_layer.py:26-62: tanh (min=-1) or sigmoid (min=0), then x**p."""
    check_metric_prediction_config(metric_type, metric_min, metric_max)
    if metric_min == -1:
        y = torch.tanh(x)
    elif metric_min == 0:
        y = 1.0 / (1.0 + torch.exp(-x))
    else:
        raise ValueError(f"metric_min={metric_min} not supported")
    p = regression_power(metric_type, metric_min, power_factor)
    return y if p == 1.0 else torch.pow(y, p)


def jigsaw_to_image(x: Tensor, h: int, w: int) -> Tensor:
    """out[b,P*i+py,P*j+px] = x[b,i*w+j,py,px]  (utils/misc/image.py:8-21)."""
    B, n, ph, pw = x.shape
    assert n == h * w
    return x.view(B, h, w, ph, pw).permute(0, 1, 3, 2, 4).reshape(B, h * ph, w * pw)


DEFAULT_CFG = dict(
    patch=14, enc_heads=6, dec_heads=8, do_self_attn=True, do_short_cut=True,
    metric_type="ssim", metric_min=0, metric_max=1, power_factor="default",
)


def forward(Wt: Dict[str, Tensor], cfg: dict, query_img: Tensor, ref_cross_imgs: Tensor,
            need_attn_weights: bool = False, need_attn_weights_head_id: int = 0,
            emulate_bf16=False, taps: Optional[dict] = None) -> Dict[str, Optional[Tensor]]:
    """CrossScoreNet.forward (task/core.py:58-117) with norm_img=False.

    emulate_bf16=True rounds every matmul operand to bf16 (fp32 accumulate) -- the precision policy of the
    HIP path -- and is used only to budget the tolerance; parity targets are always emulate_bf16=False.
    """
    c = dict(DEFAULT_CFG)
    c.update(cfg)
    if emulate_bf16 == "f16":  # budget of an fp16-operand forward (same rule: operands rounded, fp32 accumulate)
        rnd = lambda t: t.to(torch.float16).to(torch.float32)  # noqa: E731
    else:
        rnd = (lambda t: t.to(torch.bfloat16).to(torch.float32)) if emulate_bf16 else (lambda t: t)
    P = c["patch"]
    B, _, H, W = query_img.shape
    N = ref_cross_imgs.shape[1]
    h, w = H // P, W // P
    # get_featmaps: core.py:119-161
    all_imgs = torch.cat([query_img.view(B, 1, 3, H, W), ref_cross_imgs], dim=1).view(B * (1 + N), 3, H, W)
    hs = dinov2_encoder(all_imgs, Wt, c["enc_heads"], P, rnd, taps, bool(c.get("pos_interp_legacy", False)))
    if taps is not None:
        taps["last_hidden_state"] = hs.clone()
    fm = hs[:, 1:].reshape(B, 1 + N, h * w, -1)
    pe = multiview_pe(Wt, h, w, c.get("pe_interpolate_mode", "bilinear"))
    fq = fm[:, 0] + pe[None]  # core.py:87
    fr = (fm[:, 1:] + pe[None, None]).reshape(B, N * h * w, -1)  # core.py:93-98
    if taps is not None:
        taps["featmap_query"] = fq.clone()
        taps["featmap_ref"] = fr.clone()
    x, attn = decoder(fq, fr, Wt, c, rnd, need_attn_weights, need_attn_weights_head_id, taps)
    # head: cross_reference.py:45-50,82
    y = linear(x, Wt["ref_cross.head.0.weight"], Wt["ref_cross.head.0.bias"], rnd)
    y = torch.where(y >= 0, y, 0.01 * y)
    y = linear(y, Wt["ref_cross.head.2.weight"], Wt["ref_cross.head.2.bias"], rnd)
    if taps is not None:
        taps["head_pre_activation"] = y.clone()
    y = regression_layer(y, c["metric_type"], c["metric_min"], c["metric_max"], c["power_factor"])
    score = jigsaw_to_image(y.view(B, h * w, P, P), h, w)
    if attn is not None:
        attn = attn.reshape(B, h, w, N, h, w)  # cross_reference.py:91-93
    return {"score_map_ref_cross": score, "attn_weights_map_ref_cross": attn}


def to_torch(sd: Dict[str, "object"]) -> Dict[str, Tensor]:
    import numpy as np
    return {k: (torch.from_numpy(np.ascontiguousarray(v)) if not isinstance(v, torch.Tensor) else v).float()
            for k, v in sd.items()}
