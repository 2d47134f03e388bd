"""torch-tensor wrappers over the single-op C-ABI entry points (tests only)."""
import ctypes as C

import torch

from crossscore_amd import _lib


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def gemm(A, W, bias=None, epi=_lib.EPI_BIAS_F16, resid=None, out=None, pos=None, Np=0, gw=0, P=0, act=0, powp=1.0,
         K=None, ldc=None, out_f16=None, stats_out=None, ln_part=None, col_s=None, ln_eps=1e-6):
    """A:(M,lda) fp16, W:(N,ldw) fp16 -> out (allocated here unless given)."""
    lib = _lib.load()
    M, lda = A.shape
    N, ldw = W.shape
    K = K or lda
    if out is None:
        dt = torch.float16 if (epi <= _lib.EPI_BIAS_LEAKY_F16 or epi in (_lib.EPI_LN_F16, _lib.EPI_LN_GELU_F16)) else torch.float32
        out = torch.zeros((M, N), dtype=dt, device=A.device)
    ldc = ldc or out.shape[-1]
    rc = lib.cs_op_gemm(_p(A), lda, _p(W), ldw, M, N, K, _p(bias), _p(resid), resid.shape[-1] if resid is not None else 0,
                        _p(out), ldc, epi, _p(pos), Np, gw, P, act, powp, _p(out_f16), _p(stats_out),
                        stats_out.shape[1] if stats_out is not None else 0, _p(ln_part), ln_part.shape[1] if ln_part is not None else 0,
                        _p(col_s), ln_eps, _stream())
    _lib.check(rc)
    return out


def head_score(A, W, bias, B, gh, gw, P, act=0, powp=1.0, want_mean=True, cnt=None):
    """The head's last linear + activation + jigsaw and the per-image mean from the same launch -> (score (B, gh P, gw P), mean (B,), counters)."""
    lib = _lib.load()
    M, lda = A.shape
    Np = gh * gw
    score = torch.full((B, gh * P, gw * P), 777.0, dtype=torch.float32, device=A.device)
    sp = 4 * ((P * P + 127) // 128 if (P * P) % 192 else (P * P) // 192)
    part = torch.full((M, sp), float("nan"), dtype=torch.float32, device=A.device) if want_mean else None
    if want_mean and cnt is None:
        cnt = torch.zeros((B,), dtype=torch.int32, device=A.device)
    mean = torch.full((B,), 777.0, dtype=torch.float32, device=A.device) if want_mean else None
    _lib.check(lib.cs_op_head_score(_p(A), lda, _p(W), W.shape[1], M, lda, _p(bias), _p(score), Np, gw, P, act, powp, _p(part),
                                    _p(cnt) if want_mean else None, _p(mean), _stream()))
    return score, mean, cnt


def ln_finalize(part, Cc, eps=1e-6):
    """(M, sp, 2) partial sums of the 256-tile GEMM's residual epilogue -> (ceil(M / 256) * 256, 1, 2) finalised (mean, rstd) rows."""
    lib = _lib.load()
    M, sp, _ = part.shape
    Mpad = (M + 255) // 256 * 256
    stat = torch.full((Mpad, 1, 2), 777.0, dtype=torch.float32, device=part.device)
    _lib.check(lib.cs_op_ln_finalize(_p(part), M, Mpad, sp, Cc, eps, _p(stat), _stream()))
    return stat


def prescale_q(Q, dh):
    """Q * log2(e)/sqrt(dh) in fp32, rounded to fp16 once: what the forward's Q projections emit (the factor is folded into their weights)."""
    return (Q.float() * (1.4426950408889634 / dh ** 0.5)).to(torch.float16)


def attention(Q, K, V, heads, dh, lse=False, q_scale=0.0):
    """Q:(B,Lq,heads*dh) K,V:(B,Lk,heads*dh) fp16 contiguous -> O (B,Lq,heads*dh) fp16 [, lse (B,heads,Lq), base 2].
    q_scale=1: Q is prescale_q(..) already; 0: raw Q, scaled (and re-rounded) inside the kernel."""
    lib = _lib.load()
    B, Lq, Cq = Q.shape
    Lk = K.shape[1]
    O = torch.zeros((B, Lq, heads * dh), dtype=torch.float16, device=Q.device)
    L = torch.zeros((B, heads, Lq), dtype=torch.float32, device=Q.device) if lse else None
    rc = lib.cs_op_attention(_p(Q), _p(K), _p(V), _p(O), Q.stride(1), K.stride(1), V.stride(1), O.stride(1), Q.stride(0), K.stride(0),
                             V.stride(0), O.stride(0), B, heads, Lq, Lk, dh, q_scale, _p(L), _stream())
    _lib.check(rc)
    return (O, L) if lse else O


def attention_weights(Q, K, heads, dh, lse, head, q_scale=0.0):
    lib = _lib.load()
    B, Lq, _ = Q.shape
    Lk = K.shape[1]
    out = torch.zeros((B, Lq, Lk), dtype=torch.float32, device=Q.device)
    rc = lib.cs_op_attention_weights(_p(Q), _p(K), Q.stride(1), K.stride(1), Q.stride(0), K.stride(0), B, heads, Lq, Lk, dh, q_scale,
                                     _p(lse), head, _p(out), _stream())
    _lib.check(rc)
    return out


def layernorm(x, g, b, eps, want_f32=True, want_f16=True):
    lib = _lib.load()
    M, Cc = x.shape
    of = torch.zeros_like(x) if want_f32 else None
    ob = torch.zeros((M, Cc), dtype=torch.float16, device=x.device) if want_f16 else None
    _lib.check(lib.cs_op_layernorm(_p(x), M, Cc, _p(g), _p(b), eps, _p(of), _p(ob), _stream()))
    return of, ob


def im2col(x, P, Kp):
    lib = _lib.load()
    I, _, H, W = x.shape
    out = torch.zeros((I * (H // P) * (W // P), Kp), dtype=torch.float16, device=x.device)
    _lib.check(lib.cs_op_im2col(_p(x), _p(out), I, H, W, P, Kp, _stream()))
    return out


def pos_bicubic(pos, G, gh, gw, legacy=None):
    lib = _lib.load()
    Cc = pos.shape[-1]
    out = torch.zeros((1 + gh * gw, Cc), dtype=torch.float32, device=pos.device)
    if legacy is None:
        _lib.check(lib.cs_op_pos_bicubic(_p(pos), G, Cc, gh, gw, _p(out), _stream()))
    else:
        _lib.check(lib.cs_op_pos_bicubic_ex(_p(pos), G, Cc, gh, gw, int(legacy), _p(out), _stream()))
    return out


def pe_bilinear(pe, gh, gw):
    lib = _lib.load()
    ph, pw, Cc = pe.shape
    out = torch.zeros((gh * gw, Cc), dtype=torch.float32, device=pe.device)
    _lib.check(lib.cs_op_pe_bilinear(_p(pe), ph, pw, Cc, gh, gw, _p(out), _stream()))
    return out


def linear_layernorm(A, W, bias, resid, gamma, beta, eps, want_f32=True, want_f16=True):
    """LN(resid + A W^T + bias) in one launch (csrc/rowln.hip): A (M,C) fp16, W (C,C) fp16 -> (out_f32, out_f16)"""
    lib = _lib.load()
    M, Cc = A.shape
    of = torch.zeros((M, Cc), dtype=torch.float32, device=A.device) if want_f32 else None
    oh = torch.zeros((M, Cc), dtype=torch.float16, device=A.device) if want_f16 else None
    _lib.check(lib.cs_op_linear_layernorm(_p(A), _p(W), _p(bias), _p(resid), _p(gamma), _p(beta), eps, _p(of), _p(oh), M, Cc, _stream()))
    return of, oh


def linear_layernorm_linear(A, W, bias, resid, gamma, beta, eps, W2, bias2, act2, want_f32=True, want_f16=False, out2=None):
    """the same with the sub-block's next linear in the launch: -> (out_f32, out_f16, out2 (M, n2) fp16); out2 may be A itself"""
    lib = _lib.load()
    M, Cc = A.shape
    n2 = W2.shape[0]
    of = torch.zeros((M, Cc), dtype=torch.float32, device=A.device) if want_f32 else None
    oh = torch.zeros((M, Cc), dtype=torch.float16, device=A.device) if want_f16 else None
    o2 = torch.zeros((M, n2), dtype=torch.float16, device=A.device) if out2 is None else out2
    _lib.check(lib.cs_op_linear_layernorm_linear(_p(A), _p(W), _p(bias), _p(resid), _p(gamma), _p(beta), eps, _p(of), _p(oh), _p(W2), _p(bias2),
                                                 n2, int(act2), _p(o2), M, Cc, _stream()))
    return of, oh, o2


def pe_interp(pe, gh, gw, mode):
    """mode 0 bilinear, 1 bicubic (align_corners=True): model.pos_enc.multi_view.interpolate_mode"""
    lib = _lib.load()
    ph, pw, Cc = pe.shape
    out = torch.zeros((gh * gw, Cc), dtype=torch.float32, device=pe.device)
    _lib.check(lib.cs_op_pe_interp(_p(pe), ph, pw, Cc, gh, gw, int(mode), _p(out), _stream()))
    return out


def pack_f16(w, ldo=None, row_scale=None, col_scale=None):
    lib = _lib.load()
    rows, K = w.shape
    ldo = ldo or K
    out = torch.zeros((rows, ldo), dtype=torch.float16, device=w.device)
    _lib.check(lib.cs_op_pack_f16(_p(w), rows, K, _p(out), ldo, _p(row_scale), _p(col_scale), _stream()))
    return out


def ln_fold_consts(w_packed, w, beta, bias):
    lib = _lib.load()
    N, K = w.shape
    s = torch.zeros(N, device=w.device)
    c = torch.zeros(N, device=w.device)
    _lib.check(lib.cs_op_ln_fold_consts(_p(w_packed), w_packed.shape[1], _p(w), _p(beta), _p(bias), N, K, _p(s), _p(c), _stream()))
    return s, c


def column_tiles(N):
    return _lib.load().cs_gemm_column_tiles(N)


def panel_pack(wo, ls1, w1, g2, w2, ls2):
    """fp32 weights -> the unit stream of the encoder token-panel kernel (uint8 tensor of cs_panel_image_bytes)."""
    lib = _lib.load()
    n = lib.cs_panel_image_bytes(1 if wo is not None else 0)
    img = torch.zeros(n, dtype=torch.uint8, device=w1.device)
    _lib.check(lib.cs_op_panel_pack(_p(wo), _p(ls1), _p(w1), _p(g2), _p(w2), _p(ls2), _p(img), _stream()))
    return img


def encoder_panel(x, attn_o, img, bo, b1, b2, want_u=True, eps=1e-6):
    """In place on x (M,384) fp32; returns u (M,384) fp16 or None."""
    lib = _lib.load()
    M = x.shape[0]
    u = torch.zeros((M, x.shape[1]), dtype=torch.float16, device=x.device) if want_u else None
    _lib.check(lib.cs_op_encoder_panel(_p(x), _p(attn_o), _p(img), _p(bo), _p(b1), _p(b2), _p(u), M, eps, _stream()))
    return u


def patch_embed_fused(x, w, bias, pos, P):
    """one-launch form (csrc/patch.hip): same contract as patch_embed(centred=True)"""
    lib = _lib.load()
    I, _, H, W = x.shape
    C_ = w.shape[0]
    Np = (H // P) * (W // P)
    out = torch.full((I * (1 + Np), C_), 7.0, dtype=torch.float32, device=x.device)
    _lib.check(lib.cs_op_patch_embed_fused(_p(x), _p(w), _p(bias), _p(pos), I, H, W, P, C_, _p(out), _stream()))
    return out


def patch_embed_fused_u8(imgs_u8, rs, crop, mean, std, w, bias, pos, P):
    """imgs_u8 (I, h, row_bytes) uint8 device, rows of w*3 bytes (+ padding) -> token rows as patch_embed_fused of the input stage's output"""
    import ctypes as C
    lib = _lib.load()
    I, h, row = imgs_u8.shape
    y0, x0, H, W, in_w = crop
    C_ = w.shape[0]
    Np = (H // P) * (W // P)
    out = torch.full((I * (1 + Np), C_), 7.0, dtype=torch.float32, device=imgs_u8.device)
    _lib.check(lib.cs_op_patch_embed_fused_u8(_p(imgs_u8), I, h, in_w, row, rs[0], rs[1], y0, x0, H, W, (C.c_float * 3)(*mean), (C.c_float * 3)(*std),
                                              _p(w), _p(bias), _p(pos), P, C_, _p(out), _stream()))
    return out


def patch_embed(x, w, bias, pos, P, centred):
    """(I,3,H,W) images -> (I * (1 + Np), C) fp32 token rows (patch rows written, CLS rows left at 7.0)."""
    lib = _lib.load()
    I, _, H, W = x.shape
    C_ = w.shape[0]
    Np = (H // P) * (W // P)
    out = torch.full((I * (1 + Np), C_), 7.0, dtype=torch.float32, device=x.device)
    _lib.check(lib.cs_op_patch_embed(_p(x), _p(w), _p(bias), _p(pos), I, H, W, P, C_, int(centred), _p(out), _stream()))
    return out
