"""End-to-end parity of the HIP forward (CrossScoreNet -> C ABI -> gfx950 kernels) against the fp32 oracle and
the committed golden vectors generated from the reference.  Tolerance (BASELINE.json north_star): score-map
MAE < 1e-3 vs the fp32 reference; SURVEY.md 8c's target for an fp32-output path is 2e-4, which the fp16-operand MFMA path
(fp32 accumulation / softmax / LayerNorm / residual stream / output) meets with margin (measured 0.9e-4 - 1.2e-4)."""
import dataclasses
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from crossscore_amd import synth  # noqa: E402
from crossscore_amd.config import model_config  # noqa: E402
from crossscore_amd.model import CrossScoreNet  # noqa: E402
from oracle import crossscore_oracle as orc  # noqa: E402

MAE_TOL = 1e-3     # the north-star bound
MAE_TARGET = 2e-4  # SURVEY.md 8c target, asserted on the reference's own goldens and the real backbones
MAX_TOL = 5e-3     # single-pixel worst case of an fp16-operand forward on peaky synthetic weights (measured 6e-4 - 8e-4)
TINY = "synthetic/dinov2-tiny"
# optional attention-probability output (a16): the kernel recomputes one head's logits from the fp16 q / k operands with the fused kernel's
# fp32 log-sum-exp.  Measured on MI355X (r3): tiny net g0 head 3 max |dP| 1.02e-3 (peaky rows of a 60-key softmax); ViT-S 1369 x 2738
# max |dP| 8.4e-5, max per-row L1 1.28e-3.  The bounds are 3 x the measured values (they were 2e-2 / 0.05).
AW_MAX_TOL_TINY = 3.1e-3
AW_MAX_TOL = 2.6e-4
AW_L1_TOL = 4.0e-3


def _net(backbone, seed, **over):
    cfg = model_config(**{"backbone.from_pretrained": backbone, **over})
    net = CrossScoreNet(cfg)
    arch = net.arch
    sd = synth.make_state_dict(arch, seed)
    net.load_numpy_state_dict(sd)
    return net.cuda(), arch, sd


def _oracle(arch, sd, q, r, need_w=False, head=0, **cfgover):
    W = orc.to_torch(sd)
    return orc.forward(W, dict(enc_heads=arch.enc_heads, **cfgover), torch.from_numpy(q), torch.from_numpy(r), need_w, head)


def _compare(score, ref):
    d = (score.cpu() - ref).abs()
    return float(d.mean()), float(d.max())


@pytest.mark.parametrize("tag,backbone,dtype", [("g9_swiglu_base_width", "synthetic/dinov2-swiglu-2l", "fp16"),
                                               ("g9_swiglu_base_width", "synthetic/dinov2-swiglu-2l", "bf16"),
                                               ("g9_swiglu_giant_width", "synthetic/dinov2-giant-2l", "fp16")])  # hidden 1536, decoder heads of 192
def test_swiglu_backbone_vs_reference_golden(golden_dir, tag, backbone, dtype):
    """The SwiGLU MLP of facebook/dinov2-giant (HF Dinov2SwiGLUFFN, Dinov2Config.use_swiglu_ffn; task/core.py:39-40 takes any from_pretrained):
    weights_in GEMM, the gate as one elementwise launch, weights_out GEMM with the residual epilogue -- against the imported reference's score
    map and last encoder layer (g9), and the launch census says which kernels ran."""
    g = np.load(os.path.join(golden_dir, tag + ".npz"))
    net, arch, sd = _net(backbone, int(g["seed"]))
    assert arch.swiglu
    net.operand_dtype = dtype
    net.debug_capture(True)
    q, r = synth.make_inputs(int(g["B"]), int(g["N"]), int(g["H"]), int(g["W"]), int(g["seed"]))
    out = net(torch.from_numpy(q).cuda(), torch.from_numpy(r).cuda(), False, 0, False)
    torch.cuda.synchronize()
    mae, mx = _compare(out["score_map_ref_cross"], torch.from_numpy(g["score"]))
    last = net.debug_read(f"enc_layer_{arch.enc_layers - 1}")[0].float().cpu().numpy()
    rel = float(np.abs(last - g["enc_last_img0"]).mean() / np.sqrt((g["enc_last_img0"] ** 2).mean()))
    k = net.forward_stats()["kernels"]
    print(f"{tag} ({dtype}): score MAE {mae:.2e} max {mx:.2e}; last encoder layer mean |d| / RMS {rel:.2e}; {k}")
    bound = 8.0 if dtype == "bf16" else 1.0
    assert mae < MAE_TARGET * bound and mx < MAX_TOL * bound and rel < 1.5e-3 * bound
    assert k.get("silu_mul", 0) == 2 * arch.enc_layers and "panel" not in k and "ln_stats" not in k  # (two encoder chunks x two layers)


def test_tiny_nonsquare_vs_oracle_and_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "g0_tiny_all.npz"))
    net, arch, sd = _net(TINY, int(g["seed"]))
    q, r = synth.make_inputs(2, 2, 75, 90, int(g["seed"]))
    out = net(torch.from_numpy(q).cuda(), torch.from_numpy(r).cuda(), True, 3, False)
    torch.cuda.synchronize()
    assert out["score_map_ref_cross"].shape == (2, 70, 84)
    mae, mx = _compare(out["score_map_ref_cross"], torch.from_numpy(g["score"]))
    assert mae < MAE_TOL and mx < MAX_TOL, (mae, mx)
    aw = out["attn_weights_map_ref_cross"]
    assert aw.shape == (2, 5, 6, 2, 5, 6)
    aw_err = float((aw.cpu() - torch.from_numpy(g["attn_head3"])).abs().max())
    print(f"g0 head-3 attention probabilities: max abs error {aw_err:.2e}")
    assert aw_err < AW_MAX_TOL_TINY  # probabilities in [0,1]
    assert (aw.reshape(2, 30, -1).sum(-1) - 1).abs().max() < 1e-4
    # need_attn_weights must not change the score map (same kernels, extra output only)
    out2 = net(torch.from_numpy(q).cuda(), torch.from_numpy(r).cuda(), False, 0, False)
    assert out2["attn_weights_map_ref_cross"] is None
    assert torch.equal(out2["score_map_ref_cross"], out["score_map_ref_cross"])


@pytest.mark.parametrize("name,over,cfgover", [
    ("no_self_attn", {"decoder_do_self_attn": False}, dict(do_self_attn=False)),
    ("no_short_cut", {"decoder_do_short_cut": False}, dict(do_short_cut=False)),
    ("tanh", {"predict.metric.min": -1}, dict(metric_min=-1)),
    ("mae_pow2", {"predict.metric.type": "mae"}, dict(metric_type="mae")),
    ("mse_pow4", {"predict.metric.type": "mse"}, dict(metric_type="mse")),
    ("scalar_p", {"predict.metric.power_factor": 0.5}, dict(power_factor=0.5)),
])
def test_tiny_flag_variants_vs_golden(golden_dir, name, over, cfgover):
    g = np.load(os.path.join(golden_dir, "g5_tiny_flags.npz"))
    net, arch, sd = _net(TINY, int(g["seed"]), **over)
    q, r = synth.make_inputs(1, 3, 70, 70, int(g["seed"]))
    out = net(torch.from_numpy(q).cuda(), torch.from_numpy(r).cuda(), False, 0, False)
    torch.cuda.synchronize()
    mae, mx = _compare(out["score_map_ref_cross"], torch.from_numpy(g[name]))
    print(f"{name}: MAE {mae:.2e} max {mx:.2e}")
    assert mae < MAE_TOL and mx < 2 * MAX_TOL, (name, mae, mx)  # one bound for every variant (tanh has 4x the sigmoid's slope)


def _check_compact(g, score):
    P = 14
    s = score.cpu().numpy()
    B, Hs, Ws = s.shape
    assert tuple(g["shape"]) == (B, Hs, Ws)
    grid = s.reshape(B, Hs // P, P, Ws // P, P).mean(axis=(2, 4), dtype=np.float64)
    rows = s[:, g["rows_idx"], :]
    mae_rows = float(np.abs(rows - g["rows"]).mean())
    mae_grid = float(np.abs(grid - g["patch_mean"]).mean())
    dmean = float(np.abs(s.mean(axis=(1, 2), dtype=np.float64) - g["mean"]).max())
    return mae_rows, mae_grid, dmean


@pytest.mark.parametrize("name,backbone", [
    ("g1_vits_518_n5", "facebook/dinov2-small"),       # BASELINE cfg-2 shape at B=1
    ("g4_vits_518x690_n2", "facebook/dinov2-small"),   # non-square: encoder bicubic + PE bilinear, output 518x686
    ("g2_vitb_518_n10", "facebook/dinov2-base"),       # BASELINE cfg-3 shape at B=1 (decoder dh=96)
    ("g3_vits_1036_n5", "facebook/dinov2-small"),      # BASELINE cfg-5 shape at B=1 (T=5477, Lk=27380)
])
def test_full_size_vs_reference_goldens(golden_dir, name, backbone):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    net, arch, sd = _net(backbone, int(g["seed"]))
    q, r = synth.make_inputs(int(g["B"]), int(g["N"]), int(g["H"]), int(g["W"]), int(g["seed"]))
    out = net(torch.from_numpy(q).cuda(), torch.from_numpy(r).cuda(), False, 0, False, return_mean=True)
    torch.cuda.synchronize()
    mae_rows, mae_grid, dmean = _check_compact(g, out["score_map_ref_cross"])
    print(f"{name}: MAE(rows)={mae_rows:.2e} MAE(patch means)={mae_grid:.2e} |dmean|={dmean:.2e}")
    assert mae_rows < MAE_TARGET and mae_grid < MAE_TARGET and dmean < 1e-4
    # fused per-image mean == mean of the map (the value the CSV writer consumes)
    assert (out["score_mean_ref_cross"] - out["score_map_ref_cross"].mean(dim=(-1, -2))).abs().max() < 1e-5


@pytest.mark.parametrize("B,N,H,W", [(1, 1, 28, 28), (1, 4, 56, 42), (3, 1, 70, 98), (2, 3, 98, 57), (5, 2, 42, 42), (1, 6, 112, 84),
                                     (4, 1, 31, 45), (2, 5, 85, 71)])
def test_tiny_shape_sweep_vs_oracle(B, N, H, W):
    """Ragged everything: single-row-panel GEMMs, 1..6 references, sizes that are not multiples of 14 (trailing pixels ignored,
    HF:141-149), odd batch sizes over two lanes.  Tiny backbone, fp32 oracle on the host."""
    net, arch, sd = _net(TINY, 11)
    q, r = synth.make_inputs(B, N, H, W, 100 + B * 7 + N)
    out = net(torch.from_numpy(q).cuda(), torch.from_numpy(r).cuda(), False, 0, False)["score_map_ref_cross"]
    torch.cuda.synchronize()
    ref = _oracle(arch, sd, q, r)["score_map_ref_cross"]
    assert tuple(out.shape) == tuple(ref.shape) == (B, 14 * (H // 14), 14 * (W // 14))
    mae, mx = _compare(out, ref)
    assert mae < MAE_TOL and mx < 2 * MAX_TOL, (mae, mx)
    again = net(torch.from_numpy(q).cuda(), torch.from_numpy(r).cuda(), False, 0, False)["score_map_ref_cross"]
    assert torch.equal(out, again)


def test_dinov2_large_width_vs_oracle():
    """Every kernel shape of facebook/dinov2-large (C = 1024, 16 encoder heads of 64, decoder heads of 128, fc1 4096 wide: the
    256 x 256-tile GEMM for the 1024-wide linears, the 128-row one for fc1, im2col + GEMM for the patch embedding) on a two-layer encoder of
    that width against the fp32 oracle, with the attention-weights map, and the registered large architecture itself."""
    net, arch, sd = _net("synthetic/dinov2-wide", 21)
    assert (arch.hidden, arch.enc_heads, arch.hidden // arch.dec_heads) == (1024, 16, 128)
    q, r = synth.make_inputs(2, 2, 84, 98, 21)
    out = net(torch.from_numpy(q).cuda(), torch.from_numpy(r).cuda(), True, 5, False)
    torch.cuda.synchronize()
    ref = _oracle(arch, sd, q, r, True, 5)
    mae, mx = _compare(out["score_map_ref_cross"], ref["score_map_ref_cross"])
    assert mae < MAE_TOL and mx < 2 * MAX_TOL, (mae, mx)
    aw = (out["attn_weights_map_ref_cross"].cpu() - ref["attn_weights_map_ref_cross"]).abs()
    assert aw.max() < AW_MAX_TOL_TINY and torch.isfinite(out["attn_weights_map_ref_cross"]).all(), float(aw.max())
    big = synth.BACKBONES["facebook/dinov2-large"]
    assert (big.hidden, big.enc_layers, big.enc_heads) == (1024, 24, 16)


def test_dinov2_large_width_bf16_operands_vs_oracle():
    """The ViT-L two-layer net with bfloat16 operands: dh = 128 decoder attention, dh = 64 encoder attention, the 256 x 256-tile GEMM's residual /
    GELU epilogues and the im2col patch path, all in their bf16 forms, against the fp32 oracle inside north_star's bound."""
    net, arch, sd = _net("synthetic/dinov2-wide", 21)
    net.operand_dtype = "bf16"
    q, r = synth.make_inputs(2, 2, 84, 98, 21)
    out = net(torch.from_numpy(q).cuda(), torch.from_numpy(r).cuda(), False, 0, False)["score_map_ref_cross"]
    torch.cuda.synchronize()
    ref = _oracle(arch, sd, q, r)["score_map_ref_cross"]
    mae, mx = _compare(out, ref)
    print(f"dinov2-wide, bf16 operands: MAE {mae:.2e} max {mx:.2e}")
    assert torch.isfinite(out).all() and net.nonfinite_count() == 0
    assert mae < MAE_TOL and mx < 3e-2, (mae, mx)


def test_structured_images_vs_oracle():
    """Natural-image-like inputs (smooth patterns + mild noise, ImageNet-normalised uint8 pixels) are the hard case for 16-bit
    operands: a patch is mostly its mean, so the rounding error of the patch-embedding weights adds up coherently.  The
    mean-centred patch embedding + fp16 operands keep the score-map MAE well inside the bound (plain bf16 operands: 1.0e-3)."""
    from oracle import preprocess_oracle as po

    net, arch, sd = _net("facebook/dinov2-small", 1)
    rng = np.random.Generator(np.random.PCG64(1))
    yy, xx = np.mgrid[0:518, 0:518]

    def img(i):
        a = np.stack([127 + 100 * np.sin(xx / (17.0 + i) + i), 127 + 100 * np.cos(yy / (23.0 + i)), (xx + yy + 31 * i) % 256], axis=2)
        return (a + rng.normal(0, 8, a.shape)).clip(0, 255).astype(np.uint8)

    ims = np.stack([po.preprocess_u8(img(i)) for i in range(6)])
    q, r = ims[:1], ims[None, 1:]
    out = net(torch.from_numpy(q).cuda(), torch.from_numpy(r).cuda(), False, 0, False)["score_map_ref_cross"]
    torch.cuda.synchronize()
    ref = _oracle(arch, sd, q, r)["score_map_ref_cross"]
    mae, mx = _compare(out, ref)
    print(f"structured images: score-map MAE {mae:.3e} max {mx:.3e}")
    assert mae < MAE_TARGET and mx < MAX_TOL, (mae, mx)


def test_attention_weights_full_size_vs_oracle():
    """need_attn_weights at the real token counts (ViT-S, 518^2, N=2: 1369 x 2738 weights per query): rows are probability
    distributions, agree with the oracle's explicit softmax, and asking for them does not change the score map."""
    net, arch, sd = _net("facebook/dinov2-small", 2)
    q, r = synth.make_inputs(1, 2, 518, 518, 2)
    tq, tr = torch.from_numpy(q).cuda(), torch.from_numpy(r).cuda()
    out = net(tq, tr, True, 5, False)
    plain = net(tq, tr, False, 0, False)
    torch.cuda.synchronize()
    aw = out["attn_weights_map_ref_cross"]
    assert tuple(aw.shape) == (1, 37, 37, 2, 37, 37)
    assert torch.equal(out["score_map_ref_cross"], plain["score_map_ref_cross"])
    rows = aw.reshape(1369, -1)
    assert (rows.sum(-1) - 1).abs().max() < 1e-4 and rows.min() >= 0
    ref = _oracle(arch, sd, q, r, need_w=True, head=5)["attn_weights_map_ref_cross"].reshape(1369, -1)
    d = (rows.cpu() - ref).abs()
    print(f"attention probabilities 1369 x 2738: max abs error {float(d.max()):.2e}, max per-row L1 {float(d.sum(-1).max()):.2e}")
    assert float(d.sum(-1).max()) < AW_L1_TOL, float(d.sum(-1).max())   # total-variation-like distance per row (fp16 logits)
    assert float(d.max()) < AW_MAX_TOL


def test_forward_on_a_side_stream_is_stream_ordered():
    """All work is ordered on the caller's stream (internal lanes fork from and join to it): producing the inputs and consuming
    the outputs on a non-default stream without any device synchronisation gives the same bits."""
    net, arch, sd = _net(TINY, 7)
    q, r = synth.make_inputs(3, 2, 56, 70, 7)
    tq, tr = torch.from_numpy(q).cuda(), torch.from_numpy(r).cuda()
    base = net(tq, tr, False, 0, False)["score_map_ref_cross"].clone()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            q2 = tq * 1.0                                  # produced on the side stream just before the call
            out = net(q2, tr, False, 0, False)["score_map_ref_cross"]
            acc = out + 0.0                                # consumed on the side stream right after it
    side.synchronize()
    assert torch.equal(acc, base)


def test_calls_alternating_between_two_streams_do_not_race():
    """The handle's workspace is shared by all calls; a call that arrives on another stream waits for the previous call."""
    net, arch, sd = _net(TINY, 9)
    ins = [synth.make_inputs(2, 3, 70, 84, 20 + i) for i in range(2)]
    tens = [(torch.from_numpy(q).cuda(), torch.from_numpy(r).cuda()) for q, r in ins]
    want = [net(q, r, False, 0, False)["score_map_ref_cross"].clone() for q, r in tens]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    got = []
    for it in range(6):
        k = it % 2
        with torch.cuda.stream(streams[k]):
            got.append((k, net(tens[k][0], tens[k][1], False, 0, False)["score_map_ref_cross"]))
    torch.cuda.synchronize()
    for k, g in got:
        assert torch.equal(g, want[k])


def test_fp16_operand_range_on_scaled_up_weights():
    """16-bit MFMA operands are IEEE half: 11 significant bits, but finite only up to 65504.  What is stored as fp16 on this
    path: LayerNorm outputs (bounded by sqrt(C)), q / k / v and the other projections' outputs, softmax probabilities, attention
    outputs (convex combinations of v) and the MLP hidden (GELU output).  Scale the weights that feed the unbounded ones -- the
    q/k/v projections x8 and fc1 x64, inputs x4 -- far beyond trained-model magnitudes (the oracle's hidden activations then
    reach ~1e3..1e4, q/k ~1e2): the score map must stay finite and inside the parity bound against the fp32 oracle."""
    net, arch, sd = _net(TINY, 21)
    sd = {k: v.copy() for k, v in sd.items()}
    for k in sd:
        if k.startswith("backbone.encoder.layer."):
            if ".attention.attention." in k:
                sd[k] = sd[k] * 8.0
            if ".mlp.fc1." in k:
                sd[k] = sd[k] * 64.0
    net.load_numpy_state_dict(sd)
    q, r = synth.make_inputs(2, 3, 70, 84, 5)
    q, r = q * 4.0, r * 4.0
    taps = {}
    ref = orc.forward(orc.to_torch(sd), dict(enc_heads=arch.enc_heads), torch.from_numpy(q), torch.from_numpy(r), taps=taps)["score_map_ref_cross"]
    out = net(torch.from_numpy(q).cuda(), torch.from_numpy(r).cuda(), False, 0, False)["score_map_ref_cross"]
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()
    mae, mx = _compare(out, ref)
    print(f"scaled-up weights: MAE {mae:.2e} max {mx:.2e}")
    assert mae < MAE_TOL, (mae, mx)


def test_bf16_operand_mode_matches_oracle():
    """cs_config.operand_dtype = bf16 (north_star / BASELINE cfg-2 say bf16; the path's default is fp16): same kernels with the bf16
    MFMA forms and bf16 activations / weights.  8 significant bits instead of 11: the budget the oracle's operand-rounding emulation
    predicts is 8e-4 score-map MAE, inside north_star's 1e-3."""
    for backbone, seed, (B, N, H, W) in ((TINY, 3, (2, 2, 75, 90)), ("facebook/dinov2-small", 1, (1, 5, 518, 518))):
        net, arch, sd = _net(backbone, seed)
        net.operand_dtype = "bf16"
        q, r = synth.make_inputs(B, N, H, W, seed)
        out = net(torch.from_numpy(q).cuda(), torch.from_numpy(r).cuda(), False, 0, False)["score_map_ref_cross"]
        torch.cuda.synchronize()
        ref = _oracle(arch, sd, q, r)["score_map_ref_cross"]
        mae, mx = _compare(out, ref)
        print(f"bf16 operands, {backbone}: MAE {mae:.2e} max {mx:.2e}")
        assert torch.isfinite(out).all() and mae < 1e-3 and mx < 2e-2, (backbone, mae, mx)
        assert net.nonfinite_count() == 0
        # the cached-reference mode carries bf16 tokens and stays bit-identical to the full forward
        tok = net.encode_references(torch.from_numpy(r).cuda().reshape(-1, 3, H, W)).reshape(B, N, -1, arch.hidden)
        assert tok.dtype == torch.bfloat16
        assert torch.equal(net.forward_cached(torch.from_numpy(q).cuda(), tok)["score_map_ref_cross"], out)
        net.operand_dtype = "fp16"
        net._mark_dirty()
        out16 = net(torch.from_numpy(q).cuda(), torch.from_numpy(r).cuda(), False, 0, False)["score_map_ref_cross"]
        assert _compare(out16, ref)[0] < mae  # (and the default mode is the more accurate one)


def test_fp16_overflow_is_reported_and_bf16_mode_survives():
    """DINOv2-style outliers at ViT-S width: two residual channels carry +-300 (through the position embeddings), and
    eight hidden units of two encoder layers' MLPs are scaled so that their activations reach 2e5 > 65504 (fc1 rows x S, the matching fc2
    columns x 1 / S, so the fp32 model stays sane; S from the oracle's own activations).  fp16 operands: the hidden overflows to inf, the score map turns NaN and the
    handle REPORTS it (cs_nonfinite_count > 0) -- nothing clamps silently.  bf16 operands: finite, and inside north_star's bound."""
    net, arch, sd = _net("facebook/dinov2-small", 1)
    sd = {k: v.copy() for k, v in sd.items()}
    sd["backbone.embeddings.position_embeddings"][0, :, 5] += 300.0
    sd["backbone.embeddings.position_embeddings"][0, :, 200] -= 300.0
    q, r = synth.make_inputs(1, 5, 518, 518, 1)
    tq, tr = torch.from_numpy(q).cuda(), torch.from_numpy(r).cuda()
    # scale chosen from the fp32 oracle itself: the eight units' activations must reach 2e5 (3 x the largest finite half)
    for l in (4, 9):
        taps = {}
        orc.forward(orc.to_torch(sd), dict(enc_heads=arch.enc_heads), torch.from_numpy(q), torch.from_numpy(r), taps=taps)
        S = 2.0e5 / float(taps[f"enc_mlp_hidden_absmax_{l}"][:8].max())
        p = f"backbone.encoder.layer.{l}.mlp."
        sd[p + "fc1.weight"][:8] *= S
        sd[p + "fc1.bias"][:8] *= S
        sd[p + "fc2.weight"][:, :8] /= S
    taps = {}
    ref = orc.forward(orc.to_torch(sd), dict(enc_heads=arch.enc_heads), torch.from_numpy(q), torch.from_numpy(r), taps=taps)["score_map_ref_cross"]
    hid = max(float(taps[f"enc_mlp_hidden_absmax_{l}"][:8].max()) for l in (4, 9))
    print(f"oracle: largest hidden activation of the scaled units {hid:.3g}")
    assert hid > 1.5e5 and torch.isfinite(ref).all()
    net.load_numpy_state_dict(sd)
    out16 = net(tq, tr, False, 0, False)["score_map_ref_cross"]
    torch.cuda.synchronize()
    bad = net.nonfinite_count()
    print(f"fp16 operands with outliers: {bad} non-finite score values, finite {bool(torch.isfinite(out16).all())}")
    assert bad > 0 and not torch.isfinite(out16).all()
    assert net.nonfinite_count() == 0  # (the count is reset by the query)
    net.operand_dtype = "bf16"
    net._mark_dirty()
    outb = net(tq, tr, False, 0, False)["score_map_ref_cross"]
    torch.cuda.synchronize()
    mae, mx = _compare(outb, ref)
    print(f"bf16 operands with outliers: MAE {mae:.2e} max {mx:.2e}")
    assert torch.isfinite(outb).all() and net.nonfinite_count() == 0
    assert mae < 1e-3, (mae, mx)


def test_decoder_fused_linear_layernorm_matches_two_launch_form():
    """ViT-S width: the decoder closes its sub-blocks with ONE launch each (linear + residual + LayerNorm + the sub-block's next linear,
    csrc/rowln.hip) instead of a GEMM, a LayerNorm launch and another GEMM; all three forms of the forward are compared.  Same operands and fp32 arithmetic in another summation order; the normalised rows are then rounded to fp16 for the next
    projection, where a last-bit difference becomes one fp16 ulp (5e-4 relative) of an activation: the score maps of the two forms agree to
    1e-3 max / 1e-4 mean (measured 3.1e-4 / 4e-5), and both hold the oracle bound; with decoder_do_short_cut off (no residual into the LayerNorm) as well."""
    from crossscore_amd import _lib
    lib = _lib.load()
    for over in ({}, {"decoder_do_short_cut": False}, {"decoder_do_self_attn": False}):
        net, arch, sd = _net("facebook/dinov2-small", 5, **over)
        q, r = synth.make_inputs(1, 2, 126, 154, 5)
        tq, tr = torch.from_numpy(q).cuda(), torch.from_numpy(r).cuda()
        fused = net(tq, tr, False, 0, False)["score_map_ref_cross"].clone()
        try:
            lib.cs_debug_rowln_enable(2)  # linear + LayerNorm in one launch, the next linear as a GEMM of its own
            mid = net(tq, tr, False, 0, False)["score_map_ref_cross"].clone()
            lib.cs_debug_rowln_enable(0)  # GEMM + LayerNorm + GEMM
            two = net(tq, tr, False, 0, False)["score_map_ref_cross"].clone()
        finally:
            lib.cs_debug_rowln_enable(1)
        torch.cuda.synchronize()
        for other in (mid, two):
            d = (fused - other).abs()
            assert float(d.max()) < 1e-3 and float(d.mean()) < 1e-4, (over, float(d.max()), float(d.mean()))
        cfgover = {k.replace("decoder_", ""): v for k, v in over.items()}
        ref = _oracle(arch, sd, q, r, **cfgover)["score_map_ref_cross"]
        mae, mx = _compare(fused, ref)
        assert mae < MAE_TOL and mx < MAX_TOL, (over, mae, mx)


def test_multiview_pe_bicubic_mode_end_to_end():
    """model.pos_enc.multi_view.interpolate_mode=bicubic (VERDICT r3 missing #4): the config key the reference passes to F.interpolate
    (positional_encoding.py:61-69).  Tiny net at 75x90 (5x6 patch grid != 40x40: the resize is live) against the oracle in that mode; the
    bilinear default differs measurably, and a mode torch rejects is rejected."""
    net, arch, sd = _net(TINY, 13, **{"pos_enc.multi_view.interpolate_mode": "bicubic"})
    q, r = synth.make_inputs(2, 2, 75, 90, 13)
    out = net(torch.from_numpy(q).cuda(), torch.from_numpy(r).cuda(), False, 0, False)["score_map_ref_cross"]
    torch.cuda.synchronize()
    ref = _oracle(arch, sd, q, r, pe_interpolate_mode="bicubic")["score_map_ref_cross"]
    mae, mx = _compare(out, ref)
    assert mae < MAE_TOL and mx < MAX_TOL, (mae, mx)
    # the same run by the reference itself (tests/golden/make_golden.py --only g7: same seed, shapes and mode)
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g7_tiny_pe_bicubic.npz"))
    assert int(g["seed"]) == 13 and (int(g["B"]), int(g["N"]), int(g["H"]), int(g["W"])) == (2, 2, 75, 90)
    mae_g, mx_g = _compare(out, torch.from_numpy(g["score"]))
    assert mae_g < MAE_TOL and mx_g < MAX_TOL, (mae_g, mx_g)
    ref_bil = _oracle(arch, sd, q, r)["score_map_ref_cross"]
    assert float((ref - ref_bil).abs().mean()) > 3 * mae
    # any other mode: F.interpolate(..., align_corners=True) raises in the reference -- at the first forward that has to resize the PE table, not
    # at construction, and never for a patch grid equal to the table's 40 x 40 (positional_encoding.py:51-56 adds the parameter as it is)
    netn, _, _ = _net(TINY, 13, **{"pos_enc.multi_view.interpolate_mode": "nearest"})
    with pytest.raises(ValueError, match="align_corners"):
        netn(torch.from_numpy(q).cuda(), torch.from_numpy(r).cuda(), False, 0, False)
    q40, r40 = synth.make_inputs(1, 1, 560, 560, 13)
    s_near = netn(torch.from_numpy(q40).cuda(), torch.from_numpy(r40).cuda(), False, 0, False)["score_map_ref_cross"]
    s_bil = _net(TINY, 13)[0](torch.from_numpy(q40).cuda(), torch.from_numpy(r40).cuda(), False, 0, False)["score_map_ref_cross"]
    torch.cuda.synchronize()
    assert torch.equal(s_near, s_bil)


def test_pos_embed_scale_factor_interpolation_option():
    """model.backbone.pos_embed_interpolation=scale_factor: the encoder position-embedding resize of the reference's pinned
    transformers 4.33.3 (scale_factor=((h+0.1)/G, (w+0.1)/G)) instead of the installed size=(h, w).  The HIP table follows the oracle's
    restatement of it, and the two conventions really differ on a resized grid (so the switch is live)."""
    net, arch, sd = _net(TINY, 31, **{"backbone.pos_embed_interpolation": "scale_factor"})
    q, r = synth.make_inputs(1, 2, 75, 90, 6)
    out = net(torch.from_numpy(q).cuda(), torch.from_numpy(r).cuda(), False, 0, False)["score_map_ref_cross"]
    torch.cuda.synchronize()
    ref_legacy = _oracle(arch, sd, q, r, pos_interp_legacy=True)["score_map_ref_cross"]
    ref_size = _oracle(arch, sd, q, r)["score_map_ref_cross"]
    mae, mx = _compare(out, ref_legacy)
    assert mae < MAE_TOL and mx < 2 * MAX_TOL, (mae, mx)
    assert float((ref_legacy - ref_size).abs().mean()) > 3 * mae


def test_pos_embed_scale_factor_end_to_end_golden():
    """g6: the imported reference with its embeddings' interpolate_pos_encoding replaced by the pinned 4.33.3 call (torch's
    F.interpolate(scale_factor=..)); the HIP path with model.backbone.pos_embed_interpolation=scale_factor must match its score map."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g6_pos_legacy.npz"))
    net, arch, sd = _net(TINY, int(g["seed"]), **{"backbone.pos_embed_interpolation": "scale_factor"})
    q, r = synth.make_inputs(int(g["B"]), int(g["N"]), int(g["H"]), int(g["W"]), int(g["input_seed"]))
    out = net(torch.from_numpy(q).cuda(), torch.from_numpy(r).cuda(), False, 0, False)["score_map_ref_cross"]
    torch.cuda.synchronize()
    mae, mx = _compare(out, torch.from_numpy(g["score"]))
    assert mae < MAE_TOL and mx < 2 * MAX_TOL, (mae, mx)


def test_batch8_vits_matches_oracle_per_item_and_is_batch_invariant():
    """cfg-2 (ViT-S, 518^2, N=5, B=8): item 3 of the batch against the oracle; every item must equal the same item run
    alone (batch shard equivalence: shards are independent, so results are bitwise identical)."""
    net, arch, sd = _net("facebook/dinov2-small", 1)
    q, r = synth.make_inputs(8, 5, 518, 518, 1)
    tq, tr = torch.from_numpy(q).cuda(), torch.from_numpy(r).cuda()
    both = net(tq, tr, False, 0, False, return_mean=True)
    full, full_mean = both["score_map_ref_cross"], both["score_mean_ref_cross"]
    part = net(tq[2:4], tr[2:4], False, 0, False)["score_map_ref_cross"]
    torch.cuda.synchronize()
    assert torch.equal(full[2:4], part)
    assert torch.equal(full, net(tq, tr, False, 0, False)["score_map_ref_cross"])  # asking for the means does not change the map
    assert (full_mean.double() - full.double().mean(dim=(-1, -2))).abs().max() < 2e-6
    for i in range(8):  # EVERY item alone (a one-item shard: other chunking, other panel / tile boundaries) gives the same bits
        alone = net(tq[i:i + 1], tr[i:i + 1], False, 0, False, return_mean=True)
        assert torch.equal(full[i:i + 1], alone["score_map_ref_cross"]), i
        # ... and so does its mean, which the head launch forms from per-patch-row partials in an order fixed inside the image
        assert torch.equal(full_mean[i:i + 1], alone["score_mean_ref_cross"]), i
    for i in (0, 3, 7):  # first, middle and last item of the batch against the fp32 oracle
        ref = _oracle(arch, sd, q[i:i + 1], r[i:i + 1])["score_map_ref_cross"]
        mae, mx = _compare(full[i:i + 1], ref)
        print(f"cfg-2 item {i}: MAE={mae:.2e} max={mx:.2e}")
        assert mae < MAE_TARGET and mx < MAX_TOL, (i, mae, mx)
    # determinism: two runs bitwise equal
    again = net(tq, tr, False, 0, False)["score_map_ref_cross"]
    assert torch.equal(full, again)


def test_encoder_chunking_is_invisible():
    """Images-per-encoder-pass is a cache-residency knob only: any chunk size gives bitwise the same map."""
    net, arch, sd = _net(TINY, 2)
    q, r = synth.make_inputs(3, 2, 70, 98, 2)
    tq, tr = torch.from_numpy(q).cuda(), torch.from_numpy(r).cuda()
    base = net(tq, tr, False, 0, False)["score_map_ref_cross"].clone()
    for chunk in (1, 2, 4, 9):
        net.enc_chunk_images = chunk
        net._mark_dirty()
        assert torch.equal(net(tq, tr, False, 0, False)["score_map_ref_cross"], base), chunk


def test_lane_calibration_and_stream_redraw_keep_results_bitwise():
    """CrossScoreNet.calibrate_lanes (times the two-lane forward against the one-lane one on the same handle and re-draws the lane streams
    if it does not win) and the two entry points under it, cs_set_lanes / cs_redraw_lane_streams: same launches whatever the lanes and
    whichever streams carry them, so the score map stays bitwise the same."""
    from crossscore_amd import _lib

    net, arch, sd = _net(TINY, 5)
    q, r = synth.make_inputs(4, 2, 70, 98, 5)
    tq, tr = torch.from_numpy(q).cuda(), torch.from_numpy(r).cuda()
    base = net(tq, tr, False, 0, False)["score_map_ref_cross"].clone()
    cal = net.calibrate_lanes(tq, tr, tries=2, steps=2)
    assert cal["one_lane_s"] > 0 and 1 <= len(cal["lanes_s"]) <= 2 and all(v > 0 for v in cal["lanes_s"])
    assert torch.equal(net(tq, tr, False, 0, False)["score_map_ref_cross"], base)
    lib = _lib.load()
    _lib.check(lib.cs_set_lanes(net._handle, 1))
    one = net(tq, tr, False, 0, False)["score_map_ref_cross"].clone()
    _lib.check(lib.cs_set_lanes(net._handle, 0))
    _lib.check(lib.cs_redraw_lane_streams(net._handle))
    again = net(tq, tr, False, 0, False)["score_map_ref_cross"]
    torch.cuda.synchronize()
    assert torch.equal(one, base) and torch.equal(again, base)
    with pytest.raises((ValueError, RuntimeError)):
        _lib.check(lib.cs_set_lanes(net._handle, 99))
    net.lanes = 1
    net._mark_dirty()
    assert net.calibrate_lanes(tq, tr) == {"one_lane_s": None, "lanes_s": []}  # nothing to calibrate with one lane


def test_reference_token_cache_is_bit_identical():
    """encode each reference image once, gather per query -> the same score map bit for bit (encoder results
    per image do not depend on the batch they ran in), including when two queries share a reference."""
    net, arch, sd = _net(TINY, 4)
    q, r = synth.make_inputs(3, 2, 70, 98, 4)
    r[2, 0] = r[0, 1]  # query 2 reuses a reference of query 0
    tq, tr = torch.from_numpy(q).cuda(), torch.from_numpy(r).cuda()
    full = net(tq, tr, True, 5, False)
    pool = torch.stack([tr[0, 0], tr[0, 1], tr[1, 0], tr[1, 1], tr[2, 1]])     # 5 distinct reference images
    tok = net.encode_references(pool)
    assert tok.shape == (5, 5 * 7, arch.hidden) and tok.dtype == torch.float16
    idx = torch.tensor([[0, 1], [2, 3], [1, 4]], device="cuda")
    cached = net.forward_cached(tq, tok[idx], True, 5, return_mean=True)
    torch.cuda.synchronize()
    assert torch.equal(cached["score_mean_ref_cross"], net(tq, tr, False, 0, False, return_mean=True)["score_mean_ref_cross"])
    assert torch.equal(cached["score_map_ref_cross"], full["score_map_ref_cross"])
    assert torch.equal(cached["attn_weights_map_ref_cross"], full["attn_weights_map_ref_cross"])


@pytest.mark.parametrize("depth,backbone,H", [(2, "facebook/dinov2-small", 224), (3, TINY, 98), (1, TINY, 98)])
def test_batches_in_flight_are_bit_identical_to_one_at_a_time(depth, backbone, H):
    """crossscore_amd.pipeline.ForwardPipeline: `depth` replicas over the same parameters, fed round-robin on their own streams (one
    batch's decoder beside the next batch's encoder).  Seven different batches, results fetched one submit later as the predict loop
    does: every score map equals the plain forward's bit for bit, also through the reference-token cache entry."""
    from crossscore_amd.pipeline import ForwardPipeline

    net, arch, sd = _net(backbone, 5)
    batches = []
    for i in range(7):
        q, r = synth.make_inputs(2 + (i % 2), 3, H, H, 100 + i)  # batch size alternates: the workspaces are re-planned in flight
        batches.append((torch.from_numpy(q).cuda(), torch.from_numpy(r).cuda()))
    want = [net(q, r, False, 0, False)["score_map_ref_cross"].clone() for q, r in batches]
    toks = [net.encode_references(r.reshape(-1, 3, H, H)).reshape(r.shape[0], r.shape[1], -1, arch.hidden) for _, r in batches]
    torch.cuda.synchronize()
    pipe = ForwardPipeline(net, depth=depth)
    assert len(pipe.nets) == depth and all(p.data_ptr() == q.data_ptr() for n in pipe.nets[1:] for p, q in zip(n.parameters(), net.parameters()))
    got, prev = [], None
    for q, r in batches:
        t = pipe.submit(q, r, False, 0, False)
        if prev is not None:
            got.append(pipe.result(prev)["score_map_ref_cross"])
        prev = t
    got.append(pipe.result(prev)["score_map_ref_cross"])
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(zip(got, want)):
        assert torch.equal(a, b), i
    got, prev = [], None
    for (q, _), tok in zip(batches, toks):
        t = pipe.submit_cached(q, tok)
        if prev is not None:
            got.append(pipe.result(prev)["score_map_ref_cross"])
        prev = t
    got.append(pipe.result(prev)["score_map_ref_cross"])
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(zip(got, want)):
        assert torch.equal(a, b), i
    if depth > 1:  # new weights loaded into the module reach every replica (shared tensors, handles re-packed)
        sd2 = synth.make_state_dict(arch, 6)
        net.load_numpy_state_dict(sd2)
        q, r = batches[0]
        want2 = net(q, r, False, 0, False)["score_map_ref_cross"].clone()
        assert not torch.equal(want2, want[0])
        for _ in range(depth):
            assert torch.equal(pipe.result(pipe.submit(q, r, False, 0, False))["score_map_ref_cross"], want2)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_half_precision_checkpoints_load_through_the_typed_entry(dtype):
    """cs_set_weight_typed (SURVEY.md 8b's dtype argument): a module converted with .half() / .bfloat16() hands its 16-bit tensors to the
    library as they are; the result equals the fp32 module loaded with the same (rounded) values, bit for bit."""
    net, arch, sd = _net(TINY, 3)
    q, r = synth.make_inputs(2, 2, 70, 98, 3)
    tq, tr = torch.from_numpy(q).cuda(), torch.from_numpy(r).cuda()
    rounded = {k: torch.from_numpy(v).to(dtype).to(torch.float32) for k, v in sd.items()}
    net.load_state_dict(rounded, strict=True)
    want = net(tq, tr, False, 0, False)["score_map_ref_cross"].clone()
    net16 = CrossScoreNet(model_config(**{"backbone.from_pretrained": TINY}))
    net16.load_state_dict(rounded, strict=True)
    net16 = net16.cuda().to(dtype)
    assert next(net16.parameters()).dtype == dtype
    got = net16(tq, tr, False, 0, False)["score_map_ref_cross"]
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    # converting the module after its handle was built re-packs the weights
    net16 = net16.float()
    assert torch.equal(net16(tq, tr, False, 0, False)["score_map_ref_cross"], want)


def test_layernorm_fold_matches_separate_layernorm_path():
    """ln_fold=1 (LayerNorm applied inside the consuming GEMM epilogue) and the default separate-LayerNorm path are two
    roundings of the same fp32 math: both within tolerance of the oracle; the fold must not be a silent no-op."""
    net, arch, sd = _net("facebook/dinov2-small", 1)
    q, r = synth.make_inputs(1, 2, 518, 518, 1)
    tq, tr = torch.from_numpy(q).cuda(), torch.from_numpy(r).cuda()
    plain = net(tq, tr, False, 0, False)["score_map_ref_cross"].clone()
    net.ln_fold = 1
    net._mark_dirty()
    folded = net(tq, tr, False, 0, False)["score_map_ref_cross"].clone()
    ref = _oracle(arch, sd, q, r)["score_map_ref_cross"]
    torch.cuda.synchronize()
    mae_f, _ = _compare(folded, ref)
    mae_p, _ = _compare(plain, ref)
    print(f"ln_fold MAE={mae_f:.2e}  separate-LN MAE={mae_p:.2e}  fold-vs-plain MAE={float((folded - plain).abs().mean()):.2e}")
    assert mae_f < MAE_TOL and mae_p < MAE_TOL
    assert not torch.equal(folded, plain)


def test_bad_inputs_raise():
    net, arch, sd = _net(TINY, 2)
    q = torch.zeros(1, 3, 70, 70, device="cuda")
    with pytest.raises(ValueError):
        net(q, torch.zeros(1, 2, 3, 70, 84, device="cuda"), False, 0, False)  # size mismatch
    with pytest.raises(ValueError):
        net(q, torch.zeros(1, 2, 3, 70, 70, device="cuda"), True, 8, False)   # head id out of range (8 heads)
    with pytest.raises(ValueError):
        net(torch.zeros(1, 3, 10, 70, device="cuda"), torch.zeros(1, 2, 3, 10, 70, device="cuda"), False, 0, False)  # < one patch


@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
def test_four_wave_panel_kernel_through_the_forward(golden_dir, dtype):
    """The round-6 token-panel kernel (csrc/panel4.hip: 4 waves, one per SIMD; cs_debug_panel_impl(1), opt-in) through the whole forward at the
    BASELINE cfg-2 shape (B = 1): item 0 against the reference's own golden g1 at the same bounds as the default path, the launch census shows
    that it -- not the 8-wave kernel -- ran (12 layers x 2 chunks), the score map stays within the operand-rounding distance of the default
    path's, and the forward repeats bit-identically."""
    from crossscore_amd import _lib
    lib = _lib.load()
    g = np.load(os.path.join(golden_dir, "g1_vits_518_n5.npz"))
    q, r = synth.make_inputs(int(g["B"]), int(g["N"]), int(g["H"]), int(g["W"]), int(g["seed"]))
    tq, tr = torch.from_numpy(q).cuda(), torch.from_numpy(r).cuda()
    net8, arch, sd = _net("facebook/dinov2-small", int(g["seed"]))
    net8.operand_dtype = dtype
    ref8 = net8(tq, tr, False, 0, False)["score_map_ref_cross"]
    lib.cs_debug_panel_impl(1)
    try:
        net4, _, _ = _net("facebook/dinov2-small", int(g["seed"]))   # (a handle packs its panel images for the kernel selected at cs_finalize)
        net4.operand_dtype = dtype
        out = net4(tq, tr, False, 0, False)["score_map_ref_cross"]
        torch.cuda.synchronize()
        k = net4.forward_stats()["kernels"]
        again = net4(tq, tr, False, 0, False)["score_map_ref_cross"]
    finally:
        lib.cs_debug_panel_impl(0)
    assert k.get("panel4", 0) >= 12 and "panel" not in k, k
    assert torch.equal(out, again)
    mae_rows, mae_grid, dmean = _check_compact(g, out)
    d = (out - ref8).abs()
    print(f"panel4 forward ({dtype}): MAE(rows) vs golden {mae_rows:.2e}, vs the 8-wave kernel's forward mean {float(d.mean()):.2e} max {float(d.max()):.2e}")
    assert mae_rows < (MAE_TOL if dtype == "bf16" else MAE_TARGET) and mae_grid < (MAE_TOL if dtype == "bf16" else MAE_TARGET)
    assert net4.nonfinite_count() == 0
