"""Pins the CPU oracle (oracle/crossscore_oracle.py) against golden vectors produced by the reference code
itself (tests/golden/make_golden.py, run in the build container).  fp32 vs fp32: tolerance 2e-5 abs."""
import dataclasses
import os

import numpy as np
import pytest
import torch

from crossscore_amd import synth
from oracle import crossscore_oracle as orc

TOL = 2e-5
TINY = synth.BACKBONES["synthetic/dinov2-tiny"]


def _golden(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_g0_tiny_all_intermediates(golden_dir):
    g = _golden(golden_dir, "g0_tiny_all.npz")
    W = orc.to_torch(synth.make_state_dict(TINY, int(g["seed"])))
    q, r = synth.make_inputs(int(g["B"]), int(g["N"]), int(g["H"]), int(g["W"]), int(g["seed"]))
    taps = {}
    out = orc.forward(W, dict(enc_heads=TINY.enc_heads), torch.from_numpy(q), torch.from_numpy(r), True, 3, taps=taps)
    checked = 0
    for k in g.files:
        if k in taps:
            assert np.abs(taps[k].numpy() - g[k]).max() < TOL, k
            checked += 1
    assert checked >= 9
    assert out["score_map_ref_cross"].shape == (2, 70, 84)  # floor-drop of 75%14, 90%14
    assert np.abs(out["score_map_ref_cross"].numpy() - g["score"]).max() < TOL
    assert out["attn_weights_map_ref_cross"].shape == (2, 5, 6, 2, 5, 6)
    assert np.abs(out["attn_weights_map_ref_cross"].numpy() - g["attn_head3"]).max() < TOL
    # rows of the returned attention sum to one; score independent of need_attn_weights
    s = out["attn_weights_map_ref_cross"].reshape(2, 30, -1).sum(-1)
    assert torch.allclose(s, torch.ones_like(s), atol=1e-5)
    assert np.abs(out["score_map_ref_cross"].numpy() - g["score_no_weights"]).max() < TOL


def test_g8_vits_width_all_intermediates(golden_dir):
    """g8: the ViT-S width (C = 384, two encoder layers), every module output of the reference (tests/golden/make_golden.py): the oracle's
    taps at the width whose HIP kernels carry the benchmark (tests/test_hip_stages.py compares those kernels with the same arrays)."""
    arch = synth.BACKBONES["synthetic/dinov2-small-2l"]
    g = _golden(golden_dir, "g8_vits_width_all.npz")
    W = orc.to_torch(synth.make_state_dict(arch, int(g["seed"])))
    q, r = synth.make_inputs(int(g["B"]), int(g["N"]), int(g["H"]), int(g["W"]), int(g["seed"]))
    taps = {}
    out = orc.forward(W, dict(enc_heads=arch.enc_heads), torch.from_numpy(q), torch.from_numpy(r), True, 3, taps=taps)
    checked = 0
    for k in g.files:
        if k in taps:
            # residual-stream values reach |x| ~ 30 at this width: 2e-5 relative to the stage's largest magnitude
            assert np.abs(taps[k].numpy() - g[k]).max() < TOL * max(1.0, float(np.abs(g[k]).max())), k
            checked += 1
    assert checked >= 8
    assert np.abs(out["score_map_ref_cross"].numpy() - g["score"]).max() < TOL
    assert np.abs(out["attn_weights_map_ref_cross"].numpy() - g["attn_head3"]).max() < TOL


VARIANTS = {
    "no_self_attn": dict(do_self_attn=False),
    "no_short_cut": dict(do_short_cut=False),
    "tanh": dict(metric_min=-1),
    "mae_pow2": dict(metric_type="mae"),
    "mse_pow4": dict(metric_type="mse"),
    "scalar_p": dict(power_factor=0.5),
}


@pytest.mark.parametrize("tag,key", [("g9_swiglu_base_width", "synthetic/dinov2-swiglu-2l"), ("g9_swiglu_giant_width", "synthetic/dinov2-giant-2l")])
def test_g9_swiglu_mlp(golden_dir, tag, key):
    """g9: the SwiGLU MLP of facebook/dinov2-giant (HF Dinov2SwiGLUFFN: weights_in -> chunk -> silu(x1) * x2 -> weights_out) at the giant's and at
    the base width, from the imported reference: the last encoder layer's rows of image 0 and the score map."""
    arch = synth.BACKBONES[key]
    assert arch.swiglu and arch.ffn_hidden == {768: 2048, 1536: 4096}[arch.hidden]
    g = _golden(golden_dir, tag + ".npz")
    W = orc.to_torch(synth.make_state_dict(arch, int(g["seed"])))
    q, r = synth.make_inputs(int(g["B"]), int(g["N"]), int(g["H"]), int(g["W"]), int(g["seed"]))
    taps = {}
    out = orc.forward(W, dict(enc_heads=arch.enc_heads), torch.from_numpy(q), torch.from_numpy(r), False, 0, taps=taps)
    last = taps[f"enc_layer_{arch.enc_layers - 1}"][0].numpy()
    assert np.abs(last - g["enc_last_img0"]).max() < TOL * max(1.0, float(np.abs(g["enc_last_img0"]).max()))
    assert np.abs(out["score_map_ref_cross"].numpy() - g["score"]).max() < TOL


@pytest.mark.parametrize("name", sorted(VARIANTS))
def test_g5_flag_variants(golden_dir, name):
    g = _golden(golden_dir, "g5_tiny_flags.npz")
    over = VARIANTS[name]
    arch = dataclasses.replace(TINY, do_self_attn=over.get("do_self_attn", True))
    W = orc.to_torch(synth.make_state_dict(arch, int(g["seed"])))
    q, r = synth.make_inputs(int(g["B"]), int(g["N"]), int(g["H"]), int(g["W"]), int(g["seed"]))
    out = orc.forward(W, dict(enc_heads=TINY.enc_heads, **over), torch.from_numpy(q), torch.from_numpy(r))
    assert np.abs(out["score_map_ref_cross"].numpy() - g[name]).max() < TOL


def test_g7_multiview_pe_bicubic_mode(golden_dir):
    """model.pos_enc.multi_view.interpolate_mode = bicubic, run by the REFERENCE (positional_encoding.py:61-69 hands the key to F.interpolate
    with align_corners=True): its score map and the featmaps behind the PE pin the oracle's own bicubic restatement
    (bicubic_resize_grid_align_corners) end to end, next to the F.interpolate comparison below."""
    g = _golden(golden_dir, "g7_tiny_pe_bicubic.npz")
    W = orc.to_torch(synth.make_state_dict(TINY, int(g["seed"])))
    q, r = synth.make_inputs(int(g["B"]), int(g["N"]), int(g["H"]), int(g["W"]), int(g["seed"]))
    taps = {}
    out = orc.forward(W, dict(enc_heads=TINY.enc_heads, pe_interpolate_mode="bicubic"), torch.from_numpy(q), torch.from_numpy(r), taps=taps)
    for k in ("featmap_query", "featmap_ref"):
        assert np.abs(taps[k].numpy() - g[k]).max() < TOL, k
    assert np.abs(out["score_map_ref_cross"].numpy() - g["score"]).max() < TOL
    # and the mode matters: the bilinear default is measurably different on these featmaps
    taps_bil = {}
    orc.forward(W, dict(enc_heads=TINY.enc_heads), torch.from_numpy(q), torch.from_numpy(r), taps=taps_bil)
    assert np.abs(taps_bil["featmap_query"].numpy() - g["featmap_query"]).max() > 100 * TOL


def _compact_check(g, score):
    P = 14
    B, Hs, Ws = score.shape
    assert tuple(g["shape"]) == (B, Hs, Ws)
    grid = score.reshape(B, Hs // P, P, Ws // P, P).mean(axis=(2, 4), dtype=np.float64)
    assert np.abs(grid - g["patch_mean"]).max() < TOL
    assert np.abs(score[:, g["rows_idx"], :] - g["rows"]).max() < TOL
    assert np.abs(score.mean(axis=(1, 2), dtype=np.float64) - g["mean"]).max() < TOL


@pytest.mark.parametrize("name,backbone", [
    ("g1_vits_518_n5", "facebook/dinov2-small"),
    ("g4_vits_518x690_n2", "facebook/dinov2-small"),
])
def test_full_size_goldens(golden_dir, name, backbone):
    g = _golden(golden_dir, name + ".npz")
    arch = synth.BACKBONES[backbone]
    W = orc.to_torch(synth.make_state_dict(arch, int(g["seed"])))
    q, r = synth.make_inputs(int(g["B"]), int(g["N"]), int(g["H"]), int(g["W"]), int(g["seed"]))
    out = orc.forward(W, dict(enc_heads=arch.enc_heads), torch.from_numpy(q), torch.from_numpy(r))
    _compact_check(g, out["score_map_ref_cross"].numpy())


@pytest.mark.parametrize("name,backbone", [
    ("g2_vitb_518_n10", "facebook/dinov2-base"),
    ("g3_vits_1036_n5", "facebook/dinov2-small"),
])
def test_full_size_goldens_vitb_and_1036(golden_dir, name, backbone):
    """ViT-B / 1036^2 pins of the oracle (10-20 s each on 8 cores): part of the default CPU suite (they used to hide behind CS_SLOW)."""
    test_full_size_goldens(golden_dir, name, backbone)


G6_TABLES = [("tiny", "synthetic/dinov2-tiny", 5, 6), ("tiny", "synthetic/dinov2-tiny", 7, 4), ("small", "facebook/dinov2-small", 37, 49),
             ("small", "facebook/dinov2-small", 74, 74), ("small", "facebook/dinov2-small", 20, 31), ("base", "facebook/dinov2-base", 37, 49)]


@pytest.mark.parametrize("tag,backbone,h,w", G6_TABLES)
def test_legacy_pos_embed_table_matches_torch_scale_factor_golden(golden_dir, tag, backbone, h, w):
    """The scale_factor form of the encoder position-embedding resize (the reference's pinned transformers 4.33.3, environment.yaml:340):
    g6 holds F.interpolate(scale_factor=((h + 0.1) / G, (w + 0.1) / G), bicubic, align_corners=False) executed by torch on the synthetic
    tables; the oracle's restatement (encoder_pos_embed(legacy=True)) must reproduce it."""
    g = _golden(golden_dir, "g6_pos_legacy.npz")
    arch = synth.BACKBONES[backbone]
    W = orc.to_torch(synth.make_state_dict(arch, int(g[f"table_{tag}_{h}x{w}_seed"])))
    tab = orc.encoder_pos_embed(W, h, w, 14 * h, 14 * w + 1, legacy=True).numpy()  # (H != W: the resize branch even on the native grid)
    assert np.abs(tab[g[f"table_{tag}_{h}x{w}_rows_idx"]] - g[f"table_{tag}_{h}x{w}_rows"]).max() < 2e-6
    assert np.abs(tab.mean(axis=1, dtype=np.float64) - g[f"table_{tag}_{h}x{w}_chmean"]).max() < 2e-6
    # and the two conventions really differ (the switch is live)
    tab_size = orc.encoder_pos_embed(W, h, w, 14 * h, 14 * w + 1, legacy=False).numpy()
    assert np.abs(tab_size - tab).max() > 1e-3


def test_legacy_pos_embed_end_to_end_golden(golden_dir):
    """g6 end to end: the imported reference with its embeddings' interpolate_pos_encoding replaced by the 4.33.3 call (tiny net,
    75 x 90 -> 5 x 6 patches); the oracle with pos_interp_legacy=True must match its last_hidden_state and score map."""
    g = _golden(golden_dir, "g6_pos_legacy.npz")
    W = orc.to_torch(synth.make_state_dict(TINY, int(g["seed"])))
    q, r = synth.make_inputs(int(g["B"]), int(g["N"]), int(g["H"]), int(g["W"]), int(g["input_seed"]))
    out = orc.forward(W, dict(enc_heads=TINY.enc_heads, pos_interp_legacy=True), torch.from_numpy(q), torch.from_numpy(r))
    assert np.abs(out["score_map_ref_cross"].numpy() - g["score"]).max() < TOL
    out_size = orc.forward(W, dict(enc_heads=TINY.enc_heads), torch.from_numpy(q), torch.from_numpy(r))
    assert np.abs(out_size["score_map_ref_cross"].numpy() - g["score"]).max() > 10 * TOL


@pytest.mark.parametrize("gh,gw", [(5, 6), (37, 37), (74, 74), (37, 49), (1, 7)])
def test_multiview_pe_bicubic_mode_matches_torch(gh, gw):
    """model.pos_enc.multi_view.interpolate_mode=bicubic: positional_encoding.py:61-69 calls F.interpolate(PE (1,C,40,40), scale_factor=((h+1e-4)/40,
    (w+1e-4)/40), mode=<cfg>, align_corners=True); the oracle's restatement against that very torch call (torch is third-party, not the
    reference: SURVEY.md 8c), and the modes torch rejects are rejected."""
    rng = np.random.Generator(np.random.PCG64(gh * 100 + gw))
    PE = torch.from_numpy(rng.standard_normal((1, 40, 40, 16), dtype=np.float32))
    want = torch.nn.functional.interpolate(PE.permute(0, 3, 1, 2), scale_factor=((gh + 1e-4) / 40, (gw + 1e-4) / 40), mode="bicubic",
                                           align_corners=True)[0].permute(1, 2, 0)
    assert want.shape[:2] == (gh, gw)
    got = orc.multiview_pe({"pos_enc_fn.PE": PE}, gh, gw, "bicubic").reshape(gh, gw, 16)
    assert (got - want).abs().max() < 5e-6  # fp32 summation order of the 16 taps
    bil = orc.multiview_pe({"pos_enc_fn.PE": PE}, gh, gw, "bilinear").reshape(gh, gw, 16)
    if gh > 1:
        assert (bil - want).abs().max() > 1e-3  # (the two modes really differ)
    with pytest.raises(ValueError):
        orc.multiview_pe({"pos_enc_fn.PE": PE}, gh, gw, "nearest")
    with pytest.raises(ValueError):
        torch.nn.functional.interpolate(PE.permute(0, 3, 1, 2), size=(gh, gw), mode="nearest", align_corners=True)


def test_regression_layer_config_errors():
    # model/regression_layer.py:65-81 smoke grid + utils/check_config.py:1-28
    x = torch.zeros(3)
    with pytest.raises(ValueError):
        orc.regression_layer(x, "psnr", 0, 1, "default")
    with pytest.raises(ValueError):
        orc.regression_layer(x, "mae", -1, 1, "default")
    with pytest.raises(ValueError):
        orc.regression_layer(x, "ssim", 0, 2, "default")
    assert orc.regression_power("ssim", 0, "default") == 1.0
    assert orc.regression_power("mae", 0, "default") == 2.0
    assert orc.regression_power("mse", 0, "default") == 4.0
    assert orc.regression_power("ssim", -1, 5) == 1.0
    assert orc.regression_power("ssim", 0, 1.5) == 1.5
