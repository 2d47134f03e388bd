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


VARIANTS = {
    "no_self_attn": dict(do_self_attn=False),
    "no_short_cut": dict(do_short_cut=False),
    "tanh": dict(metric_min=-1),
    "mae_pow2": dict(metric_type="mae"),
    "mse_pow4": dict(metric_type="mse"),
    "scalar_p": dict(power_factor=0.5),
}


@pytest.mark.parametrize("name", sorted(VARIANTS))
def test_g5_flag_variants(golden_dir, name):
    g = _golden(golden_dir, "g5_tiny_flags.npz")
    over = VARIANTS[name]
    arch = dataclasses.replace(TINY, do_self_attn=over.get("do_self_attn", True))
    W = orc.to_torch(synth.make_state_dict(arch, int(g["seed"])))
    q, r = synth.make_inputs(int(g["B"]), int(g["N"]), int(g["H"]), int(g["W"]), int(g["seed"]))
    out = orc.forward(W, dict(enc_heads=TINY.enc_heads, **over), torch.from_numpy(q), torch.from_numpy(r))
    assert np.abs(out["score_map_ref_cross"].numpy() - g[name]).max() < TOL


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


@pytest.mark.skipif(not os.environ.get("CS_SLOW"), reason="ViT-B / 1036^2 oracle runs take ~10-20 s each; set CS_SLOW=1")
@pytest.mark.parametrize("name,backbone", [
    ("g2_vitb_518_n10", "facebook/dinov2-base"),
    ("g3_vits_1036_n5", "facebook/dinov2-small"),
])
def test_full_size_goldens_slow(golden_dir, name, backbone):
    test_full_size_goldens(golden_dir, name, backbone)


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
