"""Input stage: oracle vs torch-generated goldens (CPU), HIP kernel vs oracle and goldens (GPU)."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import preprocess_oracle as po  # noqa: E402

GOLD = os.path.join(HERE, "golden")
CASES = ["p0_down_45x60_s37", "p1_down_120x90_s40_crop", "p2_up_20x30_s28", "p3_noresize_33x47", "p4_540x720_s518"]


def _image(g):
    h, w, seed = int(g["h"]), int(g["w"]), int(g["seed"])
    rng = np.random.Generator(np.random.PCG64(seed))
    img = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
    img[..., 1] = ((np.arange(h)[:, None] * 3 + np.arange(w)[None, :] * 2) % 256).astype(np.uint8)
    return img


def _check_against_golden(g, y, tol):
    if "out" in g.files:
        assert y.shape == g["out"].shape
        assert np.abs(y - g["out"]).max() <= tol
    else:
        assert np.abs(y[:, ::97, :] - g["rows"]).max() <= tol
        assert np.abs(y[:, :, ::101] - g["cols"]).max() <= tol
        assert np.abs(y.mean(axis=(1, 2), dtype=np.float64) - g["mean"]).max() <= 1e-6


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_transforms(name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    img = _image(g)
    crop = tuple(int(v) for v in g["crop"])
    y = po.preprocess_u8(img, tuple(int(v) for v in g["rs"]), crop if crop[0] >= 0 else None)
    _check_against_golden(g, y, 1.5e-6)  # fp32 vs fp32: differences are summation-order ulps of values of magnitude <= 2.7


def test_resized_output_size_rule():
    assert po.resized_output_size(540, 720, 518) == (518, 690)
    assert po.resized_output_size(720, 540, 518) == (690, 518)
    assert po.resized_output_size(518, 518, 518) == (518, 518)
    assert po.resized_output_size(45, 60, 37) == (37, 49)


def test_filter_table_rows_sum_to_one():
    for n_in, n_out in ((720, 690), (60, 49), (20, 28), (90, 40)):
        xmin, xsize, w = po.aa_axis_table(n_in, n_out)
        assert np.allclose(w.sum(axis=1), 1.0, atol=1e-6)
        assert (xmin >= 0).all() and (xmin + xsize <= n_in).all() and (xsize >= 1).all()


# ------------------------------------------------------------------------------------------------------------------- GPU
def _hip_preprocess(img, rs, crop, mean=po.IMAGENET_MEAN, std=po.IMAGENET_STD, pad_row=0):
    import ctypes as C
    import torch
    from crossscore_amd import _lib

    lib = _lib.load()
    h, w, _ = img.shape
    row = w * 3 + pad_row
    buf = np.zeros((h, row), np.uint8)
    buf[:, : w * 3] = img.reshape(h, w * 3)
    d_img = torch.from_numpy(buf).cuda()
    y0, x0, oh, ow = crop if crop is not None else (0, 0, rs[0], rs[1])
    out = torch.empty((3, oh, ow), dtype=torch.float32, device="cuda")
    scratch = torch.empty((h * rs[1] * 3,), dtype=torch.float32, device="cuda") if tuple(rs) != (h, w) else None
    m = (C.c_float * 3)(*mean)
    s = (C.c_float * 3)(*std)
    _lib.check(lib.cs_op_preprocess_u8(C.c_void_p(d_img.data_ptr()), h, w, row, rs[0], rs[1], y0, x0, oh, ow, m, s, C.c_void_p(out.data_ptr()),
                                      C.c_void_p(scratch.data_ptr()) if scratch is not None else None,
                                      C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    return out.cpu().numpy()


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_preprocess_vs_golden_and_oracle(name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    img = _image(g)
    rs = tuple(int(v) for v in g["rs"])
    crop = tuple(int(v) for v in g["crop"])
    crop = crop if crop[0] >= 0 else None
    y = _hip_preprocess(img, rs, crop, pad_row=5)
    _check_against_golden(g, y, 2e-6)
    ref = po.preprocess_u8(img, rs, crop)
    assert np.abs(y - ref).max() <= 2e-6
    if rs == img.shape[:2]:  # no resize: the same two IEEE divisions and one subtraction -> bit identical
        assert np.array_equal(y, ref)


@pytest.mark.gpu
def test_hip_preprocess_feeds_forward_bit_identically():
    """Without a resize the kernel's output IS the reference's input tensor, so the score map does not change by a bit."""
    import torch
    from crossscore_amd import synth
    from crossscore_amd.config import model_config
    from crossscore_amd.model import CrossScoreNet

    net = CrossScoreNet(model_config(**{"backbone.from_pretrained": "synthetic/dinov2-tiny"}))
    net.load_numpy_state_dict(synth.make_state_dict(net.arch, 3))
    net = net.cuda()
    rng = np.random.Generator(np.random.PCG64(5))
    imgs = rng.integers(0, 256, size=(3, 70, 84, 3), dtype=np.uint8)  # 1 query + 2 refs
    ref_in = np.stack([po.preprocess_u8(im) for im in imgs])
    hip_in = np.stack([_hip_preprocess(im, (70, 84), None) for im in imgs])
    assert np.array_equal(ref_in, hip_in)
    a = net(torch.from_numpy(ref_in[:1]).cuda(), torch.from_numpy(ref_in[None, 1:]).cuda(), False, 0, False)["score_map_ref_cross"]
    b = net(torch.from_numpy(hip_in[:1]).cuda(), torch.from_numpy(hip_in[None, 1:]).cuda(), False, 0, False)["score_map_ref_cross"]
    assert torch.equal(a, b)


@pytest.mark.gpu
def test_hip_preprocess_bad_arguments():
    img = np.zeros((20, 30, 3), np.uint8)
    with pytest.raises(Exception):
        _hip_preprocess(img, (20, 30), (0, 0, 21, 30))  # crop window outside the image
    with pytest.raises(Exception):
        _hip_preprocess(img, (20, 30), None, std=(0.2, 0.0, 0.2))
