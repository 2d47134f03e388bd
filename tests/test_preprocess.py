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
@pytest.mark.parametrize("backbone,dtype", [("synthetic/dinov2-small-2l", "fp16"), ("synthetic/dinov2-small-2l", "bf16"), ("synthetic/dinov2-base-2l", "fp16")])
def test_forwards_fed_from_uint8_are_bit_identical(backbone, dtype):
    """cs_forward_u8 / cs_encode_references_u8 / cs_forward_cached_u8 (uint8 in, tokens out: the input stage inside the patch-embedding launch)
    against cs_op_preprocess_u8 + the fp32 entry points: score maps, means, attention-weight maps and reference tokens bit for bit; images of
    different decoded sizes in one call, a placeholder reference (data = NULL = zeros before T.Normalize), a geometry the one-pass form refuses."""
    import torch
    from crossscore_amd.config import model_config
    from crossscore_amd.data import InputStage
    from crossscore_amd.model import CrossScoreNet, U8Image
    from crossscore_amd import synth

    net = CrossScoreNet(model_config(**{"backbone.from_pretrained": backbone}))
    net.load_numpy_state_dict(synth.make_state_dict(net.arch, 3))
    net.operand_dtype = dtype
    net = net.cuda()
    dev = torch.device("cuda:0")
    stage = InputStage(dev, resize_short_side=70, integer_patches=True)
    rng = np.random.Generator(np.random.PCG64(11))
    B, N = 2, 2
    # 4:3 images of two decoded sizes, both -> 70 x 93 -> the 70 x 84 window
    sizes = [(120, 160), (90, 120), (120, 160), (150, 200), (90, 120), (120, 160)]
    imgs = [rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8) for h, w in sizes]
    size = stage.geometry(*sizes[0])[1][2:]
    assert all(stage.geometry(h, w)[1][2:] == size for h, w in sizes) and size == (70, 84)
    q32 = torch.empty((B, 3) + size, device=dev)
    r32 = torch.empty((B, N, 3) + size, device=dev)
    for b in range(B):
        stage(imgs[b], q32[b])
    refs_u8 = []
    for b in range(B):
        for n in range(N):
            if (b, n) == (1, 1):  # a placeholder of a short reference list
                r32[b, n] = stage.zero_image_value[:, None, None]
                refs_u8.append(stage.placeholder(size))
            else:
                stage(imgs[B + b * N + n], r32[b, n])
                refs_u8.append(stage.describe(imgs[B + b * N + n]))
    q_u8 = stage.batch([stage.describe(imgs[b]) for b in range(B)], size)
    r_u8 = stage.batch(refs_u8, size)
    assert net.u8_input_supported(q_u8.images[0], size)
    want = net(q32, r32, True, 3, False, return_mean=True)
    got = net.forward_u8(q_u8, r_u8, True, 3, True)
    torch.cuda.synchronize()
    for k in ("score_map_ref_cross", "score_mean_ref_cross", "attn_weights_map_ref_cross"):
        assert torch.equal(got[k], want[k]), k
    tok32 = net.encode_references(r32.reshape((B * N, 3) + size))
    tok_u8 = net.encode_references_u8(r_u8)
    assert torch.equal(tok_u8, tok32)
    cached = net.forward_cached_u8(q_u8, tok_u8.reshape((B, N) + tuple(tok_u8.shape[1:])), False, 0, True)
    torch.cuda.synchronize()
    assert torch.equal(cached["score_map_ref_cross"], want["score_map_ref_cross"]) and torch.equal(cached["score_mean_ref_cross"], want["score_mean_ref_cross"])
    stats = net.forward_stats()
    assert stats["kernels"].get("patch_u8", 0) >= 1 and "patch" not in stats["kernels"], stats["kernels"]
    # the filter-table cache holds 64 geometries and is dropped as a whole behind a device synchronisation when a 65th arrives: 70 other geometries
    # later the same call rebuilds what it needs and gives the same bits
    for k in range(70):
        net.u8_input_supported(U8Image(None, 200 + k, 260, (70 + 14 * (k % 3), 98), 0, 0), (70, 98), dev)
    again = net.forward_u8(q_u8, r_u8, True, 3, True)
    torch.cuda.synchronize()
    assert torch.equal(again["score_map_ref_cross"], want["score_map_ref_cross"])
    # a 114 x down-scale: 14 pixel rows reach 1 800 source rows, more than the launch holds for even one patch per run (1 170)
    far = U8Image(None, 8000, 9600, (70, 84), 0, 0)
    assert not net.u8_input_supported(far, size, dev)


@pytest.mark.gpu
@pytest.mark.parametrize("h,w,short,crop,pad", [
    (70, 84, 0, None, 0),              # no resize (identity tables), 5 x 6 patches
    (90, 131, 70, None, 3),            # down 1.29 x, window 70 x 98 of the 70 x 101 image (integer patches), padded source rows
    (45, 60, 98, (5, 10, 84, 112), 0), # up 2.2 x, crop window off the corner
    (300, 410, 56, None, 1),           # down 5.4 x: 14 pixel rows reach 87 source rows
    (777, 1036, 518, None, 0),         # the BASELINE geometry from a 4:3 photo: 518 x 686 window, 37 x 49 patches -> two runs per patch row
    (2100, 2100, 518, None, 0),        # down 4.05 x at full width: the run is cut so that the source rows fit the LDS buffer
])
def test_patch_embedding_straight_from_uint8_is_bit_identical(h, w, short, crop, pad):
    """SURVEY.md 8f-4 as worded (uint8 in, tokens out; task/predict.py:68-93, nvs_dataset.py:218-241): the one-pass form repeats the operations of
    cs_op_preprocess_u8 inside the patch-embedding launch, so its token rows are those of the two-launch path bit for bit -- whatever the scale,
    the crop, the row padding and the number of runs a patch row is cut into."""
    import torch
    import hip_helpers as hh

    P, Cc = 14, 384
    rng = np.random.Generator(np.random.PCG64(h * 7 + w))
    imgs = rng.integers(0, 256, size=(2, h, w, 3), dtype=np.uint8)
    imgs[1, ..., 1] = ((np.arange(h)[:, None] * 3 + np.arange(w)[None, :] * 2) % 256).astype(np.uint8)
    rs = po.resized_output_size(h, w, short) if short else (h, w)
    if crop is None:
        crop = (0, 0, rs[0] - rs[0] % P, rs[1] - rs[1] % P)
    if crop[3] % 2:
        crop = (crop[0], crop[1], crop[2], crop[3] - P)  # (the one-launch patch embedding takes even widths)
    y0, x0, H, W = crop
    two = np.stack([_hip_preprocess(im, rs, crop, pad_row=pad) for im in imgs])
    wgt = torch.from_numpy((rng.standard_normal((Cc, 3, P, P)) / 24).astype(np.float32)).cuda()
    bias = torch.from_numpy(rng.standard_normal(Cc).astype(np.float32)).cuda()
    pos = torch.from_numpy(rng.standard_normal((1 + (H // P) * (W // P), Cc)).astype(np.float32)).cuda()
    ref = hh.patch_embed_fused(torch.from_numpy(two).cuda(), wgt, bias, pos, P)
    buf = np.zeros((2, h, w * 3 + pad), np.uint8)
    buf[:, :, : w * 3] = imgs.reshape(2, h, w * 3)
    got = hh.patch_embed_fused_u8(torch.from_numpy(buf).cuda(), rs, (y0, x0, H, W, w), po.IMAGENET_MEAN, po.IMAGENET_STD, wgt, bias, pos, P)
    torch.cuda.synchronize()
    assert torch.equal(got, ref)


@pytest.mark.gpu
def test_hip_preprocess_bad_arguments():
    img = np.zeros((20, 30, 3), np.uint8)
    with pytest.raises(Exception):
        _hip_preprocess(img, (20, 30), (0, 0, 21, 30))  # crop window outside the image
    with pytest.raises(Exception):
        _hip_preprocess(img, (20, 30), None, std=(0.2, 0.0, 0.2))
