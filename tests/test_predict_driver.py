"""Predict-compatible driver: host logic on CPU; the whole image-directory -> PNG/CSV run on the GPU against
the oracle pipeline (oracle input transforms -> fp32 oracle forward -> oracle writers)."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from crossscore_amd import data as csdata  # noqa: E402
from crossscore_amd import synth  # noqa: E402
from crossscore_amd.config import load_config  # noqa: E402
from oracle import preprocess_oracle as po  # noqa: E402
from oracle import writers_oracle as wo  # noqa: E402

TINY = "synthetic/dinov2-tiny"


def _make_scene(root, n_query=3, n_ref=4, h=70, w=84, seed=7):
    """<root>/<method>/<dataset>/<res>/<scene>/test/ours_1000/{renders,gt}/frame_XXXXX.png, the depth the reference's CSV
    grouping indexes into (score_summariser.py:180-212)."""
    from PIL import Image

    rng = np.random.Generator(np.random.PCG64(seed))
    base = os.path.join(root, "gaussian", "mfr", "res_540", "s00001", "test", "ours_1000")
    qd, rd = os.path.join(base, "renders"), os.path.join(base, "gt")
    os.makedirs(qd)
    os.makedirs(rd)
    yy, xx = np.mgrid[0:h, 0:w]
    for d, n, off in ((qd, n_query, 0), (rd, n_ref, 100)):
        for i in range(n):
            img = np.stack([(xx * 3 + i * 17 + off) % 256, (yy * 2 + i * 29) % 256, (xx + yy + i * 11) % 256], axis=2).astype(np.uint8)
            img = (img.astype(np.int32) + rng.integers(-20, 21, size=img.shape)).clip(0, 255).astype(np.uint8)
            Image.fromarray(img).save(os.path.join(d, f"frame_{i:05}.png"))
    return qd, rd


# ------------------------------------------------------------------------------------------------------------------- CPU
def test_resize_rule_and_geometry():
    st = csdata.InputStage.__new__(csdata.InputStage)
    st.resize_short_side, st.crop_size, st.integer_patches, st.patch = 518, None, False, 14
    assert st.geometry(540, 720) == ((518, 690), (0, 0, 518, 690))
    st.crop_size = 518
    assert st.geometry(540, 720) == ((518, 690), (0, 0, 518, 518))
    st.crop_size, st.integer_patches, st.resize_short_side = None, True, -1
    assert st.geometry(75, 90) == ((75, 90), (0, 0, 70, 84))
    assert csdata.resized_output_size(720, 540, 518) == po.resized_output_size(720, 540, 518) == (690, 518)


def test_reference_sampling_rules(tmp_path):
    qd, rd = _make_scene(str(tmp_path), n_query=2, n_ref=4)
    items = csdata.SimpleReferenceItems(qd, rd, {"strategy": "random", "cross": 3, "deterministic": True})
    assert len(items) == 2 and items.query_paths == sorted(items.query_paths)
    it = items[1]
    assert it["query/img"].endswith("frame_00001.png") and it["reference/cross/imgs"] == items.reference_paths[:3]
    np.random.seed(1)
    a = csdata.sample_references(items.reference_paths, 3, False)
    np.random.seed(1)
    assert a == np.random.choice(items.reference_paths, 3, replace=False).tolist()  # the reference's call on the same RNG state
    padded = csdata.sample_references(items.reference_paths[:2], 5, True)
    assert sorted(padded) == sorted(items.reference_paths[:2] + [csdata.EMPTY] * 3)
    with pytest.raises(NotImplementedError):
        csdata.SimpleReferenceItems(qd, rd, {"strategy": "nearest", "cross": 3, "deterministic": True})


def test_out_dir_naming(tmp_path, monkeypatch):
    from crossscore_amd.predict import resolve_out_dir

    monkeypatch.chdir(tmp_path)
    cfg = load_config("default_predict")
    assert resolve_out_dir(cfg, now="T") == "log/T/predict_empty_ckpt/T"
    cfg.trainer.ckpt_path_to_load = str(tmp_path / "run" / "ckpt" / "last.ckpt")
    cfg.alias = "abc"
    assert resolve_out_dir(cfg, now="T") == f"{tmp_path}/run/predict/T_abc"
    cfg.logger.predict.out_dir = str(tmp_path / "explicit")
    assert resolve_out_dir(cfg, now="T") == f"{tmp_path}/explicit_abc"


def test_writer_oracle_matches_matplotlib():
    import matplotlib
    import matplotlib.pyplot as plt

    rng = np.random.Generator(np.random.PCG64(3))
    score = rng.uniform(-0.2, 1.2, size=(37, 41)).astype(np.float32)
    score[0, :4] = (0.0, 1.0, -0.5, 2.0)
    for vr in ((0, 1), (-1, 1)):
        ref = (matplotlib.colormaps["turbo"](plt.Normalize(vmin=vr[0], vmax=vr[1])(score))[:, :, :3] * 255.0).astype(np.uint8)  # gray2rgb
        assert np.array_equal(wo.rgb(score, vr, wo.turbo_table()), ref)
    s01 = np.clip(score, 0, 1)
    assert np.array_equal(wo.gray16(s01, [0, 1]), (s01 * 65535).astype(np.int32).astype(np.uint16))
    assert np.array_equal(wo.gray16(s01, [-1, 1]), ((s01 + 1) * 32767).astype(np.int32).astype(np.uint16))


def test_attn2rgb_matches_matplotlib():
    import matplotlib
    import matplotlib.pyplot as plt
    from crossscore_amd.writers import attn2rgb, colormap_table

    rng = np.random.Generator(np.random.PCG64(4))
    a = rng.dirichlet(np.ones(37 * 37)).astype(np.float32).reshape(37, 37)
    a[0, :3] = (0.0, 1.0, 1e-9)
    eps = 1e-8                                              # utils/misc/image.py:55-77, with colormaps[...] for the removed cm.get_cmap
    m = a.clip(0, 1); m = (m + eps).clip(0, 1); m = np.log(m) - np.log(eps)
    ref = (matplotlib.colormaps["turbo"](plt.Normalize(vmin=0, vmax=-np.log(eps))(m))[:, :, :3] * 255.0).astype(np.uint8)
    assert np.array_equal(attn2rgb(a, colormap_table("turbo")), ref)


def test_summariser_csv_format(tmp_path):
    from crossscore_amd.writers import ScoreSummariser, name_stem

    p = "/data/gaussian/mfr/res_540/s00001/test/ours_1000/renders/frame_00002.png"
    assert name_stem(p) == "s00001_test_ours_1000_renders_frame_00002"
    s = ScoreSummariser("ssim", 0, tmp_path)
    batch = {"item_paths": {"query/img": [p, p.replace("00002", "00001")]}}
    s.update(batch, {"score_map_ref_cross": torch.tensor([[[0.25, 0.75]], [[0.5, 0.5]]])})
    (path,) = s.summarise()
    assert path.endswith("score_summary/mfr/gaussian.csv")
    lines = open(path).read().splitlines()
    assert lines[0] == "scene_name,rendered_dir,image_name,pred_ssim_0_1"
    assert lines[1] == "s00001,data/gaussian/mfr/res_540/s00001/test/ours_1000,00001.png,0.5000"   # sorted by image name
    assert lines[2] == "s00001,data/gaussian/mfr/res_540/s00001/test/ours_1000,00002.png,0.5000"
    with pytest.raises(ValueError):
        s.update(batch, {"score_map_a": torch.zeros(2, 1, 1), "score_map_b": torch.zeros(2, 1, 1)})


# ------------------------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("colour_mode", ["gray", "rgb"])
def test_predict_run_matches_oracle_pipeline(tmp_path, colour_mode):
    from PIL import Image
    from crossscore_amd.config import model_config
    from crossscore_amd.model import CrossScoreNet
    from crossscore_amd.predict import predict
    from oracle import crossscore_oracle as orc

    qd, rd = _make_scene(str(tmp_path / "data"), n_query=3, n_ref=4)
    arch = CrossScoreNet(model_config(**{"backbone.from_pretrained": TINY})).arch
    sd = synth.make_state_dict(arch, 5)
    ckpt = tmp_path / "run" / "ckpt" / "last.ckpt"
    os.makedirs(ckpt.parent)
    torch.save({"state_dict": {"model." + k: torch.from_numpy(v) for k, v in sd.items()}, "epoch": 3}, ckpt)  # Lightning layout
    cfg = load_config("default_predict", [f"data.dataset.query_dir={qd}", f"data.dataset.reference_dir={rd}",
                                          f"trainer.ckpt_path_to_load={ckpt}", f"model.backbone.from_pretrained={TINY}",
                                          "this_main.resize_short_side=56", "data.neighbour_config.cross=2",
                                          "data.neighbour_config.deterministic=True", "data.loader.validation.batch_size=2",
                                          f"logger.predict.write.config.score_map_colour_mode={colour_mode}",
                                          "logger.predict.write.flag.item_path_json=True", "model.need_attn_weights=True",
                                          "logger.predict.write.flag.attn_weights=True", "model.need_attn_weights_head_id=1"])
    res = predict(cfg, now="NOW")
    out_dir = str(tmp_path / "run" / "predict" / "NOW")
    assert res["out_dir"] == out_dir and len(res["rows"]) == 3

    # the oracle pipeline on the CPU: reference transforms -> fp32 forward -> reference writers
    W = orc.to_torch(sd)
    refs_paths = sorted(os.listdir(rd))[:2]
    r = np.stack([po.preprocess_u8(np.array(Image.open(os.path.join(rd, p))), (56, 67)) for p in refs_paths])[None]
    table = wo.turbo_table()
    means = []
    for i, qp in enumerate(sorted(os.listdir(qd))):
        q = po.preprocess_u8(np.array(Image.open(os.path.join(qd, qp))), (56, 67))[None]
        score = orc.forward(W, dict(enc_heads=arch.enc_heads, pos_interp_legacy=True), torch.from_numpy(q), torch.from_numpy(r), False, 0)["score_map_ref_cross"][0].numpy()
        means.append(float(score.mean()))
        B, b = divmod(i, 2)
        name = f"r0_B{B:04}_b{b:03}_s00001_test_ours_1000_renders_frame_{i:05}.png"
        got = np.array(Image.open(os.path.join(out_dir, "batch", "score_map_ref_cross", name)))
        assert got.shape[:2] == (56, 56)  # 4 x 4 patches of 14
        if colour_mode == "gray":
            want = wo.gray16(score, [-1, 1])
            assert got.dtype == np.uint16 or got.dtype == np.int32
            assert np.abs(got.astype(np.int64) - want.astype(np.int64)).mean() / 32767 < 1e-3  # score-map MAE bound, in map units
        else:
            want = wo.rgb(score, [0, 1], table)
            assert got.shape == want.shape and (np.abs(got.astype(int) - want.astype(int)).max(axis=2) > 24).mean() < 0.02
        assert os.path.exists(os.path.join(out_dir, "batch", "image_query", name))
        assert len(os.listdir(os.path.join(out_dir, "batch", "image_reference", name[:-4], "cross"))) == 2
        att = sorted(os.listdir(os.path.join(out_dir, "batch", "attn_weights", name[:-4], "cross")))
        assert att == ["ref00_s00001_test_ours_1000_gt_frame_00000.png", "ref01_s00001_test_ours_1000_gt_frame_00001.png"]
        assert np.array(Image.open(os.path.join(out_dir, "batch", "attn_weights", name[:-4], "cross", att[0]))).shape == (4, 4, 3)
    # de-normalised query image round-trips to the resized uint8 image (u8 truncation: at most one level off)
    q0 = np.array(Image.open(os.path.join(out_dir, "batch", "image_query", "r0_B0000_b000_s00001_test_ours_1000_renders_frame_00000.png")))
    x0 = po.preprocess_u8(np.array(Image.open(os.path.join(qd, "frame_00000.png"))), (56, 67))
    back = ((x0 * np.array(po.IMAGENET_STD, np.float32)[:, None, None] + np.array(po.IMAGENET_MEAN, np.float32)[:, None, None]) * 255.0)
    assert np.abs(q0.astype(int) - back.transpose(1, 2, 0).astype(np.uint8).astype(int)).max() <= 1
    csv_path = os.path.join(out_dir, "score_summary", "mfr", "gaussian.csv")
    lines = open(csv_path).read().splitlines()
    assert lines[0] == "scene_name,rendered_dir,image_name,pred_ssim_0_1" and len(lines) == 4
    for i, ln in enumerate(lines[1:]):
        f = ln.split(",")
        assert f[0] == "s00001" and f[2] == f"{i:05}.png" and abs(float(f[3]) - means[i]) < 2e-3
    assert os.path.exists(os.path.join(out_dir, "batch", "item_path_json", "r0_B0001.json"))


@pytest.mark.gpu
def test_reference_token_cache_gives_identical_outputs(tmp_path):
    """this_main.cache_reference_tokens on / off: same files, same bytes in the score maps, same CSV."""
    from PIL import Image
    from crossscore_amd.config import model_config
    from crossscore_amd.model import CrossScoreNet
    from crossscore_amd.predict import predict

    qd, rd = _make_scene(str(tmp_path / "data"), n_query=5, n_ref=4)
    arch = CrossScoreNet(model_config(**{"backbone.from_pretrained": TINY})).arch
    sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict(arch, 6).items()}
    outs = {}
    for flag in (True, False):
        cfg = load_config("default_predict", [f"data.dataset.query_dir={qd}", f"data.dataset.reference_dir={rd}",
                                              f"model.backbone.from_pretrained={TINY}", "this_main.resize_short_side=56",
                                              "data.neighbour_config.cross=3", "data.loader.validation.batch_size=2",
                                              f"logger.predict.out_dir={tmp_path}/out_{flag}", f"this_main.cache_reference_tokens={flag}",
                                              "logger.predict.write.config.score_map_colour_mode=gray"])
        np.random.seed(0)
        outs[flag] = predict(cfg, state_dict=sd, now="T")
    a, b = outs[True], outs[False]
    assert [r[:3] for r in a["rows"]] == [r[:3] for r in b["rows"]] and [r[3] for r in a["rows"]] == [r[3] for r in b["rows"]]
    fa = sorted(f[len(a["out_dir"]):] for f in a["files"])
    fb = sorted(f[len(b["out_dir"]):] for f in b["files"])
    assert fa == fb and len(fa) > 10
    for rel in fa:
        if rel.endswith(".png"):
            assert np.array_equal(np.array(Image.open(a["out_dir"] + rel)), np.array(Image.open(b["out_dir"] + rel))), rel


@pytest.mark.gpu
@pytest.mark.parametrize("use_cache", [True, False])
def test_fused_input_stage_gives_identical_outputs(tmp_path, use_cache):
    """this_main.fused_input_stage (uint8 in, tokens out: SURVEY.md 8f-4 as worded) on / off at the ViT-S width, with and without the reference
    token cache: same files, same bytes in the score maps, the same CSV values.  "auto" takes the one-pass form exactly when the writers do not
    ask for the processed images; insisting on it while they do is an error."""
    from PIL import Image
    from crossscore_amd.predict import predict

    back = "synthetic/dinov2-small-2l"
    qd, rd = _make_scene(str(tmp_path / "data"), n_query=5, n_ref=4, h=70, w=90)  # -> 56 x 72 (the one-launch patch embedding takes even widths)
    from crossscore_amd.config import model_config
    from crossscore_amd.model import CrossScoreNet
    arch = CrossScoreNet(model_config(**{"backbone.from_pretrained": back})).arch
    sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict(arch, 6).items()}
    common = [f"data.dataset.query_dir={qd}", f"data.dataset.reference_dir={rd}", f"model.backbone.from_pretrained={back}",
              "this_main.resize_short_side=56", "data.neighbour_config.cross=3", "data.loader.validation.batch_size=2",
              f"this_main.cache_reference_tokens={use_cache}", "logger.predict.write.config.score_map_colour_mode=gray",
              "logger.predict.write.flag.image_query=False", "logger.predict.write.flag.image_reference=False"]
    outs = {}
    for mode in ("auto", False):
        np.random.seed(0)
        outs[mode] = predict(load_config("default_predict", common + [f"logger.predict.out_dir={tmp_path}/out_{mode}", f"this_main.fused_input_stage={mode}"]),
                             state_dict=sd, now="T")
    a, b = outs["auto"], outs[False]
    assert a["input_stage"] == "one-pass (uint8 in, tokens out)" and b["input_stage"].startswith("two-launch")
    assert [r[:3] for r in a["rows"]] == [r[:3] for r in b["rows"]] and [r[3] for r in a["rows"]] == [r[3] for r in b["rows"]]
    fa = sorted(f[len(a["out_dir"]):] for f in a["files"])
    fb = sorted(f[len(b["out_dir"]):] for f in b["files"])
    assert fa == fb and len(fa) > 5
    for rel in fa:
        if rel.endswith(".png"):
            assert np.array_equal(np.array(Image.open(a["out_dir"] + rel)), np.array(Image.open(b["out_dir"] + rel))), rel
    # the reference's default flags write the processed images: "auto" then stays with the two-launch stage, True is refused
    with_imgs = [c for c in common if "flag.image" not in c]
    c = predict(load_config("default_predict", with_imgs + [f"logger.predict.out_dir={tmp_path}/out_imgs"]), state_dict=sd, now="T")
    assert c["input_stage"].startswith("two-launch") and [r[3] for r in c["rows"]] == [r[3] for r in a["rows"]]
    with pytest.raises(ValueError):
        predict(load_config("default_predict", with_imgs + [f"logger.predict.out_dir={tmp_path}/out_x", "this_main.fused_input_stage=True"]), state_dict=sd, now="T")


@pytest.mark.gpu
def test_output_stage_kernels_bit_exact():
    from crossscore_amd.writers import ScoreMapEncoder

    rng = np.random.Generator(np.random.PCG64(9))
    score = rng.uniform(-0.1, 1.1, size=(2, 57, 63)).astype(np.float32)
    score[0, 0, :4] = (0.0, 1.0, -0.5, 2.0)
    t = torch.from_numpy(score).cuda()
    for mtype, mmin, vr_i, vr_v in (("ssim", 0, [-1, 1], [0, 1]), ("ssim", -1, [-1, 1], [-1, 1]), ("mae", 0, [0, 1], [0, 1])):
        s = np.clip(score, 0, 1) if vr_i == [0, 1] else score
        g = ScoreMapEncoder(mtype, mmin, 1, "gray", t.device)(torch.from_numpy(s).cuda())
        assert np.array_equal(g, wo.gray16(s, vr_i))
        c = ScoreMapEncoder(mtype, mmin, 1, "rgb", t.device)(t)
        assert np.array_equal(c, wo.rgb(score, vr_v, wo.turbo_table()))
