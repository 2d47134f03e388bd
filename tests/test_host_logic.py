"""CPU-side checks: config tree, drop-in module surface / state-dict layout, C-ABI exports, batch sharding."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from crossscore_amd import _lib, synth
from crossscore_amd.config import load_config, model_config
from crossscore_amd.model import CrossScoreNet, regression_activation
from crossscore_amd.parallel import shard_bounds

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_config_defaults_match_reference_keys():
    cfg = load_config()
    assert cfg.model.patch_size == 14 and cfg.model.do_reference_cross is True
    assert cfg.model.backbone.from_pretrained == "facebook/dinov2-small"
    assert (cfg.model.pos_enc.multi_view.h, cfg.model.pos_enc.multi_view.w) == (40, 40)
    assert cfg.model.pos_enc.multi_view.interpolate_mode == "bilinear"
    assert cfg.model.predict.metric.type == "ssim" and cfg.model.predict.metric.min == 0 and cfg.model.predict.metric.max == 1
    assert cfg.model.predict.metric.power_factor == "default"
    assert cfg.model.need_attn_weights is False and cfg.model.need_attn_weights_head_id == 0
    assert cfg.data.loader.validation.batch_size == 8 and cfg.data.neighbour_config.cross == 5
    assert cfg.this_main.resize_short_side == 518 and cfg.trainer.precision == "16-mixed"
    assert cfg.lightning.seed == 1


def test_config_overrides():
    cfg = load_config(overrides=["model.backbone.from_pretrained=facebook/dinov2-base", "trainer.devices=[0,1]",
                                 "model.predict.metric.power_factor=1.5", "alias=run1"])
    assert cfg.model.backbone.from_pretrained == "facebook/dinov2-base"
    assert cfg.trainer.devices == [0, 1] and cfg.model.predict.metric.power_factor == 1.5 and cfg.alias == "run1"
    with pytest.raises(ValueError):
        load_config(overrides=["novalue"])


def test_regression_activation_table():
    assert regression_activation("ssim", 0, 1, "default") == (0, 1.0)
    assert regression_activation("mae", 0, 1, "default") == (0, 2.0)
    assert regression_activation("mse", 0, 1, "default") == (0, 4.0)
    assert regression_activation("ssim", -1, 1, 5) == (1, 1.0)
    assert regression_activation("ssim", 0, 1, 1.5) == (0, 1.5)
    for bad in [("psnr", 0, 1), ("mae", -1, 1), ("ssim", 0, 2), ("ssim", 0.5, 1)]:
        with pytest.raises(ValueError):
            regression_activation(*bad, "default")


def test_state_dict_layout_small():
    net = CrossScoreNet(model_config())
    sd = net.state_dict()
    spec = {n: s for n, s, _, _ in synth.state_dict_spec(synth.BACKBONES["facebook/dinov2-small"])}
    assert set(sd) == set(spec)
    assert all(tuple(sd[k].shape) == spec[k] for k in spec)
    # spot checks against the checkpoint layout listed in SURVEY.md 8b
    assert tuple(sd["backbone.embeddings.position_embeddings"].shape) == (1, 1370, 384)
    assert tuple(sd["backbone.encoder.layer.11.mlp.fc1.weight"].shape) == (1536, 384)
    assert tuple(sd["ref_cross.attn.layers.1.multihead_attn.in_proj_weight"].shape) == (1152, 384)
    assert tuple(sd["ref_cross.head.2.weight"].shape) == (196, 384)
    assert tuple(sd["pos_enc_fn.PE"].shape) == (1, 40, 40, 384)
    assert sum(v.numel() for k, v in sd.items() if k != "img_mean_std") == 25_855_684 - 0  # SURVEY.md section 5 param count
    assert torch.allclose(net.img_mean_std, torch.tensor(synth.IMAGENET_MEAN_STD))
    assert not any(p.requires_grad for p in net.parameters())


def test_state_dict_strict_load_and_lightning_prefix(tmp_path):
    arch = synth.BACKBONES["synthetic/dinov2-tiny"]
    net = CrossScoreNet(model_config(**{"backbone.from_pretrained": "synthetic/dinov2-tiny"}))
    sd = synth.make_state_dict(arch, 3)
    net.load_numpy_state_dict(sd)
    assert np.array_equal(net.state_dict()["ref_cross.head.2.bias"].numpy(), sd["ref_cross.head.2.bias"])
    bad = dict(sd)
    bad.pop("pos_enc_fn.PE")
    with pytest.raises(RuntimeError):
        net.load_numpy_state_dict(bad)
    # Lightning checkpoint layout: state_dict keys prefixed "model."
    from crossscore_amd.model import load_lightning_checkpoint
    path = str(tmp_path / "fake.ckpt")
    torch.save({"state_dict": {"model." + k: torch.from_numpy(v) for k, v in sd.items()}, "hyper_parameters": {}, "epoch": 3}, path)
    net2 = CrossScoreNet(model_config(**{"backbone.from_pretrained": "synthetic/dinov2-tiny"}))
    net2.load_state_dict(load_lightning_checkpoint(path), strict=True)
    assert torch.equal(net2.state_dict()["pos_enc_fn.PE"], net.state_dict()["pos_enc_fn.PE"])


def test_no_self_attn_variant_has_no_self_attn_keys():
    net = CrossScoreNet(model_config(**{"backbone.from_pretrained": "synthetic/dinov2-tiny", "decoder_do_self_attn": False}))
    assert not any("self_attn" in k for k in net.state_dict())


def test_do_reference_cross_off_returns_an_empty_result_like_the_reference():
    """model.do_reference_cross=False (task/core.py:55,89): no CrossReferenceNet in the module (no `ref_cross.*` state-dict entries) and an EMPTY
    result dict from forward -- the reference computes the features and drops them; here nothing is launched (works on CPU tensors too)."""
    net = CrossScoreNet(model_config(do_reference_cross=False))
    keys = set(net.state_dict().keys())
    assert keys and not any(k.startswith("ref_cross.") for k in keys) and "pos_enc_fn.PE" in keys and "backbone.layernorm.weight" in keys
    full = set(CrossScoreNet(model_config()).state_dict().keys())
    assert keys == {k for k in full if not k.startswith("ref_cross.")}
    q = torch.zeros(2, 3, 28, 28)
    assert net(q, torch.zeros(2, 3, 3, 28, 28), False, 0, False) == {}
    assert net(q, None, False, 0, False) == {}  # (get_featmaps takes ref_cross_imgs=None: task/core.py:130)
    with pytest.raises(ValueError):
        net(q, torch.zeros(2, 3, 3, 28, 42), False, 0, False)  # images that would not concatenate


def test_forward_refuses_cpu_tensors():
    net = CrossScoreNet(model_config(**{"backbone.from_pretrained": "synthetic/dinov2-tiny"}))
    q = torch.zeros(1, 3, 70, 70)
    with pytest.raises(RuntimeError):
        net(q, torch.zeros(1, 2, 3, 70, 70), False, 0, False)
    with pytest.raises(ValueError):
        net(q, None, False, 0, False)


def test_pipeline_refuses_a_cpu_module_and_bad_depth():
    """crossscore_amd.pipeline.ForwardPipeline: batches in flight need the module on a GPU (no CPU fallback) and a depth >= 1."""
    from crossscore_amd.pipeline import ForwardPipeline

    net = CrossScoreNet(model_config(**{"backbone.from_pretrained": "synthetic/dinov2-tiny"}))
    with pytest.raises(RuntimeError):
        ForwardPipeline(net, depth=2)
    with pytest.raises(ValueError):
        ForwardPipeline(net, depth=0)


def test_invalid_metric_config_raises_like_reference():
    with pytest.raises(ValueError):
        CrossScoreNet(model_config(**{"predict.metric.type": "psnr"}))
    with pytest.raises(ValueError):
        CrossScoreNet(model_config(**{"predict.metric.type": "mae", "predict.metric.min": -1}))


def test_library_exports_every_declared_symbol():
    """The .so loads and exports exactly what include/crossscore_hip.h declares (no compute without a GPU)."""
    lib = _lib.load()
    hdr = open(os.path.join(REPO, "include", "crossscore_hip.h")).read()
    declared = set(re.findall(r"\b(cs_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name
    # cs_create validates on the host before touching the device
    cc = _lib.CsConfig(hidden=100, enc_layers=1, enc_heads=1, mlp_ratio=4, patch=14, pos_grid=5, pe_h=40, pe_w=40, dec_layers=2,
                       dec_heads=8, do_self_attn=1, do_short_cut=1, act=0, pow_p=1.0, enc_chunk_images=0, ln_fold=0, lanes=0)
    assert not lib.cs_create(ctypes.byref(cc))
    assert b"hidden" in lib.cs_last_error()


def test_product_package_never_imports_oracle():
    src_dir = os.path.join(REPO, "crossscore_amd")
    for root, _, files in os.walk(src_dir):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(root, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, f


def test_shard_bounds_cover_batch():
    for B in (1, 7, 8, 128, 130):
        for G in (1, 2, 3, 4, 8):
            spans = [shard_bounds(B, G, r) for r in range(G)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(hi - lo for lo, hi in spans) - min(hi - lo for lo, hi in spans) <= 1
    assert shard_bounds(128, 8, 3) == (48, 64)


def test_synth_shard_inputs_bitwise_equal_full_batch():
    q, r = synth.make_inputs(4, 2, 28, 42, 9)
    q1, r1 = synth.make_inputs_shard(2, 4, 2, 28, 42, 9)
    assert np.array_equal(q[2:], q1) and np.array_equal(r[2:], r1)


def _run_workers(world, script):
    env = dict(os.environ, PYTHONPATH=REPO + os.pathsep + os.environ.get("PYTHONPATH", ""))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", "29731", script]
    return subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)


def test_gloo_world2_shard_gather(tmp_path):
    """world_size-2 gloo run of the N>1 control path: shard bounds, ragged all_gather of per-image means,
    max-over-ranks timing.  Each rank 'scores' its shard with a stand-in (the seeded input mean) so the gathered
    vector must equal the single-process result bit for bit."""
    script = tmp_path / "worker.py"
    script.write_text(
        "import torch, numpy as np\n"
        "from crossscore_amd import synth\n"
        "from crossscore_amd.parallel import init_from_env, shard_bounds, gather_means, max_over_ranks, barrier\n"
        "rank, local, world = init_from_env('gloo')\n"
        "B = 5\n"
        "lo, hi = shard_bounds(B, world, rank)\n"
        "q, r = synth.make_inputs_shard(lo, hi, 2, 28, 28, 4)\n"
        "local_means = torch.from_numpy(q.reshape(hi - lo, -1).mean(1))\n"
        "barrier()\n"
        "allm = gather_means(local_means, B)\n"
        "t = max_over_ranks(float(rank + 1))\n"
        "qf, _ = synth.make_inputs(B, 2, 28, 28, 4)\n"
        "ref = torch.from_numpy(qf.reshape(B, -1).mean(1))\n"
        "assert torch.equal(allm, ref), (allm, ref)\n"
        "assert t == float(world)\n"
        "if rank == 0: print('GLOO_OK', world)\n")
    res = _run_workers(2, str(script))
    assert res.returncode == 0 and "GLOO_OK 2" in res.stdout, res.stdout[-2000:]


def test_bench_launches_its_own_ranks_and_takes_the_slowest(tmp_path):
    """`python bench.py --gpus 2` with no launcher around it must start two ranks itself (task/predict.py:119-135: one process
    per device).  Rehearsed on the CPU with gloo and a stub forward: the barrier / max-over-ranks / rank-0 line are the real ones."""
    import json

    env = dict(os.environ, PYTHONPATH=REPO + os.pathsep + os.environ.get("PYTHONPATH", ""))
    env.pop("WORLD_SIZE", None)
    res = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--plumbing-test", "--steps", "4", "--warmup", "1",
                          "--global-batch", "5"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout  # rank 0 prints ONE line
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 4 and r["data"] == "stub"
    assert r["ms_per_step"] >= 4.0  # rank 1's stub step sleeps 4 ms, rank 0's 2 ms: the slowest rank sets the time
    assert r["gathered_means"] == [0.0, 0.0, 1.0, 1.0]
    # the line proves by itself which ranks took part: one record per rank, gathered over the process group
    seen = r["ranks_seen"]
    assert [x["rank"] for x in seen] == [0, 1] and [x["local_rank"] for x in seen] == [0, 1]
    assert len({x["pid"] for x in seen}) == 2 and all(x["ms_per_step"] > 0 for x in seen)
    assert seen[1]["ms_per_step"] > seen[0]["ms_per_step"]  # rank 1's stub is the slow one
    # every rank's record carries what a forward costs its CPU thread (cs_forward_stats in the real line: VERDICT r4 next #6)
    assert all("host_enqueue_ms_per_forward" in x and "launches_per_forward" in x for x in seen)
    assert seen[1]["host_enqueue_ms_per_forward"] > seen[0]["host_enqueue_ms_per_forward"] > 0
    assert r["process_group"]["backend"] == "gloo" and r["process_group"]["world_size"] == 2
    # the strong-scaling leg (BASELINE configs[3]: a FIXED global batch sharded over the ranks, task/predict.py:119-135): ragged shards by
    # parallel.shard_bounds, every rank's shard in the line, throughput = global batch over the slowest rank's time
    s4 = r["scaling_cfg4"]
    assert s4["mode"] == "strong" and s4["global_batch"] == 5
    assert [x["shard"] for x in s4["ranks_seen"]] == [[0, 3], [3, 5]] and [x["items"] for x in s4["ranks_seen"]] == [3, 2]
    assert s4["ms_per_step"] >= 3.0 and s4["ranks_seen"][0]["ms_per_step"] > s4["ranks_seen"][1]["ms_per_step"]  # rank 0 holds 3 items at 1 ms
    assert abs(s4["value"] - 5 / (s4["ms_per_step"] / 1e3)) < 1e-6 * s4["value"]


def test_bench_names_the_failing_rank_on_stderr():
    """A rank that dies inside bench.py prints ONE line with its rank / local rank / pid / device / source position before the launcher's own
    summary, so that a driver-side N-GPU run is debuggable from the tail of its output."""
    env = dict(os.environ, PYTHONPATH=REPO + os.pathsep + os.environ.get("PYTHONPATH", ""), CS_PLUMBING_FAIL_RANK="1")
    env.pop("WORLD_SIZE", None)
    res = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--plumbing-test", "--steps", "2", "--warmup", "1"],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert res.returncode != 0
    lines = [ln for ln in res.stderr.splitlines() if ln.startswith("bench.py FAILED")]
    mine = [ln for ln in lines if "rank=1 local_rank=1 world=2" in ln]
    assert len(mine) == 1 and "rehearsed failure" in mine[0] and "plumbing_test" in mine[0], res.stderr[-2000:]
    # rank 0 is either ended by the launcher (no line) or sees its peer's connection close inside the next collective and says so itself:
    # at most one line per rank, and the only other rank is 0
    others = [ln for ln in lines if ln not in mine]
    assert len(others) <= 1 and all("rank=0 local_rank=0 world=2" in ln and "rehearsed failure" not in ln for ln in others), res.stderr[-2000:]


def test_predict_config_defaults_to_the_reference_environments_pos_embed_resize():
    """default_predict.yaml (the config of the task/predict.py stand-in) selects the scale_factor form of the pinned transformers 4.33.3;
    a bare model config keeps the installed release's size form; an override wins."""
    from crossscore_amd.config import load_config
    assert load_config("default_predict").model.backbone.pos_embed_interpolation == "scale_factor"
    assert load_config("default_predict").model.backbone.from_pretrained == "facebook/dinov2-small"  # (the group's keys survive the merge)
    assert "pos_embed_interpolation" not in model_config().model.backbone
    assert load_config("default_predict", ["model.backbone.pos_embed_interpolation=size"]).model.backbone.pos_embed_interpolation == "size"
    assert CrossScoreNet(load_config("default_predict"))._pos_legacy and not CrossScoreNet(model_config())._pos_legacy


def test_asm_audits_of_the_inline_asm_kernels(tmp_path):
    """gemm256.hip keeps inline-asm global loads in flight (residual rows) and both gemm256.hip and panel.hip read LDS by inline asm behind
    counted waits: hipcc does not know when those registers are written.  The audits scan the build's .s for any instruction that touches
    such a register before its marker / wait (a compiler copy there faulted on the GPU in r3: phi copies of in-flight registers)."""
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    from crossscore_amd import build as b
    for src, audits in (("gemm256.hip", ("asm_audit_gl.py", "asm_audit.py")), ("panel.hip", ("asm_audit.py",))):
        out = str(tmp_path / (src + ".s"))
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"] + b.EXTRA_FLAGS.get(src, []) + [
            "-S", "--cuda-device-only", "-o", out, os.path.join(REPO, "crossscore_amd", "csrc", src)]
        res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
        assert res.returncode == 0, res.stdout[-2000:]
        text = open(out).read()
        lines = text.splitlines()
        spills = [int(ln.split(":")[1]) for ln in lines if ".vgpr_spill_count:" in ln]
        # no spilled register in gemm256.hip; panel.hip (two roles in one kernel at the 256-register limit, r4): at most one scratch slot of <= 2
        # dwords per kernel -- the thread id carried past the B-wave path for the A-wave path, stored once at entry -- and never inside a loop
        assert spills and max(spills) <= (2 if src == "panel.hip" else 0), (src, spills)
        block = ""
        for ln in lines:
            if ln.startswith(".LBB") or ln.startswith("_Z"):
                block = ln
            if "scratch_" in ln:
                assert "Loop" not in block, (src, "scratch access inside a loop", block, ln)
        for a in audits:
            r = subprocess.run([sys.executable, os.path.join(REPO, "tools", a), out], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
            assert "violations: 0" in r.stdout, (src, a, r.stdout[-1500:])


def test_four_wave_panel_kernel_has_no_private_segment(tmp_path):
    """csrc/panel4.hip runs its loop at the 512-register limit.  Round 6 measured the SAME loop instructions at half the speed in a build that
    used scratch (27 dwords spilled OUTSIDE the loop: EXPERIMENTS.md); every variant of the kernel must therefore build with no private
    segment and no spilled register."""
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    from crossscore_amd import build as b
    out = str(tmp_path / "panel4.s")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"] + b.EXTRA_FLAGS.get("panel4.hip", []) + [
        "-S", "--cuda-device-only", "-o", out, os.path.join(REPO, "crossscore_amd", "csrc", "panel4.hip")]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:]
    name, seen = None, {}
    for ln in open(out):
        ln = ln.strip()
        if ln.startswith(".name:"):
            name = ln.split()[-1]
        elif ln.startswith(".private_segment_fixed_size:") and name and "cs_panel4_kernel" in name:
            seen[name] = int(ln.split()[-1])
    assert len(seen) == 4 and all(v == 0 for v in seen.values()), seen
    assert "scratch_" not in open(out).read().split("cs_panel4_pack_kernel")[0] or all(v == 0 for v in seen.values())


def test_import_does_not_edit_the_environment():
    """Importing the package must not change process-wide runtime configuration; configure_runtime() is the
    explicit call, and an explicit setting wins."""
    code = ("import os; os.environ.pop('GPU_MAX_HW_QUEUES', None); import crossscore_amd; "
            "assert 'GPU_MAX_HW_QUEUES' not in os.environ; assert crossscore_amd.configure_runtime(6); "
            "assert os.environ['GPU_MAX_HW_QUEUES'] == '6'; os.environ['GPU_MAX_HW_QUEUES'] = '3'; crossscore_amd.configure_runtime(8); "
            "assert os.environ['GPU_MAX_HW_QUEUES'] == '3'; print('OK')")
    env = dict(os.environ, PYTHONPATH=REPO + os.pathsep + os.environ.get("PYTHONPATH", ""))
    res = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert res.returncode == 0 and "OK" in res.stdout, res.stdout[-2000:]


def test_bench_refuses_more_gpus_than_visible():
    """A 1-GPU (here: 0-GPU) box must not print an N-GPU line."""
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    res = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1"], env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert res.returncode != 0 and "--gpus 2" in res.stderr and not res.stdout.strip()


def test_committed_bench_line_follows_the_contract():
    """profiles/r05_bench.json is a verbatim bench.py line: the keys the driver and the judge read must be there."""
    import json

    path = os.path.join(REPO, "profiles", "r05_bench.json")
    r = json.load(open(path))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in r, k
    assert r["metric"].startswith("query-images/sec") and r["unit"] == "query-images/sec" and r["higher_is_better"] is True
    assert r["scaling"] == "weak" and r["vs_baseline"] is None and r["dtype"] == "fp16" and r["data"] == "synthetic"
    assert "workload" in r["config"] and "model" not in r["config"]
    rf = r["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] in ("GB/s", "TFLOP/s") and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    assert (rf["bound"] == "hbm") == (rf["unit"] == "GB/s") and (rf["traffic"] is None or rf["traffic"] > 0)
    cb = r["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0 and isinstance(cb["sample"], str)
    assert r["score_map_mae"] < 2e-4
    # the eager-reference legs and the scaling configuration ride in the same line
    assert set(r["eager_baseline"]) >= {"fp32_sdpa", "fp16_autocast_sdpa"} and set(r["target_10x_met"]) == set(r["eager_baseline"])
    assert r["scaling_cfg4"]["value"] > 0 and "traffic_source" in rf
    # batches in flight: the line says how many, what the stream calibration saw, and what one batch at a time gives
    assert r["config"]["batches_in_flight"] >= 1 and "in flight" in r["config"]["parallelism"]
    if r["config"]["batches_in_flight"] > 1:
        assert r["one_batch_at_a_time"]["value"] > 0 and r["config"]["stream_calibration_ms"]["in_flight_per_stream_set_tried"]
    # round 3: the line proves what ran -- every rank with its device, the process group, a median over repeats, the overflow report
    assert len(r["ranks_seen"]) == r["n_gpus"] and {"rank", "local_rank", "device"} <= set(r["ranks_seen"][0])
    assert "backend" in r["process_group"] and r["nonfinite_score_values"] == 0
    assert len(r["value_repeats"]) == 5 and min(r["value_repeats"]) <= r["value_median_of_5"] <= max(r["value_repeats"])
    # round 4: both 16-bit operand types driver-measured in the same line (BASELINE words cfg-2 "bf16"), the strong-scaling form of
    # BASELINE configs[3] with every rank's shard, and how long batches were really in flight (HIP events per forward)
    legs = r["dtype_legs"]
    assert set(legs) == {"fp16", "bf16"} and legs[r["dtype"]]["value"] == r["value"]
    assert legs["fp16"]["score_map_mae"] < 2e-4 and legs["bf16"]["score_map_mae"] < 1e-3 and legs["bf16"]["value"] > 0
    s4 = r["scaling_cfg4"]
    assert s4["mode"] == "strong" and s4["global_batch"] == 128 and sum(x["items"] for x in s4["ranks_seen"]) == 128
    assert s4["weak_16_per_gpu"]["mode"] == "weak" and s4["weak_16_per_gpu"]["value"] > 0
    f = r["batches_in_flight_measured"]
    assert abs(f["fraction_two_in_flight"] + f["fraction_one_in_flight"] + f["fraction_idle"] - 1.0) < 1e-6 and f["fraction_two_in_flight"] > 0.5
    # round 5: what a forward costs the rank's CPU thread (cs_forward_stats), in the line and in every rank's record
    assert 0 < r["host_enqueue_ms_per_forward"] < r["ms_per_step"] and r["launches_per_forward"] == sum(r["launches_by_kernel"].values())
    assert all("host_enqueue_ms_per_forward" in x and x["launches_per_forward"] > 0 for x in r["ranks_seen"])
    assert r["roofline"]["traffic_source"].startswith("committed profile profiles/r0")
