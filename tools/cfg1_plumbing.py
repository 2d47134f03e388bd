"""BASELINE.json configs[0] as a plumbing run of THIS build: 1 query + 5 references at 540x720 (resized to 518x690 like MFR_subset_demo
frames) through `crossscore_amd.predict` (image directory -> GPU input stage -> forward -> PNG/CSV), checked against the oracle pipeline
(oracle transforms + fp32 oracle forward) on the host.  Seeded synthetic ViT-S weights in a Lightning-layout checkpoint (the released
checkpoint and MFR_subset_demo are not in the image).  Prints one JSON line."""
import json, os, sys, tempfile, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from PIL import Image
from crossscore_amd import synth
from crossscore_amd.config import load_config, model_config
from crossscore_amd.model import CrossScoreNet
from crossscore_amd.predict import predict
from oracle import crossscore_oracle as orc, preprocess_oracle as po

root = tempfile.mkdtemp(prefix="cfg1_")
base = os.path.join(root, "data", "gaussian", "mfr", "res_540", "s00000", "test", "ours_1000")
qd, rd = os.path.join(base, "renders"), os.path.join(base, "gt")
os.makedirs(qd); os.makedirs(rd)
rng = np.random.Generator(np.random.PCG64(1))
yy, xx = np.mgrid[0:540, 0:720]
def img(i):
    a = np.stack([127 + 100 * np.sin(xx / (17.0 + i) + i), 127 + 100 * np.cos(yy / (23.0 + i)), (xx + yy + 31 * i) % 256], axis=2)
    return (a + rng.normal(0, 8, a.shape)).clip(0, 255).astype(np.uint8)
Image.fromarray(img(0)).save(os.path.join(qd, "frame_00000.png"))
for i in range(5): Image.fromarray(img(i + 1)).save(os.path.join(rd, f"frame_{i:05}.png"))
arch = CrossScoreNet(model_config()).arch
sd = synth.make_state_dict(arch, 1)
ckpt = os.path.join(root, "run", "ckpt", "synthetic.ckpt"); os.makedirs(os.path.dirname(ckpt))
torch.save({"state_dict": {"model." + k: torch.from_numpy(v) for k, v in sd.items()}}, ckpt)
cfg = load_config("default_predict", [f"data.dataset.query_dir={qd}", f"data.dataset.reference_dir={rd}", f"trainer.ckpt_path_to_load={ckpt}",
                                      "data.neighbour_config.deterministic=True", "logger.predict.write.config.score_map_colour_mode=gray"])
t0 = time.perf_counter(); res = predict(cfg, now="RUN"); t_run = time.perf_counter() - t0
png = [f for f in res["files"] if "/score_map_ref_cross/" in f][0]
got = np.array(Image.open(png)).astype(np.float64) / 32767 - 1
q = po.preprocess_u8(np.array(Image.open(os.path.join(qd, "frame_00000.png"))), (518, 690))[None]
r = np.stack([po.preprocess_u8(np.array(Image.open(os.path.join(rd, f"frame_{i:05}.png"))), (518, 690)) for i in range(5)])[None]
torch.set_num_threads(min(os.cpu_count() or 1, 32))
t0 = time.perf_counter()
ref = orc.forward(orc.to_torch(sd), dict(enc_heads=arch.enc_heads, pos_interp_legacy=True), torch.from_numpy(q), torch.from_numpy(r), False, 0)["score_map_ref_cross"][0].numpy()
t_cpu = time.perf_counter() - t0
print(json.dumps({"workload": "cfg1 plumbing: 1 query + 5 refs, 540x720 PNG -> 518x690, ViT-S, via crossscore_amd.predict",
                  "score_map_shape": list(got.shape), "score_map_mae_vs_oracle_pipeline": float(np.abs(got - ref).mean()),
                  "csv_row": res["rows"][0], "oracle_mean": float(ref.mean()), "files_written": len(res["files"]),
                  "driver_wall_s_incl_model_build_png_io": round(t_run, 2), "model_query_images_per_sec_first_batch": round(res["query_images_per_sec"], 2),
                  "cpu_oracle_forward_s": round(t_cpu, 2), "cpu_threads": torch.get_num_threads()}))
