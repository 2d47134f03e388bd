"""End-to-end predict driver on a synthetic image directory: 32 query PNGs + 20 reference PNGs at 540x720, ViT-S, 5 references per
query, batch 8 -- wall time per stage with the reference-token cache on / off and with / without PNG outputs."""
import json, os, sys, tempfile, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from PIL import Image
from crossscore_amd import synth
from crossscore_amd.config import load_config, model_config
from crossscore_amd.model import CrossScoreNet
from crossscore_amd.predict import predict
root = tempfile.mkdtemp(prefix="e2e_")
base = os.path.join(root, "data", "gaussian", "mfr", "res_540", "s00000", "test", "ours_1000")
qd, rd = os.path.join(base, "renders"), os.path.join(base, "gt"); os.makedirs(qd); os.makedirs(rd)
rng = np.random.Generator(np.random.PCG64(1)); yy, xx = np.mgrid[0:540, 0:720]
def img(i):
    a = np.stack([127 + 100 * np.sin(xx / (17.0 + i) + i), 127 + 100 * np.cos(yy / (23.0 + i)), (xx + yy + 31 * i) % 256], axis=2)
    return (a + rng.normal(0, 8, a.shape)).clip(0, 255).astype(np.uint8)
for i in range(32): Image.fromarray(img(i)).save(os.path.join(qd, f"frame_{i:05}.png"))
for i in range(20): Image.fromarray(img(100 + i)).save(os.path.join(rd, f"frame_{i:05}.png"))
arch = CrossScoreNet(model_config()).arch
sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict(arch, 1).items()}
# write: "all" = the reference's default flags (score maps + the processed query / reference images), "maps" = score maps only, False = nothing;
# fused = this_main.fused_input_stage (uint8 in, tokens out; "auto" takes it when no processed image is written)
for rnd in range(2):  # (the first round pays table builds, stream probes and page-ins)
    for cache in (True, False):
        for write, fused in (("all", "auto"), ("maps", False), ("maps", "auto"), (False, False), (False, "auto")):
            over = [f"data.dataset.query_dir={qd}", f"data.dataset.reference_dir={rd}", f"this_main.cache_reference_tokens={cache}",
                    f"logger.predict.out_dir={root}/out_{rnd}_{cache}_{write}_{fused}", f"logger.predict.write.flag.batch={bool(write)}",
                    f"this_main.fused_input_stage={fused}"]
            if write == "maps":
                over += ["logger.predict.write.flag.image_query=False", "logger.predict.write.flag.image_reference=False"]
            t0 = time.perf_counter(); res = predict(load_config("default_predict", over), state_dict=sd, now="T"); dt = time.perf_counter() - t0
            print(json.dumps({"round": rnd, "cache_reference_tokens": cache, "write_png": write, "input_stage": res["input_stage"].split(" ")[0],
                              "wall_s": round(dt, 2), "query_images_per_sec_wall": round(32 / dt, 1),
                              "query_images_per_sec_loop": round(res["query_images_per_sec"], 1), "files": len(res["files"])}), flush=True)
