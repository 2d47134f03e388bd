"""cfg-2 item-0 score-map MAE vs the committed golden (g1) + timing."""
import os, sys, time, numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from crossscore_amd import synth
from crossscore_amd.config import model_config
from crossscore_amd.model import CrossScoreNet
g = np.load(os.path.join(REPO, "tests", "golden", "g1_vits_518_n5.npz"))
print({k: g[k].shape for k in g.files})
net = CrossScoreNet(model_config()); net.load_numpy_state_dict(synth.make_state_dict(net.arch, int(g["seed"]))); net = net.cuda()
q, r = synth.make_inputs(1, 5, 518, 518, int(g["seed"]))
q8, r8 = synth.make_inputs(8, 5, 518, 518, 1); tq8, tr8 = torch.from_numpy(q8).cuda(), torch.from_numpy(r8).cuda()
for tp in (1,):
    out = net(torch.from_numpy(q).cuda(), torch.from_numpy(r).cuda(), False, 0, False)["score_map_ref_cross"][0].cpu().numpy()
    grid = out.reshape(37, 14, 37, 14).mean(axis=(1, 3))
    msg = f"fp16 operands: patch-mean grid MAE {np.abs(grid - g['patch_mean'][0]).mean():.3e}"
    msg += f" full-resolution rows MAE {np.abs(out[g['rows_idx']] - g['rows'][0]).mean():.3e}"
    for _ in range(3): net(tq8, tr8, False, 0, False)
    torch.cuda.synchronize(); t = time.time()
    for _ in range(20): net(tq8, tr8, False, 0, False)
    torch.cuda.synchronize(); dt = (time.time() - t) / 20
    print(msg, f"| cfg-2 {dt * 1e3:.2f} ms -> {8 / dt:.1f} q/s", flush=True)
