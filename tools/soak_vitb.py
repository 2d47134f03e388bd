"""ViT-B soak (LayerNorm-folded 256-tile GEMMs): 60 cfg-4 steps, determinism; 120 batches two in flight against the plain forward, both operand types."""
import os, sys, time, torch
REPO = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, REPO)
from crossscore_amd import synth
from crossscore_amd.config import model_config
from crossscore_amd.model import CrossScoreNet
from crossscore_amd.pipeline import ForwardPipeline
for dt in ("fp16", "bf16"):
    net = CrossScoreNet(model_config(**{"backbone.from_pretrained": "facebook/dinov2-base"})); net.load_numpy_state_dict(synth.make_state_dict(net.arch, 1)); net.operand_dtype = dt; net = net.cuda()
    ins = []
    for i in range(2):
        q, r = synth.make_inputs(16, 5, 518, 518, 20 + i); ins.append((torch.from_numpy(q).cuda(), torch.from_numpy(r).cuda()))
    want = [net(a, b, False, 0, False)["score_map_ref_cross"].clone() for a, b in ins]
    bad = 0
    torch.cuda.synchronize(); t = time.time()
    for i in range(60):
        bad += int(not torch.equal(net(*ins[i % 2], False, 0, False)["score_map_ref_cross"], want[i % 2]))
    torch.cuda.synchronize()
    print(dt, f"60 plain steps {(time.time() - t) / 60 * 1e3:.1f} ms each, differing: {bad}, nonfinite {net.nonfinite_count()}", flush=True)
    pipe = ForwardPipeline(net, depth=2); pipe.calibrate(*ins[0])
    bad, queue = 0, []
    torch.cuda.synchronize(); t = time.time()
    for i in range(120):
        queue.append((i % 2, pipe.submit(ins[i % 2][0], ins[i % 2][1], False, 0, False)))
        if len(queue) >= 2:
            k, tk = queue.pop(0); bad += int(not torch.equal(pipe.result(tk)["score_map_ref_cross"], want[k]))
    while queue:
        k, tk = queue.pop(0); bad += int(not torch.equal(pipe.result(tk)["score_map_ref_cross"], want[k]))
    torch.cuda.synchronize()
    print(dt, f"120 batches two in flight: {(time.time() - t) / 120 * 1e3:.1f} ms per batch ({16 / ((time.time() - t) / 120):.0f} q/s), differing: {bad}", flush=True)
    del pipe, net
    torch.cuda.empty_cache()
