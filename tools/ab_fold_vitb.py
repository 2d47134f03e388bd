"""ViT-B (cfg-4 shape, 16 items): the LayerNorm-folded GEMM epilogues (ln_fold=1) against the separate LayerNorm kernels, one batch at a
time and two in flight."""
import sys, time, torch, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from crossscore_amd import synth
from crossscore_amd.config import model_config
from crossscore_amd.model import CrossScoreNet
from crossscore_amd.pipeline import ForwardPipeline
net = CrossScoreNet(model_config(**{"backbone.from_pretrained": "facebook/dinov2-base"})); net.load_numpy_state_dict(synth.make_state_dict(net.arch, 1)); net = net.cuda()
q, r = synth.make_inputs(16, 5, 518, 518, 1); tq = torch.from_numpy(q).cuda(); tr = torch.from_numpy(r).cuda()
base = None
for fold in (2, 0, 2, 0):  # 2 = separate LayerNorm launches, 0 = the default (r5): folded into the 256-tile GEMM epilogues
    net.ln_fold = fold; net.lanes = 0; net._mark_dirty()
    for _ in range(2): out = net(tq, tr, False, 0, False)["score_map_ref_cross"]
    torch.cuda.synchronize(); t = time.time()
    for _ in range(5): net(tq, tr, False, 0, False)
    torch.cuda.synchronize(); one = (time.time() - t) / 5
    pipe = ForwardPipeline(net, depth=2)
    pipe.calibrate(tq, tr, steps=3)
    torch.cuda.synchronize(); t = time.time()
    for _ in range(8): tk = pipe.submit(tq, tr, False, 0, False)
    torch.cuda.synchronize(); two = (time.time() - t) / 8
    if base is None: base = out.clone()
    print(f"ln_fold={fold}: one at a time {one*1e3:.1f} ms, two in flight {two*1e3:.1f} ms ({16/two:.0f} q/s); max |diff| vs separate LayerNorms {(out - base).abs().max().item():.2e}", flush=True)
    del pipe
