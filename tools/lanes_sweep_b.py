"""Lane / chunk sweep for the ViT-B workloads (cfg-3: N=10, B=8 -> 88 images; cfg-4: N=5, B=16 -> 96 images)."""
import sys, time, torch, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from crossscore_amd import synth
from crossscore_amd.config import model_config
from crossscore_amd.model import CrossScoreNet
net = CrossScoreNet(model_config(**{"backbone.from_pretrained": "facebook/dinov2-base"})); net.load_numpy_state_dict(synth.make_state_dict(net.arch, 1)); net = net.cuda()
for (B, N) in ((8, 10), (16, 5)):
    q, r = synth.make_inputs(B, N, 518, 518, 1); tq = torch.from_numpy(q).cuda(); tr = torch.from_numpy(r).cuda()
    for lanes, chunk in [tuple(int(v) for v in x.split(":")) for x in os.environ.get("CS_SWEEP", "2:0,2:11,2:12,2:16,2:22,2:23,2:24,1:0,3:16,2:0").split(",")]:
        net.lanes = lanes; net.enc_chunk_images = chunk; net._mark_dirty()
        for _ in range(2): net(tq, tr, False, 0, False)
        torch.cuda.synchronize(); t = time.time()
        for _ in range(6): net(tq, tr, False, 0, False)
        torch.cuda.synchronize(); dt = (time.time() - t) / 6
        print(f"B={B} N={N} lanes={lanes} chunk={chunk}: {dt*1e3:.2f} ms -> {B/dt:.1f} q/s", flush=True)
