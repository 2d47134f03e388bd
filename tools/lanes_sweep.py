import sys, time, torch, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from crossscore_amd import synth
from crossscore_amd.config import model_config
from crossscore_amd.model import CrossScoreNet
net = CrossScoreNet(model_config()); net.load_numpy_state_dict(synth.make_state_dict(net.arch, 1)); net = net.cuda()
q, r = synth.make_inputs(8, 5, 518, 518, 1); tq = torch.from_numpy(q).cuda(); tr = torch.from_numpy(r).cuda()
base = None
for lanes, chunk in [tuple(int(v) for v in x.split(":")) for x in os.environ.get("CS_SWEEP", "2:0,3:23,2:23,2:11,3:11,4:11,2:12,1:23,1:48,2:0").split(",")]:
    net.lanes = lanes; net.enc_chunk_images = chunk; net._mark_dirty()
    for _ in range(3): out = net(tq, tr, False, 0, False)["score_map_ref_cross"]
    torch.cuda.synchronize(); t = time.time()
    for _ in range(10): net(tq, tr, False, 0, False)
    torch.cuda.synchronize(); dt = (time.time() - t) / 10
    if base is None: base = out.clone()
    print(f"lanes={lanes} chunk={chunk}: {dt*1e3:.2f} ms -> {8/dt:.1f} q/s  bitwise_equal={torch.equal(out, base)}", flush=True)
