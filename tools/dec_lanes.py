import os, sys, time, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from crossscore_amd import synth
from crossscore_amd.config import model_config
from crossscore_amd.model import CrossScoreNet
net = CrossScoreNet(model_config()); net.load_numpy_state_dict(synth.make_state_dict(net.arch, 1)); net = net.cuda()
q, r = synth.make_inputs(8, 5, 518, 518, 1); tq = torch.from_numpy(q).cuda(); tr = torch.from_numpy(r).cuda()
base = None
for dl in (2, 4, 3, 1, 2, 4):
    os.environ["CS_DEC_LANES"] = str(dl)
    for tp in (0, 1):
        net.tail_precision = tp; net._mark_dirty()
        for _ in range(3): out = net(tq, tr, False, 0, False)["score_map_ref_cross"]
        torch.cuda.synchronize(); t = time.time()
        for _ in range(20): net(tq, tr, False, 0, False)
        torch.cuda.synchronize(); dt = (time.time() - t) / 20
        print(f"dec_lanes={dl} tail_precision={tp}: {dt*1e3:.2f} ms -> {8/dt:.1f} q/s", flush=True)
