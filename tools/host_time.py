import sys, time, torch, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from crossscore_amd import synth
from crossscore_amd.config import model_config
from crossscore_amd.model import CrossScoreNet
net = CrossScoreNet(model_config()); net.load_numpy_state_dict(synth.make_state_dict(net.arch, 1)); net = net.cuda()
q, r = synth.make_inputs(8, 5, 518, 518, 1); tq = torch.from_numpy(q).cuda(); tr = torch.from_numpy(r).cuda()
for lanes, chunk in ((2, 0), (3, 0), (4, 0), (1, 48), (4, 6)):
    net.lanes = lanes; net.enc_chunk_images = chunk; net._mark_dirty()
    for _ in range(3): net(tq, tr, False, 0, False)
    torch.cuda.synchronize()
    host = []; tot = []
    for _ in range(10):
        t0 = time.perf_counter(); net(tq, tr, False, 0, False); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        host.append(t1 - t0); tot.append(t2 - t0)
    print(f"lanes={lanes} chunk={chunk}: host enqueue {1e3*sum(host)/10:.2f} ms, total {1e3*sum(tot)/10:.2f} ms", flush=True)
