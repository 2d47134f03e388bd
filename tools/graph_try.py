"""Does a hipGraph replay of the whole forward beat eager enqueue?  Captures CrossScoreNet.forward (its internal lanes fork / join with
events, which a stream capture follows) with torch.cuda.CUDAGraph after a warm-up, and times replays against eager calls."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from crossscore_amd import synth
from crossscore_amd.config import model_config
from crossscore_amd.model import CrossScoreNet
net = CrossScoreNet(model_config()); net.load_numpy_state_dict(synth.make_state_dict(net.arch, 1)); net = net.cuda()
q, r = synth.make_inputs(8, 5, 518, 518, 1); tq = torch.from_numpy(q).cuda(); tr = torch.from_numpy(r).cuda()
for _ in range(3): ref = net(tq, tr, False, 0, False)["score_map_ref_cross"]
torch.cuda.synchronize()
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
eager = timeit(lambda: net(tq, tr, False, 0, False))
print(f"eager: {eager:.3f} ms/step", flush=True)
try:
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(2): net(tq, tr, False, 0, False)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            out = net(tq, tr, False, 0, False)["score_map_ref_cross"]
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    print("graph output equals eager:", bool(torch.equal(out, ref)), flush=True)
    rep = timeit(g.replay)
    print(f"graph replay: {rep:.3f} ms/step ({100 * (rep / eager - 1):+.1f} %)", flush=True)
except Exception as e:
    print("capture failed:", repr(e)[:500], flush=True)
