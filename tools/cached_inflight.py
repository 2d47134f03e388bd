"""Reference-token cache mode (queries scored against pre-encoded reference tokens) with one batch at a time and with batches in flight."""
import os, sys, time, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from crossscore_amd import synth
from crossscore_amd.config import model_config
from crossscore_amd.model import CrossScoreNet
from crossscore_amd.pipeline import ForwardPipeline
net = CrossScoreNet(model_config()); net.load_numpy_state_dict(synth.make_state_dict(net.arch, 1)); net = net.cuda()
q, r = synth.make_inputs(8, 5, 518, 518, 1); tq = torch.from_numpy(q).cuda(); tr = torch.from_numpy(r).cuda()
tok = net.encode_references(tr.reshape(-1, 3, 518, 518)).reshape(8, 5, -1, net.arch.hidden)
K = 40
for _ in range(3): base = net.forward_cached(tq, tok)["score_map_ref_cross"]
torch.cuda.synchronize(); t = time.time()
for _ in range(K): net.forward_cached(tq, tok)
torch.cuda.synchronize(); one = (time.time() - t) / K
print(f"one at a time (2 lanes): {one*1e3:.2f} ms ({8/one:.0f} q/s)")
for depth in (2, 3):
    pipe = ForwardPipeline(net, depth=depth)
    for _ in range(2 * depth): tk = pipe.submit_cached(tq, tok)
    torch.cuda.synchronize(); t = time.time()
    for _ in range(K): tk = pipe.submit_cached(tq, tok)
    torch.cuda.synchronize(); two = (time.time() - t) / K
    print(f"{depth} in flight: {two*1e3:.2f} ms ({8/two:.0f} q/s), bitwise {torch.equal(pipe.result(tk)['score_map_ref_cross'], base)}")
    del pipe
