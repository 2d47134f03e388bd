"""How long is the decoder + head of a cfg-2 step?  forward_cached (query encoder + decoder) minus the encoder of 8 images."""
import os, sys, time, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from crossscore_amd import synth
from crossscore_amd.config import model_config
from crossscore_amd.model import CrossScoreNet
net = CrossScoreNet(model_config()); net.load_numpy_state_dict(synth.make_state_dict(net.arch, 1)); net = net.cuda()
q, r = synth.make_inputs(8, 5, 518, 518, 1); tq = torch.from_numpy(q).cuda(); tr = torch.from_numpy(r).cuda()
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time() - t0) / n * 1e3
for tp in (1,):
    tok = net.encode_references(tr.reshape(-1, 3, 518, 518)).reshape(8, 5, -1, net.arch.hidden)
    full = t(lambda: net(tq, tr, False, 0, False))
    cached = t(lambda: net.forward_cached(tq, tok))
    enc8 = t(lambda: net.encode_references(tq))
    enc40 = t(lambda: net.encode_references(tr.reshape(-1, 3, 518, 518)))
    print(f"full {full:.2f} ms | cached (enc 8 + decoder) {cached:.2f} | encoder of 8 images {enc8:.2f} | of 40 {enc40:.2f} -> decoder+head ~ {cached - enc8:.2f} ms", flush=True)
