"""Stability: 300 cfg-2 steps (drift, determinism of the last vs the first result) and a batch of 40 items (sub-batching path)."""
import os, sys, time, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from crossscore_amd import synth
from crossscore_amd.config import model_config
from crossscore_amd.model import CrossScoreNet
net = CrossScoreNet(model_config()); net.load_numpy_state_dict(synth.make_state_dict(net.arch, 1)); net = net.cuda()
q, r = synth.make_inputs(8, 5, 518, 518, 1); tq = torch.from_numpy(q).cuda(); tr = torch.from_numpy(r).cuda()
first = net(tq, tr, False, 0, False)["score_map_ref_cross"].clone()
ts = []
for blk in range(6):
    torch.cuda.synchronize(); t = time.time()
    for _ in range(50): out = net(tq, tr, False, 0, False)["score_map_ref_cross"]
    torch.cuda.synchronize(); ts.append((time.time() - t) / 50 * 1e3)
print("ms/step per block of 50:", ["%.2f" % v for v in ts], "identical:", bool(torch.equal(out, first)), flush=True)
q, r = synth.make_inputs(40, 5, 518, 518, 2); tq = torch.from_numpy(q).cuda(); tr = torch.from_numpy(r).cuda()
big = net(tq, tr, False, 0, False)["score_map_ref_cross"]
part = net(tq[17:19], tr[17:19], False, 0, False)["score_map_ref_cross"]
torch.cuda.synchronize()
print("B=40:", tuple(big.shape), "items 17-18 identical to the same items alone:", bool(torch.equal(big[17:19], part)), "finite:", bool(torch.isfinite(big).all()), flush=True)
# batches in flight: 600 batches through a depth-3 pipeline with changing inputs, every result against the plain forward of the same input
from crossscore_amd.pipeline import ForwardPipeline
inputs = []
for i in range(4):
    q, r = synth.make_inputs(8, 5, 518, 518, 10 + i); inputs.append((torch.from_numpy(q).cuda(), torch.from_numpy(r).cuda()))
want = [net(a, b, False, 0, False)["score_map_ref_cross"].clone() for a, b in inputs]
pipe = ForwardPipeline(net, depth=3)
pipe.calibrate(*inputs[0])
bad, queue = 0, []
torch.cuda.synchronize(); t = time.time()
for i in range(600):
    queue.append((i % 4, pipe.submit(inputs[i % 4][0], inputs[i % 4][1], False, 0, False)))
    if len(queue) >= 3:
        k, tk = queue.pop(0)
        bad += int(not torch.equal(pipe.result(tk)["score_map_ref_cross"], want[k]))
while queue:
    k, tk = queue.pop(0)
    bad += int(not torch.equal(pipe.result(tk)["score_map_ref_cross"], want[k]))
torch.cuda.synchronize()
print(f"600 batches, 3 in flight: {(time.time() - t) / 600 * 1e3:.2f} ms per batch, results differing from the plain forward: {bad}", flush=True)
