"""Batches in flight (crossscore_amd.pipeline.ForwardPipeline): replicas of one module fed round-robin, so one batch's decoder
overlaps the next batch's encoder.  Prints the throughput beside the one-handle loop (same weights, same inputs)."""
import os, sys, time, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from crossscore_amd import synth
from crossscore_amd.config import model_config
from crossscore_amd.model import CrossScoreNet
from crossscore_amd.pipeline import ForwardPipeline
sd = None
NF = int(os.environ.get("CS_INFLIGHT", "2"))
nets = []
for i in range(1):
    net = CrossScoreNet(model_config())
    if sd is None: sd = synth.make_state_dict(net.arch, 1)
    net.load_numpy_state_dict(sd); nets.append(net.cuda())
q, r = synth.make_inputs(8, 5, 518, 518, 1); tq = torch.from_numpy(q).cuda(); tr = torch.from_numpy(r).cuda()
K = 24
for lanes, chunk in [tuple(int(v) for v in x.split(":")) for x in os.environ.get("CS_SWEEP", "2:0,1:0").split(",")]:
    for n in nets: n.lanes = lanes; n.enc_chunk_images = chunk; n._mark_dirty()
    for _ in range(3): base = nets[0](tq, tr, False, 0, False)["score_map_ref_cross"]
    torch.cuda.synchronize(); t = time.time()
    for _ in range(K): nets[0](tq, tr, False, 0, False)
    torch.cuda.synchronize(); one = (time.time() - t) / K
    pipe = ForwardPipeline(nets[0], depth=NF, lanes=lanes)  # the product class (its own replicas and streams)
    tickets = [None] * NF
    for i in range(2 * NF):
        tickets[i % NF] = pipe.submit(tq, tr, False, 0, False)
    torch.cuda.synchronize(); t = time.time()
    for i in range(K):
        tickets[i % NF] = pipe.submit(tq, tr, False, 0, False)
    torch.cuda.synchronize(); two = (time.time() - t) / K
    outs = [pipe.result(tk)["score_map_ref_cross"] for tk in tickets]
    del pipe
    print(f"lanes={lanes} chunk={chunk} inflight={NF}: one in flight {one*1e3:.2f} ms/step ({8/one:.0f} q/s); two in flight {two*1e3:.2f} ms/step ({8/two:.0f} q/s); "
          f"bitwise {all(torch.equal(o, base) for o in outs)}", flush=True)
