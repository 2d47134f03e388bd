"""Sum of per-kernel GPU time (HIP events, one lane, kernels in isolation) as a function of images per encoder pass:
does a smaller working set (Infinity-Cache residency) make the kernels themselves faster?"""
import sys, os, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from crossscore_amd import synth
from crossscore_amd.config import model_config
from crossscore_amd.model import CrossScoreNet
net = CrossScoreNet(model_config()); net.load_numpy_state_dict(synth.make_state_dict(net.arch, 1)); net = net.cuda()
q, r = synth.make_inputs(8, 5, 518, 518, 1); tq = torch.from_numpy(q).cuda(); tr = torch.from_numpy(r).cuda()
fams = list(range(7)) + [19, 20, 32]
names = {0: "gemm_bias", 1: "gemm_gelu", 4: "gemm_resid", 20: "attn64", 19: "attn48", 32: "misc"}
for chunk in (48, 24, 16, 12, 8, 6, 4, 3):
    net.enc_chunk_images = chunk; net.lanes = 1; net._mark_dirty()
    for _ in range(2): net(tq, tr, False, 0, False)
    net.profile_enable(True)
    for _ in range(3): net(tq, tr, False, 0, False)
    tot = 0; parts = []
    for f in fams:
        ms, n, fl = net.profile_read(f)
        tot += ms / 3
        if f in names: parts.append(f"{names[f]}={ms/3:.2f}")
    net.profile_enable(False)
    print(f"chunk={chunk:2d}: sum of kernel time {tot:.2f} ms/step | " + " ".join(parts), flush=True)
