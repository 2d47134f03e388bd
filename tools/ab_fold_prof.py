import sys, os, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from crossscore_amd import synth
from crossscore_amd.config import model_config
from crossscore_amd.model import CrossScoreNet
net = CrossScoreNet(model_config()); net.load_numpy_state_dict(synth.make_state_dict(net.arch, 1)); net = net.cuda()
q, r = synth.make_inputs(8, 5, 518, 518, 1); tq = torch.from_numpy(q).cuda(); tr = torch.from_numpy(r).cuda()
names = {0: "bias", 1: "gelu", 4: "resid", 5: "patch", 7: "ln_bias", 8: "ln_gelu", 9: "resid_ln", 19: "attn48", 20: "attn64", 32: "misc"}
for fold in (0, 1):
    net.ln_fold = fold; net.lanes = 1; net.enc_chunk_images = 48; net._mark_dirty()
    for _ in range(2): net(tq, tr, False, 0, False)
    net.profile_enable(True)
    for _ in range(3): net(tq, tr, False, 0, False)
    parts = []; tot = 0
    for f, nm in names.items():
        ms, n, fl = net.profile_read(f)
        if n: parts.append(f"{nm}={ms/3:.2f}ms/{n//3}"); tot += ms / 3
    net.profile_enable(False)
    print(f"ln_fold={fold}: total {tot:.2f} | " + " ".join(parts), flush=True)
