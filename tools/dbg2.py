import sys, os, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from crossscore_amd import synth
from crossscore_amd.config import model_config
from crossscore_amd.model import CrossScoreNet
net = CrossScoreNet(model_config(**{"backbone.from_pretrained": "synthetic/dinov2-tiny"})); net.load_numpy_state_dict(synth.make_state_dict(net.arch, 2)); net = net.cuda()
q, r = synth.make_inputs(3, 2, 70, 98, 2); tq = torch.from_numpy(q).cuda(); tr = torch.from_numpy(r).cuda()
for fold in (1, 0):
    for lanes in (1, 2):
        net.ln_fold = fold; net.lanes = lanes
        outs = {}
        for chunk in (0, 1, 2, 3, 4, 9, 9, 3):
            net.enc_chunk_images = chunk; net._mark_dirty()
            o = net(tq, tr, False, 0, False)["score_map_ref_cross"].clone(); torch.cuda.synchronize()
            outs.setdefault(chunk, []).append(o)
        base = outs[0][0]
        print(f"fold={fold} lanes={lanes}:", {k: [float((v - base).abs().max()) for v in vs] for k, vs in outs.items()})
