"""A/B of one environment variable on the same box: alternating subprocesses, cfg-2 step time.  usage: ab_env.py VAR [value]"""
import os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
child = r'''
import sys, time, torch
sys.path.insert(0, %r)
from crossscore_amd import synth
from crossscore_amd.config import model_config
from crossscore_amd.model import CrossScoreNet
import os
net = CrossScoreNet(model_config()); net.load_numpy_state_dict(synth.make_state_dict(net.arch, 1)); net = net.cuda()
net.lanes = int(os.environ.get('CS_AB_LANES', '0')); net.enc_chunk_images = int(os.environ.get('CS_AB_CHUNK', '0'))
q, r = synth.make_inputs(8, 5, 518, 518, 1); tq = torch.from_numpy(q).cuda(); tr = torch.from_numpy(r).cuda()
for _ in range(4): net(tq, tr, False, 0, False)
torch.cuda.synchronize(); t = time.time()
for _ in range(30): net(tq, tr, False, 0, False)
torch.cuda.synchronize(); print((time.time() - t) / 30 * 1e3)
''' % REPO
var = sys.argv[1]; val = sys.argv[2] if len(sys.argv) > 2 else "1"
res = {"off": [], "on": []}
for rep in range(3):
    for k in ("off", "on"):
        env = dict(os.environ); env.pop(var, None)
        if k == "on": env[var] = val
        out = subprocess.run([sys.executable, "-c", child], capture_output=True, text=True, env=env)
        try: res[k].append(float(out.stdout.strip().splitlines()[-1]))
        except Exception: print(out.stderr[-500:])
for k in res: print(var, k, ["%.3f" % v for v in res[k]], "min %.3f ms" % min(res[k]))
