"""A/B of two builds of the library on the same box: alternating subprocesses, cfg-2 step time.  usage: ab_libs.py <libA.so> <libB.so>"""
import os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
child = r'''
import sys, time, torch
sys.path.insert(0, %r)
from crossscore_amd import _lib
_lib.LIB_PATH = sys.argv[1]
from crossscore_amd import synth
from crossscore_amd.config import model_config
from crossscore_amd.model import CrossScoreNet
net = CrossScoreNet(model_config()); net.load_numpy_state_dict(synth.make_state_dict(net.arch, 1)); net = net.cuda()
q, r = synth.make_inputs(8, 5, 518, 518, 1); tq = torch.from_numpy(q).cuda(); tr = torch.from_numpy(r).cuda()
for _ in range(4): net(tq, tr, False, 0, False)
torch.cuda.synchronize(); t = time.time()
for _ in range(30): net(tq, tr, False, 0, False)
torch.cuda.synchronize(); thr = (time.time() - t) / 30 * 1e3
lat = []
for _ in range(20):
    torch.cuda.synchronize(); t = time.time(); net(tq, tr, False, 0, False); torch.cuda.synchronize(); lat.append((time.time() - t) * 1e3)
print(thr, min(lat))
''' % REPO
libs = sys.argv[1:3]
res = {l: [] for l in libs}
for rep in range(3):
    for l in libs:
        out = subprocess.run([sys.executable, "-c", child, l], capture_output=True, text=True)
        try:
            a, b = out.stdout.strip().splitlines()[-1].split()
            res[l].append((float(a), float(b)))
        except Exception: print(out.stderr[-500:])
for l in libs: print(l, ["%.3f" % v[0] for v in res[l]], "min %.3f ms back-to-back | single synchronous forward min %.3f ms" % (min(v[0] for v in res[l]), min(v[1] for v in res[l])))
