"""A/B of two library builds on the attention kernel alone, same box, alternating subprocesses: encoder shape (48 images x 6 heads, 1370 tokens,
dh 64) and the decoder's cross-attention (8 x 8 heads, 1369 x 6845, dh 48), HIP events.  usage: attn_ab_libs.py <libA.so> <libB.so>"""
import os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
child = r'''
import sys, torch, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r + "/tests")
from crossscore_amd import _lib
_lib.LIB_PATH = sys.argv[1]
import hip_helpers as hh
g = np.random.default_rng(0)
res = []
for (B, heads, Lq, Lk, dh) in ((48, 6, 1370, 1370, 64), (8, 8, 1369, 6845, 48)):
    C = heads * dh
    Q = torch.from_numpy(1.5 * g.standard_normal((B, Lq, C), dtype=np.float32)).cuda().half()
    K = torch.from_numpy(1.5 * g.standard_normal((B, Lk, C), dtype=np.float32)).cuda().half()
    V = torch.from_numpy(g.standard_normal((B, Lk, C), dtype=np.float32)).cuda().half()
    Q = hh.prescale_q(Q, dh)
    for _ in range(3): hh.attention(Q, K, V, heads, dh, q_scale=1.0)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): hh.attention(Q, K, V, heads, dh, q_scale=1.0)
    b.record(); torch.cuda.synchronize()
    res.append(1e3 * a.elapsed_time(b) / 20)
print(*res)
''' % (REPO, REPO)
libs = sys.argv[1:3]
out = {l: [] for l in libs}
for rep in range(4):
    for l in libs:
        r = subprocess.run([sys.executable, "-c", child, l], capture_output=True, text=True)
        try: out[l].append([float(v) for v in r.stdout.strip().splitlines()[-1].split()])
        except Exception: print(r.stderr[-800:])
for l in libs: print(l, " | ".join("dh64 %.1f us, dh48 %.1f us" % tuple(v) for v in out[l]))
