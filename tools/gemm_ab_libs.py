"""A/B of two library builds on the ViT-B encoder GEMM shapes, same box, alternating subprocesses, HIP events.
usage: gemm_ab_libs.py <libA.so> <libB.so>   (CS_GAB_ROWS=131520,30140)"""
import os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
child = r'''
import os, sys, torch
sys.path.insert(0, %r); sys.path.insert(0, %r + "/tests")
from crossscore_amd import _lib
_lib.LIB_PATH = sys.argv[1]
import hip_helpers as hh
res = []
for M in [int(v) for v in os.environ.get("CS_GAB_ROWS", "131520,30140").split(",")]:
    for (N, K, epi) in ((2304, 768, _lib.EPI_BIAS_F16), (768, 768, _lib.EPI_RESID_F32), (3072, 768, _lib.EPI_BIAS_GELU_F16), (768, 3072, _lib.EPI_RESID_F32)):
        A = torch.randn(M, K, device="cuda").half(); W = (torch.randn(N, K, device="cuda") / K ** 0.5).half(); b = torch.randn(N, device="cuda")
        r = torch.randn(M, N, device="cuda") if epi == _lib.EPI_RESID_F32 else None
        o = torch.empty(M, N, device="cuda", dtype=torch.float32 if r is not None else torch.float16)
        for _ in range(3): hh.gemm(A, W, b, epi, resid=r, out=o)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): hh.gemm(A, W, b, epi, resid=r, out=o)
        e1.record(); torch.cuda.synchronize()
        res.append(1e3 * e0.elapsed_time(e1) / 20)
print(*res)
''' % (REPO, REPO)
libs = sys.argv[1:3]
out = {l: [] for l in libs}
for rep in range(3):
    for l in libs:
        r = subprocess.run([sys.executable, "-c", child, l], capture_output=True, text=True)
        try: out[l].append([float(v) for v in r.stdout.strip().splitlines()[-1].split()])
        except Exception: print(r.stderr[-800:])
print("columns per row count: qkv, out-proj, fc1, fc2 (us)")
for l in libs:
    for v in out[l]: print(os.path.basename(l), " ".join("%7.1f" % x for x in v))
