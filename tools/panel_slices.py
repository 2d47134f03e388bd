"""Which hidden slices does a panel-kernel build get wrong?  Runs the kernel 48 times on one panel with fc2 masked to one 32-wide hidden
slice (so the output is that slice's contribution alone) and prints the error per slice; then with fc1 masked to one 16-wide k-step."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
import hip_helpers as hh
from test_hip_panel import _make, _reference
dev = torch.device("cuda:0")
M = 128
x, o, w = _make(M, 1, dev)
def run(w):
    img = hh.panel_pack(w["wo"], w["ls1"], w["w1"], w["g2"], w["w2"], w["ls2"])
    xk = x.clone()
    hh.encoder_panel(xk, o, img, w["bo"], w["b1"], w["b2"])
    torch.cuda.synchronize()
    ref, _ = _reference(x, o, w, True, emulate=True)
    return (xk - ref).abs()
errs = []
for s in range(48):
    wm = dict(w); m = torch.zeros_like(w["w2"]); m[:, 32 * s:32 * s + 32] = 1; wm["w2"] = w["w2"] * m
    errs.append(float(run(wm).max()))
print("fc2 slice mask:", " ".join(f"{e:.0e}" for e in errs))
errs = []
for k in range(24):
    wm = dict(w); m = torch.zeros_like(w["w1"]); m[:, 16 * k:16 * k + 16] = 1; wm["w1"] = w["w1"] * m
    errs.append(float(run(wm).max()))
print("fc1 k-step mask:", " ".join(f"{e:.0e}" for e in errs))
d = run(w)
print("full: max", float(d.max()), "per pair:", [float(d[32 * p:32 * p + 32].max()) for p in range(4)])
for rep in range(3):
    print("repeat", rep, float(run(w).max()))
