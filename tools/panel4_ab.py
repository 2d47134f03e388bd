"""Same-box A/B of the two token-panel kernels (csrc/panel.hip, 8 waves; csrc/panel4.hip, 4 waves) on the cfg-2 launch (48 images x 1370 rows):
correctness of each against the fp32 restatement, then alternating timed rounds in ONE process (HIP events around batches of launches)."""
import os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import hip_helpers as hh
from crossscore_amd import _lib
from test_hip_panel import _make, _reference
lib = _lib.load()
dev = torch.device("cuda:0")
M = int(os.environ.get("CS_PANEL_M", 48 * 1370))
bf = int(os.environ.get("CS_PANEL_BF16", "0"))
x, o, w = _make(M, 21, dev)
lib.cs_debug_set_op_operand_dtype(bf)
ob = o.float().to(torch.bfloat16).view(torch.float16) if bf else o
imgs = {}
for impl in (0, 1):
    lib.cs_debug_panel_impl(impl)
    imgs[impl] = hh.panel_pack(w["wo"], w["ls1"], w["w1"], w["g2"], w["w2"], w["ls2"])
    xk = x.clone()
    u = hh.encoder_panel(xk, ob, imgs[impl], w["bo"], w["b1"], w["b2"])
    torch.cuda.synchronize()
    if not bf:
        ref_x, ref_u = _reference(x[:8192], o[:8192], w, True, emulate=True)
        d = (xk[:8192] - ref_x).abs()
        du = (u[:8192].float() - ref_u).abs()
        print(f"impl {impl}: x mean |d| {float(d.mean()):.2e} max {float(d.max()):.2e}; u max {float(du.max()):.2e}; finite {bool(torch.isfinite(xk).all())}", flush=True)
    else:
        print(f"impl {impl}: finite {bool(torch.isfinite(xk).all())}", flush=True)
xs = x.clone()
u = torch.empty((M, 384), dtype=torch.float16, device=dev)
def run(impl, n):
    lib.cs_debug_panel_impl(impl)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        hh.encoder_panel(xs, ob, imgs[impl], w["bo"], w["b1"], w["b2"])
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for impl in (0, 1): run(impl, 5)
res = {0: [], 1: []}
for r in range(6):
    for impl in (0, 1):
        res[impl].append(run(impl, 20))
for impl in (0, 1):
    v = sorted(res[impl])
    print(f"impl {impl}: us per launch (M = {M}): median {v[len(v)//2]:.1f} min {v[0]:.1f} all {[round(t,1) for t in res[impl]]}")
lib.cs_debug_panel_impl(0); lib.cs_debug_set_op_operand_dtype(0)
