"""How much of a token-panel launch is the clock the chip holds: both kernels (8 waves / 4 waves) on the cfg-2 launch with the usual random
operands and with all-zero operands (same instructions, same cycle counts -- the MFMAs multiply zeros -- at whatever clock the chip then
holds).  One process, alternating; 20 launches per timing."""
import os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import hip_helpers as hh
from crossscore_amd import _lib
from test_hip_panel import _make
lib = _lib.load()
dev = torch.device("cuda:0")
M = 48 * 1370
x, o, w = _make(M, 21, dev)
def run(impl, zero, n=20):
    lib.cs_debug_panel_impl(impl)
    ww = {k: (torch.zeros_like(v) if zero else v) for k, v in w.items()}
    img = hh.panel_pack(ww["wo"], ww["ls1"], ww["w1"], ww["g2"], ww["w2"], ww["ls2"])
    xs = torch.zeros_like(x) if zero else x.clone()
    oo = torch.zeros_like(o) if zero else o
    for _ in range(5): hh.encoder_panel(xs, oo, img, ww["bo"], ww["b1"], ww["b2"])
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): hh.encoder_panel(xs, oo, img, ww["bo"], ww["b1"], ww["b2"])
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n
for rnd in range(2):
    for impl in (0, 1):
        r, z = run(impl, False), run(impl, True)
        print(f"{'8-wave' if impl == 0 else '4-wave'} kernel: random operands {r:.1f} us, all-zero operands {z:.1f} us  ({r / z:.2f} x)", flush=True)
lib.cs_debug_panel_impl(0)
