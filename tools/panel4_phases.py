"""Phase clocks of the four-wave token-panel kernel (csrc/panel4.hip): runs the -DCS_P4_STAMP variant built by
`tools/build_variant.sh p4stamp panel4.hip -DCS_P4_STAMP` and prints, averaged over the first 64 workgroups x 4 waves, the shader cycles of every
phase and the per-tick split of the MLP loop (S1 = fc1 + GELU, wait for the fetches, barrier, S2 = fc2 + fetch issue).  Shares, not times: the
stamps fence the schedule."""
import ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, os.path.join(R, "tools", "_var", os.environ.get("CS_VARIANT", "p4stamp")))
import numpy as np, torch
import crossscore_amd
assert "_var" in crossscore_amd.__file__, crossscore_amd.__file__
import hip_helpers as hh
from crossscore_amd import _lib
from test_hip_panel import _make
lib = _lib.load(); lib.cs_debug_panel_impl(1)
dev = torch.device("cuda:0")
M = int(os.environ.get("CS_PANEL_M", 48 * 1370))
x, o, w = _make(M, 21, dev)
img = hh.panel_pack(w["wo"], w["ls1"], w["w1"], w["g2"], w["w2"], w["ls2"])
for _ in range(5): hh.encoder_panel(x, o, img, w["bo"], w["b1"], w["b2"])
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (64 * 4 * 16))()
lib.cs_panel4_debug_read.argtypes = [ctypes.c_void_p]
assert lib.cs_panel4_debug_read(buf) == 0
d = np.frombuffer(buf, dtype=np.uint64).reshape(64, 4, 16).astype(np.float64)
names = ["x rows + attn_o DMA + barrier", "out-projection", "LayerNorm + norm2(x) hand-off + b2", "MLP prologue (W1 0/1, W2 0, fc1 of tick 0)", "MLP loop (24 ticks)", "x stores", "LayerNorm + u stores"]
tot = (d[..., 7] - d[..., 0]).mean()
print(f"workgroup total {tot:.0f} cycles")
for k, n in enumerate(names):
    v = (d[..., k + 1] - d[..., k]).mean()
    print(f"  {n:45s} {v:9.0f}  {100 * v / tot:5.1f} %")
s1, vm, bar, s2 = (d[..., 8 + k].mean() / 24 for k in range(4))
print(f"per tick: S1 {s1:.0f}  fetch wait {vm:.0f}  barrier {bar:.0f}  S2 {s2:.0f}  total {s1 + vm + bar + s2:.0f}   (MFMA floor 2 x 50 / 48 x 32 = 3136)")
print("per-wave S1:", (d[..., 8].mean(0) / 24).round(0), " barrier:", (d[..., 10].mean(0) / 24).round(0), " S2:", (d[..., 11].mean(0) / 24).round(0))

if hasattr(lib, "cs_panel4_debug_read2"):
    b2 = (ctypes.c_ulonglong * (64 * 4 * 8))()
    lib.cs_panel4_debug_read2.argtypes = [ctypes.c_void_p]
    if lib.cs_panel4_debug_read2(b2) == 0:
        e = np.frombuffer(b2, dtype=np.uint64).reshape(64, 4, 8).astype(np.float64)
        if e[..., 0].min() > 0:
            # S1 of iteration 6 (fc1 of tick 7): stamps at gaps 0, 8, .., 40, then before the hand-off write; entry 7 = start of the iteration
            print("S1 of one iteration, cycles from its start to gap 0 / 8 / 16 / 24 / 32 / 40 / before the hand-off write:",
                  [(e[..., k] - e[..., 7]).mean().round(0) for k in range(7)])
