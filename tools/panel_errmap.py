"""Where is a panel-kernel build wrong?  max |x_out - reference| per (wave pair = 32-row group of a panel, 32-column output tile), plus the
same for the normalised rows; M rows (default 384 = three panels)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
import hip_helpers as hh
from test_hip_panel import _make, _reference
dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 384
x, o, w = _make(M, 1, dev)
img = hh.panel_pack(w["wo"], w["ls1"], w["w1"], w["g2"], w["w2"], w["ls2"])
xk = x.clone()
u = hh.encoder_panel(xk, o, img, w["bo"], w["b1"], w["b2"])
torch.cuda.synchronize()
ref, uref = _reference(x, o, w, True, emulate=True)
d = (xk - ref).abs()
print("max err", float(d.max()), "mean", float(d.mean()))
rows = torch.arange(M, device=dev)
for pnl in range((M + 127) // 128):
    for pair in range(4):
        sel = (rows // 128 == pnl) & ((rows % 128) // 32 == pair)
        if sel.any():
            print(f"panel {pnl} pair {pair}:", " ".join(f"{float(d[sel][:, 32 * t:32 * t + 32].max()):8.1e}" for t in range(12)))
