"""The four-wave token-panel kernel alone (M = 256 x 128 rows = one round, 6 launches): target of rocprofv3 --pmc passes."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
import hip_helpers as hh
from crossscore_amd import _lib
from test_hip_panel import _make
lib = _lib.load(); lib.cs_debug_panel_impl(int(os.environ.get("CS_PANEL_IMPL", "1")))
dev = torch.device("cuda:0")
x, o, w = _make(int(os.environ.get("CS_PANEL_M", 256 * 128)), 1, dev)
img = hh.panel_pack(w["wo"], w["ls1"], w["w1"], w["g2"], w["w2"], w["ls2"])
for _ in range(6):
    hh.encoder_panel(x, o, img, w["bo"], w["b1"], w["b2"])
    x.zero_().add_(1.0)
torch.cuda.synchronize()
