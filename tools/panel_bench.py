"""Times the encoder token-panel kernel alone (HIP events) at the row counts of cfg-2: 48 and 12 images of 1370 tokens."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import hip_helpers as hh
from test_hip_panel import _make

dev = torch.device("cuda:0")
for M in (48 * 1370, 24 * 1370, 12 * 1370):
    x, o, w = _make(M, 1, dev)
    img = hh.panel_pack(w["wo"], w["ls1"], w["w1"], w["g2"], w["w2"], w["ls2"])
    for outproj in (True, False):
        im = img if outproj else hh.panel_pack(None, None, w["w1"], w["g2"], w["w2"], w["ls2"])
        for _ in range(3):
            hh.encoder_panel(x, o if outproj else None, im, w["bo"] if outproj else None, w["b1"], w["b2"])
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        n = 20
        for _ in range(n):
            hh.encoder_panel(x, o if outproj else None, im, w["bo"] if outproj else None, w["b1"], w["b2"])
        b.record()
        torch.cuda.synchronize()
        us = 1e3 * a.elapsed_time(b) / n
        fl = 4.0 * M * 384 * 1536 + (2.0 * M * 384 * 384 if outproj else 0)
        print(f"M={M} outproj={outproj}: {us:.1f} us  {fl / us / 1e6:.0f} TFLOP/s", flush=True)
