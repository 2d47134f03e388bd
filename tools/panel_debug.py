"""Where does the panel kernel differ from the fp16-emulating torch restatement?  (debug aid)"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
import hip_helpers as hh
from test_hip_panel import _make, _reference

dev = torch.device("cuda:0")
for outproj in (False, True):
    M = 128
    x, o, w = _make(M, 3, dev)
    img = hh.panel_pack(w["wo"] if outproj else None, w["ls1"] if outproj else None, w["w1"], w["g2"], w["w2"], w["ls2"])
    xk = x.clone()
    u = hh.encoder_panel(xk, o if outproj else None, img, w["bo"] if outproj else None, w["b1"], w["b2"])
    torch.cuda.synchronize()
    rx, ru = _reference(x, o, w, outproj, True)
    e = (xk - rx).abs()
    print(f"outproj={outproj}: max {e.max().item():.4f} mean {e.mean().item():.5f} frac>0.02 {(e > 0.02).float().mean().item():.4f}")
    print(" per 16-row block max:", [round(v, 3) for v in e.view(8, 16, 384).amax(dim=(1, 2)).tolist()])
    print(" per row-in-block (m) max:", [round(v, 3) for v in e.view(8, 16, 384).amax(dim=(0, 2)).tolist()])
    print(" per 16-col tile max:", [round(v, 3) for v in e.view(128, 24, 16).amax(dim=(0, 2)).tolist()])
    print(" per col-in-tile max:", [round(v, 3) for v in e.view(128, 24, 16).amax(dim=(0, 1)).tolist()])
    # isolate stages: no MLP contribution (w2 = 0) and no out-proj contribution
    w0 = dict(w); w0["w2"] = torch.zeros_like(w["w2"])
    img0 = hh.panel_pack(w0["wo"] if outproj else None, w0["ls1"] if outproj else None, w0["w1"], w0["g2"], w0["w2"], w0["ls2"])
    xk0 = x.clone()
    hh.encoder_panel(xk0, o if outproj else None, img0, w0["bo"] if outproj else None, w0["b1"], w0["b2"])
    rx0, _ = _reference(x, o, w0, outproj, True)
    print("  with W2 = 0 (out-proj + residual + b2 only): max err", (xk0 - rx0).abs().max().item())
    eu = (u.float() - ru).abs()
    print("  u: max", eu.max().item(), "mean", eu.mean().item())
