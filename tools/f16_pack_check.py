"""Where does cs_op_pack_f16 differ from torch's fp32 -> fp16 rounding?  (debug aid)"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
import hip_helpers as hh
g = torch.Generator().manual_seed(1)
Wf = (torch.randn(700, 384, generator=g) / 384 ** 0.5).cuda()
s = (0.5 + 0.2 * torch.randn(700, generator=g)).cuda()
for sc in (None, s):
    W = hh.pack_f16(Wf, row_scale=sc)
    ref = (Wf * sc[:, None] if sc is not None else Wf).to(torch.float16)
    torch.cuda.synchronize()
    bad = (W != ref)
    print("scale" if sc is not None else "plain", "mismatches", int(bad.sum()), "of", W.numel())
    if bad.any():
        i = bad.nonzero()[:8]
        for r, c in i.tolist():
            x = (Wf[r, c] * (sc[r] if sc is not None else 1.0))
            print(f"  fp32 {x.item():.9e} kernel {W[r, c].item():.9e} torch {ref[r, c].item():.9e}  bits {W[r, c].view(torch.int16).item()} vs {ref[r, c].view(torch.int16).item()}")
