"""bf16 operand mode of the attention kernel at op level: raw-Q scaling inside the kernel vs Q pre-scaled by the caller."""
import math, os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from crossscore_amd import _lib
import hip_helpers as hh
lib = _lib.load(); dev = "cuda"
g = torch.Generator().manual_seed(0)
for mode in (0, 1):
    lib.cs_debug_set_op_operand_dtype(mode)
    dt = torch.bfloat16 if mode else torch.float16
    for dh, heads in ((64, 6), (48, 8), (16, 8)):
        B, Lq, Lk = 2, 300, 500
        Q = torch.randn(B, Lq, heads * dh, generator=g).to(dev).to(dt); K = torch.randn(B, Lk, heads * dh, generator=g).to(dev).to(dt); V = torch.randn(B, Lk, heads * dh, generator=g).to(dev).to(dt)
        qh, kh, vh = (t.float().view(B, -1, heads, dh).transpose(1, 2) for t in (Q, K, V))
        ref = (torch.softmax(qh @ kh.transpose(-1, -2) / math.sqrt(dh), -1) @ vh).transpose(1, 2).reshape(B, Lq, heads * dh)
        O0 = hh.attention(Q.view(torch.float16), K.view(torch.float16), V.view(torch.float16), heads, dh).view(dt)
        Qs = (Q.float() * (1.4426950408889634 / math.sqrt(dh))).to(dt)
        O1 = hh.attention(Qs.view(torch.float16), K.view(torch.float16), V.view(torch.float16), heads, dh, q_scale=1.0).view(dt)
        torch.cuda.synchronize()
        for nm, O in (("raw q", O0), ("prescaled", O1)):
            d = (O.float() - ref).abs(); print(f"mode {mode} dh {dh} {nm}: max {float(d.max()):.3e} mean {float(d.mean()):.3e}", flush=True)
lib.cs_debug_set_op_operand_dtype(0)
