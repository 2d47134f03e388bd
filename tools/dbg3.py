import sys, os, math, torch, numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from crossscore_amd import _lib
import hip_helpers as hh
torch.manual_seed(0)
dev = "cuda"
Cc, K = 128, 128
Mbig, Msm = 324, 36
A = torch.randn(Mbig, K, device=dev).bfloat16(); W = (torch.randn(Cc, K, device=dev) / K ** 0.5).bfloat16(); b = torch.randn(Cc, device=dev)
r = torch.randn(Mbig, Cc, device=dev)
sp = 4 * hh.column_tiles(Cc)
def prod(M):
    xb = torch.zeros((M, Cc), dtype=torch.bfloat16, device=dev); st = torch.zeros((M, sp, 2), device=dev); out = r[:M].clone()
    hh.gemm(A[:M].contiguous(), W, b, _lib.EPI_RESID_F32_LN, resid=out, out=out, out_bf16=xb, stats_out=st)
    torch.cuda.synchronize(); return out, xb, st
o1, xb1, st1 = prod(Mbig); o2, xb2, st2 = prod(Msm)
print("producer: out", float((o1[:Msm] - o2).abs().max()), "xb", float((xb1[:Msm].float() - xb2.float()).abs().max()), "stats", float((st1[:Msm] - st2).abs().max()))
# consumer
N = 384
Wf = torch.randn(N, Cc, device=dev) / Cc ** 0.5; gam = 1 + 0.2 * torch.randn(Cc, device=dev); bet = 0.1 * torch.randn(Cc, device=dev); bb = 0.1 * torch.randn(N, device=dev)
Wp = hh.pack_bf16(Wf, col_scale=gam); s, c = hh.ln_fold_consts(Wp, Wf, bet, bb)
def cons(M, xb, st):
    o = hh.gemm(xb[:M].contiguous(), Wp, c, _lib.EPI_LN_BF16, ln_part=st[:M].contiguous(), col_s=s); torch.cuda.synchronize(); return o
c1 = cons(Mbig, xb1, st1); c2 = cons(Msm, xb1, st1)
print("consumer: ", float((c1[:Msm].float() - c2.float()).abs().max()))
ref = torch.nn.functional.layer_norm(o1, (Cc,), gam, bet, 1e-6) @ Wf.t() + bb
print("consumer vs ref big:", float((c1.float() - ref).abs().max()), " small:", float((c2.float() - ref[:Msm]).abs().max()))
for M in (36, 128, 129, 256, 300, 324):
    cc = cons(M, xb1, st1); print(M, float((cc.float() - ref[:M]).abs().max()), float((cc.float() - c1[:M].float()).abs().max()))
