"""Patch embedding: the one-launch kernel (csrc/patch.hip) against im2col + GEMM, timed inside a forward-sized chunk.
Uses the handle-free op entry points, which allocate and synchronise: so the kernels themselves are timed through rocprofv3
(`rocprofv3 --kernel-trace --stats -- python tools/patch_bench.py`), and this script only drives them."""
import os, sys, math, torch, numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
if os.environ.get("CS_PB_ROOT"): sys.path.insert(0, os.environ["CS_PB_ROOT"])  # a variant build (tools/patch_ab.py)
import hip_helpers as hh
I, H, W, C, P = int(os.environ.get("CS_PB_IMGS", 48)), 518, 518, int(os.environ.get("CS_PB_C", 384)), 14
g = np.random.default_rng(0)
x = torch.from_numpy(g.standard_normal((I, 3, H, W), dtype=np.float32)).cuda()
w = torch.from_numpy(g.standard_normal((C, 3, P, P), dtype=np.float32) / math.sqrt(588)).cuda()
b = torch.zeros(C, device="cuda"); pos = torch.zeros(1 + 37 * 37, C, device="cuda")
for _ in range(5):
    a = hh.patch_embed_fused(x, w, b, pos, P)
    c = hh.patch_embed(x, w, b, pos, P, 1)
torch.cuda.synchronize()
print("max diff", float((a - c).abs().max()))
