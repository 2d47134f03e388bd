"""The token-panel kernel alone (48 images = 65 760 rows, 6 launches) and the 256x256x64 GEMM on the ViT-B QKV shape (131 520 x 2304 x 768,
6 launches): the targets of rocprofv3 --pmc passes (tools/profile_round.sh; summarised by tools/summarise_kernel_pmc.py)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
import hip_helpers as hh
from crossscore_amd import _lib
from test_hip_panel import _make
dev = torch.device("cuda:0")
x, o, w = _make(48 * 1370, 1, dev)
img = hh.panel_pack(w["wo"], w["ls1"], w["w1"], w["g2"], w["w2"], w["ls2"])
for _ in range(6):
    hh.encoder_panel(x, o, img, w["bo"], w["b1"], w["b2"])
    x.zero_().add_(1.0)
M, N, K = 131520, 2304, 768
A = torch.randn(M, K, device=dev).half(); W = (torch.randn(N, K, device=dev) / K ** 0.5).half(); b = torch.randn(N, device=dev)
out = torch.empty(M, N, device=dev, dtype=torch.float16)
for _ in range(6):
    hh.gemm(A, W, b, _lib.EPI_BIAS_F16, out=out)
torch.cuda.synchronize()
