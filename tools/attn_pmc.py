"""Encoder-shaped attention launches only (48 images x 6 heads, 1370 tokens, dh 64): the target of rocprofv3 --pmc passes."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
import hip_helpers as hh
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
B, H, L, dh = 48, 6, 1370, 64
Q = (torch.randn(B, L, H * dh, generator=g) * 1.5).to(dev).to(torch.float16)
K = (torch.randn(B, L, H * dh, generator=g) * 1.5).to(dev).to(torch.float16)
V = torch.randn(B, L, H * dh, generator=g).to(dev).to(torch.float16)
for _ in range(6):
    hh.attention(Q, K, V, H, dh)
torch.cuda.synchronize()
