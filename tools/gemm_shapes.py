"""Runs the four encoder GEMM shapes (and the encoder attention) a few times: target for rocprofv3 --pmc passes."""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from crossscore_amd import _lib
import hip_helpers as hh
dev = "cuda"
shapes = {"qkv": (65760, 1152, 384, _lib.EPI_BIAS_F16), "outproj": (65760, 384, 384, _lib.EPI_RESID_F32),
          "fc1": (65760, 1536, 384, _lib.EPI_BIAS_GELU_F16), "fc2": (65760, 384, 1536, _lib.EPI_RESID_F32)}
for sn, (M, N, K, epi) in shapes.items():
    A = torch.randn(M, K, device=dev).bfloat16(); W = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
    b = torch.randn(N, device=dev)
    resid = torch.randn(M, N, device=dev) if epi == _lib.EPI_RESID_F32 else None
    o = torch.empty(M, N, device=dev, dtype=torch.float32 if epi == _lib.EPI_RESID_F32 else torch.float16)
    for _ in range(3):
        hh.gemm(A, W, b, epi, resid=resid, out=o)
    torch.cuda.synchronize()
qkv = torch.randn(48, 1370, 1152, device=dev).bfloat16()
for _ in range(3):
    hh.attention(qkv[:, :, :384], qkv[:, :, 384:768], qkv[:, :, 768:], 6, 64)
torch.cuda.synchronize()
