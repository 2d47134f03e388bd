"""(Needs tools/experiments/attn_two_blocks_per_wave.patch applied to crossscore_amd/csrc/attention.hip: round 4's experiment, not in the product.)
Attention at dh = 64 with one or two 32-row query blocks per wave (CS_ATTN_QW=1|2, read once per process): encoder shape (48 images, 6 heads,
1370 tokens) and the 1036-pixel shape (12 images, 5477 tokens); time per launch and the error against an fp32 reference on one image."""
import os, sys, math, subprocess
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("CS_ATTN_CHILD"):
    sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
    import torch
    import hip_helpers as hh
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(3)
    for (B, heads, T) in ((48, 6, 1370), (24, 6, 1370), (12, 6, 5477)):
        dh = 64; C = heads * dh
        qkv = (torch.randn(B, T, 3 * C, generator=g) * 1.2).to(dev).half()
        q = hh.prescale_q(qkv[:, :, :C], dh); qkv[:, :, :C] = q
        Q, K, V = qkv[:, :, :C], qkv[:, :, C:2 * C], qkv[:, :, 2 * C:]
        O = hh.attention(Q, K, V, heads, dh, q_scale=1.0)
        qh, kh, vh = (t[:1].float().view(1, T, heads, dh).transpose(1, 2) for t in (Q, K, V))
        ref = (torch.softmax(qh @ kh.transpose(-1, -2) * math.log(2.0), -1) @ vh).transpose(1, 2).reshape(1, T, C)
        err = (O[:1].float() - ref).abs()
        for _ in range(3): hh.attention(Q, K, V, heads, dh, q_scale=1.0)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): hh.attention(Q, K, V, heads, dh, q_scale=1.0)
        b.record(); torch.cuda.synchronize()
        us = 1e3 * a.elapsed_time(b) / 20
        print(f"  QW={os.environ.get('CS_ATTN_QW')} B={B} T={T}: {us:7.1f} us  {4.0 * B * heads * T * T * dh / us / 1e6:5.0f} TFLOP/s   max err {float(err.max()):.1e} mean {float(err.mean()):.1e}", flush=True)
    sys.exit(0)
for rep in range(2):
    for qw in ("1", "2"):
        subprocess.call([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, CS_ATTN_CHILD="1", CS_ATTN_QW=qw))
