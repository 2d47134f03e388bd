"""Is the attention kernel held back by the clock the chip gives up for operand toggling (as the token-panel kernels are, tools/panel_zero_data.py)?
The same launches on random operands, on all-zero operands and on constant operands, alternating in one process (HIP events, 20 launches each)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
import hip_helpers as hh
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for name, B, H, Lq, Lk, dh in (("encoder dh64", 48, 6, 1370, 1370, 64), ("cross dh48", 8, 8, 1369, 6845, 48)):
    data = {}
    data["random"] = [(torch.randn(B, L, H * dh, generator=g) * s).to(dev).to(torch.float16) for L, s in ((Lq, 1.5), (Lk, 1.5), (Lk, 1.0))]
    data["zeros"] = [torch.zeros_like(t) for t in data["random"]]
    data["ones"] = [torch.full_like(t, 0.25) for t in data["random"]]
    for rnd in range(2):
        for kind, (Q, K, V) in data.items():
            for _ in range(5): hh.attention(Q, K, V, H, dh)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20): hh.attention(Q, K, V, H, dh)
            b.record(); torch.cuda.synchronize()
            print(f"{name} {kind:7s}: {1e3 * a.elapsed_time(b) / 20:7.1f} us", flush=True)
