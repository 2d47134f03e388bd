"""Times the attention kernel alone at the three sites of cfg-2 (HIP events): encoder self-attention (48 images x 6 heads, 1370 tokens,
dh 64), decoder self-attention (8 x 8 heads, 1369, dh 48), cross-attention (8 x 8 heads, 1369 x 6845, dh 48).
CS_ATTN_VARIANTS="A=1,B=2+C=3": also builds the library with -DCS_ATTN_<..> per variant (experiment macros a kernel under study may
read) into scratch directories and times each in turn."""
import os, subprocess, sys, shutil, tempfile
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("CS_ATTN_CHILD"):
    sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, os.environ["CS_ATTN_CHILD"])
    import torch
    import hip_helpers as hh
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    for name, B, H, Lq, Lk, dh in (("encoder dh64", 48, 6, 1370, 1370, 64), ("dec self dh48", 8, 8, 1369, 1369, 48), ("cross dh48", 8, 8, 1369, 6845, 48)):
        Q = (torch.randn(B, Lq, H * dh, generator=g) * 1.5).to(dev).to(torch.float16)
        K = (torch.randn(B, Lk, H * dh, generator=g) * 1.5).to(dev).to(torch.float16)
        V = torch.randn(B, Lk, H * dh, generator=g).to(dev).to(torch.float16)
        for _ in range(3): O = hh.attention(Q, K, V, H, dh)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): hh.attention(Q, K, V, H, dh)
        b.record(); torch.cuda.synchronize()
        us = 1e3 * a.elapsed_time(b) / 20
        fl = 4.0 * B * H * Lq * Lk * dh
        # spot check against torch on one (batch, head)
        q, k, v = Q[0, :, :dh].float(), K[0, :, :dh].float(), V[0, :, :dh].float()
        ref = torch.softmax(q @ k.t() / dh ** 0.5, -1) @ v
        err = (O[0, :, :dh].float() - ref).abs().max().item()
        print(f"  {name}: {us:7.1f} us  {fl / us / 1e6:6.0f} TFLOP/s  max err {err:.1e}", flush=True)
    sys.exit(0)
sys.path.insert(0, R)
from crossscore_amd import build
variants = [v for v in os.environ.get("CS_ATTN_VARIANTS", "").split(",") if v]
srcs = build.SOURCES
for var in [None] + variants:
    if var is None:
        pkgroot = R
    else:
        tmp = tempfile.mkdtemp(prefix="attn_var_")
        pkg = os.path.join(tmp, "crossscore_amd")
        shutil.copytree(os.path.join(R, "crossscore_amd"), pkg, ignore=shutil.ignore_patterns("*.so", "build", "__pycache__"))
        shutil.copytree(os.path.join(R, "include"), os.path.join(tmp, "include"))
        objs, procs = [], []
        for s in srcs:
            o = os.path.join(tmp, s + ".o"); objs.append(o)
            cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"]
            cmd += build.EXTRA_FLAGS.get(s, [])
            if s == "attention.hip": cmd += ["-fno-slp-vectorize" if d == "NOSLP" else "-DCS_ATTN_" + d for d in var.split("+")]
            procs.append(subprocess.Popen(cmd + ["-c", os.path.join(pkg, "csrc", s), "-o", o]))
        for pr in procs: assert pr.wait() == 0
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(pkg, "libcrossscore_hip.so")] + objs)
        pkgroot = tmp
    print("variant", var or "(in-tree build)", flush=True)
    subprocess.call([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, CS_ATTN_CHILD=pkgroot))
