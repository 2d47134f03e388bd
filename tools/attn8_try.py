"""(Needs tools/experiments/attn_eight_wave_and_mfma_rowsum.patch applied to crossscore_amd/csrc/attention.hip: round 4's experiments, not in the product.)
Attention at dh = 64 / 48 with the four-wave kernel (CS_ATTN8=0) and the eight-wave two-group form (CS_ATTN8=1, read once per process):
encoder shape (48 images, 6 heads, 1370 tokens), the 1036-pixel shape (12 images, 5477 tokens), the decoder's cross-attention shape
(16 queries x 8 heads x 1369 x 6845 at dh = 48) and a ragged small one; time per launch and the error against an fp32 reference on one image."""
import os, sys, math, subprocess
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("CS_ATTN_CHILD"):
    sys.path.insert(0, os.environ["CS_ATTN_CHILD"]); sys.path.insert(0, os.path.join(R, "tests"))
    import torch
    import hip_helpers as hh
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(3)
    for (B, heads, dh, Lq, Lk) in ((48, 6, 64, 1370, 1370), (24, 6, 64, 1370, 1370), (12, 6, 64, 5477, 5477), (16, 8, 48, 1369, 6845), (3, 2, 64, 300, 77)):
        C = heads * dh
        Q = (torch.randn(B, Lq, C, generator=g) * 1.2).to(dev).half()
        K = (torch.randn(B, Lk, C, generator=g) * 1.2).to(dev).half()
        V = (torch.randn(B, Lk, C, generator=g) * 1.2).to(dev).half()
        Q = hh.prescale_q(Q, dh)
        O, L = hh.attention(Q, K, V, heads, dh, lse=True, q_scale=1.0)
        bi = B - 1
        qh, kh, vh = (t[bi:bi + 1].float().view(1, -1, heads, dh).transpose(1, 2) for t in (Q, K, V))
        sc = qh @ kh.transpose(-1, -2) * math.log(2.0)
        ref = (torch.softmax(sc, -1) @ vh).transpose(1, 2).reshape(1, Lq, C)
        err = (O[bi:bi + 1].float() - ref).abs()
        lerr = (L[bi] - torch.logsumexp(sc[0], -1) / math.log(2.0)).abs().max()
        for _ in range(3): hh.attention(Q, K, V, heads, dh, q_scale=1.0)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): hh.attention(Q, K, V, heads, dh, q_scale=1.0)
        b.record(); torch.cuda.synchronize()
        us = 1e3 * a.elapsed_time(b) / 20
        print(f"  {os.environ.get('CS_ATTN_LABEL')} B={B} h={heads} dh={dh} Lq={Lq} Lk={Lk}: {us:7.1f} us  {4.0 * B * heads * Lq * Lk * dh / us / 1e6:5.0f} TFLOP/s   "
              f"max err {float(err.max()):.1e} mean {float(err.mean()):.1e} lse err {float(lerr):.1e}", flush=True)
    sys.exit(0)
# variants: the in-tree build with CS_ATTN8=0/1, then one scratch build per entry of CS_ATTN_VARIANTS ("MSUM,X+Y": -DCS_ATTN_MSUM, -DCS_ATTN_X -DCS_ATTN_Y)
import shutil, tempfile
sys.path.insert(0, R)
from crossscore_amd import build
roots = [("four-wave", R, "0")] + ([("eight-wave", R, "1")] if os.environ.get("CS_ATTN_TRY8") else []) + ([("pipelined", R, "pipe")] if os.environ.get("CS_ATTN_TRYPIPE") else [])
for var in [v for v in os.environ.get("CS_ATTN_VARIANTS", "").split(",") if v]:
    tmp = tempfile.mkdtemp(prefix="attn_var_")
    pkg = os.path.join(tmp, "crossscore_amd")
    shutil.copytree(os.path.join(R, "crossscore_amd"), pkg, ignore=shutil.ignore_patterns("*.so", "build", "__pycache__"))
    shutil.copytree(os.path.join(R, "include"), os.path.join(tmp, "include"))
    objs, procs = [], []
    for src in build.SOURCES:
        o = os.path.join(tmp, src + ".o"); objs.append(o)
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"] + build.EXTRA_FLAGS.get(src, [])
        if src == "attention.hip": cmd += ["-DCS_ATTN_" + d for d in var.split("+")]
        procs.append(subprocess.Popen(cmd + ["-c", os.path.join(pkg, "csrc", src), "-o", o]))
    for pr in procs: assert pr.wait() == 0
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(pkg, "libcrossscore_hip.so")] + objs)
    roots.append(("four-wave+" + var, tmp, "0"))
for rep in range(2):
    for label, root, e in roots:
        extra = {"CS_ATTN8": "0", "CS_ATTN_PIPE": "1"} if e == "pipe" else {"CS_ATTN8": e}
        subprocess.call([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, CS_ATTN_CHILD=root, CS_ATTN_LABEL=label, **extra))
