"""Phase clocks of the attention tile loop: builds the library with -DCS_ATTN_STAMP into a scratch directory, runs the encoder-shaped
launch (48 images x 6 heads, 1370 tokens, dh 64; CS_ATTN_SHAPE=cross for the decoder cross-attention) and prints, averaged over the
first 64 workgroups x 4 waves, the cycles per tile each phase took (s_memtime, 100 MHz-independent shader clock counts)."""
import ctypes, os, shutil, subprocess, sys, tempfile
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("CS_ATTN_CHILD"):
    sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, os.environ["CS_ATTN_CHILD"])
    import numpy as np, torch
    import hip_helpers as hh
    from crossscore_amd import _lib
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    B, H, Lq, Lk, dh = (8, 8, 1369, 6845, 48) if os.environ.get("CS_ATTN_SHAPE") == "cross" else (48, 6, 1370, 1370, 64)
    Q = (torch.randn(B, Lq, H * dh, generator=g) * 1.5).to(dev).to(torch.float16)
    K = (torch.randn(B, Lk, H * dh, generator=g) * 1.5).to(dev).to(torch.float16)
    V = torch.randn(B, Lk, H * dh, generator=g).to(dev).to(torch.float16)
    for _ in range(3): hh.attention(Q, K, V, H, dh)
    torch.cuda.synchronize()
    lib = _lib.load()
    buf = (ctypes.c_ulonglong * (64 * 4 * 8))()
    lib.cs_attn_debug_read.argtypes = [ctypes.c_void_p]
    assert lib.cs_attn_debug_read(buf) == 0
    d = np.frombuffer(buf, dtype=np.uint64).reshape(64, 4, 8).astype(np.float64)
    nt = (Lk + 63) // 64
    d = d[: min(64, (B * H * ((Lq + 127) // 128) + 49) // 50)]
    names = ["K reads + QK^T + max", "exp2 / sum / pack", "V reads + PV issue", "tile write (vmcnt)", "barrier"]
    per = d[..., :5].mean((0, 1)) / nt
    print(f"tiles {nt:.0f}; cycles per tile per wave: total {d[..., 6].mean() / nt:.0f}")
    for n, v in zip(names, per): print(f"  {n:24s} {v:7.0f}")
    us = d[..., 7].mean() / 100.0
    print(f"  loop wall time per workgroup {us:.1f} us -> in-kernel clock {d[..., 6].mean() / us / 1e3:.2f} GHz; workgroups {B * H * ((Lq + 127) // 128)}")
    st = (d[:, 0, 5] - d[:, 0, 5].min()) / 100.0
    dur = d[:, 0, 7] / 100.0
    print("  sampled workgroups (every 50th): start us", st.round(0).astype(int).tolist())
    print("  duration us", dur.round(0).astype(int).tolist())
    span = (st + dur).max()
    print(f"  span {span:.0f} us; mean workgroups in flight ~ {dur.sum() * 50 / span:.0f}")
    print("  per-wave spread of the barrier wait:", (d[..., 4] / nt).mean(0).round(0))
    sys.exit(0)
sys.path.insert(0, R)
from crossscore_amd import build
tmp = tempfile.mkdtemp(prefix="attn_ph_")
pkg = os.path.join(tmp, "crossscore_amd")
shutil.copytree(os.path.join(R, "crossscore_amd"), pkg, ignore=shutil.ignore_patterns("*.so", "build", "__pycache__"))
shutil.copytree(os.path.join(R, "include"), os.path.join(tmp, "include"))
objs, procs = [], []
for s in build.SOURCES if hasattr(build, "SOURCES") else ["api.hip", "gemm.hip", "attention.hip", "elementwise.hip", "preprocess.hip", "panel.hip"]:
    o = os.path.join(tmp, s + ".o"); objs.append(o)
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"] + build.EXTRA_FLAGS.get(s, [])
    if s == "attention.hip": cmd += ["-DCS_ATTN_STAMP"] + ["-DCS_ATTN_" + d for d in os.environ.get("CS_ATTN_DEFS", "").split("+") if d]
    procs.append(subprocess.Popen(cmd + ["-c", os.path.join(pkg, "csrc", s), "-o", o]))
for pr in procs: assert pr.wait() == 0
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(pkg, "libcrossscore_hip.so")] + objs)
sys.exit(subprocess.call([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, CS_ATTN_CHILD=tmp)))
