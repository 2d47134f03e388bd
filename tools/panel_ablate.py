"""Where does a panel-kernel chunk's time go?  Builds libcrossscore_hip with -DCS_PANEL_ABLATE (timing-only variants of cs_panel_kernel
selected by the CS_PANEL_ABL environment variable; their results are wrong by design) into a scratch directory and times each at one full
round of workgroups (256 panels).  bits: 1 no GELU arithmetic, 2 no LDS-DMA after the prologue, 4 A waves skip their MFMAs, 8 B waves skip
theirs, 16 no per-chunk s_barrier, 32 no weight-fragment LDS reads."""
import os, subprocess, sys, shutil, tempfile
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("CS_ABL_PKG"):  # child: crossscore_amd comes from the scratch copy that holds the ablation library
    sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, os.environ["CS_ABL_PKG"])
    import torch
    import hip_helpers as hh
    from test_hip_panel import _make
    dev = torch.device("cuda:0")
    M = 256 * 128
    x, o, w = _make(M, 1, dev)
    img = hh.panel_pack(w["wo"], w["ls1"], w["w1"], w["g2"], w["w2"], w["ls2"])
    res = {}
    for rnd in range(3):
        # (variants with BOTH MFMA streams removed -- bits 4 and 8 together -- spill 600 bytes of scratch since the round-5 kernel and one of them
        #  faulted on the GPU box: a timing-only build must not run at all if it is not memory-safe; they are refused here)
        abls = [int(v) for v in os.environ.get('CS_PANEL_ABLS', '0,1,2,16,32').split(',')]
        assert not any((a & 12) == 12 for a in abls), "ABL variants without any MFMAs are not memory-safe any more (round 5)"
        for abl in abls:
            os.environ["CS_PANEL_ABL"] = str(abl)
            for _ in range(2):
                hh.encoder_panel(x, o, img, w["bo"], w["b1"], w["b2"])
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10):
                hh.encoder_panel(x, o, img, w["bo"], w["b1"], w["b2"])
            b.record(); torch.cuda.synchronize()
            res.setdefault(abl, []).append(1e2 * a.elapsed_time(b))
            x.zero_().add_(1.0)  # ablated variants may write junk: keep the rows finite
    for abl, v in res.items():
        print(f"ABL={abl:2d}: min {min(v):7.1f} us  median {sorted(v)[len(v)//2]:7.1f} us", flush=True)
    # phase stamps of the real kernel (ABL = 0): per role, median over blocks < 64 and the four pairs
    import ctypes, numpy as np
    from crossscore_amd import _lib
    os.environ["CS_PANEL_ABL"] = "0"
    for Mrows in (M, 129 * 128):
        xs, os_, ws = _make(Mrows, 1, dev)
        for _ in range(3):
            hh.encoder_panel(xs, os_, img, w["bo"], w["b1"], w["b2"])
        torch.cuda.synchronize()
        buf = np.zeros(64 * 8 * 16, dtype=np.uint64)
        lib = ctypes.CDLL(os.path.join(os.environ["CS_ABL_PKG"], "crossscore_amd", "libcrossscore_hip.so"))
        assert lib.cs_panel_debug_read(buf.ctypes.data_as(ctypes.c_void_p)) == 0
        d = buf.reshape(64, 8, 16).astype(np.int64)
        names = ["start->ring/bias|x loaded", "T0", "out-proj phase", "LN hand-off", "MLP", "tail/epilogue"]
        for role, sl in (("A", slice(0, 4)), ("B", slice(4, 8))):
            ph = np.diff(d[:, sl, 0:6], axis=2).reshape(-1, 5)
            clk = (d[:, sl, 5] - d[:, sl, 0]) / np.maximum(d[:, sl, 9] - d[:, sl, 8], 1) * 100.0
            print(f"M={Mrows} role {role}: phase cycles median", dict(zip(["prologue", "outproj", "handoff", "mlp", "tail"], np.median(ph, axis=0).astype(int).tolist())),
                  f"total {int(np.median(ph.sum(1)))} cycles; clock ~{np.median(clk):.0f} MHz", flush=True)
    # who is late at the MLP phase's unit boundaries (units 40 .. 55 of blocks 0 .. 3): arrival relative to the first arriver, and the release
    if hasattr(lib, "cs_panel_bar_read"):
        os.environ["CS_PANEL_ABL"] = "0"
        xs, os_, ws = _make(M, 1, dev)
        for _ in range(3):
            hh.encoder_panel(xs, os_, img, w["bo"], w["b1"], w["b2"])
        torch.cuda.synchronize()
        bb = np.zeros(4 * 8 * 16 * 2, dtype=np.uint64)
        assert lib.cs_panel_bar_read(bb.ctypes.data_as(ctypes.c_void_p)) == 0
        b = bb.reshape(4, 8, 16, 2).astype(np.int64)
        for blk in range(2):
            print(f"block {blk}: per boundary v = 40..55: arrival of waves A0..A3 B0..B3 relative to the first (cycles) | release - last arrival | interval since the previous release")
            prev = None
            for v in range(16):
                arr = b[blk, :, v, 0]; rel = b[blk, :, v, 1]
                first = arr.min()
                print(f"  v={40 + v} (gap {6 if v % 2 == 0 else 18}):", " ".join(f"{int(a - first):5d}" for a in arr), "|", int(rel.min() - arr.max()), "|", (int(rel.min() - prev) if prev is not None else "-"))
                prev = rel.min()
    # where the transitions' time goes (ABL = 64: s_memtime around the LDS drain, the vmcnt wait and the barrier of every transition)
    os.environ["CS_PANEL_ABL"] = "64"
    xs, os_, ws = _make(M, 1, dev)
    for _ in range(3):
        hh.encoder_panel(xs, os_, img, w["bo"], w["b1"], w["b2"])
    torch.cuda.synchronize()
    assert lib.cs_panel_debug_read(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    d = buf.reshape(64, 8, 16).astype(np.int64)
    for role, sl in (("A", slice(0, 4)), ("B", slice(4, 8))):
        print(f"transitions, role {role}: cycles summed over the launch: LDS drain {int(np.median(d[:, sl, 10]))}, vmcnt {int(np.median(d[:, sl, 11]))}, "
              f"barrier {int(np.median(d[:, sl, 12]))}; launch {int(np.median(d[:, sl, 5] - d[:, sl, 0]))}", flush=True)
    sys.exit(0)
# parent: build the ablation library next to a copy of the package, then run the child against it
tmp = tempfile.mkdtemp(prefix="panel_abl_")
pkg = os.path.join(tmp, "crossscore_amd")
shutil.copytree(os.path.join(R, "crossscore_amd"), pkg, ignore=shutil.ignore_patterns("*.so", "build", "__pycache__"))
shutil.copytree(os.path.join(R, "include"), os.path.join(tmp, "include"))
srcs = ["api.hip", "gemm.hip", "gemm256.hip", "attention.hip", "elementwise.hip", "preprocess.hip", "panel.hip", "patch.hip", "rowln.hip"]
objs = []
procs = []
for s in srcs:
    o = os.path.join(tmp, s + ".o"); objs.append(o)
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"]
    if s == "panel.hip":
        cmd += ["-fno-slp-vectorize", "-DCS_PANEL_ABLATE"] + os.environ.get("CS_PANEL_EXTRA", "").split()
    procs.append(subprocess.Popen(cmd + ["-c", os.path.join(pkg, "csrc", s), "-o", o]))
for p in procs:
    assert p.wait() == 0
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(pkg, "libcrossscore_hip.so")] + objs)
sys.exit(subprocess.call([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, CS_ABL_PKG=tmp)))
