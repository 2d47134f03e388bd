"""Where does a panel-kernel chunk's time go?  Builds libcrossscore_hip with -DCS_PANEL_ABLATE (timing-only variants of cs_panel_kernel
selected by the CS_PANEL_ABL environment variable; their results are wrong by design) into a scratch directory and times each at one full
round of workgroups (256 panels).  bits: 1 no GELU arithmetic, 2 no LDS-DMA after the prologue, 4 A waves skip their MFMAs, 8 B waves skip
theirs, 16 no per-chunk s_barrier, 32 no weight-fragment LDS reads."""
import os, subprocess, sys, shutil, tempfile
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
    import torch
    import hip_helpers as hh
    from test_hip_panel import _make
    dev = torch.device("cuda:0")
    M = 256 * 128
    x, o, w = _make(M, 1, dev)
    img = hh.panel_pack(w["wo"], w["ls1"], w["w1"], w["g2"], w["w2"], w["ls2"])
    res = {}
    for rnd in range(3):
        for abl in (0, 1, 2, 3, 4, 8, 12, 16, 32, 5, 47):
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
    sys.exit(0)
# parent: build the ablation library next to a copy of the package, then run the child against it
tmp = tempfile.mkdtemp(prefix="panel_abl_")
pkg = os.path.join(tmp, "crossscore_amd")
shutil.copytree(os.path.join(R, "crossscore_amd"), pkg, ignore=shutil.ignore_patterns("*.so", "build", "__pycache__"))
srcs = ["api.hip", "gemm.hip", "attention.hip", "elementwise.hip", "preprocess.hip", "panel.hip"]
objs = []
procs = []
for s in srcs:
    o = os.path.join(tmp, s + ".o"); objs.append(o)
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"]
    if s == "panel.hip":
        cmd += ["-fno-slp-vectorize", "-DCS_PANEL_ABLATE"]
    procs.append(subprocess.Popen(cmd + ["-c", os.path.join(pkg, "csrc", s), "-o", o]))
for p in procs:
    assert p.wait() == 0
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(pkg, "libcrossscore_hip.so")] + objs)
env = dict(os.environ, PYTHONPATH=tmp + os.pathsep + os.path.join(R, "tests"))
# the child imports crossscore_amd from the scratch copy (first on PYTHONPATH), tests/ helpers from the repo
code = f"import sys; sys.argv=['x','child']; sys.path.insert(0, {tmp!r}); exec(open({os.path.abspath(__file__)!r}).read().replace('sys.path.insert(0, R);', ''))"
sys.exit(subprocess.call([sys.executable, "-c", code], env=env))
