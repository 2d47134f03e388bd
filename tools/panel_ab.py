"""A/B of panel-kernel builds on one box: CS_PANEL_VARIANTS="A,B+C,file:tools/_ab/panel_old.hip" builds the library with -DCS_PANEL_<..> per variant (experiment macros
panel.hip may read) into scratch directories and times the kernel alone (HIP events) at 48 and 24 images beside the in-tree build."""
import os, subprocess, sys, shutil, tempfile
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("CS_PANEL_CHILD"):
    sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, os.environ["CS_PANEL_CHILD"])
    import torch
    import hip_helpers as hh
    from test_hip_panel import _make, _reference
    dev = torch.device("cuda:0")
    for M in (48 * 1370, 24 * 1370):
        x, o, w = _make(M, 1, dev)
        img = hh.panel_pack(w["wo"], w["ls1"], w["w1"], w["g2"], w["w2"], w["ls2"])
        xk = x.clone()
        hh.encoder_panel(xk, o, img, w["bo"], w["b1"], w["b2"])
        ref, _ = _reference(x[:512], o[:512], w, True, emulate=True)
        err = (xk[:512] - ref).abs().max().item()
        for _ in range(3): hh.encoder_panel(x, o, img, w["bo"], w["b1"], w["b2"])
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): hh.encoder_panel(x, o, img, w["bo"], w["b1"], w["b2"])
        b.record(); torch.cuda.synchronize()
        us = 1e3 * a.elapsed_time(b) / 20
        print(f"  M={M}: {us:7.1f} us  {(4.0 * M * 384 * 1536 + 2.0 * M * 384 * 384) / us / 1e6:5.0f} TFLOP/s  max err (512 rows) {err:.1e}", flush=True)
    sys.exit(0)
sys.path.insert(0, R)
from crossscore_amd import build
for var in [None] + [v for v in os.environ.get("CS_PANEL_VARIANTS", "").split(",") if v]:
    if var is None:
        pkgroot = R
    else:
        tmp = tempfile.mkdtemp(prefix="panel_var_")
        pkg = os.path.join(tmp, "crossscore_amd")
        shutil.copytree(os.path.join(R, "crossscore_amd"), pkg, ignore=shutil.ignore_patterns("*.so", "build", "__pycache__"))
        shutil.copytree(os.path.join(R, "include"), os.path.join(tmp, "include"))
        defs = var.split("+")
        if var.startswith("file:"):  # another version of the kernel's source (e.g. last round's, exported to tools/_ab/), optionally "+MACRO+.."
            shutil.copy(os.path.join(R, defs[0][5:]), os.path.join(pkg, "csrc", "panel.hip"))
            defs = defs[1:]
        objs, procs = [], []
        for s in build.SOURCES:
            o = os.path.join(tmp, s + ".o"); objs.append(o)
            cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"] + build.EXTRA_FLAGS.get(s, [])
            if s == "panel.hip": cmd += ["-DCS_PANEL_" + d for d in defs]
            procs.append(subprocess.Popen(cmd + ["-c", os.path.join(pkg, "csrc", s), "-o", o]))
        for pr in procs: assert pr.wait() == 0
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(pkg, "libcrossscore_hip.so")] + objs)
        pkgroot = tmp
    print("variant", var or "(in-tree build)", flush=True)
    subprocess.call([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, CS_PANEL_CHILD=pkgroot))
