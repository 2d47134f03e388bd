"""Timing-only ablations / variants of the one-launch patch embedding: CS_PF_VARIANTS="NOLOAD,NOMMA+NOSTORE" builds the library with
-DCS_PF_<..> per variant (macros patch.hip reads) into scratch package roots and prints them; time each with
  rocprofv3 --kernel-trace --stats --output-format csv -d <out> -- python3 tools/patch_bench.py     (CS_PB_ROOT=<root> selects the build)
Ablated builds compute wrong results on purpose."""
import os, subprocess, sys, shutil, tempfile
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from crossscore_amd import build
base = os.environ.get("CS_PF_OUT", tempfile.mkdtemp(prefix="pf_var_"))
for var in [v for v in os.environ.get("CS_PF_VARIANTS", "").split(",") if v]:
    tmp = os.path.join(base, var.replace("+", "_"))
    pkg = os.path.join(tmp, "crossscore_amd")
    shutil.rmtree(tmp, ignore_errors=True)
    shutil.copytree(os.path.join(R, "crossscore_amd"), pkg, ignore=shutil.ignore_patterns("*.so", "build", "__pycache__"))
    shutil.copytree(os.path.join(R, "include"), os.path.join(tmp, "include"))
    objs, procs = [], []
    for s in build.SOURCES:
        o = os.path.join(tmp, s + ".o"); objs.append(o)
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"] + build.EXTRA_FLAGS.get(s, [])
        if s == "patch.hip": cmd += ["-DCS_PF_" + d for d in var.split("+")]
        procs.append(subprocess.Popen(cmd + ["-c", os.path.join(pkg, "csrc", s), "-o", o]))
    for pr in procs: assert pr.wait() == 0
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(pkg, "libcrossscore_hip.so")] + objs)
    for o in objs: os.remove(o)
    print(var, tmp, flush=True)
