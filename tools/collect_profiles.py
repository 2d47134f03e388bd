"""Copies the summaries of one tools/profile_round.sh run (gpurun_out/prof_<tag>/) into profiles/<tag>_* and stamps the traffic table with
the commit it was measured at.  usage: collect_profiles.py r03"""
import json, os, shutil, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
O = os.path.join(R, "gpurun_out", "prof_" + tag)
P = os.path.join(R, "profiles")
commit = subprocess.run(["git", "-C", R, "rev-parse", "--short", "HEAD"], stdout=subprocess.PIPE, text=True).stdout.strip()
def line(src):  # last JSON line of a bench output
    return json.loads([l for l in open(src).read().splitlines() if l.startswith("{")][-1])
for src, dst in (("bench.json", "bench.json"), ("bench_bf16.json", "bench_bf16.json"), ("bench_cfg3.json", "bench_cfg3.json"),
                 ("bench_cfg4.json", "bench_cfg4.json"), ("bench_cfg5.json", "bench_cfg5.json")):
    json.dump(line(os.path.join(O, src)), open(os.path.join(P, f"{tag}_{dst}"), "w"), indent=1)
for src, dst in (("default_kernel_stats.csv", "kernel_stats.csv"), ("alone_kernel_stats.csv", "kernel_stats_alone.csv"),
                 ("alone_cfg4_kernel_stats.csv", "kernel_stats_alone_cfg4.csv"), ("attn_pmc.json", "attn_pmc.json"), ("kernel_pmc.json", "kernel_pmc.json"),
                 ("timeline.json", "timeline.json"), ("gemm_vs_blas.log", "gemm_vs_blas.txt"), ("mfma_peak.log", "mfma_peak.txt"),
                 ("gemm256_phases.log", "gemm256_phases.txt")):
    keep = {"mfma_peak.txt": "CUs busy", "gemm256_phases.txt": "M half", "gemm_vs_blas.txt": "M="}.get(dst)
    if keep:  # text logs: only the result lines (compiler warnings and loader notices stay behind)
        open(os.path.join(P, f"{tag}_{dst}"), "w").write("".join(l for l in open(os.path.join(O, src)) if keep in l))
    else:
        shutil.copy(os.path.join(O, src), os.path.join(P, f"{tag}_{dst}"))
for src, dst, keep in (("panel4_ab.log", "panel4_ab.txt", "impl"), ("panel4_pmc.txt", "panel4_pmc.txt", ""), ("panel4_phases.txt", "panel4_phases.txt", "")):
    if os.path.exists(os.path.join(O, src)):
        open(os.path.join(P, f"{tag}_{dst}"), "w").write("".join(l for l in open(os.path.join(O, src)) if keep in l and "amdgpu.ids" not in l))
import glob
for f in glob.glob(os.path.join(O, "stats_panel4", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(P, f"{tag}_kernel_stats_panel4.csv"))
for src, dst, cmd in (("alone_hbm_traffic.json", "hbm_traffic.json", "bench.py --no-cpu-baseline --no-eager --no-cfg4 --inflight 1 --lanes 1 --chunk 48 --steps 3 --warmup 1"),
                      ("alone_cfg4_hbm_traffic.json", "hbm_traffic_cfg4.json", "bench.py --workload cfg4 --no-cpu-baseline --no-eager --no-cfg4 --inflight 1 --lanes 1 --chunk 96 --steps 2 --warmup 1")):
    t = json.load(open(os.path.join(O, src)))
    t["_meta"] = {"commit": commit, "command": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE -- python3 {cmd} (tools/profile_round.sh)",
                  "note": "FETCH_SIZE counts 64 B per 128-B request on gfx950 (fetch_bytes_corrected = 2 x); hbm_bytes_per_launch = fetch_bytes_corrected + write_bytes"}
    json.dump(t, open(os.path.join(P, f"{tag}_{dst}"), "w"), indent=1, sort_keys=True)
# register / spill / in-flight-load audit of the hand-scheduled kernels, from a fresh -S build (CPU only: hipcc cross-compiles)
import tempfile
sys.path.insert(0, R)
from crossscore_amd.build import EXTRA_FLAGS  # the flags the library is built with
tmp = tempfile.mkdtemp(prefix="audit_")
lines = [f"asm audit at {commit}: hipcc --offload-arch=gfx950 -O3 -S (crossscore_amd/build.py's flags per file) of the MFMA kernels (tools/asm_audit.py, tools/asm_audit_gl.py)"]
for src in ("gemm256.hip", "panel.hip", "panel4.hip", "patch.hip", "attention.hip", "gemm.hip"):
    out = os.path.join(tmp, src + ".s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"] + EXTRA_FLAGS.get(src, []) +
                          ["-S", "--cuda-device-only", "-o", out, os.path.join(R, "crossscore_amd", "csrc", src)], stderr=subprocess.DEVNULL)
    name = None
    for l in open(out):
        l = l.strip()
        if l.startswith(".name:"): name = l.split()[-1]
        elif l.startswith(".vgpr_count:"): vg = l.split()[-1]
        elif l.startswith(".vgpr_spill_count:") and name:
            lines.append(f"{src:14s} {subprocess.run(['c++filt', name], stdout=subprocess.PIPE, text=True).stdout.strip()[:110]:110s} vgpr {vg:>3s} spill {l.split()[-1]}")
    if src in ("gemm256.hip", "panel.hip"):
        for tool in (("asm_audit.py", "asm_audit_gl.py") if src == "gemm256.hip" else ("asm_audit.py",)):
            r = subprocess.run([sys.executable, os.path.join(R, "tools", tool), out], stdout=subprocess.PIPE, text=True)
            lines.append(f"{src:14s} {tool}: {r.stdout.strip().splitlines()[-1]}")
open(os.path.join(P, f"{tag}_asm_audit.txt"), "w").write("\n".join(lines) + "\n")
print("collected into", P, "at", commit)
