"""Builds libcrossscore_hip.so for gfx950 with hipcc (in-tree, next to this file)."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libcrossscore_hip.so")
SOURCES = ["api.hip", "gemm.hip", "gemm256.hip", "attention.hip", "elementwise.hip", "preprocess.hip", "panel.hip", "panel4.hip", "patch.hip", "rowln.hip"]


# panel.hip: its GELU arithmetic shares one wave's issue stream with the MFMAs; SLP-packed v_pk_fma_f32 (dependent-issue nops)
# costs more there than scalar fma chains
EXTRA_FLAGS = {"panel.hip": ["-fno-slp-vectorize"], "panel4.hip": ["-fno-slp-vectorize"]}


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(os.path.dirname(HERE), "include", "crossscore_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    for s in SOURCES:
        o = os.path.join(HERE, "build", s.replace(".hip", ".o"))
        objs.append(o)
        cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"] + EXTRA_FLAGS.get(s, []) + ["-c", os.path.join(CSRC, s), "-o", o]
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd) + "\n" + out)
        if verbose and out.strip():
            print(out, file=sys.stderr)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed: " + r.stdout)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
