"""Where a K tile of the 256 x 256 x 64-tile GEMM spends its cycles: builds libcrossscore_hip with -DCS_G256_STAMP (s_memtime stamps around the
load segment, the two barriers and the MFMA cluster of each of the four phases; a diagnostic build -- its own run time is not quoted) into a
scratch directory and prints, per wave group (M half 0 / 1), the share of each segment, the cycles per K tile and the in-kernel clock."""
import os, subprocess, sys, shutil, tempfile
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("CS_G256_CHILD"):
    sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, os.environ["CS_G256_CHILD"])
    import ctypes, numpy as np, torch
    import hip_helpers as hh
    from crossscore_amd import _lib
    dev = "cuda"
    lib = ctypes.CDLL(os.path.join(os.environ["CS_G256_CHILD"], "crossscore_amd", "libcrossscore_hip.so"))
    for (M, N, K, epi, name) in ((131520, 2304, 768, _lib.EPI_BIAS_F16, "qkvB"), (131520, 768, 3072, _lib.EPI_RESID_F32, "fc2B"), (131520, 3072, 768, _lib.EPI_BIAS_GELU_F16, "fc1B")):
        A = torch.randn(M, K, device=dev).half(); W = (torch.randn(N, K, device=dev) / K ** 0.5).half(); b = torch.randn(N, device=dev)
        res = torch.randn(M, N, device=dev) if epi == _lib.EPI_RESID_F32 else None
        o = torch.empty(M, N, device=dev, dtype=torch.float32 if res is not None else torch.float16)
        for _ in range(3): hh.gemm(A, W, b, epi, resid=res, out=o)
        torch.cuda.synchronize()
        buf = np.zeros(64 * 8 * 20, dtype=np.uint64)
        assert lib.cs_gemm256_debug_read(buf.ctypes.data_as(ctypes.c_void_p)) == 0
        d = buf.reshape(64, 8, 20).astype(np.float64)
        T = K // 64
        tiles = np.ceil((M / 256) * (N / 256) / 256)
        for grp, sl in (("M half 0 (waves 0-3)", slice(0, 4)), ("M half 1 (waves 4-7)", slice(4, 8))):
            x = d[:, sl, :].reshape(-1, 20)
            tot = x[:, 17]; clk = np.median(x[:, 17] / np.maximum(x[:, 18], 1)) * 0.1
            loop = x[:, :16].sum(1)
            sh = np.median(x[:, :17] / tot[:, None], axis=0)
            names = ["load", "bar", "mfma", "bar"]
            per_phase = " | ".join("P%d " % (p + 1) + " ".join(f"{names[k]} {100 * sh[4 * p + k]:4.1f}" for k in range(4)) for p in range(4))
            print(f"{name} {grp}: {per_phase} | seam+epilogue {100 * sh[16]:4.1f} %   K tile = {np.median(loop) / (tiles * T):6.0f} cycles (stamped), clock {clk:.2f} GHz", flush=True)
    sys.exit(0)
sys.path.insert(0, R)
from crossscore_amd import build
tmp = tempfile.mkdtemp(prefix="g256_")
pkg = os.path.join(tmp, "crossscore_amd")
shutil.copytree(os.path.join(R, "crossscore_amd"), pkg, ignore=shutil.ignore_patterns("*.so", "build", "__pycache__"))
shutil.copytree(os.path.join(R, "include"), os.path.join(tmp, "include"))
objs, procs = [], []
for s in build.SOURCES:
    o = os.path.join(tmp, s + ".o"); objs.append(o)
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"] + build.EXTRA_FLAGS.get(s, [])
    if s == "gemm256.hip": cmd += ["-DCS_G256_STAMP"] + os.environ.get("CS_G256_EXTRA", "").split()
    procs.append(subprocess.Popen(cmd + ["-c", os.path.join(pkg, "csrc", s), "-o", o]))
for pr in procs: assert pr.wait() == 0
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(pkg, "libcrossscore_hip.so")] + objs)
sys.exit(subprocess.call([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, CS_G256_CHILD=tmp)))
