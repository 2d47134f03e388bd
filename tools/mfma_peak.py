"""What this box's chip sustains on bare fp16 MFMA loops (random data, operands in registers, 2 waves per SIMD, every CU busy) for the two
MFMA shapes the kernels use: v_mfma_f32_32x32x16_f16 (panel, attention) and v_mfma_f32_16x16x32_f16 (the GEMMs).  Builds tools/mfma_peak.hip
with hipcc on the box.  The in-kernel clock is s_memtime / s_memrealtime x 100 MHz."""
import ctypes as C, os, subprocess, sys, tempfile
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
so = os.path.join(tempfile.mkdtemp(prefix="mfma_peak_"), "libmfma_peak.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-Wno-unused-value", "-Wno-unused-result", "-o", so, os.path.join(R, "tools", "mfma_peak.hip")])
lib = C.CDLL(so)
src = (torch.randn(65536 * 8, device="cuda") * 1.0).to(torch.float16)
out = torch.zeros(512 * 512, device="cuda"); clk = torch.zeros(512, dtype=torch.int64, device="cuda"); ms = C.c_float()
for blocks in (256, 128, 32):
    for shape, flop_per_iter in ((32, 16 * 2 * 32 * 32 * 16), (16, 32 * 2 * 16 * 16 * 32)):
        iters = 20000
        rc = lib.mfma_peak_run(shape, blocks, iters, C.c_void_p(src.data_ptr()), C.c_void_p(out.data_ptr()), C.c_void_p(clk.data_ptr()), C.byref(ms))
        torch.cuda.synchronize()
        c = clk.cpu().numpy().reshape(-1, 2)[:blocks]
        ghz = float(np.median(c[:, 0] / np.maximum(c[:, 1], 1))) * 0.1
        tf = blocks * 8 * iters * flop_per_iter / (ms.value * 1e-3) / 1e12
        cyc = float(np.median(c[:, 0])) / (iters * (16 if shape == 32 else 32))
        print(f"{blocks:3d} CUs busy, {shape}x{shape}: {tf:7.1f} TFLOP/s  ({ms.value:.2f} ms), in-kernel clock {ghz:.2f} GHz, {cyc:.1f} cycles per MFMA per wave", flush=True)
