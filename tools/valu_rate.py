"""Issue cost (cycles per wave-instruction) of v_exp_f32 / v_exp_f16 / v_cvt_f16_f32 / v_rcp_f32 / v_log_f32 / v_floor_f32 with one and
three waves per SIMD (8 independent chains per wave).  Builds tools/valu_rate.hip with hipcc on the box."""
import ctypes as C, os, subprocess, tempfile
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
so = os.path.join(tempfile.mkdtemp(prefix="valu_rate_"), "libvalu_rate.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-Wno-unused-value", "-Wno-unused-result", "-o", so, os.path.join(R, "tools", "valu_rate.hip")])
lib = C.CDLL(so)
out = torch.zeros(1 << 20, device="cuda"); clk = torch.zeros(4096, dtype=torch.int64, device="cuda")
names = ["v_exp_f32", "v_exp_f16", "v_cvt_f16_f32", "v_rcp_f32", "v_log_f32", "v_floor_f32", "v_fma_f32", "v_pk_fma_f32", "v_fma_f32, one dependent chain", "v_pk_fma_f32, one dependent chain", "v_pk_fma_f32, two dependent chains"]
iters = 20000
for threads, label in ((256, "1 wave per SIMD"), (512, "2 waves per SIMD"), (768, "3 waves per SIMD")):
    for w, n in enumerate(names):
        assert lib.valu_rate_run(w, 256, threads, iters, C.c_void_p(out.data_ptr()), C.c_void_p(clk.data_ptr())) == 0
        c = float(np.median(clk.cpu().numpy()[:256]))
        waves_per_simd = threads // 256
        print(f"{label}: {n:38s} {c / (iters * 8):6.2f} cycles per instruction and wave, {c / (iters * 8 * waves_per_simd):6.2f} per instruction on the SIMD", flush=True)
