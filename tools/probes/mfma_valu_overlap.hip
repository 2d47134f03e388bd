// Does vector work hide behind MFMAs?  Per "gap": one v_mfma_f32_32x32x16_f16 (independent accumulators, 4 in rotation) followed by NV plain
// VALU instructions (v_fma_f32 on independent registers) and NE v_exp_f32, all inline asm in program order (nothing for the compiler to move).
// Prints shader cycles per gap (s_memtime) for 1, 2 and 3 waves per SIMD (workgroups of 256 threads, 1 .. 3 per CU: 24 KiB of LDS each caps it),
// all CUs busy.  Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 tools/probes/mfma_valu_overlap.hip -o /tmp/mvo && /tmp/mvo
// Second table: the same instruction multiset as two SEPARATE streams (one wave only MFMAs, its SIMD partner only VALU): what two waves overlap.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__device__ unsigned long long g_out[4096];

template <int NV, int NE, int MODE>  // MODE 0: every wave runs MFMA + VALU; 1: even waves MFMA only, odd waves VALU only (512-thread workgroup)
__global__ void k(int iters, float seed) {
  extern __shared__ char smem[];
  h8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(seed + i); b[i] = (_Float16)(seed * 0.5f + i); }
  f16v acc[4];
  for (int g = 0; g < 4; ++g) for (int e = 0; e < 16; ++e) acc[g][e] = seed;
  float x[12], y[4];
  for (int i = 0; i < 12; ++i) x[i] = seed + i;
  for (int i = 0; i < 4; ++i) y[i] = seed - i;
  const float c = 1.0001f, d = 0.5f;
  const int wv = threadIdx.x >> 6;
  const bool do_m = MODE == 0 || (wv & 4) == 0, do_v = MODE == 0 || (wv & 4) != 0;  // waves w and w + 4 share a SIMD
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (do_m) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[g]) : "v"(a), "v"(b));
      if (do_v) {
#pragma unroll
        for (int i = 0; i < NV; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(c), "v"(d));
#pragma unroll
        for (int i = 0; i < NE; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(y[i]));
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int g = 0; g < 4; ++g) for (int e = 0; e < 16; ++e) s += acc[g][e];
  for (int i = 0; i < 12; ++i) s += x[i];
  for (int i = 0; i < 4; ++i) s += y[i];
  if (s == 12345.678f) smem[0] = 1;  // keep everything alive
  if ((threadIdx.x & 63) == 0 && blockIdx.x < 64) g_out[blockIdx.x * 8 + wv] = t1 - t0;
}

template <int NV, int NE, int MODE>
double run(int wgs_per_cu) {
  const int iters = 2000, threads = MODE ? 512 : 256;
  const int lds = MODE ? 150 * 1024 : (wgs_per_cu == 1 ? 150 * 1024 : wgs_per_cu == 2 ? 70 * 1024 : 48 * 1024);  // caps the workgroups per CU
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<NV, NE, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL((k<NV, NE, MODE>), dim3(256 * wgs_per_cu), dim3(threads), lds, 0, iters, 1.0f);
  hipDeviceSynchronize();
  unsigned long long h[512];
  hipMemcpyFromSymbol(h, HIP_SYMBOL(g_out), sizeof(h));
  double s = 0; int n = 0;
  for (int b = 0; b < 64; ++b) for (int w = 0; w < threads / 64; ++w) { s += (double)h[b * 8 + w]; ++n; }
  return s / n / (iters * 4.0);
}
#define ROW(NV, NE) printf("  %2d fma + %d exp per MFMA: %6.1f  %6.1f  %6.1f   | split over two waves: %6.1f\n", NV, NE, run<NV, NE, 0>(1), run<NV, NE, 0>(2), run<NV, NE, 0>(3), run<NV, NE, 1>(1));
int main() {
  printf("cycles per MFMA gap (32x32x16 f16 = 32 cycles of matrix pipe); waves per SIMD: 1, 2, 3\n");
  ROW(0, 0) ROW(3, 0) ROW(5, 0) ROW(8, 0) ROW(12, 0) ROW(3, 1) ROW(4, 2) ROW(6, 4) ROW(0, 4)
  return 0;
}
