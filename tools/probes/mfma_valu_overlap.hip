// Does vector work hide behind MFMAs?  Per "gap": one v_mfma_f32_32x32x16_f16 (independent accumulators, 4 in rotation) followed by NV plain
// VALU instructions (v_fma_f32 on independent registers) and NE v_exp_f32, all inline asm in program order (nothing for the compiler to move).
// Prints shader cycles per gap (s_memtime) for 1, 2 and 3 waves per SIMD (one workgroup of 256 / 512 / 768 threads per CU), all CUs busy.  Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 tools/probes/mfma_valu_overlap.hip -o /tmp/mvo && /tmp/mvo
// Second table: the same instruction multiset as two SEPARATE streams (one wave only MFMAs, its SIMD partner only VALU): what two waves overlap.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__device__ unsigned long long g_out[512];

template <int NV, int NE, int MODE>  // MODE 0: every wave runs MFMA + VALU; 1: waves 0-3 MFMAs only, waves 4-7 (their SIMD partners) the vector work only
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
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (MODE == 0) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[g]) : "v"(a), "v"(b));
#pragma unroll
        for (int i = 0; i < NV; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(c), "v"(d));
#pragma unroll
        for (int i = 0; i < NE; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(y[i]));
      }
    }
  } else if ((wv & 4) == 0) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int g = 0; g < 4; ++g) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[g]) : "v"(a), "v"(b));
    }
  } else {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
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
  if ((threadIdx.x & 63) == 0 && blockIdx.x < 32) g_out[blockIdx.x * 16 + wv] = t1 - t0;
}

// waves_per_simd waves per SIMD inside ONE workgroup per CU (the whole LDS keeps a second workgroup away); returns cycles per gap of waves [w0, w1)
template <int NV, int NE, int MODE>
double run(int waves_per_simd, int w0, int w1) {
  const int iters = 2000, threads = 256 * waves_per_simd;
  const int lds = 150 * 1024;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<NV, NE, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL((k<NV, NE, MODE>), dim3(256), dim3(threads), lds, 0, iters, 1.0f);
  hipDeviceSynchronize();
  unsigned long long h[512];
  hipMemcpyFromSymbol(h, HIP_SYMBOL(g_out), sizeof(h));
  double s = 0; int n = 0;
  for (int b = 0; b < 32; ++b) for (int w = w0; w < w1; ++w) { s += (double)h[b * 16 + w]; ++n; }
  return s / n / (iters * 4.0);
}
#define ROW(NV, NE) printf("  %2d fma + %d exp per MFMA: %6.1f  %6.1f  %6.1f   | split: MFMA wave %6.1f, vector wave %6.1f\n", NV, NE, run<NV, NE, 0>(1, 0, 4), \
    run<NV, NE, 0>(2, 0, 8), run<NV, NE, 0>(3, 0, 12), run<NV, NE, 1>(2, 0, 4), run<NV, NE, 1>(2, 4, 8));
int main() {
  printf("cycles per MFMA gap per wave (32x32x16 f16 = 32 cycles of matrix pipe); waves per SIMD: 1, 2, 3 (every wave the same stream);\n"
         "split: two waves per SIMD, one issues only the MFMAs, its partner only the vector instructions (cycles per gap of each)\n");
  ROW(0, 0) ROW(3, 0) ROW(5, 0) ROW(8, 0) ROW(12, 0) ROW(3, 1) ROW(4, 2) ROW(6, 4) ROW(0, 4)
  return 0;
}
