// Bare MFMA loops on random fp16 data, operands in registers: what the chip sustains per MFMA shape (MI355X_MICROARCH.md, DVFS give-back
// item 7: the clock the chip holds under load depends on the shape).  WAVES waves per SIMD, NACC independent accumulators per wave.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int SHAPE>
__global__ __launch_bounds__(512, 2) void mfma_loop(const h8* __restrict__ src, float* __restrict__ out, int iters, unsigned long long* clk) {
  const int tid = threadIdx.x + blockIdx.x * blockDim.x;
  h8 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i] = src[(tid * 8 + i) & 65535]; b[i] = src[(tid * 8 + 4 + i) & 65535]; }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  float sink = 0.f;
  if constexpr (SHAPE == 32) {
    f16v acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + u) & 3], b[i], acc[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) sink += acc[i][0] + acc[i][7];
  } else {
    f4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(i + u) & 3], b[i & 3], acc[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) sink += acc[i][0] + acc[i][3];
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[tid] = sink;
  if (threadIdx.x == 0 && blockIdx.x < 256) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

extern "C" int mfma_peak_run(int shape, int blocks, int iters, const void* src, float* out, unsigned long long* clk, float* ms_out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0, 0);
    if (shape == 32) hipLaunchKernelGGL(mfma_loop<32>, dim3(blocks), dim3(512), 0, 0, (const h8*)src, out, iters, clk);
    else hipLaunchKernelGGL(mfma_loop<16>, dim3(blocks), dim3(512), 0, 0, (const h8*)src, out, iters, clk);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    hipEventElapsedTime(ms_out, e0, e1);
  }
  return (int)hipGetLastError();
}
