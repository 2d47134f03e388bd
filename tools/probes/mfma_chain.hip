// Cycles per v_mfma_f32_32x32x16 when N accumulators are updated round-robin by one wave per SIMD (N = 1: a single dependent accumulation
// chain), f16 and bf16 forms, operands in registers.  Build: hipcc --offload-arch=gfx950 -O3 -o mfma_chain mfma_chain.hip ; run: ./mfma_chain
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
template <int N, bool BF, bool RANDOM, int FILL = 0>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters) {
  f16v acc[N];
  for (int i = 0; i < N; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = threadIdx.x * 0.001f + i;
  h8 a, b; unsigned filler = threadIdx.x, filler2 = 7;
  unsigned rs = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  for (int e = 0; e < 8; ++e) {
    rs = rs * 1664525u + 1013904223u; a[e] = RANDOM ? (_Float16)(((int)(rs >> 8) % 2001 - 1000) * 0.001f) : (_Float16)0.0f;
    rs = rs * 1664525u + 1013904223u; b[e] = RANDOM ? (_Float16)(((int)(rs >> 8) % 2001 - 1000) * 0.001f) : (_Float16)0.0f;
  }
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 48 / N; ++u)
#pragma unroll
      for (int i = 0; i < N; ++i) {
        if constexpr (BF) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b8, a), __builtin_bit_cast(b8, b), acc[i], 0, 0, 0);
        else acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
        if constexpr (FILL == 1) asm volatile("v_mov_b32 %0, %0" : "+v"(filler));
        if constexpr (FILL == 2) asm volatile("s_nop 0");
        if constexpr (FILL == 3) asm volatile("v_pk_fma_f16 %0, %0, %0, %0\n\tv_pk_fma_f16 %1, %1, %1, %1\n\tv_pk_fma_f16 %0, %0, %0, %0\n\tv_pk_fma_f16 %1, %1, %1, %1" : "+v"(filler), "+v"(filler2));
        if constexpr (FILL == 4) asm volatile("s_waitcnt lgkmcnt(7)");
        __builtin_amdgcn_sched_barrier(0);
      }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < N; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s + filler + filler2;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int N, bool BF, bool RANDOM, int FILL = 0> void run(float* out, unsigned long long* cyc, int blocks) {
  const int iters = 500;
  for (int r = 0; r < 20; ++r) k<N, BF, RANDOM, FILL><<<blocks, 256>>>(out, cyc, iters);
  hipDeviceSynchronize();
  unsigned long long h[256];
  hipMemcpy(h, cyc, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
  double m = 0; for (int i = 0; i < blocks; ++i) m += h[i]; m /= blocks;
  printf("%s %s N=%2d accumulators, filler %d, %3d blocks: %.1f cycles per MFMA\n", RANDOM ? "random" : "zeros ", BF ? "bf16" : "f16 ", N, FILL, blocks, m / (iters * 48.0));
}
int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
  for (int blocks : {256}) {
    run<1, false, true, 0>(out, cyc, blocks); run<1, false, true, 1>(out, cyc, blocks); run<1, false, true, 2>(out, cyc, blocks); run<1, false, true, 3>(out, cyc, blocks); run<1, false, true, 4>(out, cyc, blocks);
    run<2, false, true, 1>(out, cyc, blocks); run<2, false, true, 3>(out, cyc, blocks); run<4, false, true, 3>(out, cyc, blocks); run<12, false, true, 3>(out, cyc, blocks);
  }
  return 0;
}
