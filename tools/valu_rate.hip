// Issue cost of a few vector instructions on one wave per SIMD (tools/valu_rate.py): dependent-free streams of 8 independent chains,
// timed with s_memtime.  Not part of the product.
#include <hip/hip_runtime.h>
#define CHAIN8(OP)                                                                                         \
  asm volatile(OP " %0, %0\n\t" OP " %1, %1\n\t" OP " %2, %2\n\t" OP " %3, %3\n\t" OP " %4, %4\n\t" OP " %5, %5\n\t" OP " %6, %6\n\t" OP " %7, %7" \
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7))
template <int WHICH>
__global__ void rate_kernel(float* out, unsigned long long* clk, int iters) {
  float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    if constexpr (WHICH == 0) CHAIN8("v_exp_f32");
    if constexpr (WHICH == 1) CHAIN8("v_exp_f16");
    if constexpr (WHICH == 2) CHAIN8("v_cvt_f16_f32");
    if constexpr (WHICH == 3) CHAIN8("v_rcp_f32");
    if constexpr (WHICH == 4) CHAIN8("v_log_f32");
    if constexpr (WHICH == 5) CHAIN8("v_floor_f32");
    if constexpr (WHICH == 6)
      asm volatile("v_fma_f32 %0, %0, %0, %1\n\tv_fma_f32 %1, %1, %1, %2\n\tv_fma_f32 %2, %2, %2, %3\n\tv_fma_f32 %3, %3, %3, %4\n\t"
                   "v_fma_f32 %4, %4, %4, %5\n\tv_fma_f32 %5, %5, %5, %6\n\tv_fma_f32 %6, %6, %6, %7\n\tv_fma_f32 %7, %7, %7, %0"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    if constexpr (WHICH == 8)  // one dependent chain
      asm volatile("v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %0, %0, %0, %0\n\t"
                   "v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %0, %0, %0, %0" : "+v"(a0));
    if constexpr (WHICH == 9)
      asm volatile("v_pk_fma_f32 %0, %0, %0, %0\n\tv_pk_fma_f32 %0, %0, %0, %0\n\tv_pk_fma_f32 %0, %0, %0, %0\n\tv_pk_fma_f32 %0, %0, %0, %0\n\t"
                   "v_pk_fma_f32 %0, %0, %0, %0\n\tv_pk_fma_f32 %0, %0, %0, %0\n\tv_pk_fma_f32 %0, %0, %0, %0\n\tv_pk_fma_f32 %0, %0, %0, %0" : "+v"(p0));
    if constexpr (WHICH == 10)  // two dependent packed chains interleaved
      asm volatile("v_pk_fma_f32 %0, %0, %0, %0\n\tv_pk_fma_f32 %1, %1, %1, %1\n\tv_pk_fma_f32 %0, %0, %0, %0\n\tv_pk_fma_f32 %1, %1, %1, %1\n\t"
                   "v_pk_fma_f32 %0, %0, %0, %0\n\tv_pk_fma_f32 %1, %1, %1, %1\n\tv_pk_fma_f32 %0, %0, %0, %0\n\tv_pk_fma_f32 %1, %1, %1, %1" : "+v"(p0), "+v"(p1));
    if constexpr (WHICH == 7)
      asm volatile("v_pk_fma_f32 %0, %0, %0, %1\n\tv_pk_fma_f32 %1, %1, %1, %2\n\tv_pk_fma_f32 %2, %2, %2, %3\n\tv_pk_fma_f32 %3, %3, %3, %0\n\t"
                   "v_pk_fma_f32 %0, %0, %0, %1\n\tv_pk_fma_f32 %1, %1, %1, %2\n\tv_pk_fma_f32 %2, %2, %2, %3\n\tv_pk_fma_f32 %3, %3, %3, %0"
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0[0] + p1[1] + p2[0] + p3[1];
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}
extern "C" int valu_rate_run(int which, int blocks, int threads, int iters, float* out, unsigned long long* clk) {
  switch (which) {
    case 0: hipLaunchKernelGGL(rate_kernel<0>, dim3(blocks), dim3(threads), 0, 0, out, clk, iters); break;
    case 1: hipLaunchKernelGGL(rate_kernel<1>, dim3(blocks), dim3(threads), 0, 0, out, clk, iters); break;
    case 2: hipLaunchKernelGGL(rate_kernel<2>, dim3(blocks), dim3(threads), 0, 0, out, clk, iters); break;
    case 3: hipLaunchKernelGGL(rate_kernel<3>, dim3(blocks), dim3(threads), 0, 0, out, clk, iters); break;
    case 4: hipLaunchKernelGGL(rate_kernel<4>, dim3(blocks), dim3(threads), 0, 0, out, clk, iters); break;
    case 5: hipLaunchKernelGGL(rate_kernel<5>, dim3(blocks), dim3(threads), 0, 0, out, clk, iters); break;
    case 6: hipLaunchKernelGGL(rate_kernel<6>, dim3(blocks), dim3(threads), 0, 0, out, clk, iters); break;
    case 7: hipLaunchKernelGGL(rate_kernel<7>, dim3(blocks), dim3(threads), 0, 0, out, clk, iters); break;
    case 8: hipLaunchKernelGGL(rate_kernel<8>, dim3(blocks), dim3(threads), 0, 0, out, clk, iters); break;
    case 9: hipLaunchKernelGGL(rate_kernel<9>, dim3(blocks), dim3(threads), 0, 0, out, clk, iters); break;
    case 10: hipLaunchKernelGGL(rate_kernel<10>, dim3(blocks), dim3(threads), 0, 0, out, clk, iters); break;
  }
  return (int)hipDeviceSynchronize();
}
