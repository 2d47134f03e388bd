// Issue cost of a few vector instructions on one wave per SIMD (tools/valu_rate.py): dependent-free streams of 8 independent chains,
// timed with s_memtime.  Not part of the product.
#include <hip/hip_runtime.h>
#define CHAIN8(OP)                                                                                         \
  asm volatile(OP " %0, %0\n\t" OP " %1, %1\n\t" OP " %2, %2\n\t" OP " %3, %3\n\t" OP " %4, %4\n\t" OP " %5, %5\n\t" OP " %6, %6\n\t" OP " %7, %7" \
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7))
template <int WHICH>
__global__ void rate_kernel(float* out, unsigned long long* clk, int iters) {
  float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    if constexpr (WHICH == 0) CHAIN8("v_exp_f32");
    if constexpr (WHICH == 1) CHAIN8("v_exp_f16");
    if constexpr (WHICH == 2) CHAIN8("v_cvt_f16_f32");
    if constexpr (WHICH == 3) CHAIN8("v_rcp_f32");
    if constexpr (WHICH == 4) CHAIN8("v_log_f32");
    if constexpr (WHICH == 5) CHAIN8("v_floor_f32");
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
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
  }
  return (int)hipDeviceSynchronize();
}
