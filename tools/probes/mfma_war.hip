// Does a vector instruction that overwrites an MFMA's A / B source register shortly after the MFMA was issued corrupt the product?
// D = A x B with known operands; N independent VALU fillers after the v_mfma_f32_32x32x16_f16, then v_mov_b32 into the first register of
// the B (or A) operand; compare D with the clean product.  Build: hipcc --offload-arch=gfx950 -O3 -o mfma_war mfma_war.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f16v __attribute__((ext_vector_type(16)));
template <int N, int WHICH, int BUSY>
__global__ __launch_bounds__(64) void k(float* out) {
  // A = v[0:3], B = v[4:7]: every half = 1.0 (0x3c00); C = 0 -> D = 16 everywhere.  Overwrite with 0 afterwards.
  f16v d;
  asm volatile(
      "v_mov_b32 v0, 0x3c003c00\n\tv_mov_b32 v1, 0x3c003c00\n\tv_mov_b32 v2, 0x3c003c00\n\tv_mov_b32 v3, 0x3c003c00\n\t"
      "v_mov_b32 v4, 0x3c003c00\n\tv_mov_b32 v5, 0x3c003c00\n\tv_mov_b32 v6, 0x3c003c00\n\tv_mov_b32 v7, 0x3c003c00\n\t"
      "v_mov_b32 v8, 0\n\ts_nop 7\n\t"
      ".if %3\n\tv_mfma_f32_32x32x16_f16 a[0:15], v[0:3], v[0:3], 0\n\tv_mfma_f32_32x32x16_f16 a[16:31], v[0:3], v[0:3], 0\n\tv_mfma_f32_32x32x16_f16 a[0:15], v[0:3], v[0:3], 0\n\t.endif\n\t"
      "v_mfma_f32_32x32x16_f16 %0, v[0:3], v[4:7], 0\n\t"
      ".rept %1\n\tv_mov_b32 v9, v8\n\t.endr\n\t"
      ".if %2 == 0\n\tv_mov_b32 v4, 0\n\tv_mov_b32 v5, 0\n\tv_mov_b32 v6, 0\n\tv_mov_b32 v7, 0\n\t.else\n\tv_mov_b32 v0, 0\n\tv_mov_b32 v1, 0\n\tv_mov_b32 v2, 0\n\tv_mov_b32 v3, 0\n\t.endif\n\t"
      "s_nop 15\n\ts_nop 15\n\ts_nop 15"
      : "=v"(d) : "n"(N), "n"(WHICH), "n"(BUSY) : "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "a0", "a31");
  float s = 0; for (int r = 0; r < 16; ++r) s += d[r];
  out[threadIdx.x] = s;
}
template <int N, int WHICH, int BUSY = 0> void run(float* out) {
  k<N, WHICH, BUSY><<<1, 64>>>(out); hipDeviceSynchronize();
  float h[64]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  float mn = 1e9, mx = -1e9; for (int i = 0; i < 64; ++i) { mn = h[i] < mn ? h[i] : mn; mx = h[i] > mx ? h[i] : mx; }
  printf("%s%s overwritten %2d fillers after the MFMA: per-lane sum of D min %.0f max %.0f (clean: 256)\n", BUSY ? "(pipe busy: three MFMAs queued in front) " : "", WHICH ? "A" : "B", N, mn, mx);
}
int main() {
  float* out; hipMalloc(&out, 256);
  run<0, 0>(out); run<1, 0>(out); run<2, 0>(out); run<4, 0>(out); run<6, 0>(out); run<8, 0>(out); run<12, 0>(out); run<16, 0>(out);
  run<0, 1>(out); run<1, 1>(out); run<2, 1>(out); run<4, 1>(out); run<6, 1>(out); run<8, 1>(out); run<12, 1>(out);
  run<0, 0, 1>(out); run<1, 0, 1>(out); run<2, 0, 1>(out); run<4, 0, 1>(out); run<6, 0, 1>(out); run<8, 0, 1>(out); run<12, 0, 1>(out); run<16, 0, 1>(out); run<24, 0, 1>(out);
  run<0, 1, 1>(out); run<2, 1, 1>(out); run<4, 1, 1>(out); run<8, 1, 1>(out); run<16, 1, 1>(out); run<24, 1, 1>(out);
  return 0;
}
