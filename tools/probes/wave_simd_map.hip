// Which SIMD does wave w of a workgroup run on?  Prints HW_ID fields (gfx9: wave_id [3:0], simd_id [5:4], cu_id [11:8], se_id [15:13]) per wave
// for workgroups of 256 and 512 threads.  Build and run on the GPU box:  hipcc --offload-arch=gfx950 -O2 tools/probes/wave_simd_map.hip -o /tmp/wsm && /tmp/wsm
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void probe(unsigned* out) {
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = id;
}
int main() {
  unsigned* d;
  hipMalloc(&d, 4096 * sizeof(unsigned));
  for (int threads : {256, 512}) {
    const int nb = 3, nw = threads / 64;
    hipMemset(d, 0, 4096 * sizeof(unsigned));
    hipLaunchKernelGGL(probe, dim3(nb), dim3(threads), 0, 0, d);
    unsigned h[64];
    hipMemcpy(h, d, nb * nw * sizeof(unsigned), hipMemcpyDeviceToHost);
    for (int b = 0; b < nb; ++b) {
      printf("threads %d block %d: simd of waves 0..%d =", threads, b, nw - 1);
      for (int w = 0; w < nw; ++w) printf(" %u", (h[b * nw + w] >> 4) & 3);
      printf("   (cu %u se %u; wave slots", (h[b * nw] >> 8) & 15, (h[b * nw] >> 13) & 7);
      for (int w = 0; w < nw; ++w) printf(" %u", h[b * nw + w] & 15);
      printf(")\n");
    }
  }
  return 0;
}
