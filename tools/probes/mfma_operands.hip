// Cycles per v_mfma_f32_32x32x16_f16 for explicit register placements (one wave per SIMD, 512-register kernel): accumulators in AGPRs,
// A and B operands in chosen VGPR quads.  Reproduces the fc1 loop of csrc/panel4.hip (two accumulators alternating, A cycling over 8 quads,
// B over 24 quads).  Build: hipcc --offload-arch=gfx950 -O3 -o mfma_operands mfma_operands.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#define STR2(x) #x
#define STR(x) STR2(x)
#define MF(acc, a, b) "v_mfma_f32_32x32x16_f16 a[" STR(acc) ":" STR(acc+15) "], v[" STR(a) ":" STR(a+3) "], v[" STR(b) ":" STR(b+3) "], a[" STR(acc) ":" STR(acc+15) "]\n\t"
// MODE 0: A cycles v[0..31] (8 quads), B cycles v[64..159] (24 quads), acc alternates a0 / a16   (the panel4 fc1 pattern)
// MODE 1: same but B fixed v[64:67]        MODE 2: same but A fixed v[0:3]        MODE 3: A and B fixed
// MODE 4: as 0 with B quads placed at v[58..153] (the compiler's placement)
template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned long long* cyc, int iters) {
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
  for (int it = 0; it < iters; ++it) {
    if constexpr (MODE == 0)
      asm volatile(MF(0,0,64) MF(16,4,64) MF(0,8,68) MF(16,12,68) MF(0,16,72) MF(16,20,72) MF(0,24,76) MF(16,28,76)
                   MF(0,0,80) MF(16,4,80) MF(0,8,84) MF(16,12,84) MF(0,16,88) MF(16,20,88) MF(0,24,92) MF(16,28,92)
                   MF(0,0,96) MF(16,4,96) MF(0,8,100) MF(16,12,100) MF(0,16,104) MF(16,20,104) MF(0,24,108) MF(16,28,108)
                   MF(0,0,112) MF(16,4,112) MF(0,8,116) MF(16,12,116) MF(0,16,120) MF(16,20,120) MF(0,24,124) MF(16,28,124)
                   MF(0,0,128) MF(16,4,128) MF(0,8,132) MF(16,12,132) MF(0,16,136) MF(16,20,136) MF(0,24,140) MF(16,28,140)
                   MF(0,0,144) MF(16,4,144) MF(0,8,148) MF(16,12,148) MF(0,16,152) MF(16,20,152) MF(0,24,156) MF(16,28,156) ::: "memory");
    else if constexpr (MODE == 1)
      asm volatile(MF(0,0,64) MF(16,4,64) MF(0,8,64) MF(16,12,64) MF(0,16,64) MF(16,20,64) MF(0,24,64) MF(16,28,64)
                   MF(0,0,64) MF(16,4,64) MF(0,8,64) MF(16,12,64) MF(0,16,64) MF(16,20,64) MF(0,24,64) MF(16,28,64)
                   MF(0,0,64) MF(16,4,64) MF(0,8,64) MF(16,12,64) MF(0,16,64) MF(16,20,64) MF(0,24,64) MF(16,28,64)
                   MF(0,0,64) MF(16,4,64) MF(0,8,64) MF(16,12,64) MF(0,16,64) MF(16,20,64) MF(0,24,64) MF(16,28,64)
                   MF(0,0,64) MF(16,4,64) MF(0,8,64) MF(16,12,64) MF(0,16,64) MF(16,20,64) MF(0,24,64) MF(16,28,64)
                   MF(0,0,64) MF(16,4,64) MF(0,8,64) MF(16,12,64) MF(0,16,64) MF(16,20,64) MF(0,24,64) MF(16,28,64) ::: "memory");
    else if constexpr (MODE == 2)
      asm volatile(MF(0,0,64) MF(16,0,64) MF(0,0,68) MF(16,0,68) MF(0,0,72) MF(16,0,72) MF(0,0,76) MF(16,0,76)
                   MF(0,0,80) MF(16,0,80) MF(0,0,84) MF(16,0,84) MF(0,0,88) MF(16,0,88) MF(0,0,92) MF(16,0,92)
                   MF(0,0,96) MF(16,0,96) MF(0,0,100) MF(16,0,100) MF(0,0,104) MF(16,0,104) MF(0,0,108) MF(16,0,108)
                   MF(0,0,112) MF(16,0,112) MF(0,0,116) MF(16,0,116) MF(0,0,120) MF(16,0,120) MF(0,0,124) MF(16,0,124)
                   MF(0,0,128) MF(16,0,128) MF(0,0,132) MF(16,0,132) MF(0,0,136) MF(16,0,136) MF(0,0,140) MF(16,0,140)
                   MF(0,0,144) MF(16,0,144) MF(0,0,148) MF(16,0,148) MF(0,0,152) MF(16,0,152) MF(0,0,156) MF(16,0,156) ::: "memory");
    else if constexpr (MODE == 3)
      asm volatile(MF(0,0,64) MF(16,0,64) MF(0,0,64) MF(16,0,64) MF(0,0,64) MF(16,0,64) MF(0,0,64) MF(16,0,64)
                   MF(0,0,64) MF(16,0,64) MF(0,0,64) MF(16,0,64) MF(0,0,64) MF(16,0,64) MF(0,0,64) MF(16,0,64)
                   MF(0,0,64) MF(16,0,64) MF(0,0,64) MF(16,0,64) MF(0,0,64) MF(16,0,64) MF(0,0,64) MF(16,0,64)
                   MF(0,0,64) MF(16,0,64) MF(0,0,64) MF(16,0,64) MF(0,0,64) MF(16,0,64) MF(0,0,64) MF(16,0,64)
                   MF(0,0,64) MF(16,0,64) MF(0,0,64) MF(16,0,64) MF(0,0,64) MF(16,0,64) MF(0,0,64) MF(16,0,64)
                   MF(0,0,64) MF(16,0,64) MF(0,0,64) MF(16,0,64) MF(0,0,64) MF(16,0,64) MF(0,0,64) MF(16,0,64) ::: "memory");
    else
      asm volatile(MF(0,20,58) MF(16,24,58) MF(0,28,62) MF(16,16,62) MF(0,12,66) MF(16,8,66) MF(0,4,70) MF(16,0,70)
                   MF(0,20,74) MF(16,24,74) MF(0,28,78) MF(16,16,78) MF(0,12,82) MF(16,8,82) MF(0,4,86) MF(16,0,86)
                   MF(0,20,90) MF(16,24,90) MF(0,28,94) MF(16,16,94) MF(0,12,98) MF(16,8,98) MF(0,4,102) MF(16,0,102)
                   MF(0,20,106) MF(16,24,106) MF(0,28,110) MF(16,16,110) MF(0,12,114) MF(16,8,114) MF(0,4,118) MF(16,0,118)
                   MF(0,20,122) MF(16,24,122) MF(0,28,126) MF(16,16,126) MF(0,12,130) MF(16,8,130) MF(0,4,134) MF(16,0,134)
                   MF(0,20,138) MF(16,24,138) MF(0,28,142) MF(16,16,142) MF(0,12,146) MF(16,8,146) MF(0,4,150) MF(16,0,150) ::: "memory");
  }
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  // the kernel declares the registers it names (v0..v159, a0..a31) through one clobbering statement
  asm volatile("" ::: "v0", "v31", "v64", "v159", "a0", "a31");
}
template <int MODE> void run(unsigned long long* cyc, const char* what) {
  const int iters = 200, blocks = 256;
  k<MODE><<<blocks, 256>>>(cyc, iters); k<MODE><<<blocks, 256>>>(cyc, iters);
  hipDeviceSynchronize();
  unsigned long long h[256];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double m = 0; for (int i = 0; i < blocks; ++i) m += h[i]; m /= blocks;
  printf("%-60s %.1f cycles per MFMA\n", what, m / (iters * 48.0));
}

// k2: the gap pattern of panel4's S1 -- MFMA (AGPR accumulators, chain), then FILL: 0 nothing, 1 four v_pk_fma_f16, 2 a ds_read_b128 that
// refills the A operand just used + lgkmcnt wait, 3 both; BIG: clobber v255 / a255 (512-register allocation)
#define FILLV "v_pk_fma_f16 v200, v200, v200, v200\n\tv_pk_fma_f16 v201, v201, v201, v201\n\tv_pk_fma_f16 v200, v200, v200, v200\n\tv_pk_fma_f16 v201, v201, v201, v201\n\t"
#define G0(acc, a, b) MF(acc, a, b)
#define G1(acc, a, b) MF(acc, a, b) FILLV
#define G2(acc, a, b) MF(acc, a, b) "ds_read_b128 v[" STR(a) ":" STR(a+3) "], v202\n\ts_waitcnt lgkmcnt(7)\n\t"
#define G3(acc, a, b) MF(acc, a, b) "ds_read_b128 v[" STR(a) ":" STR(a+3) "], v202\n\t" FILLV "s_waitcnt lgkmcnt(7)\n\t"
#define ROW(G) G(0,0,64) G(16,4,64) G(0,8,68) G(16,12,68) G(0,16,72) G(16,20,72) G(0,24,76) G(16,28,76)
template <int FILL, bool BIG>
__global__ __launch_bounds__(256) void k2(unsigned long long* cyc, int iters) {
  __shared__ char lds[4096];
  unsigned long long t0, t1;
  asm volatile("v_mov_b32 v202, %0" :: "v"((unsigned)(size_t)lds + (threadIdx.x & 63) * 16) : "v202");
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
  for (int it = 0; it < iters; ++it) {
    if constexpr (FILL == 0) asm volatile(ROW(G0) ROW(G0) ROW(G0) ROW(G0) ROW(G0) ROW(G0) ::: "memory");
    if constexpr (FILL == 1) asm volatile(ROW(G1) ROW(G1) ROW(G1) ROW(G1) ROW(G1) ROW(G1) ::: "memory");
    if constexpr (FILL == 2) asm volatile(ROW(G2) ROW(G2) ROW(G2) ROW(G2) ROW(G2) ROW(G2) ::: "memory");
    if constexpr (FILL == 3) asm volatile(ROW(G3) ROW(G3) ROW(G3) ROW(G3) ROW(G3) ROW(G3) ::: "memory");
  }
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  asm volatile("" ::: "v0", "v31", "v64", "v159", "v200", "v201", "v202", "a0", "a31");
  if constexpr (BIG) asm volatile("" ::: "v255", "a255");
}
template <int FILL, bool BIG> void run2(unsigned long long* cyc, const char* what) {
  const int iters = 200, blocks = 256;
  k2<FILL, BIG><<<blocks, 256>>>(cyc, iters); k2<FILL, BIG><<<blocks, 256>>>(cyc, iters);
  hipDeviceSynchronize();
  unsigned long long h[256];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double m = 0; for (int i = 0; i < blocks; ++i) m += h[i]; m /= blocks;
  printf("%-70s %.1f cycles per MFMA\n", what, m / (iters * 48.0));
}

// k3: as k2 FILL = 3 with the B operands high in the register file (v160..v255) and SGPR constants in the VALU fillers
#define FILLS "v_pk_fma_f16 v56, v56, s20, 1.0 op_sel_hi:[1,1,0] clamp\n\tv_pk_fma_f16 v57, v57, s20, 1.0 op_sel_hi:[1,1,0] clamp\n\tv_pk_fma_f16 v56, v56, v58, s21\n\tv_pk_fma_f16 v57, v57, v58, s21\n\t"
#define H3(acc, a, b) MF(acc, a, b) "ds_read_b128 v[" STR(a) ":" STR(a+3) "], v59\n\t" FILLS "s_waitcnt lgkmcnt(7)\n\t"
#define ROWH(G, b0) G(0,0,b0) G(16,4,b0) G(0,8,b0+4) G(16,12,b0+4) G(0,16,b0+8) G(16,20,b0+8) G(0,24,b0+12) G(16,28,b0+12)
__global__ __launch_bounds__(256) void k3(unsigned long long* cyc, int iters) {
  __shared__ char lds[4096];
  unsigned long long t0, t1;
  asm volatile("v_mov_b32 v59, %0\n\ts_mov_b32 s20, 0xb400b400\n\ts_mov_b32 s21, 0x3e8a3e8a" :: "v"((unsigned)(size_t)lds + (threadIdx.x & 63) * 16) : "v59", "s20", "s21");
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
  for (int it = 0; it < iters; ++it)
    asm volatile(ROWH(H3, 160) ROWH(H3, 176) ROWH(H3, 192) ROWH(H3, 208) ROWH(H3, 224) ROWH(H3, 240) ::: "memory");
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  asm volatile("" ::: "v0", "v31", "v56", "v57", "v58", "v59", "v160", "v255", "a0", "a31", "a255", "s20", "s21");
}

// k4: 48 MFMAs over 12 accumulator tiles round-robin (the fc2 pattern of panel4): ACCV = 0 accumulators in AGPRs a[32:223], 1 in VGPRs v[32:223];
// AOP = 0 A operand in VGPRs, 1 in AGPRs a[224:..]
#define MFX(ACC, acc, AR, a, b) "v_mfma_f32_32x32x16_f16 " ACC "[" STR(acc) ":" STR(acc+15) "], " AR "[" STR(a) ":" STR(a+3) "], v[" STR(b) ":" STR(b+3) "], " ACC "[" STR(acc) ":" STR(acc+15) "]\n\t"
#define GRP(ACC, AR, a0, b) MFX(ACC,32,AR,a0,b) MFX(ACC,48,AR,a0+4,b) MFX(ACC,64,AR,a0+8,b) MFX(ACC,80,AR,a0,b+4) MFX(ACC,96,AR,a0+4,b+4) MFX(ACC,112,AR,a0+8,b+4) \
                            MFX(ACC,128,AR,a0,b+8) MFX(ACC,144,AR,a0+4,b+8) MFX(ACC,160,AR,a0+8,b+8) MFX(ACC,176,AR,a0,b+12) MFX(ACC,192,AR,a0+4,b+12) MFX(ACC,208,AR,a0+8,b+12)
template <int ACCV, int AOP>
__global__ __launch_bounds__(256) void k4(unsigned long long* cyc, int iters) {
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
  for (int it = 0; it < iters; ++it) {
    if constexpr (ACCV == 0 && AOP == 0) asm volatile(GRP("a", "v", 0, 16) GRP("a", "v", 0, 16) GRP("a", "v", 0, 16) GRP("a", "v", 0, 16) ::: "memory");
    if constexpr (ACCV == 0 && AOP == 1) asm volatile(GRP("a", "a", 224, 16) GRP("a", "a", 236, 16) GRP("a", "a", 224, 16) GRP("a", "a", 236, 16) ::: "memory");
    if constexpr (ACCV == 1 && AOP == 0) asm volatile(GRP("v", "v", 0, 16) GRP("v", "v", 0, 16) GRP("v", "v", 0, 16) GRP("v", "v", 0, 16) ::: "memory");
    if constexpr (ACCV == 1 && AOP == 1) asm volatile(GRP("v", "a", 224, 16) GRP("v", "a", 236, 16) GRP("v", "a", 224, 16) GRP("v", "a", 236, 16) ::: "memory");
  }
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  asm volatile("" ::: "v0", "v31", "v255", "a0", "a255");
}
template <int ACCV, int AOP> void run4(unsigned long long* cyc, const char* what) {
  k4<ACCV, AOP><<<256, 256>>>(cyc, 200); k4<ACCV, AOP><<<256, 256>>>(cyc, 200); hipDeviceSynchronize();
  unsigned long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double m = 0; for (int i = 0; i < 256; ++i) m += h[i]; m /= 256;
  printf("%-70s %.1f cycles per MFMA\n", what, m / (200 * 48.0));
}

// k5: the two patterns of a panel4 tick back to back in one loop: 48 MFMAs on two alternating accumulators a[0:31] (fc1), then 48 over twelve
// accumulators a[32:223] (fc2); WITHV: 32 v_accvgpr_read of a[0:31] + 16 cvt between them (the hand-off's pack)
template <int WITHV>
__global__ __launch_bounds__(256) void k5(unsigned long long* cyc, int iters) {
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
  for (int it = 0; it < iters; ++it) {
    asm volatile(ROW(G0) ROW(G0) ROW(G0) ROW(G0) ROW(G0) ROW(G0) ::: "memory");
    if constexpr (WITHV) asm volatile("s_nop 15\n\ts_nop 7\n\tv_accvgpr_read_b32 v200, a0\n\tv_accvgpr_read_b32 v201, a1\n\tv_accvgpr_read_b32 v200, a2\n\tv_accvgpr_read_b32 v201, a3\n\tv_accvgpr_read_b32 v200, a16\n\tv_accvgpr_read_b32 v201, a17" ::: "memory");
    asm volatile(GRP("a", "v", 0, 16) GRP("a", "v", 0, 16) GRP("a", "v", 0, 16) GRP("a", "v", 0, 16) ::: "memory");
  }
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  asm volatile("" ::: "v0", "v31", "v200", "v201", "v255", "a0", "a255");
}
template <int WITHV> void run5(unsigned long long* cyc, const char* what) {
  k5<WITHV><<<256, 256>>>(cyc, 200); k5<WITHV><<<256, 256>>>(cyc, 200); hipDeviceSynchronize();
  unsigned long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double m = 0; for (int i = 0; i < 256; ++i) m += h[i]; m /= 256;
  printf("%-70s %.1f cycles per MFMA\n", what, m / (200 * 96.0));
}

// k6: k2<3> + the fc2 group per iteration, on RANDOM register and LDS contents (finite halves): does the cycle count depend on the data?
#define SETV(n) "v_mov_b32 v" STR(n) ", %0\n\tv_mul_lo_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %2\n\tv_and_b32 %0, %0, %3\n\t"
template <int RANDOM>
__global__ __launch_bounds__(256) void k6(unsigned long long* cyc, int iters) {
  __shared__ unsigned lds[2048];
  unsigned long long t0, t1;
  for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = RANDOM ? (((i + 1) * 2654435761u) & 0xbbffbbffu) : 0u;  // halves with exponent <= 14: |x| < 2
  __syncthreads();
  asm volatile("v_mov_b32 v202, %0" :: "v"((unsigned)(size_t)lds + (threadIdx.x & 63) * 16) : "v202");
  // every operand register from the (random or zero) LDS contents
#define LD(n, off) "ds_read_b128 v[" STR(n) ":" STR(n+3) "], v202 offset:" STR(off) "\n\t"
#define LD4(n, off) LD(n, off) LD(n+4, off+1024) LD(n+8, off+2048) LD(n+12, off+3072)
  asm volatile(LD4(0, 0) LD4(16, 16) LD4(64, 32) LD4(80, 48) LD4(96, 64) LD4(112, 80) LD4(128, 96) LD4(144, 112) "ds_read_b64 v[200:201], v202 offset:128\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
  for (int it = 0; it < iters; ++it) {
    asm volatile(ROW(G3) ROW(G3) ROW(G3) ROW(G3) ROW(G3) ROW(G3) ::: "memory");
    asm volatile(GRP("a", "v", 0, 64) GRP("a", "v", 12, 80) GRP("a", "v", 0, 96) GRP("a", "v", 12, 112) ::: "memory");
  }
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  asm volatile("" ::: "v0", "v31", "v64", "v159", "v200", "v201", "v202", "v255", "a0", "a255");
}
template <int RANDOM> void run6(unsigned long long* cyc, const char* what) {
  for (int r = 0; r < 30; ++r) k6<RANDOM><<<256, 256>>>(cyc, 400);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); for (int r = 0; r < 50; ++r) k6<RANDOM><<<256, 256>>>(cyc, 400); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double m = 0; for (int i = 0; i < 256; ++i) m += h[i]; m /= 256;
  printf("%-70s %.1f cycles per MFMA; %.1f us per launch -> %.2f GHz\n", what, m / (400 * 96.0), ms * 1e3 / 50, m / (ms * 1e3 / 50) / 1e3);
}

// k7: the same work as k6 on v_mfma_f32_16x16x32_f16: twice the MFMA instructions (4 accumulator registers each), the same fragment reads and
// vector fillers per FLOP -- which shape does the chip run faster BY WALL on random operands (cdna_hip_programming.md rule 28)?
#define MF16(acc, a, b) "v_mfma_f32_16x16x32_f16 a[" STR(acc) ":" STR(acc+3) "], v[" STR(a) ":" STR(a+3) "], v[" STR(b) ":" STR(b+3) "], a[" STR(acc) ":" STR(acc+3) "]\n\t"
// one "32x32x16 gap" = two 16x16x32 MFMAs + the gap's ds_read + 4 VALU + wait
#define H16(acc, a, b) MF16(acc, a, b) "ds_read_b128 v[" STR(a) ":" STR(a+3) "], v202\n\t" MF16(acc+4, a, b+4) FILLV "s_waitcnt lgkmcnt(7)\n\t"
#define ROW16(b0) H16(0,0,b0) H16(8,4,b0) H16(16,8,b0+8) H16(24,12,b0+8) H16(0,16,b0+16) H16(8,20,b0+16) H16(16,24,b0+24) H16(24,28,b0+24)
#define G16(acc, a, b) MF16(acc, a, b) MF16(acc+4, a, b+4)
#define GRP16(a0, b) G16(32,a0,b) G16(40,a0+4,b) G16(48,a0+8,b) G16(56,a0,b+8) G16(64,a0+4,b+8) G16(72,a0+8,b+8) G16(80,a0,b+16) G16(88,a0+4,b+16) G16(96,a0+8,b+16) G16(104,a0,b+24) G16(112,a0+4,b+24) G16(120,a0+8,b+24)
template <int RANDOM>
__global__ __launch_bounds__(256) void k7(unsigned long long* cyc, int iters) {
  __shared__ unsigned lds[2048];
  unsigned long long t0, t1;
  for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = RANDOM ? (((i + 1) * 2654435761u) & 0xbbffbbffu) : 0u;
  __syncthreads();
  asm volatile("v_mov_b32 v202, %0" :: "v"((unsigned)(size_t)lds + (threadIdx.x & 63) * 16) : "v202");
  asm volatile(LD4(0, 0) LD4(16, 16) LD4(64, 32) LD4(80, 48) LD4(96, 64) LD4(112, 80) LD4(128, 96) LD4(144, 112) "ds_read_b64 v[200:201], v202 offset:128\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
  for (int it = 0; it < iters; ++it) {
    asm volatile(ROW16(64) ROW16(96) ROW16(128) ROW16(64) ROW16(96) ROW16(128) ::: "memory");
    asm volatile(GRP16(0, 64) GRP16(12, 96) GRP16(0, 128) GRP16(12, 64) ::: "memory");
  }
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  asm volatile("" ::: "v0", "v31", "v64", "v159", "v200", "v201", "v202", "v255", "a0", "a255");
}
template <int RANDOM> void run7(unsigned long long* cyc, const char* what) {
  for (int r = 0; r < 30; ++r) k7<RANDOM><<<256, 256>>>(cyc, 400);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); for (int r = 0; r < 50; ++r) k7<RANDOM><<<256, 256>>>(cyc, 400); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double m = 0; for (int i = 0; i < 256; ++i) m += h[i]; m /= 256;
  printf("%-70s %.1f cycles per 32x32x16-equivalent; %.1f us per launch -> %.2f GHz\n", what, m / (400 * 96.0), ms * 1e3 / 50, m / (ms * 1e3 / 50) / 1e3);
}

// k8: k6 with the A operand's mantissas cut short (the table's first KiB, which the loop's fragment reads come from, keeps only KEEP mantissa bits
// per half; the B registers are loaded from beyond it and stay full random): does the power the MFMA draws depend on how many mantissa bits of ONE
// operand toggle -- i.e. would 16-bit weights rounded to fewer mantissa bits (bf16's 7 of fp16's 10) buy clock at the power cap?
template <int KEEP>
__global__ __launch_bounds__(256) void k8(unsigned long long* cyc, int iters) {
  __shared__ unsigned lds[2048];
  unsigned long long t0, t1;
  const unsigned cut = 0xffffu & ~((1u << (10 - KEEP)) - 1u), m2 = cut | (cut << 16);
  for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = (((i + 1) * 2654435761u) & 0xbbffbbffu) & (i < 260 ? m2 : 0xffffffffu);
  __syncthreads();
  asm volatile("v_mov_b32 v202, %0" :: "v"((unsigned)(size_t)lds + (threadIdx.x & 63) * 16) : "v202");
  asm volatile(LD4(0, 0) LD4(16, 0) LD4(64, 1056) LD4(80, 1072) LD4(96, 1088) LD4(112, 1104) LD4(128, 1120) LD4(144, 1136) "ds_read_b64 v[200:201], v202 offset:128\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
  asm volatile(LD(4, 0) LD(8, 0) LD(12, 0) LD(20, 0) LD(24, 0) LD(28, 0) "s_waitcnt lgkmcnt(0)" ::: "memory");  // every A register from the cut region
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
  for (int it = 0; it < iters; ++it) {
    asm volatile(ROW(G3) ROW(G3) ROW(G3) ROW(G3) ROW(G3) ROW(G3) ::: "memory");
    asm volatile(GRP("a", "v", 0, 64) GRP("a", "v", 12, 80) GRP("a", "v", 0, 96) GRP("a", "v", 12, 112) ::: "memory");
  }
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  asm volatile("" ::: "v0", "v31", "v64", "v159", "v200", "v201", "v202", "v255", "a0", "a255");
}
template <int KEEP> void run8(unsigned long long* cyc, const char* what) {
  for (int r = 0; r < 30; ++r) k8<KEEP><<<256, 256>>>(cyc, 400);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); for (int r = 0; r < 50; ++r) k8<KEEP><<<256, 256>>>(cyc, 400); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double m = 0; for (int i = 0; i < 256; ++i) m += h[i]; m /= 256;
  printf("%-70s %.1f cycles per MFMA; %.1f us per launch -> %.2f GHz\n", what, m / (400 * 96.0), ms * 1e3 / 50, m / (ms * 1e3 / 50) / 1e3);
}
int main() {
  unsigned long long* cyc; hipMalloc(&cyc, 256 * 8);
  run<0>(cyc, "A cycles 8 quads, B cycles 24 quads (v64..), 2 accumulators");
  run<1>(cyc, "A cycles 8 quads, B fixed");
  run<2>(cyc, "A fixed, B cycles 24 quads");
  run<3>(cyc, "A fixed, B fixed");
  run<4>(cyc, "the compiler's placement (A v0..31 shuffled, B v58..153)");
  run2<0, false>(cyc, "gap: MFMA only");
  run2<1, false>(cyc, "gap: MFMA + 4 v_pk_fma_f16");
  run2<2, false>(cyc, "gap: MFMA + ds_read_b128 into the A operand just used + lgkmcnt(7)");
  run2<3, false>(cyc, "gap: MFMA + ds_read + 4 v_pk_fma_f16 + lgkmcnt(7)");
  run2<0, true>(cyc, "512 registers: MFMA only");
  run2<1, true>(cyc, "512 registers: MFMA + 4 v_pk_fma_f16");
  run2<3, true>(cyc, "512 registers: MFMA + ds_read + 4 v_pk_fma_f16 + lgkmcnt(7)");
  {
    k3<<<256, 256>>>(cyc, 200); k3<<<256, 256>>>(cyc, 200); hipDeviceSynchronize();
    unsigned long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0; for (int i = 0; i < 256; ++i) m += h[i]; m /= 256;
    printf("%-70s %.1f cycles per MFMA\n", "512 registers, B in v160..v255, SGPR constants in the fillers", m / (200 * 48.0));
  }
  run4<0, 0>(cyc, "12 accumulators in AGPRs a[32:223], A in VGPRs");
  run4<0, 1>(cyc, "12 accumulators in AGPRs a[32:223], A in AGPRs");
  run4<1, 0>(cyc, "12 accumulators in VGPRs v[32:223], A in VGPRs");
  run4<1, 1>(cyc, "12 accumulators in VGPRs v[32:223], A in AGPRs");
  run5<0>(cyc, "48 on two accumulators + 48 over twelve, per iteration");
  run5<1>(cyc, "the same with accumulator reads between the two groups");
  run6<0>(cyc, "fc1 gaps (MFMA + ds_read + 4 VALU) + fc2 group, zero data");
  run6<1>(cyc, "fc1 gaps (MFMA + ds_read + 4 VALU) + fc2 group, random data");
  run8<10>(cyc, "32x32x16, A operand with 10 mantissa bits (full), B full random");
  run8<7>(cyc, "32x32x16, A operand with 7 mantissa bits (bf16's), B full random");
  run8<4>(cyc, "32x32x16, A operand with 4 mantissa bits, B full random");
  run8<0>(cyc, "32x32x16, A operand powers of two only, B full random");
  run8<10>(cyc, "32x32x16 again, A full");
  run8<7>(cyc, "32x32x16 again, A with 7 mantissa bits");
  run7<0>(cyc, "16x16x32: fc1 gaps + fc2 group, zero data");
  run7<1>(cyc, "16x16x32: fc1 gaps + fc2 group, random data");
  run6<1>(cyc, "32x32x16 again: fc1 gaps + fc2 group, random data");
  run7<1>(cyc, "16x16x32 again: fc1 gaps + fc2 group, random data");
  return 0;
}
