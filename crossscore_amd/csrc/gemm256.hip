// Large-tile MFMA GEMM for the K >= 512 linears of the CrossScore hot path on gfx950 (the ViT-B encoder's QKV / out-proj / fc1 / fc2
// projections and the decoder's K/V projection; HF modeling_dinov2.py:211-213,250,293-297; torch functional.py:5785-5852).
//
//   out[m][n] = epilogue( bias[n] + sum_k A[m][k] * W[n][k] )     A:[M][K] 16-bit activations, W:[N][K] 16-bit (nn.Linear layout)
//
// Why a second GEMM kernel: gemm.hip's 128 x 192 x 32 tile stages 1 byte from L2 per 77 FLOP, which at K >= 768 puts the kernel on
// the L2 -> LDS staging path (measured 0.55-0.70 x hipBLASLt on the ViT-B shapes).  This kernel stages 1 byte per
// 128 FLOP: a 256 x 256 x 64 tile per workgroup of 8 waves (2 along M x 4 along N, wave tile 128 x 64 = 8 x 4 accumulators of
// v_mfma_f32_16x16x32), one workgroup per CU.
//
// Schedule (a K tile = 4 phases; a phase = [load segment | s_barrier | 16 MFMAs | s_barrier]):
//   * the two waves of a SIMD (wave w and w + 4: the two M halves) run ONE barrier apart, so while one is in its 16-MFMA cluster
//     the other issues its LDS reads and LDS-DMA pieces (ping-pong on the matrix pipe);
//   * operands are staged by global_load_lds_dwordx4 as four 16-KiB half-tiles per K tile (A rows of the waves' first / second 64 rows,
//     W rows of the waves' first / second 32 columns), each phase issues one half-tile (2 instructions per wave);
//     phase 1 reads A half 0 + W half 0 (quadrant 00), phase 2 W half 1 (01), phase 3 A half 1 (11), phase 4 nothing (10), so a
//     half-tile's LDS slot is free one phase after its read and is refilled two phases after it (the refill of K tile t + 2 goes
//     A0 @ phase 3, W0 @ 4 of tile t, W1 @ phase 1, A1 @ 2 of tile t + 1);
//   * ONE counted s_waitcnt vmcnt(4) per K tile (phase 4: everything of the next K tile has landed, two half-tiles stay in flight),
//     never 0 inside the loop; the barrier that follows orders it for every wave's reads one phase later;
//   * LDS image: rows of 128 B (64 k), 16-byte chunk c of row r stored at chunk c ^ ((r >> 1) & 7): applied on the DMA's per-lane
//     SOURCE address and on the ds_read_b128 address, conflict free for the 16x16x32 operand read.
// The MFMA takes the W fragment as its first operand, so a lane owns 4 consecutive output columns of one row; the epilogue goes
// through a wave-private LDS patch and stores whole 128-byte lines.
#include "cs_common.h"
#include <type_traits>
#include <utility>

namespace {

constexpr int G_BM = 256, G_BN = 256, G_BK = 64;
constexpr int G_ROWB = G_BK * 2;            // 128 bytes per staged row
constexpr int G_OPND = 256 * G_ROWB;        // 32 KiB: one operand's K tile
constexpr int G_BUF = 2 * G_OPND;           // 64 KiB: A | W
constexpr int G_RING = 2 * G_BUF;           // 128 KiB
constexpr int G_PATCH_H = 16 * 144;         // fp16 patch: 16 rows x (128 + 16) B
constexpr int G_PATCH_F = 16 * 272;         // fp32 patch: 16 rows x (256 + 16) B
constexpr int G_BIAS = G_RING;              // 1 KiB behind the ring: the tile's 256 bias values (fp32)
constexpr int G_LDS = G_RING + 1024;        // the epilogue patches live in the ring (idle once the K loop is done)

typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

template <int V> using IC = std::integral_constant<int, V>;

template <bool BF>
__device__ __forceinline__ f32x4_t mfma16(const h16x8_t& w, const h16x8_t& a, const f32x4_t& c) {
  if constexpr (BF) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, w), __builtin_bit_cast(bf16x8_t, a), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_f16(w, a, c, 0, 0, 0);
}
template <bool BF>
__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
  if constexpr (BF) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    const bf16x2_t v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(uint32_t, v);
  } else {
    return pack_h16x2(lo, hi);
  }
}

#define G_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#define G_SB() __builtin_amdgcn_sched_barrier(0)

template <int EPI, bool BF>
__global__ __launch_bounds__(512, 2) void cs_gemm256_kernel(CsGemmParams p) {
  constexpr bool kHalf = EPI <= CS_EPI_BIAS_LEAKY_F16;
  static_assert(kHalf || EPI == CS_EPI_RESID_F32, "gemm256: bias->16-bit and residual fp32 epilogues only");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wv >> 2, wn = wv & 3;

  // ---- tile of this workgroup, XCD aware: blocks b, b + 8, .. share an XCD (round-robin dispatch; speed only); XCD x owns the A row
  //      panels tm == x (mod 8) and walks them n-fastest, so the blocks resident on one L2 share A panels ----
  const int tiles_n = p.N / G_BN;
  const int tiles_m = (p.M + G_BM - 1) / G_BM;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int pl = slot / tiles_n;
  const int tm = pl * 8 + xcd;
  if (tm >= tiles_m) return;  // whole workgroup, before any barrier
  const int m0 = tm * G_BM, n0 = (slot - pl * tiles_n) * G_BN;
  const int T = p.K / G_BK;   // K tiles (even: cs_gemm256_supported)

  // ---- LDS-DMA maps.  One instruction = 8 rows x 128 B; lane i writes LDS chunk (i & 7) of row (i >> 3) and fetches source chunk
  //      (i & 7) ^ ((row >> 1) & 7).  Half-tile h of A = rows {wm' * 128 + h * 64 + 0..63}; of W = rows {wn' * 64 + h * 32 + 0..31};
  //      wave wv issues pieces 2 wv, 2 wv + 1 of each half-tile. ----
  const int srow = lane >> 3;
  unsigned offA[2][2], offW[2][2];   // [half][piece] global byte offsets
  int ldsA[2][2], ldsW[2][2];        // wave-uniform LDS byte offsets inside a buffer
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int ra0 = (wv >> 2) * 128 + h * 64 + ((16 * wv) & 63) + 8 * e;   // first row of the piece (wave-uniform)
      const int ra = ra0 + srow;
      const int gca = (lane & 7) ^ ((ra >> 1) & 7);
      offA[h][e] = ((unsigned)min(m0 + ra, p.M - 1) * (unsigned)p.lda + (unsigned)gca * 8u) * 2u;
      ldsA[h][e] = ra0 * G_ROWB;
      const int rw0 = (wv >> 1) * 64 + h * 32 + ((16 * wv) & 31) + 8 * e;
      const int rw = rw0 + srow;
      const int gcw = (lane & 7) ^ ((rw >> 1) & 7);
      offW[h][e] = ((unsigned)(n0 + rw) * (unsigned)p.ldw + (unsigned)gcw * 8u) * 2u;
      ldsW[h][e] = G_OPND + rw0 * G_ROWB;
    }
  // stage half-tile (which: 0 = A half 0, 1 = W half 0, 2 = W half 1, 3 = A half 1) of K tile kt into buffer (kt & 1)
  auto stage = [&](auto WHICH_, int kt) {
    constexpr int WHICH = decltype(WHICH_)::value;
    constexpr bool isA = WHICH == 0 || WHICH == 3;
    constexpr int h = (WHICH == 0 || WHICH == 1) ? 0 : 1;
    char* base = smem + (kt & 1) * G_BUF;
    const char* src = reinterpret_cast<const char*>(isA ? p.A : p.W) + (size_t)kt * G_ROWB;
#pragma unroll
    for (int e = 0; e < 2; ++e)
      __builtin_amdgcn_global_load_lds(CS_GLOBAL_PTR(src + (isA ? offA[h][e] : offW[h][e])), CS_LDS_PTR(base + (isA ? ldsA[h][e] : ldsW[h][e])), 16, 0, 0);
  };

  // ---- fragment read addressing: lane (fr = lane & 15, cq = lane >> 4) reads row fr, chunk (4 s + cq) ^ ((fr >> 1) & 7) ----
  const int fr = lane & 15;
  const int c0 = ((lane >> 4) ^ ((fr >> 1) & 7)) * 16;
  const char* rdA = smem + wm * (128 * G_ROWB) + fr * G_ROWB;           // + buffer + (mh * 64 + 16 i) * 128 + (c0 | c0 ^ 64)
  const char* rdW = smem + G_OPND + wn * (64 * G_ROWB) + fr * G_ROWB;   // + buffer + (nh * 32 + 16 j) * 128 + ..

  f32x4_t acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  h16x8_t fa[4][2], fw[2][2][2];  // A fragments of the current M half [i][s]; W fragments of both N halves [nh][j][s]
  auto ld_a = [&](auto B_, auto MH_) {
    constexpr int B = decltype(B_)::value, MH = decltype(MH_)::value;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const char* r = rdA + B * G_BUF + (MH * 64 + 16 * i) * G_ROWB;
      fa[i][0] = *reinterpret_cast<const h16x8_t*>(r + c0);
      fa[i][1] = *reinterpret_cast<const h16x8_t*>(r + (c0 ^ 64));
    }
  };
  auto ld_w = [&](auto B_, auto NH_) {
    constexpr int B = decltype(B_)::value, NH = decltype(NH_)::value;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const char* r = rdW + B * G_BUF + (NH * 32 + 16 * j) * G_ROWB;
      fw[NH][j][0] = *reinterpret_cast<const h16x8_t*>(r + c0);
      fw[NH][j][1] = *reinterpret_cast<const h16x8_t*>(r + (c0 ^ 64));
    }
  };
  auto mma = [&](auto MH_, auto NH_) {  // one quadrant: 4 x 2 accumulators x 2 k-steps = 16 MFMAs
    constexpr int MH = decltype(MH_)::value, NH = decltype(NH_)::value;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[MH * 4 + i][NH * 2 + j] = mfma16<BF>(fw[NH][j][s], fa[i][s], acc[MH * 4 + i][NH * 2 + j]);
    __builtin_amdgcn_s_setprio(0);
  };
  // the barrier between a phase's load segment and its MFMA cluster, and the one behind the cluster
  auto bar_then_wait = [&]() {
    G_SB();
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    G_SB();
  };
  auto bar = [&]() {
    G_SB();
    __builtin_amdgcn_s_barrier();
    G_SB();
  };

  // ---- prologue: K tile 0 completely, A half 0 and W half 0 of K tile 1 (its other halves are issued by phases 1 and 2 of tile 0) ----
  stage(IC<0>{}, 0); stage(IC<1>{}, 0); stage(IC<2>{}, 0); stage(IC<3>{}, 0);
  stage(IC<0>{}, 1); stage(IC<1>{}, 1);
  // the tile's bias values -> LDS (read back in the epilogue: 16 registers less across the K loop).  Wave 0 only; its load is younger
  // than its LDS-DMA pieces, so the compiler's wait in front of the LDS write retires those too -- harmless, this is the prologue
  if (wv == 0) {
    f32x4_t b4 = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) b4 = *reinterpret_cast<const f32x4_t*>(p.bias + n0 + lane * 4);
    *reinterpret_cast<f32x4_t*>(smem + G_BIAS + lane * 16) = b4;
  }
  G_VMCNT(4);
  bar();
  if (wm == 1) bar();  // the second M half runs one barrier behind the first

  // ---- K loop, two K tiles (buffers 0 and 1) per iteration ----
  auto ktile = [&](auto B_, int t) {
    constexpr int B = decltype(B_)::value;
    const bool more1 = t + 1 < T, more2 = t + 2 < T;
    // phase 1: quadrant (0, 0)
    ld_w(IC<B>{}, IC<0>{}); ld_a(IC<B>{}, IC<0>{});
    if (more1) stage(IC<2>{}, t + 1);
    bar_then_wait(); mma(IC<0>{}, IC<0>{}); bar();
    // phase 2: quadrant (0, 1)
    ld_w(IC<B>{}, IC<1>{});
    if (more1) stage(IC<3>{}, t + 1);
    bar_then_wait(); mma(IC<0>{}, IC<1>{}); bar();
    // phase 3: quadrant (1, 1)
    ld_a(IC<B>{}, IC<1>{});
    if (more2) stage(IC<0>{}, t + 2);
    bar_then_wait(); mma(IC<1>{}, IC<1>{}); bar();
    // phase 4: quadrant (1, 0); the next K tile has landed behind this phase's first barrier
    if (more2) { stage(IC<1>{}, t + 2); G_VMCNT(4); }
    else G_VMCNT(0);
    bar_then_wait(); mma(IC<1>{}, IC<0>{}); bar();
  };
  for (int t = 0; t < T; t += 2) {
    ktile(IC<0>{}, t);
    ktile(IC<1>{}, t + 1);
  }
  if (wm == 0) bar();  // the first half's matching barrier

  // ---- epilogue.  Wave-private LDS patch (in the ring: every read of it is retired and no LDS-DMA is outstanding): a 16-row
  //      sub-tile goes in in the accumulator layout and comes out as row segments, 16 B per lane, whole 128-byte lines to memory ----
  char* patch = smem + wv * (2 * G_PATCH_F);
  f32x4_t bia[4];  // this lane's columns: 16 j + 4 (lane >> 4) .. + 3 of the wave's 64
#pragma unroll
  for (int j = 0; j < 4; ++j) bia[j] = *reinterpret_cast<const f32x4_t*>(smem + G_BIAS + (wn * 64 + 16 * j + 4 * (lane >> 4)) * 4);
  const int row_w = m0 + wm * 128;        // first row of the wave tile
  const int col_w = n0 + wn * 64;         // first column
  if constexpr (kHalf) {
    const int wr_off = fr * 144 + (lane >> 4) * 8;       // + 32 j
    const int rrow = lane >> 3, rch = lane & 7;          // read: 8 rows x 8 chunks per instruction
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      char* pp = patch + (i & 1) * G_PATCH_H;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v[4] = {acc[i][j][0] + bia[j][0], acc[i][j][1] + bia[j][1], acc[i][j][2] + bia[j][2], acc[i][j][3] + bia[j][3]};
        if constexpr (EPI == CS_EPI_BIAS_GELU_F16) gelu_erf4(v);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if constexpr (EPI == CS_EPI_BIAS_RELU_F16) v[r] = fmaxf(v[r], 0.f);
          if constexpr (EPI == CS_EPI_BIAS_LEAKY_F16) v[r] = v[r] >= 0.f ? v[r] : 0.01f * v[r];
        }
        *reinterpret_cast<u32x2_t*>(pp + wr_off + 32 * j) = u32x2_t{pack2<BF>(v[0], v[1]), pack2<BF>(v[2], v[3])};
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int rr = q * 8 + rrow;
        const f32x4_t seg = *reinterpret_cast<const f32x4_t*>(pp + rr * 144 + rch * 16);
        const int m = row_w + 16 * i + rr;
        if (m < p.M) *reinterpret_cast<f32x4_t*>(reinterpret_cast<h16_t*>(p.out) + (size_t)m * p.ldc + col_w + rch * 8) = seg;
      }
    }
  } else {
    const int wr_off = fr * 272 + (lane >> 4) * 16;      // + 64 j
    const int rrow = lane >> 4, rch = lane & 15;         // read: 4 rows x 16 chunks per instruction
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      char* pp = patch + (i & 1) * G_PATCH_F;
      f32x4_t res[4];
      if (p.resid) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int m = min(row_w + 16 * i + q * 4 + rrow, p.M - 1);
          res[q] = *reinterpret_cast<const f32x4_t*>(p.resid + (size_t)m * p.ldr + col_w + rch * 4);
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4_t*>(pp + wr_off + 64 * j) = acc[i][j] + bia[j];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int rr = q * 4 + rrow;
        f32x4_t seg = *reinterpret_cast<const f32x4_t*>(pp + rr * 272 + rch * 16);
        if (p.resid) seg += res[q];
        const int m = row_w + 16 * i + rr;
        if (m < p.M) *reinterpret_cast<f32x4_t*>(reinterpret_cast<float*>(p.out) + (size_t)m * p.ldc + col_w + rch * 4) = seg;
      }
    }
  }
}

int g_enabled = 1;

template <int EPI, bool BF>
hipError_t launch256(const CsGemmParams& p, hipStream_t st) {
  static bool attr_done[16] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return hipErrorInvalidDevice;
  if (!attr_done[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(cs_gemm256_kernel<EPI, BF>), hipFuncAttributeMaxDynamicSharedMemorySize, G_LDS);
    if (e != hipSuccess) return e;
    attr_done[dev] = true;
  }
  const int tiles_n = p.N / G_BN, tiles_m = (p.M + G_BM - 1) / G_BM;
  const int grid = ((tiles_m + 7) / 8) * tiles_n * 8;
  hipLaunchKernelGGL((cs_gemm256_kernel<EPI, BF>), dim3(grid), dim3(512), G_LDS, st, p);
  return hipGetLastError();
}

}  // namespace

extern "C" {

// tools / tests: 0 routes every GEMM to gemm.hip's kernel again
void cs_gemm256_enable(int on) { g_enabled = on; }

// Shapes this kernel takes (everything else stays with gemm.hip): whole 256-column tiles, an even number of 64-deep K tiles and
// K >= 512 (below that a tile's prologue and epilogue outweigh what the larger tile saves), the plain epilogues.
int cs_gemm256_supported(const CsGemmParams* p, int epi) {
  if (!g_enabled) return 0;
  if (epi > CS_EPI_RESID_F32) return 0;
  if (p->N % G_BN || p->K % (2 * G_BK) || p->K < 512 || p->M < G_BM) return 0;
  if (p->lda % 8 || p->ldw % 8 || p->ldc % 8) return 0;
  if ((long long)p->M * p->lda * 2 >= (1ll << 32) || (long long)p->N * p->ldw * 2 >= (1ll << 32)) return 0;
  if (epi == CS_EPI_RESID_F32 && p->resid && p->ldr % 4) return 0;
  if (p->pos || p->pmean || p->out_f16 || p->stats_out || p->ln_part) return 0;
  return 1;
}

hipError_t cs_gemm256_launch(const CsGemmParams* p, int epi, int bf16, hipStream_t st) {
#define G_CASE(E) case E: return bf16 ? launch256<E, true>(*p, st) : launch256<E, false>(*p, st);
  switch (epi) {
    G_CASE(CS_EPI_BIAS_F16) G_CASE(CS_EPI_BIAS_GELU_F16) G_CASE(CS_EPI_BIAS_RELU_F16) G_CASE(CS_EPI_BIAS_LEAKY_F16) G_CASE(CS_EPI_RESID_F32)
  }
#undef G_CASE
  return hipErrorInvalidValue;
}

}  // extern "C"
