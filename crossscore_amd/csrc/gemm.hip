// fp16 MFMA GEMM with fused epilogues for the CrossScore hot path (gfx950).
//
//   out[m][n] = epilogue( bias[n] + sum_k A[m][k] * W[n][k] )     A:[M][K] fp16 (activations), W:[N][K] fp16 (nn.Linear layout)
//
// Replaces the eager nn.Linear / Conv2d-patchify op groups K2,K4,K6,K7,K10,K12,K13,K15-K18 of SURVEY.md 2a
// (HF modeling_dinov2.py:148,211-213,250,293-297; torch functional.py:5785-5852; cross_reference.py:45-50).
//
// Structure: 128 x {192,128} x 32 tiles, 4 waves side by side in N, wave tile 128 x {48,32} of v_mfma_f32_16x16x32_f16.
// * Both operands are K-contiguous, so both fragments are 16-byte rows: staged HBM->LDS by global_load_lds_dwordx4 (no
//   VGPR round trip) into a lane-linear image whose 16-byte chunks are XOR-swizzled on the SOURCE address and on the
//   ds_read_b128 address (conflict free), 3-slot ring, counted s_waitcnt vmcnt + raw s_barrier, one barrier per K slice.
// * Operands are swapped in the MFMA (W is the "A" operand) so each lane owns 4 consecutive output columns.
// * Persistent blocks (two per CU) walk tiles in an XCD-aware order: the blocks that share an A row panel share one L2.
// * DEFERRED EPILOGUE.  The kernel is bound by the L2->LDS staging stream (measured: stream alone 60 us, MFMA loop alone
//   52 us, epilogue alone 32-80 us, and they added up because a block in its epilogue issues no LDS-DMA).  K is only
//   384-1536 here, so the epilogue is a third of a tile's time.  A finished tile's accumulators therefore move to a second
//   register set and are written out in 8 steps (one 16-row sub-tile each) interleaved with the first 8 K slices of the
//   NEXT tile: the staging stream never pauses and the epilogue's VALU/LDS/store work sits between MFMA groups.
// * Each step transposes its sub-tile through a wave-private LDS patch (inline-asm ds ops, so the compiler does not see an
//   alias with the LDS-DMA ring and drain it) and stores whole 128-byte lines; residual / position addends are read the
//   same way.  The bias is the accumulator's initial value, LayerScale is folded into the packed weights at finalize.
#include "cs_common.h"
#include <atomic>
#include <stdlib.h>
#include <type_traits>
#include <utility>

namespace {

constexpr int BK = 32;
constexpr int BM = 128;
constexpr int A_BYTES = BM * BK * 2;  // 8 KiB per slice

typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));

template <int NSUB> struct GemmCfg {
  static constexpr int BN = 64 * NSUB;
  static constexpr int STAGE_BYTES = (BM + BN) * BK * 2;  // 20 / 16 KiB
  static constexpr int NS = 3;                            // ring slots; deeper rings measured slower (tools/gemm_ns.py)
  static constexpr int D = NS - 1;                        // K slices kept in flight
  static constexpr int LPS = 2 + NSUB;                    // LDS-DMA instructions per wave per slice (2 A + NSUB W)
  static constexpr int RING = NS * STAGE_BYTES;
  static constexpr int WN = 16 * NSUB;                    // wave tile columns
  static constexpr int PROW_F = WN * 4 + 16;              // patch row (fp32), padded: conflict-free b128 writes
  static constexpr int PROW_H = WN * 2 + 16;              // patch row (fp16)
  static constexpr int PATCH = 16 * PROW_F + (NSUB == 2 ? 512 : 0);  // one 16-row patch per wave (+ room for the LN stash)
  static constexpr int BIAS_MAX = 1536;                   // the bias vector lives in LDS when N <= BIAS_MAX (every N of the path)
  static constexpr int BIAS_OFF = RING + 4 * PATCH;
  static constexpr int LDS = BIAS_OFF + BIAS_MAX * 4;     // 79.3 / 65.2 KiB -> two blocks per CU
};

#define CS_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

#ifdef CS_ABLATE  // timing-only debug builds (tools/gemm_ablate.py): bit0 no epilogue, bit1 no MFMA, bit2 no LDS-DMA
#define CS_ABL(bit) (p.ablate & (bit))
// per-wave phase clocks (issue time, s_memtime): 0 vmcnt wait, 1 barrier, 2 prefetch + LDS-DMA issue, 3 LDS reads + MFMA,
// 4 epilogue step, 5 everything between slices (tile switch, hand-over, loop control), 6 slices, 7 total
__device__ long long* g_cs_dbg = nullptr;
#define CS_T(k) do { const long long t_ = clock64(); dbg_acc[k] += t_ - dbg_last; dbg_last = t_; } while (0)
#else
#define CS_ABL(bit) 0
#define CS_T(k) do { } while (0)
#endif

// RegressionLayer activation (regression_layer.py:26-62); kept out of line so the unrolled head epilogue does not spill
__device__ __noinline__ float head_activation(float v, int act, float powp) {
  float y = act == 0 ? 1.0f / (1.0f + __expf(-v)) : tanhf(v);
  if (powp != 1.0f) y = powf(y, powp);
  return y;
}

// ---- per-image mean of the score map inside the head launch (CsGemmParams::mean_*; score_summariser.py:180-192: score_map.mean(dim=[-1, -2])) ----
// A patch row's P*P pixels are spread over the 4 lanes (lane >> 4) x the waves x the column tiles that hold its columns.  Each (wave, column tile)
// leaves ONE partial per row in mean_part[row][sp]: its four lanes' sums added pairwise (l ^ 16, then l ^ 32: the same bits in all four lanes).
// Nothing depends on where the row sits in a tile or the image in the batch, so an item's mean has the same bits alone and inside any batch.
__device__ __forceinline__ void head_row_partial(float* mean_part, int m, float ps, int sp, int slot, int lane) {
  ps += __shfl_xor(ps, 16, 64);
  ps += __shfl_xor(ps, 32, 64);
  if (lane < 16 && m >= 0) mean_part[(size_t)m * sp + slot] = ps;
}
// After a wave's eighth epilogue step of a tile: its partials of rows [r0, r1) are written.  Per image those rows belong to, the wave adds the
// row count to the image's counter; the Np * sp-th row-slot to arrive makes its wave the image's finisher: lane l sums the partials of patch rows
// l, l + 64, .. (a row's slots in ascending order), the lanes are added by a fixed xor tree, mean = sum / (Np * N).  Agent-scope fences on both
// sides (the partials of other workgroups come through other XCDs' L2s).  The finisher zeroes the counter for the next launch.
// (scalars, not the parameter block: a reference to it would put a copy of the whole block into a private segment)
__device__ __noinline__ void head_mean_arrive(const float* mean_part, unsigned* mean_cnt, float* mean_out, int Np, int N, int r0, int r1, int sp, int lane) {
  struct { const float* mean_part; unsigned* mean_cnt; float* mean_out; int Np, N; } p{mean_part, mean_cnt, mean_out, Np, N};
  __threadfence();
  for (int b = r0 / p.Np; b * p.Np < r1; ++b) {
    const int lo = max(r0, b * p.Np), hi = min(r1, (b + 1) * p.Np);
    unsigned old = 0;
    if (lane == 0) old = atomicAdd(p.mean_cnt + b, (unsigned)(hi - lo));
    old = __builtin_amdgcn_readfirstlane(old);
    if (old + (unsigned)(hi - lo) != (unsigned)p.Np * (unsigned)sp) continue;
    __threadfence();
    float a = 0.f;
    for (int r = lane; r < p.Np; r += 64) {
      const float* q = p.mean_part + ((size_t)b * p.Np + r) * sp;
      float t = q[0];
      for (int k = 1; k < sp; ++k) t += q[k];
      a += t;
    }
    for (int o = 32; o >= 1; o >>= 1) a += __shfl_xor(a, o, 64);
    if (lane == 0) {
      p.mean_out[b] = a / ((float)p.Np * (float)p.N);
      __hip_atomic_store(p.mean_cnt + b, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// ---- wave-private LDS patch access by inline asm (the compiler must not treat it as aliasing the LDS-DMA ring) ----
// (constant byte offsets go into the instruction's offset field: one address register per access group)
template <int OFF = 0>
__device__ __forceinline__ void patch_write16(unsigned addr, f32x4_t v) {
  asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}
template <int OFF = 0>
__device__ __forceinline__ void patch_write8(unsigned addr, u32x2_t v) {
  asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}
template <int N, int STRIDE>
__device__ __forceinline__ void patch_read16(unsigned addr, f32x4_t (&r)[N]) {
  static_assert(N == 1 || N == 2 || N == 4, "patch_read16");
  if constexpr (N == 1) {
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r[0]) : "v"(addr) : "memory");
  } else if constexpr (N == 2) {
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:%3\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(r[0]), "=&v"(r[1]) : "v"(addr), "n"(STRIDE) : "memory");
  } else {
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:%5\n\tds_read_b128 %2, %4 offset:%6\n\tds_read_b128 %3, %4 offset:%7\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]) : "v"(addr), "n"(STRIDE), "n"(2 * STRIDE), "n"(3 * STRIDE) : "memory");
  }
}

// LN-folded consumer step: (mu, rstd) of the lane's row (8 bytes) from the wave's stash
__device__ __forceinline__ u32x2_t patch_read8(unsigned addr) {
  u32x2_t r;
  asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r) : "v"(addr) : "memory");
  return r;
}

// Residual prefetch (PIPE kernels): 16 bytes per lane at sbase + voff, issued by inline asm so that the compiler neither
// counts it nor waits for it; the kernel's own counted s_waitcnt covers it (see k_slice), and rb_touch() orders the uses.
__device__ __forceinline__ void gload16(f32x4_t& r, unsigned voff, const float* sbase) {
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r) : "v"(voff), "s"(sbase) : "memory");
}
__device__ __forceinline__ void rb_touch(f32x4_t& r) { asm volatile("" : "+v"(r)); }

// (sum, sum of squares) of one lane's 4 values, reduced over the LPR lanes that hold a row.  One helper with explicit fma order
// for the full-tile and the ragged path: with -ffp-contract the compiler would otherwise fuse the two copies differently, and
// a row's LayerNorm statistics (hence every later fp16 rounding) would depend on which tile the row fell into.
// The total lands in the LAST lane of the row (lane % LPR == LPR-1).  LPR == 16 is one DPP row: four v_add with row_shr
// (no LDS crossbar traffic, which ds_bpermute-based shuffles would put into every epilogue step).
template <int SHR>
__device__ __forceinline__ float dpp_row_shr_add(float v) {
  // row_shr:SHR = 0x110 + SHR; bound_ctrl: lanes shifted in from outside the row read 0
  const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x110 + SHR, 0xf, 0xf, true);
  return v + __int_as_float(moved);
}
template <int LPR>
__device__ __forceinline__ void row_partial(f32x4_t v, bool on, float& a, float& b) {
  a = on ? (v[0] + v[1]) + (v[2] + v[3]) : 0.f;
  b = on ? fmaf(v[0], v[0], fmaf(v[1], v[1], fmaf(v[2], v[2], v[3] * v[3]))) : 0.f;
  if constexpr (LPR == 16) {
    a = dpp_row_shr_add<1>(a); b = dpp_row_shr_add<1>(b);
    a = dpp_row_shr_add<2>(a); b = dpp_row_shr_add<2>(b);
    a = dpp_row_shr_add<4>(a); b = dpp_row_shr_add<4>(b);
    a = dpp_row_shr_add<8>(a); b = dpp_row_shr_add<8>(b);
  } else {  // 8 lanes per row (128-wide tiles): xor butterfly, then hand the total to the row's last lane like the DPP form
#pragma unroll
    for (int o = 1; o < LPR; o <<= 1) {
      a += __shfl_xor(a, o, 64);
      b += __shfl_xor(b, o, 64);
    }
  }
}

template <int... Js, class F>
__device__ __forceinline__ void static_for(std::integer_sequence<int, Js...>, F&& f) {
  (f(std::integral_constant<int, Js>{}), ...);
}

// Patch embedding on mean-centred patches: im2col removed each patch's per-channel mean before the fp16 rounding (a smooth image
// patch is mostly its mean, and a fp16 rounding error of the weights times that mean was the largest single error term of the
// encoder on natural images); the exact contribution mean_ch * sum_taps(W[n][ch]) comes back here in fp32.
__device__ __forceinline__ f32x4_t patch_dc(const CsGemmParams& p, int m, int n) {
  const f32x4_t mu = *reinterpret_cast<const f32x4_t*>(p.pmean + (size_t)m * 4);
  f32x4_t r = mu[0] * *reinterpret_cast<const f32x4_t*>(p.wsum + n);
  r += mu[1] * *reinterpret_cast<const f32x4_t*>(p.wsum + (size_t)p.ldc + n);
  r += mu[2] * *reinterpret_cast<const f32x4_t*>(p.wsum + 2 * (size_t)p.ldc + n);
  return r;
}

template <int EPI> struct EpiTraits {
  static constexpr bool kLN = EPI == CS_EPI_LN_F16 || EPI == CS_EPI_LN_GELU_F16;           // LayerNorm-folded consumer
  static constexpr bool kHalf = EPI == CS_EPI_BIAS_F16 || EPI == CS_EPI_BIAS_GELU_F16 || EPI == CS_EPI_BIAS_RELU_F16 ||
                                EPI == CS_EPI_BIAS_LEAKY_F16 || kLN;
  static constexpr bool kResid = EPI == CS_EPI_RESID_F32 || EPI == CS_EPI_RESID_F32_LN;
  static constexpr bool kLnOut = EPI == CS_EPI_RESID_F32_LN || EPI == CS_EPI_PATCH_F32;      // may emit fp16 copy + row partials
};

// NSUB = 16-column sub-tiles per wave; block tile = 128 x (64*NSUB); 4 waves side by side in N, wave tile 128 x 16*NSUB.
// SPL = float4 loads per row of LayerNorm partial sums (LN-folded consumers only: ln_sp / 2).
// PIPE (RESID_F32 with a residual, K >= 9 slices): the residual is prefetched in the accumulator layout a slice or more ahead of
// its epilogue step and added in registers, so no step consumes a load it has just issued (that would wait for every older
// LDS-DMA: the vector-memory queue retires in order, and the K=384/1536 residual GEMMs were spending a third of their time there).
template <int EPI, int NSUB, int SPL, bool PIPE, bool BF>
__global__ __launch_bounds__(256, 2) void cs_gemm_kernel(CsGemmParams p) {
  static_assert(!PIPE || EPI == CS_EPI_RESID_F32, "PIPE is the RESID_F32 residual prefetch");
  using Cfg = GemmCfg<NSUB>;
  constexpr bool kLN = EpiTraits<EPI>::kLN, kResid = EpiTraits<EPI>::kResid, kLnOut = EpiTraits<EPI>::kLnOut;
  constexpr int BN = Cfg::BN, NS = Cfg::NS, D = Cfg::D, LPS = Cfg::LPS, STAGE_BYTES = Cfg::STAGE_BYTES;
  static_assert(D == 2, "the vmcnt bookkeeping below tracks exactly two iterations of epilogue traffic");
  constexpr bool kHalf = EpiTraits<EPI>::kHalf;
  // row-segment geometry of the epilogue steps (16 bytes per lane)
  constexpr int CPR = kHalf ? Cfg::WN * 2 / 16 : Cfg::WN * 4 / 16;         // 16-byte chunks per patch row: 6/4 (fp16), 12/8 (fp32)
  constexpr int LPR = kHalf ? (NSUB == 2 ? 4 : 8) : (NSUB == 2 ? 8 : 16);  // lanes assigned per row
  constexpr int RPI = 64 / LPR;                                            // rows per ds_read / global access
  constexpr int NRD = 16 / RPI;                                            // accesses per 16-row step: 2/1 (fp16), 4/2 (fp32)
  constexpr int PROW = kHalf ? Cfg::PROW_H : Cfg::PROW_F;
  // vector-memory instructions one full-tile epilogue step issues: stores (+ the row partials of an LN-folded consumer);
  // stores + as many addend loads for RESID / PATCH (+ fp16 copy + row partials when those are emitted)
  constexpr int VE = kHalf ? NRD : (kLnOut ? 4 * NRD : (PIPE ? NRD : 2 * NRD));
  // LN-folded consumer: behind the fp16 patch rows each wave keeps s[n], c[n] of the tile's columns and (mu, rstd) of the
  // tile's 128 rows, filled once per tile, so the epilogue steps issue no vector-memory loads (a load consumed inside a
  // step would wait for every older LDS-DMA: the queue retires in order)
  constexpr int STASH = 16 * Cfg::PROW_H;
  constexpr int MRST = STASH + 2 * Cfg::WN * 4;
  static_assert(!kLN || MRST + 128 * 8 <= Cfg::PATCH, "stash does not fit the patch");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_n = (p.N + BN - 1) / BN;
  const int tiles_m = (p.M + BM - 1) / BM;
  const int wn = wv;
  const int nk = p.K / BK;
  if constexpr (PIPE) __builtin_assume(nk >= 9);

  // ---- persistent tile walk, XCD aware: blocks b, b+8, .. share an XCD (round-robin dispatch); XCD x owns the A row
  //      panels tm == x (mod 8) and its blocks walk that list n-fastest, so concurrent blocks of one L2 share A. ----
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
  const int panels_x = (tiles_m - xcd + 7) / 8;  // row panels owned by this XCD
  const int ntile_x = panels_x * tiles_n;
  const int my_tiles = slot < ntile_x ? (ntile_x - slot + slots - 1) / slots : 0;
  const int total = my_tiles * nk;  // K slices this block consumes
  auto tile_of = [&](int idx, int& m0, int& n0) {
    const int pl = idx / tiles_n;
    m0 = (pl * 8 + xcd) * BM;
    n0 = (idx - pl * tiles_n) * BN;
  };

  // ---- LDS-DMA staging: one wave-instruction = 16 rows x 64 B.  Lane i writes LDS chunk (i&3) of row (i>>2); it
  //      fetches source chunk (i&3) ^ 2*((row>>2)&1), and ds_read_b128 applies the same XOR: conflict free. ----
  const int srow = lane >> 2;
  const int gchunk = (lane & 3) ^ (((srow >> 2) & 1) << 1);
  unsigned offA[2], offW[NSUB];  // byte offsets (the shape check keeps M*lda and N*ldw below 2^31 elements)
  auto set_tile = [&](int m0, int n0) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int ga = m0 + (j * 4 + wv) * 16 + srow;  // A tile: 8 groups of 16 rows, 2 per wave
      ga = ga < p.M ? ga : p.M - 1;
      offA[j] = (unsigned)(ga * p.lda + gchunk * 8) * 2u;
#ifdef CS_ABLATE
      if (CS_ABL(8)) {  // timing experiment: 8 lanes fetch one whole 128-byte line (the data is meaningless)
        ga = min(m0 + (j * 4 + wv) * 8 + (lane >> 3), p.M - 1);
        offA[j] = (unsigned)(ga * p.lda + (lane & 7) * 8) * 2u;
      }
#endif
    }
#pragma unroll
    for (int j = 0; j < NSUB; ++j) {
      int gw = n0 + (j * 4 + wv) * 16 + srow;  // W tile: 4*NSUB groups of 16 rows, NSUB per wave
      gw = gw < p.N ? gw : p.N - 1;
      offW[j] = (unsigned)(gw * p.ldw + gchunk * 8) * 2u;
#ifdef CS_ABLATE
      if (CS_ABL(8)) {
        gw = min(n0 + (j * 4 + wv) * 8 + (lane >> 3), p.N - 1);
        offW[j] = (unsigned)(gw * p.ldw + (lane & 7) * 8) * 2u;
      }
#endif
    }
  };
  auto stage = [&](int ring, int kt) {
    char* base = smem + ring * STAGE_BYTES;
#ifdef CS_ABLATE
    if (CS_ABL(8)) kt &= ~1;  // 64-element aligned: stays inside the row
#endif
    // uniform base (this slice's K offset) + 32-bit lane offset: the scalar-base addressing form, no per-instruction VALU
    const char* sA = reinterpret_cast<const char*>(p.A + kt * BK);
    const char* sW = reinterpret_cast<const char*>(p.W + kt * BK);
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_global_load_lds(CS_GLOBAL_PTR(sA + offA[j]), CS_LDS_PTR(base + (j * 4 + wv) * 1024), 16, 0, 0);
#pragma unroll
    for (int j = 0; j < NSUB; ++j)
      __builtin_amdgcn_global_load_lds(CS_GLOBAL_PTR(sW + offW[j]), CS_LDS_PTR(base + A_BYTES + (j * 4 + wv) * 1024), 16, 0, 0);
  };

  // ---- bias vector -> LDS once per block (read back by inline-asm ds_read at each tile start: a global load there would be
  //      consumed at once and drain the LDS-DMA queue once per tile) ----
  const unsigned bias_lds = (unsigned)(size_t)CS_LDS_PTR(smem + Cfg::BIAS_OFF);
  // (a block that owns a single tile reads its 48 bias values per wave straight from global memory: the drain that costs a
  //  multi-tile block a queue of LDS-DMA falls on the empty prologue there, and the 6 KB fill + barrier would be pure overhead)
  const bool bias_in_lds = p.bias && !kLN && p.N <= Cfg::BIAS_MAX && my_tiles > 1;
  if (bias_in_lds) {
    for (int i = tid * 4; i < p.N; i += 1024) patch_write16(bias_lds + i * 4, *reinterpret_cast<const f32x4_t*>(p.bias + i));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }

  // load cursor (runs D slices ahead of the compute cursor, across tile boundaries)
  int issued = 0, l_idx = slot, l_kt = 0, l_ring = 0;
  if (total > 0) {
    int m0, n0;
    tile_of(l_idx, m0, n0);
    set_tile(m0, n0);
  }
  auto issue_one = [&]() {
    stage(l_ring, l_kt);
    ++issued;
    l_ring = l_ring + 1 == NS ? 0 : l_ring + 1;
    if (++l_kt == nk) {
      l_kt = 0;
      l_idx += slots;
      if (l_idx < ntile_x) {
        int m0, n0;
        tile_of(l_idx, m0, n0);
        set_tile(m0, n0);
      }
    }
  };
  for (int s0 = 0; s0 < D; ++s0) {
    if (issued < total && !CS_ABL(4)) issue_one();
    else if (issued < total) ++issued;
  }

  const int frow = lane & 15;
  const int coff = ((lane >> 4) ^ (((frow >> 2) & 1) << 1)) * 16;
  const unsigned patch_lds = (unsigned)(size_t)CS_LDS_PTR(smem + Cfg::RING + wv * Cfg::PATCH);
  // epilogue-step addressing: patch write in the accumulator layout (row = lane&15, 4 columns at (lane>>4)*4 of sub-tile j),
  // patch read / global access as row segments (row = lane / LPR, 16-byte chunk = lane % LPR)
  const unsigned pw_addr = patch_lds + (lane & 15) * PROW + (lane >> 4) * (kHalf ? 8 : 16);
  const int rrow = lane / LPR, rch = lane % LPR;
  const unsigned pr_addr = patch_lds + rrow * PROW + (rch < CPR ? rch : CPR - 1) * 16;
  const unsigned stash_addr = patch_lds + STASH + (lane >> 4) * 16;  // + j*64 (s), + WN*4 + j*64 (c)

#ifdef CS_ABLATE
  long long dbg_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long dbg_last = clock64();
  const long long dbg_t0 = dbg_last;
#endif
  f32x4_t acc[8][NSUB], prv[8][NSUB];
  bool have_prev = false, prev_full = false;
  int pm0 = 0, pn0 = 0, cm0 = 0, cn0 = 0;
  int it = 0, c_ring = 0;
  // stores the epilogue steps of the last / second-to-last slice issued (-1: unknown count), and whether the last slice began
  // with residual prefetches (PIPE): the inputs of the counted wait at the top of k_slice
  int st1 = 0, st2 = 0;
  bool rl1 = false, last_dma = false;
  // PIPE: residual prefetch registers of epilogue step t (accumulator layout); live from the slice that loads them to the next
  // slice's start, so they reuse registers of `prv` sub-tiles whose step has already run
  [[maybe_unused]] f32x4_t rb[8][NSUB];
  [[maybe_unused]] const unsigned lane_roff = PIPE ? (unsigned)(((lane & 15) * p.ldr + (lane >> 4) * 4) * 4) : 0u;
  [[maybe_unused]] auto rload = [&](int m0, int n0, auto T_) {
    constexpr int T = decltype(T_)::value;
    const int r0 = min(m0 + T * 16, p.M - 16);  // clamps only on ragged tiles, whose prefetch is not used
#pragma unroll
    for (int j = 0; j < NSUB; ++j) {
      const int c0 = min(n0 + wn * Cfg::WN + j * 16, p.N - 16);
      gload16(rb[T][j], lane_roff, p.resid + (size_t)r0 * p.ldr + c0);
    }
  };
  [[maybe_unused]] auto radd = [&](auto T_) {
    constexpr int T = decltype(T_)::value;
#pragma unroll
    for (int j = 0; j < NSUB; ++j) rb_touch(rb[T][j]);
    if (prev_full) {
#pragma unroll
      for (int j = 0; j < NSUB; ++j) prv[T][j] += rb[T][j];
    }
  };

  // ---- one epilogue step: 16-row sub-tile I of the previous tile; returns the number of vector-memory instructions it
  //      issued when that is known at compile time (full tiles), else -1 ----
  auto epi_step = [&](auto I_, auto FLUSH_) -> int {
    constexpr int I = decltype(I_)::value;
    // PIPE: the residual of a full tile is already in prv, except for steps 3..7 of a block's last tile (flushed after the loop)
    constexpr bool kLoadResid = kResid && (!PIPE || (decltype(FLUSH_)::value && I >= 3));
    if (CS_ABL(1)) {
      if (prv[I][0][0] == 12345.678f) reinterpret_cast<float*>(p.out)[0] = prv[I][NSUB - 1][3];  // keep prv live
      return -1;
    }
    const int ncol0 = pn0 + wn * Cfg::WN;
    if constexpr (EPI == CS_EPI_HEAD_SCORE) {
      const int m = pm0 + I * 16 + (lane & 15);
      if (m < p.M) {
        const int b = m / p.Np;
        const int pp = m - b * p.Np;
        const int pi = pp / p.gw, pj = pp - pi * p.gw;
        const int gh = p.Np / p.gw;
        const int Ws = p.P * p.gw;
        float* dst = reinterpret_cast<float*>(p.out) + ((size_t)b * gh * p.P + (size_t)pi * p.P) * Ws + pj * p.P;
        float ps = 0.f;  // this lane's part of the patch's pixel sum, columns in ascending order
#pragma unroll
        for (int j = 0; j < NSUB; ++j) {
          const int n = ncol0 + j * 16 + (lane >> 4) * 4;
          if (n >= p.N) continue;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int nn = n + r;
            const int py = nn / p.P, px = nn - py * p.P;
            const float y = head_activation(prv[I][j][r], p.act, p.powp);
            dst[(size_t)py * Ws + px] = y;
            ps += y;
          }
        }
        if (p.mean_part) head_row_partial(p.mean_part, m, ps, 4 * tiles_n, (pn0 / BN) * 4 + wn, lane);
      } else if (p.mean_part) {
        head_row_partial(p.mean_part, -1, 0.f, 4 * tiles_n, 0, lane);  // (the cross-lane exchange is wave-wide)
      }
      if constexpr (I == 7) {
        if (p.mean_part) head_mean_arrive(p.mean_part, p.mean_cnt, p.mean_out, p.Np, p.N, pm0, min(pm0 + BM, p.M), 4 * tiles_n, lane);
      }
      return -1;
    } else {
      // 1) activation in the accumulator layout, then into the patch
      float ln_mu = 0.f, ln_rs = 1.f;
      if constexpr (kLN) {  // LayerNorm statistics of this lane's row (stashed at the tile switch)
        const u32x2_t mr = patch_read8(patch_lds + MRST + (I * 16 + (lane & 15)) * 8);
        ln_mu = __uint_as_float(mr[0]);
        ln_rs = __uint_as_float(mr[1]);
      }
      static_for(std::make_integer_sequence<int, NSUB>{}, [&](auto J_) {
        constexpr int j = decltype(J_)::value;
        float v[4] = {prv[I][j][0], prv[I][j][1], prv[I][j][2], prv[I][j][3]};
        if constexpr (kLN) {
          f32x4_t sc[2];
          patch_read16<2, Cfg::WN * 4>(stash_addr + j * 64, sc);  // s then c of these 4 columns
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaf(fmaf(-ln_mu, sc[0][r], v[r]), ln_rs, sc[1][r]);
        }
        if constexpr (EPI == CS_EPI_BIAS_GELU_F16 || EPI == CS_EPI_LN_GELU_F16) {
          gelu_erf4(v);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if constexpr (EPI == CS_EPI_BIAS_RELU_F16) v[r] = fmaxf(v[r], 0.f);
          if constexpr (EPI == CS_EPI_BIAS_LEAKY_F16) v[r] = v[r] >= 0.f ? v[r] : 0.01f * v[r];
        }
        if constexpr (kHalf) {
          patch_write8<j * 32>(pw_addr, u32x2_t{pack_o16x2<BF>(v[0], v[1]), pack_o16x2<BF>(v[2], v[3])});
        } else {
          patch_write16<j * 64>(pw_addr, f32x4_t{v[0], v[1], v[2], v[3]});
        }
      });
      // 2) whole row segments out of the patch
      f32x4_t seg[NRD];
      patch_read16<NRD, RPI * PROW>(pr_addr, seg);
      const int mrow0 = pm0 + I * 16 + rrow;
      const int n = ncol0 + rch * (kHalf ? 8 : 4);
      [[maybe_unused]] const int st_slot = (pn0 / BN) * 4 + wn;
      [[maybe_unused]] const bool ln_out = kLnOut && p.out_f16 && p.stats_out;
      if (prev_full && (!kResid || p.resid) && (!kLnOut || EPI == CS_EPI_PATCH_F32 || ln_out) &&
          (EPI != CS_EPI_PATCH_F32 || ln_out)) {
        // every row and column of the tile exists: no row tests, a fixed number (VE) of memory instructions
        if (rch < CPR) {
          if constexpr (kLoadResid) {
#pragma unroll
            for (int q = 0; q < NRD; ++q)
              seg[q] += *reinterpret_cast<const f32x4_t*>(p.resid + (size_t)(mrow0 + q * RPI) * p.ldr + n);
          } else if constexpr (EPI == CS_EPI_PATCH_F32) {
#pragma unroll
            for (int q = 0; q < NRD; ++q) {
              const int m = mrow0 + q * RPI;
              seg[q] += *reinterpret_cast<const f32x4_t*>(p.pos + (size_t)(m % p.Np + 1) * p.ldc + n);
              if (p.pmean) seg[q] += patch_dc(p, m, n);
            }
          }
#pragma unroll
          for (int q = 0; q < NRD; ++q) {
            const int m = mrow0 + q * RPI;
            if constexpr (kHalf) {
              *reinterpret_cast<f32x4_t*>(reinterpret_cast<h16_t*>(p.out) + (size_t)m * p.ldc + n) = seg[q];
            } else {
              // CS_EPI_PATCH_F32: token row m of image img lands at row m + img + 1 (CLS rows interleaved)
              const size_t row = EPI == CS_EPI_PATCH_F32 ? (size_t)(m + m / p.Np + 1) : (size_t)m;
              *reinterpret_cast<f32x4_t*>(reinterpret_cast<float*>(p.out) + row * p.ldc + n) = seg[q];
              if constexpr (kLnOut)
                *reinterpret_cast<u32x2_t*>(p.out_f16 + row * p.ldc + n) =
                    u32x2_t{pack_o16x2<BF>(seg[q][0], seg[q][1]), pack_o16x2<BF>(seg[q][2], seg[q][3])};
            }
          }
        }
        if constexpr (kLnOut) {  // per-row partial (sum, sumsq) of this wave's columns; reduction over the LPR lanes of a row
#pragma unroll
          for (int q = 0; q < NRD; ++q) {
            float a, b;
            row_partial<LPR>(seg[q], rch < CPR, a, b);
            const int m = mrow0 + q * RPI;
            const size_t row = EPI == CS_EPI_PATCH_F32 ? (size_t)(m + m / p.Np + 1) : (size_t)m;
            if (rch == LPR - 1) *reinterpret_cast<float2*>(p.stats_out + (row * p.stats_sp + st_slot) * 2) = make_float2(a, b);
          }
        }
        return kLoadResid != (kResid && !PIPE) ? -1 : VE;
      }
      // ragged tile (or no residual / LN outputs): per-row tests, data-dependent instruction count
      const bool col_ok = rch < CPR && n < p.N;
#pragma unroll
      for (int q = 0; q < NRD; ++q) {
        const int m = mrow0 + q * RPI;
        const bool ok = col_ok && m < p.M;
        [[maybe_unused]] size_t row = (size_t)m;
        if (ok) {
          if constexpr (kHalf) {
            *reinterpret_cast<f32x4_t*>(reinterpret_cast<h16_t*>(p.out) + (size_t)m * p.ldc + n) = seg[q];
          } else {
            if constexpr (kResid) {
              if (p.resid) seg[q] += *reinterpret_cast<const f32x4_t*>(p.resid + (size_t)m * p.ldr + n);
            } else {
              seg[q] += *reinterpret_cast<const f32x4_t*>(p.pos + (size_t)(m % p.Np + 1) * p.ldc + n);
              if (p.pmean) seg[q] += patch_dc(p, m, n);
              row = (size_t)(m + m / p.Np + 1);
            }
            *reinterpret_cast<f32x4_t*>(reinterpret_cast<float*>(p.out) + row * p.ldc + n) = seg[q];
            if constexpr (kLnOut)
              if (p.out_f16)
                *reinterpret_cast<u32x2_t*>(p.out_f16 + row * p.ldc + n) =
                    u32x2_t{pack_o16x2<BF>(seg[q][0], seg[q][1]), pack_o16x2<BF>(seg[q][2], seg[q][3])};
          }
        }
        if constexpr (kLnOut) {
          float a, b;
          row_partial<LPR>(seg[q], ok, a, b);
          if (EPI == CS_EPI_PATCH_F32 && m < p.M) row = (size_t)(m + m / p.Np + 1);
          if (p.stats_out && rch == LPR - 1 && m < p.M) *reinterpret_cast<float2*>(p.stats_out + (row * p.stats_sp + st_slot) * 2) = make_float2(a, b);
        }
      }
      return -1;
    }
  };

  // ---- one K slice of the current tile (+ optionally one epilogue step of the previous tile) ----
  auto k_slice = [&](auto STEP_) {
    constexpr int STEP = decltype(STEP_)::value;
    // Slice `it` must have landed.  The vector-memory queue retires in order; younger than slice `it`'s LDS-DMA are the D-1
    // later slices and whatever the epilogue steps of the last D iterations issued, so exactly that many may stay in flight.
    // Order inside a slice: [residual prefetches (PIPE)] [LDS-DMA of slice it+D] [stores of the epilogue step].  If the last
    // slice began with prefetches they must have landed too: they are older than its LDS-DMA, so only that and its stores stay.
    CS_T(5);
    const int ahead = min(total - 1 - it, D - 1);
    const int young = (st1 < 0 || (!rl1 && st2 < 0)) ? -1 : (rl1 ? st1 : st1 + st2);
    if (ahead < D - 1 || young < 0) CS_VMCNT(0);
    else if (young == 0) CS_VMCNT(LPS);
    else if (young == VE) CS_VMCNT(LPS + VE);
    else if (young == 2 * VE) CS_VMCNT(LPS + 2 * VE);
    else CS_VMCNT(0);
    CS_T(0);
    __builtin_amdgcn_s_barrier();  // every wave's part of slice `it` landed; everyone has left ring slot (it-1)%NS
    asm volatile("" ::: "memory");
    CS_T(1);
    bool rl = false;
    if constexpr (PIPE) {
      // step t's residual: loaded in slice t-2, added at the top of slice t-1 (steps 0..2: loaded in the tile's last slice,
      // added at the hand-over), so that no prefetch register is live while all of prv still is
      if constexpr (STEP >= 2 && STEP <= 6) {
        if (have_prev) radd(std::integral_constant<int, (STEP >= 2 && STEP <= 6) ? STEP + 1 : 0>{});
      }
      if constexpr (STEP >= 1 && STEP <= 5) {
        rload(pm0, pn0, std::integral_constant<int, (STEP >= 1 && STEP <= 5) ? STEP + 2 : 0>{});
        rl = true;
      }
      if constexpr (STEP == -2) {
        rload(cm0, cn0, std::integral_constant<int, 0>{});
        rload(cm0, cn0, std::integral_constant<int, 1>{});
        rload(cm0, cn0, std::integral_constant<int, 2>{});
        rl = true;
      }
    }
    last_dma = issued < total;
    if (issued < total && !CS_ABL(4)) issue_one();  // refill the slot just vacated
    else if (issued < total) ++issued;
    CS_T(2);
    if (!CS_ABL(2)) {
      const char* sa = smem + c_ring * STAGE_BYTES + frow * 64 + coff;
      const char* sw = smem + c_ring * STAGE_BYTES + A_BYTES + (wn * 16 * NSUB + frow) * 64 + coff;
      h16x8_t fw[NSUB], fa[8];
#pragma unroll
      for (int j = 0; j < NSUB; ++j) fw[j] = *reinterpret_cast<const h16x8_t*>(sw + j * 16 * 64);
#pragma unroll
      for (int i = 0; i < 8; ++i) fa[i] = *reinterpret_cast<const h16x8_t*>(sa + i * 16 * 64);
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < NSUB; ++j) acc[i][j] = mfma_16x16x32<BF>(fw[j], fa[i], acc[i][j]);
      // W fragments + LOOK A fragments up front, then one A-fragment read per MFMA group, LOOK-1 groups ahead of its use
      // (the LN-folded consumers are at the 256-register limit: one fragment less in flight avoids spills)
      constexpr int LOOK = ((kLN || PIPE) && NSUB == 3) ? 2 : 3;
      __builtin_amdgcn_sched_group_barrier(0x100, NSUB + LOOK, 0);
#pragma unroll
      for (int i = 0; i < 8 - LOOK; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, NSUB, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, LOOK * NSUB, 0);
    }
    CS_T(3);
    int ve = 0;
    if constexpr (STEP >= 0) {
      if (have_prev) ve = epi_step(std::integral_constant<int, (STEP >= 0 ? STEP : 0)>{}, std::false_type{});
    }
    CS_T(4);
    st2 = st1;
    st1 = __builtin_amdgcn_readfirstlane(ve);
    rl1 = rl;
    ++it;
    c_ring = c_ring + 1 == NS ? 0 : c_ring + 1;
  };

  for (int idx = slot; idx < ntile_x; idx += slots) {
    tile_of(idx, cm0, cn0);
    // accumulators start at the bias of their columns
#pragma unroll
    for (int j = 0; j < NSUB; ++j) {
      const int n = min(cn0 + wn * Cfg::WN + j * 16 + (lane >> 4) * 4, p.N - 4);
      f32x4_t b4[1] = {f32x4_t{0.f, 0.f, 0.f, 0.f}};
      if (bias_in_lds) patch_read16<1, 0>(bias_lds + n * 4, b4);
      else if (p.bias && !kLN) b4[0] = *reinterpret_cast<const f32x4_t*>(p.bias + n);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i][j] = b4[0];
    }
    if (p.bias && !kLN && !bias_in_lds) st1 = -1;  // global bias loads are consumed at once: the next wait drains the queue
    int kt = 0;
    // the first 8 slices carry the previous tile's 8 epilogue steps
#define CS_SLICE_WITH_STEP(E)                                                              \
    if (kt < nk) { k_slice(std::integral_constant<int, E>{}); ++kt; }                     \
    else if (have_prev) { (void)epi_step(std::integral_constant<int, E>{}, std::false_type{}); st1 = -1; }
    CS_SLICE_WITH_STEP(0) CS_SLICE_WITH_STEP(1) CS_SLICE_WITH_STEP(2) CS_SLICE_WITH_STEP(3)
    CS_SLICE_WITH_STEP(4) CS_SLICE_WITH_STEP(5) CS_SLICE_WITH_STEP(6) CS_SLICE_WITH_STEP(7)
#undef CS_SLICE_WITH_STEP
    if constexpr (PIPE) {  // nk >= 9 (launcher): the last slice prefetches the residual of this tile's steps 0..2
      for (; kt < nk - 1; ++kt) k_slice(std::integral_constant<int, -1>{});
      k_slice(std::integral_constant<int, -2>{});
      ++kt;
    } else {
      for (; kt < nk; ++kt) k_slice(std::integral_constant<int, -1>{});
    }
    // hand the finished tile to the deferred epilogue
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < NSUB; ++j) prv[i][j] = acc[i][j];
    have_prev = true;
    pm0 = cm0; pn0 = cn0;
    prev_full = (cm0 + BM <= p.M) && (cn0 + BN <= p.N);
    if constexpr (PIPE) {
      // the three prefetches are older than the last slice's LDS-DMA (if it issued one): wait for exactly those
      if (last_dma) CS_VMCNT(LPS);
      else CS_VMCNT(0);
      radd(std::integral_constant<int, 0>{});
      radd(std::integral_constant<int, 1>{});
      radd(std::integral_constant<int, 2>{});
    }
    if constexpr (kLN) {  // s[n], c[n] of the finished tile's columns and (mu, rstd) of its rows -> wave-private stash
      {
        const float invc = 1.0f / (float)p.K;
        float mr4[4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {  // lane l owns rows 2l, 2l+1 of the tile; partial sums come from the producing epilogue
          const int mrow = min(cm0 + 2 * lane + h, p.M - 1);
          const f32x4_t* pp = reinterpret_cast<const f32x4_t*>(p.ln_part + (size_t)mrow * (2 * SPL) * 2);
          float s1 = 0.f, s2 = 0.f;
#pragma unroll
          for (int q = 0; q < SPL; ++q) {
            const f32x4_t t4 = pp[q];
            s1 += t4[0] + t4[2];
            s2 += t4[1] + t4[3];
          }
          const float mu = s1 * invc;
          mr4[2 * h] = mu;
          mr4[2 * h + 1] = 1.0f / sqrtf(fmaxf(s2 * invc - mu * mu, 0.f) + p.ln_eps);
        }
        patch_write16(patch_lds + MRST + lane * 16, f32x4_t{mr4[0], mr4[1], mr4[2], mr4[3]});
      }
      if ((lane & 15) == 0) {
#pragma unroll
        for (int j = 0; j < NSUB; ++j) {
          const int n = min(cn0 + wn * Cfg::WN + j * 16 + (lane >> 4) * 4, p.N - 4);
          patch_write16(stash_addr + j * 64, *reinterpret_cast<const f32x4_t*>(p.col_s + n));
          patch_write16(stash_addr + Cfg::WN * 4 + j * 64, *reinterpret_cast<const f32x4_t*>(p.bias + n));
        }
      }
      st1 = -1;  // those loads are consumed at once: the next wait drains the queue
    }
  }
  if (have_prev) {  // flush the last tile
    (void)epi_step(std::integral_constant<int, 0>{}, std::true_type{}); (void)epi_step(std::integral_constant<int, 1>{}, std::true_type{});
    (void)epi_step(std::integral_constant<int, 2>{}, std::true_type{}); (void)epi_step(std::integral_constant<int, 3>{}, std::true_type{});
    (void)epi_step(std::integral_constant<int, 4>{}, std::true_type{}); (void)epi_step(std::integral_constant<int, 5>{}, std::true_type{});
    (void)epi_step(std::integral_constant<int, 6>{}, std::true_type{}); (void)epi_step(std::integral_constant<int, 7>{}, std::true_type{});
  }
#ifdef CS_ABLATE
  if (g_cs_dbg && lane == 0) {
    long long* d = g_cs_dbg + ((size_t)blockIdx.x * 4 + wv) * 8;
    for (int k = 0; k < 6; ++k) d[k] = dbg_acc[k];
    d[6] = it;
    d[7] = clock64() - dbg_t0;
  }
#endif
}

constexpr int CS_MAX_DEVICES = 16;
int g_num_cus[CS_MAX_DEVICES] = {};  // per device (a handle may be moved to another GPU of the process)

template <int EPI, int NSUB, int SPL, bool PIPE, bool BF>
hipError_t launch_nb(const CsGemmParams& p, hipStream_t stream) {
  constexpr int BN = 64 * NSUB;
  constexpr int LDS = GemmCfg<NSUB>::LDS;
  static std::atomic<bool> attr_done[CS_MAX_DEVICES];  // (zero-initialised; hipFuncSetAttribute is idempotent, a racing second caller only repeats it)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= CS_MAX_DEVICES) return hipErrorInvalidDevice;
  if (!attr_done[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(cs_gemm_kernel<EPI, NSUB, SPL, PIPE, BF>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return e;
    attr_done[dev] = true;
  }
  if (g_num_cus[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return hipErrorUnknown;
    g_num_cus[dev] = n;
  }
  const int tiles_n = (p.N + BN - 1) / BN;
  const int tiles_m = (p.M + BM - 1) / BM;
  // two persistent blocks per CU (LDS and registers admit two); grid is a multiple of 8 so that b%8 labels the XCD group
  // p.bpc > 2 oversubscribes the CUs (only two blocks are resident per CU; the rest queue behind them): each block then walks
  // fewer tiles and gives its slot back sooner, which pays when a second stream's kernels share the GPU (two encoder lanes:
  // -2 % step time with 3..16 blocks per CU) and costs when the kernel runs alone (+1..5 %).
  int grid = ((p.bpc > 0 ? p.bpc : 2) * g_num_cus[dev] / 8) * 8;
#ifdef CS_ABLATE
  if (const char* e = getenv("CS_GEMM_GRID")) grid = atoi(e);
#endif
  const int need = ((tiles_m + 7) / 8) * tiles_n * 8;
  if (grid > need) grid = need;
  if (grid < 8) grid = 8;
  hipLaunchKernelGGL((cs_gemm_kernel<EPI, NSUB, SPL, PIPE, BF>), dim3(grid), dim3(256), LDS, stream, p);
  return hipGetLastError();
}

// operand type at run time (CsGemmParams::bf16).  The LayerNorm-folded epilogues (an opt-in path) are built for IEEE half only.
template <int EPI, int NSUB, int SPL, bool PIPE = false>
hipError_t launch_n(const CsGemmParams& p, hipStream_t stream) {
  if constexpr (EPI < CS_EPI_LN_F16) {
    if (p.bf16) return launch_nb<EPI, NSUB, SPL, PIPE, true>(p, stream);
  }
  return launch_nb<EPI, NSUB, SPL, PIPE, false>(p, stream);
}

template <int EPI>
hipError_t launch(const CsGemmParams& p, hipStream_t stream) {
  // column tile: 192 where it divides N (every N of the path is a multiple of 384, 128 or the 196-wide head), else 128
  if constexpr (EpiTraits<EPI>::kLN) {
    if (p.N % 192 == 0) {
      if (p.ln_sp == 4) return launch_n<EPI, 3, 2>(p, stream);
      if (p.ln_sp == 8) return launch_n<EPI, 3, 4>(p, stream);
      return launch_n<EPI, 3, 8>(p, stream);
    }
    if (p.ln_sp == 4) return launch_n<EPI, 2, 2>(p, stream);
    if (p.ln_sp == 8) return launch_n<EPI, 2, 4>(p, stream);
    return launch_n<EPI, 2, 8>(p, stream);
  } else {
#ifdef CS_ABLATE
    if (const char* e = getenv("CS_GEMM_NSUB")) {  // timing experiment: force the 128-wide column tile
      if (atoi(e) == 2 && p.N % 128 == 0) {
        if constexpr (EPI == CS_EPI_RESID_F32) {
          if (p.resid && p.K / BK >= 9) return launch_n<EPI, 2, 0, true>(p, stream);
        }
        return launch_n<EPI, 2, 0>(p, stream);
      }
    }
#endif
    // Narrow outputs (N <= 384: out-proj, fc2, the decoder's C x C linears) take the 128-wide tile: 1.5x as many tiles, so
    // the persistent grid is fuller and its last round shorter (out-proj / fc2 at 12 images: 258 tiles of 192 columns leave
    // half the block slots empty; measured -21 % at 16440 rows, -14 % at 32880, -2..-12 % at 65760).  The LayerNorm-stat
    // producers keep the 192-wide tile (their partial-sum slots are sized by cs_gemm_column_tiles).
    constexpr bool kNarrowOk = EPI != CS_EPI_PATCH_F32 && EPI != CS_EPI_RESID_F32_LN;
    const bool wide = p.N % 192 == 0 && !(kNarrowOk && p.N % 128 == 0 && p.N <= 384);
    if constexpr (EPI == CS_EPI_RESID_F32) {
      // residual prefetch pipeline: needs a residual, >= 9 K slices (the 8 step-carrying slices + a last one) and 32-bit offsets
      if (p.resid && p.K / BK >= 9 && p.M >= 16 && p.N >= 16 && (long long)p.M * p.ldr * 4 < (1ll << 31)) {
        if (wide) return launch_n<EPI, 3, 0, true>(p, stream);
        return launch_n<EPI, 2, 0, true>(p, stream);
      }
    }
    if (wide) return launch_n<EPI, 3, 0>(p, stream);
    return launch_n<EPI, 2, 0>(p, stream);
  }
}

}  // namespace

// column tile the launcher picks for N (the LayerNorm partial-sum slots of a producer are 4 per column tile)
extern "C" int cs_gemm_column_tiles(int N) { return N % 192 == 0 ? N / 192 : (N + 127) / 128; }

namespace {

}  // namespace

// Host-side shape contract (checked here so a bad call fails loudly instead of faulting on the GPU).
extern "C" int cs_gemm_column_tiles(int N);
extern "C" int cs_gemm256_supported(const CsGemmParams* p, int epi);
extern "C" const char* cs_gemm_check(const CsGemmParams* p, int epi) {
  // the LayerNorm-folded forms of the 256-tile kernel (gemm256.hip) have their own statistics layout: finalised (mean, rstd) rows on the consumer
  // side (ln_sp == 1), N / 64 partial slots per row on the producer side; both operand types
  const bool ln256 = epi >= CS_EPI_LN_F16 && cs_gemm256_supported(p, epi);
  if (p->M <= 0 || p->N <= 0 || p->K <= 0) return "gemm: empty shape";
  if (p->K % 64) return "gemm: K must be a multiple of 64";
  if (p->N % 4 || p->ldc % 4) return "gemm: N and ldc must be multiples of 4";
  if (epi <= CS_EPI_BIAS_LEAKY_F16 && (p->N % 8 || p->ldc % 8)) return "gemm: fp16 outputs need N and ldc multiples of 8 (16-byte row stores)";
  if (p->lda % 8 || p->ldw % 8) return "gemm: lda/ldw must be multiples of 8 (16-byte rows)";
  if (p->lda < p->K || p->ldw < p->K) return "gemm: lda/ldw smaller than K";
  if (!p->A || !p->W || !p->out) return "gemm: null operand";
  if (p->bf16 && epi >= CS_EPI_LN_F16 && !ln256) return "gemm: the 128-row kernel's LayerNorm-folded epilogues are built for fp16 operands only";
  if (p->scale) return "gemm: a per-column scale is folded into the packed weights (cs_op_pack_f16 row_scale), it is not an epilogue operand";
  if ((epi == CS_EPI_RESID_F32 || epi == CS_EPI_RESID_F32_LN) && p->resid && p->ldr % 4) return "gemm: ldr must be a multiple of 4";
  if (epi == CS_EPI_RESID_F32_LN && (!p->out_f16 || !p->stats_out)) return "gemm: RESID_F32_LN needs out_f16 and stats_out";
  if (!ln256 && (epi == CS_EPI_RESID_F32_LN || epi == CS_EPI_PATCH_F32) && p->stats_out && p->stats_sp != 4 * cs_gemm_column_tiles(p->N))
    return "gemm: stats_sp must be 4 x the number of column tiles";
  if (!ln256 && (epi == CS_EPI_LN_F16 || epi == CS_EPI_LN_GELU_F16) &&
      (!p->ln_part || !p->col_s || !p->bias || (p->ln_sp != 4 && p->ln_sp != 8 && p->ln_sp != 16) || p->N % 8 || p->ldc % 8))
    return "gemm: LayerNorm-folded epilogue needs ln_part, col_s, bias (= c), ln_sp in {4,8,16}";
  if (epi == CS_EPI_PATCH_F32 && (!p->pos || p->Np <= 0 || p->M % p->Np)) return "gemm: bad patch epilogue params";
  if (epi == CS_EPI_HEAD_SCORE && (p->Np <= 0 || p->gw <= 0 || p->Np % p->gw || p->M % p->Np || p->N != p->P * p->P))
    return "gemm: bad head epilogue params";
  if (epi == CS_EPI_HEAD_SCORE && (!p->mean_part != !p->mean_cnt || !p->mean_part != !p->mean_out))
    return "gemm: the head's mean needs mean_part, mean_cnt and mean_out together";
  if (epi != CS_EPI_HEAD_SCORE && (p->mean_part || p->mean_cnt || p->mean_out)) return "gemm: mean_* belong to the head epilogue";
  if ((long long)p->M * p->lda >= (1ll << 31) || (long long)p->N * p->ldw >= (1ll << 31)) return "gemm: operand too large for 32-bit offsets";
  return nullptr;
}

#ifdef CS_ABLATE
extern "C" int cs_gemm_dbg_set(long long* buf) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_cs_dbg), &buf, sizeof(buf)); }
#endif

extern "C" hipError_t cs_gemm256_launch(const CsGemmParams* p, int epi, int bf16, hipStream_t st);

extern "C" hipError_t cs_gemm_launch(const CsGemmParams* p0, int epi, hipStream_t stream) {
  // K >= 512 with whole 256-column tiles (the ViT-B projections, the decoder's K/V projection at C = 768): the large-tile kernel
  if (cs_gemm256_supported(p0, epi)) return cs_gemm256_launch(p0, epi, p0->bf16, stream);
  CsGemmParams pp = *p0;
#ifdef CS_ABLATE
  if (const char* e = getenv("CS_GEMM_ABLATE")) pp.ablate = atoi(e);
#endif
  const CsGemmParams* p = &pp;
  switch (epi) {
    case CS_EPI_BIAS_F16: return launch<CS_EPI_BIAS_F16>(*p, stream);
    case CS_EPI_BIAS_GELU_F16: return launch<CS_EPI_BIAS_GELU_F16>(*p, stream);
    case CS_EPI_BIAS_RELU_F16: return launch<CS_EPI_BIAS_RELU_F16>(*p, stream);
    case CS_EPI_BIAS_LEAKY_F16: return launch<CS_EPI_BIAS_LEAKY_F16>(*p, stream);
    case CS_EPI_RESID_F32: return launch<CS_EPI_RESID_F32>(*p, stream);
    case CS_EPI_PATCH_F32: return launch<CS_EPI_PATCH_F32>(*p, stream);
    case CS_EPI_HEAD_SCORE: return launch<CS_EPI_HEAD_SCORE>(*p, stream);
    case CS_EPI_LN_F16: return launch<CS_EPI_LN_F16>(*p, stream);
    case CS_EPI_LN_GELU_F16: return launch<CS_EPI_LN_GELU_F16>(*p, stream);
    case CS_EPI_RESID_F32_LN: return launch<CS_EPI_RESID_F32_LN>(*p, stream);
  }
  return hipErrorInvalidValue;
}
