// Large-tile MFMA GEMM for the K >= 512 linears of the CrossScore hot path on gfx950 (the ViT-B encoder's QKV / out-proj / fc1 / fc2
// projections and the decoder's K/V projection; HF modeling_dinov2.py:211-213,250,293-297; torch functional.py:5785-5852).
//
//   out[m][n] = epilogue( bias[n] + sum_k A[m][k] * W[n][k] )     A:[M][K] 16-bit activations, W:[N][K] 16-bit (nn.Linear layout)
//
// Why a second GEMM kernel: gemm.hip's 128 x 192 x 32 tile stages 1 byte from L2 per 77 FLOP, which at K >= 768 puts the kernel on
// the L2 -> LDS staging path (measured 0.55-0.70 x hipBLASLt on the ViT-B shapes).  This kernel stages 1 byte per
// 128 FLOP: a 256 x 256 x 64 tile per workgroup of 8 waves (2 along M x 4 along N, wave tile 128 x 64 = 8 x 4 accumulators of
// v_mfma_f32_16x16x32), one persistent workgroup per CU walking output tiles in an XCD-aware order.
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
//   * the LDS-DMA stream does not stop at an output tile's end: the K tiles of the NEXT output tile follow in the same slots, and at the
//     seam the two pieces that would be issued in the new tile's first two phases are issued BEFORE the epilogue's stores, so that the
//     first waits of the new tile do not have to retire those stores (the vector-memory queue retires in order);
//   * LDS image: rows of 128 B (64 k), 16-byte chunk c of row r stored at chunk c ^ ((r >> 1) & 7): applied on the DMA's per-lane
//     SOURCE address and on the ds_read_b128 address, conflict free for the 16x16x32 operand read.
// The MFMA takes the W fragment as its first operand, so a lane owns 4 consecutive output columns of one row; the epilogue goes
// through a wave-private LDS patch (inline-asm ds ops: the compiler must not order them against the LDS-DMA in flight) and stores
// whole 128-byte lines; the bias vector lives in LDS and is the accumulators' initial value.
//
// LayerNorm folded into the epilogues (r5; ViT-B's encoder, no separate LayerNorm pass over the fp32 residual stream; HF modeling_dinov2.py:361-380):
//   LN = 2, producer (residual epilogue: out-projection, fc2): besides the fp32 rows it writes their 16-bit copy (the consuming projection's A
//           operand, UN-normalised) and, per row and 64-column wave slice, the partial (sum, sum of squares) of the new fp32 values
//           [rows][N / 64][2]; cs_ln_finalize_kernel (elementwise.hip) turns the partials of a row into (mean, rstd).
//   LN = 1, consumer (QKV, fc1 [+ GELU]): out = rstd[m] * (acc - mean[m] * s[n]) + c[n] with W' = W * gamma packed, s[n] = sum_k W'[n][k],
//           c[n] = b[n] + sum_k beta[k] W[n][k] (cs_finalize).  A wave's 128 (mean, rstd) pairs and its 64 s / c values travel into its
//           private patch by three LDS-DMA instructions at the START of the tile (they land under the K loop: the loop's counted waits retire
//           them, being older than everything those waits leave in flight) and are read into registers before the patch is reused.
//   fp16(x) instead of fp16(LN(x)) as the MFMA operand: the same RELATIVE rounding step per element (the row's 1/sigma scales value and error
//   alike), so an outlier channel at 300 in a row of sigma 20 carries 0.125 / 20 = 6e-3 against 3.9e-3 for the rounded normalised value 15.
#include "cs_common.h"
#include <atomic>
#include <stdlib.h>
#include <type_traits>
#include <utility>

namespace {

constexpr int G_BM = 256, G_BN = 256, G_BK = 64;
constexpr int G_ROWB = G_BK * 2;            // 128 bytes per staged row
constexpr int G_OPND = 256 * G_ROWB;        // 32 KiB: one operand's K tile
constexpr int G_WOFF = 2 * G_OPND;          // ring = [A buffer 0 | A buffer 1 | W buffer 0 | W buffer 1]: every fragment read is one of four
                                            // per-lane base addresses (operand x k-step) plus an immediate below 64 KiB
constexpr int G_RING = 4 * G_OPND;          // 128 KiB
constexpr int G_NMAX = 3072;                // widest N whose bias vector is kept in LDS (wider: read from memory at every tile start)
constexpr int G_NLIM = 8192;                // widest N taken
constexpr int G_BIAS = G_RING;              // 12 KiB
constexpr int G_PROW = 144;                 // patch row: 128 B + 16 B pad
constexpr int G_PATCH = 16 * G_PROW;        // one 16-row patch per wave
constexpr int G_PATCH0 = G_BIAS + G_NMAX * 4;
constexpr int G_LDS = G_PATCH0 + 8 * G_PATCH;   // 158 KiB

typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));

template <int V> using IC = std::integral_constant<int, V>;

#ifdef CS_G256_STAMP
// diagnostic builds only (tools/gemm256_phases.py): per (block < 64, wave) cycles summed over the K loop of
//   4 p + 0: load segment of phase p (LDS reads, LDS-DMA issue, counted wait)   4 p + 1: its barrier
//   4 p + 2: LDS wait + 16 MFMAs                                                 4 p + 3: the barrier behind them;    16: seam + epilogue
__device__ unsigned long long g_g256_dbg[64 * 8 * 20];
#define G_TS(k) do { __builtin_amdgcn_sched_barrier(0); unsigned long long now_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_) :: "memory"); \
    ph[k] += now_ - tlast; tlast = now_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define G_TS(k) do { } while (0)
#endif
#define G_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#define G_LGKM(n) do { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(n) : "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define G_SB() __builtin_amdgcn_sched_barrier(0)

// wave-private patch / bias access by inline asm (invisible to the compiler's LDS-DMA alias ordering; waits are the caller's)
template <int OFF>
__device__ __forceinline__ void pw8(unsigned addr, u32x2_t v) { asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory"); }
template <int OFF>
__device__ __forceinline__ void pw16(unsigned addr, f32x4_t v) { asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory"); }
template <int OFF>
__device__ __forceinline__ void pr16(unsigned addr, f32x4_t& v) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(v) : "v"(addr), "n"(OFF) : "memory"); }
template <int OFF>
__device__ __forceinline__ void pr8(unsigned addr, u32x2_t& v) { asm volatile("ds_read_b64 %0, %1 offset:%2" : "=&v"(v) : "v"(addr), "n"(OFF) : "memory"); }
// x + (the value SHR lanes below in the same 16-lane DPP row; 0 from outside the row): three of them leave the sum of 8 consecutive lanes in the last
template <int SHR>
__device__ __forceinline__ float dpp_row_shr_add(float v) {
  const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x110 + SHR, 0xf, 0xf, true);
  return v + __int_as_float(moved);
}
// residual rows (fp32 epilogue): 16 bytes per lane at sbase + voff by inline asm, so that the compiler neither counts nor waits for the
// load (its own wait would be vmcnt(0): every store of the previous step); the epilogue's counted waits cover it
__device__ __forceinline__ void gl16(f32x4_t& r, unsigned voff, const float* sbase) {
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(r) : "v"(voff), "s"(sbase) : "memory");
}
// first use of a register set loaded by gl16 (behind the counted wait that retires it); the marker is what tools/asm_audit.py looks for
__device__ __forceinline__ void touch(f32x4_t& r) { asm volatile("; GL16_USE %0" : "+v"(r)); }

// RES: the fp32 epilogue adds residual rows (compile time: a run-time test would put every in-flight residual register behind phi copies)
// LN: 0 plain epilogues, 1 LayerNorm-folded consumer (16-bit epilogues), 2 LayerNorm producer (residual epilogue)
template <int EPI, bool BF, bool RES, int LN = 0>
__global__ __launch_bounds__(512, 2) void cs_gemm256_kernel(CsGemmParams p) {
  constexpr bool kHalf = EPI <= CS_EPI_BIAS_LEAKY_F16;
  constexpr bool kLNc = LN == 1, kLNp = LN == 2;
  static_assert(!kLNc || kHalf, "gemm256: the LayerNorm-folded consumer has a 16-bit output");
  static_assert(!kLNp || RES, "gemm256: the LayerNorm producer is the residual epilogue");
  static_assert(!RES || EPI == CS_EPI_RESID_F32, "gemm256: residual rows belong to the fp32 epilogue");
  static_assert(kHalf || EPI == CS_EPI_RESID_F32, "gemm256: bias->16-bit and residual fp32 epilogues only");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wv >> 2, wn = wv & 3;

  // ---- persistent tile walk, XCD aware: blocks b, b + 8, .. share an XCD (round-robin dispatch; speed only); XCD x owns the A row
  //      panels tm == x (mod 8) and its blocks walk that list n-fastest, so the blocks resident on one L2 share A panels ----
  const int tiles_n = p.N / G_BN;
  const int tiles_m = (p.M + G_BM - 1) / G_BM;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
  const int panels_x = (tiles_m - xcd + 7) / 8;
  const int ntile_x = panels_x * tiles_n;
  if (slot >= ntile_x) return;  // whole workgroup, before any barrier
  const int T = p.K / G_BK;     // K tiles per output tile (even, >= 6: cs_gemm256_supported)
  auto tile_of = [&](int idx, int& m0, int& n0) {
    const int pl = idx / tiles_n;
    m0 = (pl * 8 + xcd) * G_BM;
    n0 = (idx - pl * tiles_n) * G_BN;
  };

  // ---- bias vector -> LDS once per block (plain accesses: nothing else is in flight yet) ----
  for (int i = tid * 4; i < min(p.N, G_NMAX); i += 2048) {
    f32x4_t b4 = {0.f, 0.f, 0.f, 0.f};
    if (p.bias && !kLNc) b4 = *reinterpret_cast<const f32x4_t*>(p.bias + i);  // (consumer: `bias` is c[n], applied behind the row scale)
    *reinterpret_cast<f32x4_t*>(smem + G_BIAS + i * 4) = b4;
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  G_SB();

  // ---- LDS-DMA maps.  One instruction = 8 rows x 128 B; lane i writes LDS chunk (i & 7) of row (i >> 3) and fetches source chunk
  //      (i & 7) ^ ((row >> 1) & 7).  Half-tile h of A = rows {wm' * 128 + h * 64 + 0..63}; of W = rows {wn' * 64 + h * 32 + 0..31};
  //      wave wv issues pieces 2 wv, 2 wv + 1 of each half-tile. ----
  const int srow = lane >> 3;
  unsigned offA[2][2], offW[2][2];   // [half][piece] global byte offsets of the load cursor's output tile
  auto set_offsets = [&](int m0, int n0) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int ra = (wv >> 2) * 128 + h * 64 + ((16 * wv) & 63) + 8 * e + srow;
        const int gca = (lane & 7) ^ ((ra >> 1) & 7);
        offA[h][e] = ((unsigned)min(m0 + ra, p.M - 1) * (unsigned)p.lda + (unsigned)gca * 8u) * 2u;
        const int rw = (wv >> 1) * 64 + h * 32 + ((16 * wv) & 31) + 8 * e + srow;
        const int gcw = (lane & 7) ^ ((rw >> 1) & 7);
        offW[h][e] = ((unsigned)(n0 + rw) * (unsigned)p.ldw + (unsigned)gcw * 8u) * 2u;
      }
  };
  // load cursor: the K tile whose pieces are being issued (it runs two K tiles ahead of the MFMAs, across output tiles)
  int l_idx = slot, l_kt = 0;
  bool l_valid = true;
  {
    int m0, n0;
    tile_of(l_idx, m0, n0);
    set_offsets(m0, n0);
  }
  // piece WHICH (0 = A half 0, 1 = W half 0, 2 = W half 1, 3 = A half 1) of the cursor's K tile, into buffer (l_kt & 1)
  auto stage = [&](auto WHICH_) {
    constexpr int WHICH = decltype(WHICH_)::value;
    constexpr bool isA = WHICH == 0 || WHICH == 3;
    constexpr int h = (WHICH == 0 || WHICH == 1) ? 0 : 1;
    char* base = smem + (l_kt & 1) * G_OPND + (isA ? 0 : G_WOFF);
    const char* src = reinterpret_cast<const char*>(isA ? p.A : p.W) + (size_t)l_kt * G_ROWB;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int r0 = isA ? (wv >> 2) * 128 + h * 64 + ((16 * wv) & 63) + 8 * e : (wv >> 1) * 64 + h * 32 + ((16 * wv) & 31) + 8 * e;
      __builtin_amdgcn_global_load_lds(CS_GLOBAL_PTR(src + (isA ? offA[h][e] : offW[h][e])), CS_LDS_PTR(base + r0 * G_ROWB), 16, 0, 0);
    }
  };
  auto advance = [&]() {
    if (++l_kt == T) {
      l_kt = 0;
      l_idx += slots;
      l_valid = l_idx < ntile_x;
      if (l_valid) {
        int m0, n0;
        tile_of(l_idx, m0, n0);
        set_offsets(m0, n0);
      }
    }
  };

  // ---- fragment read addressing: lane (fr = lane & 15, cq = lane >> 4) reads row fr, chunk (4 s + cq) ^ ((fr >> 1) & 7) ----
  const int fr = lane & 15, cq = lane >> 4;
  const int c0 = (cq ^ ((fr >> 1) & 7)) * 16;
  const char* rdA = smem + wm * (128 * G_ROWB) + fr * G_ROWB;           // + buffer + (mh * 64 + 16 i) * 128 + (c0 | c0 ^ 64)
  const char* rdW = smem + G_WOFF + wn * (64 * G_ROWB) + fr * G_ROWB;   // + buffer + (nh * 32 + 16 j) * 128 + ..

  f32x4_t acc[8][4];
  h16x8_t fa[4][2], fw[2][2][2];  // A fragments of the current M half [i][s]; W fragments of both N halves [nh][j][s]
  auto ld_a = [&](auto B_, auto MH_) {
    constexpr int B = decltype(B_)::value, MH = decltype(MH_)::value;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const char* r = rdA + B * G_OPND + (MH * 64 + 16 * i) * G_ROWB;
      fa[i][0] = *reinterpret_cast<const h16x8_t*>(r + c0);
      fa[i][1] = *reinterpret_cast<const h16x8_t*>(r + (c0 ^ 64));
    }
  };
  auto ld_w = [&](auto B_, auto NH_) {
    constexpr int B = decltype(B_)::value, NH = decltype(NH_)::value;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const char* r = rdW + B * G_OPND + (NH * 32 + 16 * j) * G_ROWB;
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
        for (int j = 0; j < 2; ++j) acc[MH * 4 + i][NH * 2 + j] = mfma_16x16x32<BF>(fw[NH][j][s], fa[i][s], acc[MH * 4 + i][NH * 2 + j]);
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

  // ---- wave-private patch / bias addressing, recomputed at every seam from an opaque copy of the lane id: hoisted out of the tile loop
  //      these six registers would be spilled around the K loop (which runs at the 256-register limit) ----
  const unsigned lds0 = (unsigned)(size_t)CS_LDS_PTR(smem);
  unsigned pw_addr = 0, pr_addr = 0, bias_addr = 0;
  [[maybe_unused]] unsigned ln_a = 0, ln_b = 0;  // consumer: this lane's addresses of its rows' (mean, rstd) and its columns' s / c in the patch
  int rrow = 0, rch = 0;
  auto seam_addresses = [&]() {
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const unsigned patch = lds0 + G_PATCH0 + wv * G_PATCH;
    pw_addr = patch + (ln & 15) * G_PROW + (ln >> 4) * (kHalf ? 8 : 16);   // accumulator layout: row fr, 4 columns at 4 cq of a 16-column tile
    rrow = ln >> 3; rch = ln & 7;
    pr_addr = patch + rrow * G_PROW + rch * 16;                            // row segments: 8 rows x 8 chunks per read (+ 8 rows: + 8 * G_PROW)
    bias_addr = lds0 + G_BIAS + (wn * 64 + 4 * (ln >> 4)) * 4;
    if constexpr (kLNc) { ln_a = patch + (ln & 15) * 8; ln_b = patch + 1024 + (ln >> 4) * 16; }
  };
  // consumer: rows [m0 + 128 wm, + 128) x (mean, rstd) (1 KiB; ln_part holds whole 256-row tiles), s and c of columns [n0 + 64 wn, + 64)
  // (256 B each) -> the wave's patch [0, 1024) | [1024, 1280) | [1280, 1536)
  auto ln_fetch = [&](int m0, int n0) {
    if constexpr (kLNc) {
      int ln = lane;
      asm volatile("" : "+v"(ln));
      char* patch = smem + G_PATCH0 + wv * G_PATCH;
      __builtin_amdgcn_global_load_lds(CS_GLOBAL_PTR(p.ln_part + ((size_t)(m0 + wm * 128) + 2 * ln) * 2), CS_LDS_PTR(patch), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(CS_GLOBAL_PTR(p.col_s + n0 + wn * 64 + ln), CS_LDS_PTR(patch + 1024), 4, 0, 0);
      __builtin_amdgcn_global_load_lds(CS_GLOBAL_PTR(p.bias + n0 + wn * 64 + ln), CS_LDS_PTR(patch + 1280), 4, 0, 0);
    }
  };
  auto init_acc = [&](int n0) {  // accumulators start at the bias of their columns (16 j + 4 cq .. + 3 of the wave's 64)
    f32x4_t b4[4];
    if constexpr (kLNc) {
#pragma unroll
      for (int j = 0; j < 4; ++j) b4[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    } else if (p.N <= G_NMAX) {
      const unsigned a = bias_addr + n0 * 4;
      pr16<0>(a, b4[0]); pr16<64>(a, b4[1]); pr16<128>(a, b4[2]); pr16<192>(a, b4[3]);
      G_LGKM(0);
    } else {
      // N > 3072 (dinov2-large's fc1): the vector does not fit beside the ring; four plain loads per tile (the compiler waits for them
      // with vmcnt(0), i.e. for the DMA pieces in flight too: once per output tile)
      const float* bp = p.bias ? p.bias + n0 + (int)((bias_addr - (lds0 + G_BIAS)) >> 2) : nullptr;
#pragma unroll
      for (int j = 0; j < 4; ++j) b4[j] = bp ? *reinterpret_cast<const f32x4_t*>(bp + 16 * j) : f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = b4[j];
  };

  // ---- prologue: K tiles 0 and 1 of the first output tile completely ----
  stage(IC<0>{}); stage(IC<1>{}); stage(IC<2>{}); stage(IC<3>{}); advance();
  stage(IC<0>{}); stage(IC<1>{}); stage(IC<2>{}); stage(IC<3>{}); advance();
  G_VMCNT(8);
  bar();               // K tile 0 has landed for everyone; the bias vector is in LDS
  if (wm == 1) bar();  // the second M half runs one barrier behind the first

#ifdef CS_G256_STAMP
  unsigned long long ph[20] = {};
  unsigned long long tlast = __builtin_amdgcn_s_memtime();
  const unsigned long long tbegin = tlast, rbegin = __builtin_amdgcn_s_memrealtime();
#endif
  // ---- one K tile (buffer B).  kt0: the first K tile of an output tile (its successor's late pieces were issued at the seam) ----
  // last: the output tile's last K tile.  Its final barrier is left to the caller for the second M half (wm == 1), which then runs its seam
  // (epilogue) BEFORE that barrier, i.e. beside the first half's seam instead of after it: the halves are one barrier apart, and with
  // the barrier in its usual place the first half would wait for the second's MFMAs and then the second for the first's whole epilogue.
  auto ktile = [&](auto B_, bool kt0, bool last) {
    constexpr int B = decltype(B_)::value;
    // phase 1: quadrant (0, 0)
    ld_w(IC<B>{}, IC<0>{}); ld_a(IC<B>{}, IC<0>{});
    if (!kt0 && l_valid) stage(IC<2>{});
    G_TS(0); bar_then_wait(); G_TS(1); mma(IC<0>{}, IC<0>{}); G_TS(2); bar(); G_TS(3);
    // phase 2: quadrant (0, 1)
    ld_w(IC<B>{}, IC<1>{});
    if (!kt0 && l_valid) { stage(IC<3>{}); advance(); }
    G_TS(4); bar_then_wait(); G_TS(5); mma(IC<0>{}, IC<1>{}); G_TS(6); bar(); G_TS(7);
    // phase 3: quadrant (1, 1)
    ld_a(IC<B>{}, IC<1>{});
    const bool more = l_valid;
    if (more) stage(IC<0>{});
    G_TS(8); bar_then_wait(); G_TS(9); mma(IC<1>{}, IC<1>{}); G_TS(10); bar(); G_TS(11);
    // phase 4: quadrant (1, 0); the next K tile has landed behind this phase's first barrier (the two pieces just issued stay in flight)
    if (more) { stage(IC<1>{}); G_VMCNT(4); }
    else G_VMCNT(0);
    G_TS(12); bar_then_wait(); G_TS(13); mma(IC<1>{}, IC<0>{}); G_TS(14);
    if (!(last && wm == 1)) { bar(); G_TS(15); }
  };

  for (int idx = slot; idx < ntile_x; idx += slots) {
    int cm0, cn0;
    tile_of(idx, cm0, cn0);
    seam_addresses();
    init_acc(cn0);
    ln_fetch(cm0, cn0);
    ktile(IC<0>{}, true, false);
    ktile(IC<1>{}, false, false);
    for (int t = 2; t < T; t += 2) {
      ktile(IC<0>{}, false, false);
      ktile(IC<1>{}, false, t + 2 >= T);
    }
    // ---- seam: the late pieces of the next output tile's K tile 1 go out BEFORE this tile's stores ----
    if (l_valid) { stage(IC<2>{}); stage(IC<3>{}); advance(); }
    G_SB();
    seam_addresses();
    // ---- epilogue: a 16-row sub-tile goes into the patch in the accumulator layout and comes out as row segments, 16 B per lane,
    //      whole 128-byte lines to memory.  LDS operations of one wave execute in order, so step i + 1's writes are issued right behind
    //      step i's reads and a counted lgkmcnt retires the reads. ----
    const int row_w = cm0 + wm * 128;       // first row of the wave tile
    const int col_w = cn0 + wn * 64;        // first column
    if constexpr (kHalf) {
      [[maybe_unused]] f32x4_t s4[4], c4[4];
      [[maybe_unused]] float rstd[8], nm[8];
      if constexpr (kLNc) {
        // (mean, rstd) of this lane's eight rows 16 i + fr and s / c of its sixteen columns 16 j + 4 cq ..: out of the patch before it is reused
        u32x2_t st8[8];
        pr8<0>(ln_a, st8[0]); pr8<128>(ln_a, st8[1]); pr8<256>(ln_a, st8[2]); pr8<384>(ln_a, st8[3]);
        pr8<512>(ln_a, st8[4]); pr8<640>(ln_a, st8[5]); pr8<768>(ln_a, st8[6]); pr8<896>(ln_a, st8[7]);
        pr16<0>(ln_b, s4[0]); pr16<64>(ln_b, s4[1]); pr16<128>(ln_b, s4[2]); pr16<192>(ln_b, s4[3]);
        pr16<256>(ln_b, c4[0]); pr16<320>(ln_b, c4[1]); pr16<384>(ln_b, c4[2]); pr16<448>(ln_b, c4[3]);
        G_LGKM(0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          rstd[i] = __uint_as_float(st8[i][1]);
          nm[i] = -__uint_as_float(st8[i][0]) * rstd[i];
        }
      }
      auto put = [&](auto I_) {
        constexpr int i = decltype(I_)::value;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
          if constexpr (kLNc) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaf(rstd[i], v[r], fmaf(nm[i], s4[j][r], c4[j][r]));
          }
          if constexpr (EPI == CS_EPI_BIAS_GELU_F16) gelu_erf4(v);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if constexpr (EPI == CS_EPI_BIAS_RELU_F16) v[r] = fmaxf(v[r], 0.f);
            if constexpr (EPI == CS_EPI_BIAS_LEAKY_F16) v[r] = v[r] >= 0.f ? v[r] : 0.01f * v[r];
          }
          const u32x2_t pk = {pack_o16x2<BF>(v[0], v[1]), pack_o16x2<BF>(v[2], v[3])};
          if (j == 0) pw8<0>(pw_addr, pk);
          if (j == 1) pw8<32>(pw_addr, pk);
          if (j == 2) pw8<64>(pw_addr, pk);
          if (j == 3) pw8<96>(pw_addr, pk);
        }
      };
      f32x4_t seg[2];
      auto get = [&]() { pr16<0>(pr_addr, seg[0]); pr16<8 * G_PROW>(pr_addr, seg[1]); };
      auto out = [&](auto I_) {
        constexpr int i = decltype(I_)::value;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int m = row_w + 16 * i + q * 8 + rrow;
          if (m < p.M) *reinterpret_cast<f32x4_t*>(reinterpret_cast<h16_t*>(p.out) + (size_t)m * p.ldc + col_w + rch * 8) = seg[q];
        }
      };
      put(IC<0>{}); get();
#define G_STEP(I) put(IC<I>{}); G_LGKM(4); out(IC<I - 1>{}); G_SB(); get();
      G_STEP(1) G_STEP(2) G_STEP(3) G_STEP(4) G_STEP(5) G_STEP(6) G_STEP(7)
#undef G_STEP
      G_LGKM(0); out(IC<7>{});
    } else {
      // fp32: 16 steps of (16 rows x 32 columns): sub-tile (i, nh) = accumulators [i][2 nh], [i][2 nh + 1].  The residual rows are loaded
      // FOUR steps ahead by inline asm into four register sets (the operand fragments' registers are free here).  Two steps ahead (r3's
      // first form) left one 2-KiB request per wave in flight behind each wait: 16 memory latencies per tile, as long as the tile's K loop
      // at K = 768.  Vector-memory order: L0 L1 L2 L3 | S0 L4 | S1 L5 | .. | S11 L15 | S12 | S13 | S14 | S15 (two instructions each); when
      // step t's rows are needed the younger instructions number G_RCNT(t) below (the DMA pieces issued above are older: they land first).
      // The loads' destination registers are written when the data arrives, not at the asm statement: nothing may touch them in between
      // (tools/asm_audit_gl.py checks the build's .s for that).
      // With the LayerNorm outputs a step issues 4 stores (fp32 rows, their 16-bit copies) and every second step (nh == 1) 2 more (partial sums):
      // the same bookkeeping with |S_k| = 4 + 2 (k & 1) instead of 2.
#define G_SK(k) (kLNp ? 4 + 2 * ((k) & 1) : 2)
#define G_RCNT(t) ((t) == 0 ? 6 : (t) == 1 ? 6 + G_SK(0) : (t) == 2 ? 6 + G_SK(0) + G_SK(1) : (t) == 3 ? 6 + G_SK(0) + G_SK(1) + G_SK(2) : \
                   (t) <= 12 ? 6 + G_SK((t) - 3) + G_SK((t) - 2) + G_SK((t) - 1) :                                                          \
                   (t) == 13 ? 4 + G_SK(10) + G_SK(11) + G_SK(12) : (t) == 14 ? 2 + G_SK(11) + G_SK(12) + G_SK(13) : G_SK(12) + G_SK(13) + G_SK(14))
      [[maybe_unused]] f32x4_t ra0, ra1, rb0, rb1, rc0, rc1, rd0, rd1;
      auto rload = [&](auto S_, f32x4_t& r0, f32x4_t& r1) {
        constexpr int st = decltype(S_)::value;
        constexpr int i = st >> 1, nh = st & 1;
        // per-lane row, clamped into the matrix (rows past M are never stored); 32-bit byte offsets: M * ldr * 4 < 2^32 (cs_gemm256_supported)
        const int m = row_w + 16 * i + rrow;
        const float* sb = p.resid + col_w + nh * 32;
        gl16(r0, (unsigned)(min(m, p.M - 1) * p.ldr + rch * 4) * 4u, sb);
        gl16(r1, (unsigned)(min(m + 8, p.M - 1) * p.ldr + rch * 4) * 4u, sb);
      };
      auto put = [&](auto S_) {
        constexpr int st = decltype(S_)::value;
        constexpr int i = st >> 1, nh = st & 1;
        pw16<0>(pw_addr, acc[i][2 * nh]);
        pw16<64>(pw_addr, acc[i][2 * nh + 1]);
      };
      f32x4_t seg0, seg1;
      [[maybe_unused]] float sa0 = 0.f, sb0 = 0.f, sa1 = 0.f, sb1 = 0.f;  // partial (sum, sum of squares) of this lane's columns, rows rrow / rrow + 8
      auto get = [&]() { pr16<0>(pr_addr, seg0); pr16<8 * G_PROW>(pr_addr, seg1); };
      auto out = [&](auto S_, f32x4_t& r0, f32x4_t& r1) {
        constexpr int st = decltype(S_)::value;
        constexpr int i = st >> 1, nh = st & 1;
        if constexpr (RES) { touch(r0); touch(r1); seg0 += r0; seg1 += r1; }
        const int m_a = row_w + 16 * i + rrow, m_b = m_a + 8;
        float* o = reinterpret_cast<float*>(p.out) + (size_t)m_a * p.ldc + col_w + nh * 32 + rch * 4;
        if (m_a < p.M) *reinterpret_cast<f32x4_t*>(o) = seg0;
        if (m_b < p.M) *reinterpret_cast<f32x4_t*>(o + (size_t)8 * p.ldc) = seg1;
        if constexpr (kLNp) {
          h16_t* o16 = p.out_f16 + (size_t)m_a * p.ldc + col_w + nh * 32 + rch * 4;
          if (m_a < p.M) *reinterpret_cast<u32x2_t*>(o16) = u32x2_t{pack_o16x2<BF>(seg0[0], seg0[1]), pack_o16x2<BF>(seg0[2], seg0[3])};
          if (m_b < p.M) *reinterpret_cast<u32x2_t*>(o16 + (size_t)8 * p.ldc) = u32x2_t{pack_o16x2<BF>(seg1[0], seg1[1]), pack_o16x2<BF>(seg1[2], seg1[3])};
          const float a0 = (seg0[0] + seg0[1]) + (seg0[2] + seg0[3]), a1 = (seg1[0] + seg1[1]) + (seg1[2] + seg1[3]);
          const float b0 = fmaf(seg0[0], seg0[0], fmaf(seg0[1], seg0[1], fmaf(seg0[2], seg0[2], seg0[3] * seg0[3])));
          const float b1 = fmaf(seg1[0], seg1[0], fmaf(seg1[1], seg1[1], fmaf(seg1[2], seg1[2], seg1[3] * seg1[3])));
          if constexpr (nh == 0) { sa0 = a0; sb0 = b0; sa1 = a1; sb1 = b1; }
          else {
            // both 32-column halves of the wave's 64 columns are in: sum over the eight lanes of a row (one DPP row holds two rows' lanes), the
            // row's last lane (rch == 7) stores slot (column tile) * 4 + wn
            float t0 = sa0 + a0, u0 = sb0 + b0, t1 = sa1 + a1, u1 = sb1 + b1;
            t0 = dpp_row_shr_add<1>(t0); u0 = dpp_row_shr_add<1>(u0); t1 = dpp_row_shr_add<1>(t1); u1 = dpp_row_shr_add<1>(u1);
            t0 = dpp_row_shr_add<2>(t0); u0 = dpp_row_shr_add<2>(u0); t1 = dpp_row_shr_add<2>(t1); u1 = dpp_row_shr_add<2>(u1);
            t0 = dpp_row_shr_add<4>(t0); u0 = dpp_row_shr_add<4>(u0); t1 = dpp_row_shr_add<4>(t1); u1 = dpp_row_shr_add<4>(u1);
            const int slot = (cn0 / G_BN) * 4 + wn;
            if (rch == 7 && m_a < p.M) *reinterpret_cast<float2*>(p.stats_out + ((size_t)m_a * p.stats_sp + slot) * 2) = make_float2(t0, u0);
            if (rch == 7 && m_b < p.M) *reinterpret_cast<float2*>(p.stats_out + ((size_t)m_b * p.stats_sp + slot) * 2) = make_float2(t1, u1);
          }
        }
      };
      const bool full = cm0 + G_BM <= p.M;  // a ragged tile's masked stores make the store count unknown: its waits are vmcnt(0)
      if constexpr (RES) { rload(IC<0>{}, ra0, ra1); rload(IC<1>{}, rb0, rb1); rload(IC<2>{}, rc0, rc1); rload(IC<3>{}, rd0, rd1); }
      put(IC<0>{}); get();
#define G_STEPF(S, R0, R1)                                                                     \
      put(IC<S>{}); G_LGKM(2);                                                                 \
      if constexpr (RES) { if (full) G_VMCNT(G_RCNT(S - 1)); else G_VMCNT(0); }                \
      G_SB(); out(IC<S - 1>{}, R0, R1); G_SB();                                                \
      if constexpr (RES && S + 3 < 16) rload(IC<(S + 3 < 16 ? S + 3 : 0)>{}, R0, R1);          \
      get();
      G_STEPF(1, ra0, ra1) G_STEPF(2, rb0, rb1) G_STEPF(3, rc0, rc1) G_STEPF(4, rd0, rd1) G_STEPF(5, ra0, ra1) G_STEPF(6, rb0, rb1)
      G_STEPF(7, rc0, rc1) G_STEPF(8, rd0, rd1) G_STEPF(9, ra0, ra1) G_STEPF(10, rb0, rb1) G_STEPF(11, rc0, rc1) G_STEPF(12, rd0, rd1)
      G_STEPF(13, ra0, ra1) G_STEPF(14, rb0, rb1) G_STEPF(15, rc0, rc1)
#undef G_STEPF
      G_LGKM(0);
      if constexpr (RES) { if (full) G_VMCNT(G_RCNT(15)); else G_VMCNT(0); }
      G_SB(); out(IC<15>{}, rd0, rd1);
#undef G_RCNT
#undef G_SK
    }
    G_SB();
    if (wm == 1) bar();  // (the last K tile's final barrier of the second half)
    G_TS(16);
  }
#ifdef CS_G256_STAMP
  if (blockIdx.x < 64 && lane == 0) {
    unsigned long long* d = g_g256_dbg + (blockIdx.x * 8 + wv) * 20;
    for (int k = 0; k < 17; ++k) d[k] = ph[k];
    d[17] = __builtin_amdgcn_s_memtime() - tbegin;
    d[18] = __builtin_amdgcn_s_memrealtime() - rbegin;
  }
#endif
  if (wm == 0) bar();  // the first half's matching barrier
}

int g_enabled = 1;
int g_kmin = 384;  // cs_debug_gemm256_kmin

template <int EPI, bool BF, bool RES = false, int LN = 0>
hipError_t launch256(const CsGemmParams& p, hipStream_t st) {
  static std::atomic<bool> attr_done[16];  // (zero-initialised; hipFuncSetAttribute is idempotent, a racing second caller only repeats it)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return hipErrorInvalidDevice;
  if (!attr_done[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(cs_gemm256_kernel<EPI, BF, RES, LN>), hipFuncAttributeMaxDynamicSharedMemorySize, G_LDS);
    if (e != hipSuccess) return e;
    attr_done[dev] = true;
  }
  static int num_cus[16] = {};
  if (num_cus[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return hipErrorUnknown;
    num_cus[dev] = n;
  }
  const int tiles_n = p.N / G_BN, tiles_m = (p.M + G_BM - 1) / G_BM;
  // one persistent workgroup per CU (LDS and registers admit one); the grid is a multiple of 8 so that b % 8 labels the XCD group
  int grid = (num_cus[dev] / 8) * 8;
  const int need = ((tiles_m + 7) / 8) * tiles_n * 8;
  if (grid > need) grid = need;
  if (grid < 8) grid = 8;
  hipLaunchKernelGGL((cs_gemm256_kernel<EPI, BF, RES, LN>), dim3(grid), dim3(512), G_LDS, st, p);
  return hipGetLastError();
}

}  // namespace

extern "C" {

#ifdef CS_G256_STAMP
int cs_gemm256_debug_read(unsigned long long* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_g256_dbg), sizeof(g_g256_dbg)); }
#endif

// tools / tests: 0 routes every GEMM to gemm.hip's kernel again
void cs_debug_gemm256_enable(int on) { g_enabled = on; }
// tools (tools/qkv_k384_try.py): smallest K the kernel takes; 384 by default
void cs_debug_gemm256_kmin(int k) { g_kmin = k > 0 ? k : 384; }

// Shapes this kernel takes (everything else stays with gemm.hip): whole 256-column tiles, an even number of 64-deep K tiles and
// K >= 384, the plain epilogues.
int cs_gemm256_supported(const CsGemmParams* p, int epi) {
  if (!g_enabled) return 0;
  // LayerNorm-folded forms (r5): consumers read FINALISED row statistics (ln_sp == 1: [rows rounded up to 256][2] = mean, rstd; the partial-sum
  // layout with ln_sp 4 / 8 / 16 belongs to gemm.hip's kernel), the producer writes N / 64 partial slots per row
  const bool ln_c = (epi == CS_EPI_LN_F16 || epi == CS_EPI_LN_GELU_F16) && p->ln_sp == 1 && p->ln_part && p->col_s && p->bias;
  const bool ln_p = epi == CS_EPI_RESID_F32_LN && p->resid && p->out_f16 && p->stats_out && p->stats_sp == p->N / 64;
  if (epi > CS_EPI_RESID_F32 && !ln_c && !ln_p) return 0;
  // K >= 384 (r4; 512 before): at K = 384 the large tile already wins clearly -- ViT-S QKV (32 880 x 1280 x 384) 36.9 us against 52.9 on the
  // 128-row kernel (47.5 at the unpadded 1152 columns), tools/qkv_k384_try.py.  The K loop itself takes any even number >= 4 of K tiles.
  if (p->N % G_BN || p->N > G_NLIM || p->K % (2 * G_BK) || p->K < g_kmin || p->K < 4 * G_BK || p->M < G_BM) return 0;
  if (p->lda % 8 || p->ldw % 8 || p->ldc % 8) return 0;
  if ((long long)p->M * p->lda * 2 >= (1ll << 32) || (long long)p->N * p->ldw * 2 >= (1ll << 32)) return 0;
  if ((epi == CS_EPI_RESID_F32 || ln_p) && p->resid && (p->ldr % 4 || (long long)p->M * p->ldr * 4 >= (1ll << 32))) return 0;
  if (p->pos || p->pmean) return 0;
  if (!ln_p && (p->out_f16 || p->stats_out)) return 0;
  if (!ln_c && p->ln_part) return 0;
  if (ln_c && p->N > G_NLIM) return 0;
  return 1;
}

hipError_t cs_gemm256_launch(const CsGemmParams* p, int epi, int bf16, hipStream_t st) {
#define G_CASE(E) case E: return bf16 ? launch256<E, true>(*p, st) : launch256<E, false>(*p, st);
  switch (epi) {
    G_CASE(CS_EPI_BIAS_F16) G_CASE(CS_EPI_BIAS_GELU_F16) G_CASE(CS_EPI_BIAS_RELU_F16) G_CASE(CS_EPI_BIAS_LEAKY_F16)
    case CS_EPI_RESID_F32:
      if (p->resid) return bf16 ? launch256<CS_EPI_RESID_F32, true, true>(*p, st) : launch256<CS_EPI_RESID_F32, false, true>(*p, st);
      return bf16 ? launch256<CS_EPI_RESID_F32, true, false>(*p, st) : launch256<CS_EPI_RESID_F32, false, false>(*p, st);
    case CS_EPI_RESID_F32_LN: return bf16 ? launch256<CS_EPI_RESID_F32, true, true, 2>(*p, st) : launch256<CS_EPI_RESID_F32, false, true, 2>(*p, st);
    case CS_EPI_LN_F16: return bf16 ? launch256<CS_EPI_BIAS_F16, true, false, 1>(*p, st) : launch256<CS_EPI_BIAS_F16, false, false, 1>(*p, st);
    case CS_EPI_LN_GELU_F16: return bf16 ? launch256<CS_EPI_BIAS_GELU_F16, true, false, 1>(*p, st) : launch256<CS_EPI_BIAS_GELU_F16, false, false, 1>(*p, st);
  }
#undef G_CASE
  return hipErrorInvalidValue;
}

}  // extern "C"
