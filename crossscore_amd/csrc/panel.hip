// Encoder token-panel kernel for gfx950 (ViT-S: C = 384, MLP 4C = 1536): everything a DINOv2 layer does between its attention
// and the next layer's QKV projection, in ONE launch, with the 4C hidden activations never leaving the CU.
//
//   x   += attn_o Wo'^T + bo'                       Dinov2SelfOutput + layer_scale1 + residual   (HF modeling_dinov2.py:249-252,365-370)
//   x   += GELU(LN2(x) W1'^T + b1') W2'^T + b2'     norm2 + Dinov2MLP + layer_scale2 + residual  (HF:373-378, 293-297)
//   u    = f16((x - mean(x)) * rstd(x))             norm1 of the next layer (its gamma/beta live in that layer's packed Wqkv / bias)
//
// Replaces five launches of round 1 (out-proj GEMM, LayerNorm, fc1+GELU GEMM, fc2 GEMM, LayerNorm) and their HBM round trips.
//
// Structure: a workgroup = 8 waves = 128 token rows = 4 wave PAIRS; the two waves of a pair sit on the same SIMD (waves w and w+4)
// and share 32 rows, with different roles:
//   * the A wave (waves 0-3) owns fc1: it keeps norm2(x) of its 32 rows as 24 MFMA B fragments in registers, multiplies one
//     32-wide hidden slice per "tick" (24 MFMAs) and runs the GELU of the PREVIOUS slice on the VALU between those MFMAs; the
//     activated slice goes to its partner through a 2-KiB LDS slot as two ready-made B fragments;
//   * the B wave (waves 4-7) owns the residual rows: 32 x 384 fp32 accumulators (192 registers).  It does the attention output
//     projection onto them, LayerNorm (hands norm2(x) to the A wave through LDS), then one fc2 slice per tick (24 MFMAs),
//     two ticks behind the A wave, and finally writes x and the next layer's normalised rows.
// So every SIMD has one VALU-heavy and one MFMA/LDS-only instruction stream feeding the same matrix pipe (the one-wave-per-SIMD
// version of this kernel was issue-bound: MFMA issue + GELU + LDS-DMA issue of ONE wave exceeded the MFMA pipe time 1.6x).
// MFMA v_mfma_f32_32x32x16_f16, WEIGHT fragment = A operand (32 output features x 16 k), activation fragment = B operand
// (16 k x 32 token rows): D[feature][row], a lane (j = lane & 31, h = lane >> 5) holds, for token row j, the 16 features
// rho(h, r) = (r & 3) + 8 (r >> 2) + 4 h.  A D tile is the B operand of the next product up to the fixed permutation
// kappa(h, e) = (e & 3) + 8 (e >> 2) + 4 h of the contraction index inside a 16-wide k-step, which the packed weights carry, so
// LN output -> fc1 and GELU output -> fc2 need no cross-lane movement.
// Weights stream L2 -> LDS by global_load_lds_dwordx4 as 24-KiB chunks (= 24 fragments of 1 KiB, stored in the order and the
// lane-linear layout they are read in: every ds_read_b128 / LDS-DMA piece is 64 lanes x 16 contiguous bytes, no swizzle needed)
// through a 3-slot ring, two chunks in flight, counted s_waitcnt vmcnt + one raw s_barrier per chunk.  A chunk is one half tick:
// [12 fragments for the A waves (half the K range of an fc1 slice) | 12 for the B waves (6 of the 12 output tiles of an fc2 slice)].
#include "cs_common.h"
#include <stdlib.h>
#include <type_traits>
#include <utility>

namespace {

constexpr int PC = 384;               // hidden size
constexpr int PF = 1536;              // MLP hidden
constexpr int NT = PC / 32;           // 12 output tiles of 32 features
constexpr int KS = PC / 16;           // 24 k-steps over C
constexpr int NSL = PF / 32;          // 48 hidden slices of 32
constexpr int FRAG = 1024;            // bytes of one MFMA operand fragment (64 lanes x 16 B)
constexpr int CHUNK = 24 * FRAG;      // ring chunk
constexpr int NSLOT = 3;
constexpr int OUT_CHUNKS = KS / 2;    // out-projection: 12 chunks of two k-steps x 12 tiles
constexpr int LAG = 5;                // half ticks the fc2 side runs behind the fc1 side
constexpr int MLP_CHUNKS = 2 * NSL + LAG;  // 101 half ticks
constexpr int PAD_CHUNKS = 2;         // fetched by the last transitions, never read
constexpr int PANEL_ROWS = 128;
#ifndef CS_PANEL_NA
#define CS_PANEL_NA 0
#endif
constexpr int NA_PIECES = CS_PANEL_NA;  // of a pair's 6 weight pieces per chunk (MLP phase) the A wave issues this many, the B wave the rest
// LDS map
constexpr int LDS_RING = 0;
constexpr int LDS_R = NSLOT * CHUNK;                 // 48 KiB: 3 slots x 4 pairs x 4 KiB residual-row tiles (out-projection phase), then the
                                                     //         norm2(x) hand-off, 12 fragments per pair at a time
constexpr int LDS_HB = LDS_R + 4 * 12 * FRAG;        // 24 KiB: 3 slots x 4 pairs x 2 attention-output fragments (out-projection phase), then
                                                     //         2 slots x 4 pairs x 2 activation fragments (MLP phase)
constexpr int LDS_B1 = LDS_HB + 3 * 4 * 2 * FRAG;    // fc1 bias, fp32
constexpr int LDS_BV = LDS_B1 + PF * 4;              // bo | b2, fp32
constexpr int LDS_BYTES = LDS_BV + 2 * PC * 4;       // 153 KiB
// epilogue staging (over the ring and R, idle by then): per pair 32 rows x 768 B (half a residual row, or a whole normalised f16 row), rows
// padded to 784 B so that the B waves' 16-byte writes of 16 different rows and the A waves' row-contiguous reads are both conflict-free
constexpr int ST_ROW = 784;
constexpr int ST_PAIR = 32 * ST_ROW;

#define CS_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#define CS_SB() __builtin_amdgcn_sched_barrier(0) /* nothing crosses (any other mask let hipcc move MFMAs over the asm waits) */
#define CS_LGKM(n) do { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(n) : "memory"); CS_SB(); } while (0)

template <int V> using IC = std::integral_constant<int, V>;
template <int... Js, class F>
__device__ __forceinline__ void static_for(std::integer_sequence<int, Js...>, F&& f) {
  (f(IC<Js>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void sfor(F&& f) { static_for(std::make_integer_sequence<int, N>{}, f); }

// exact-erf GELU, the degree-7 minimax fit of Phi of cs_common.h (gelu_erf4), one value per call in plain fma form: in the A wave the
// VALU work shares the wave's issue stream with its MFMAs, so it must be schedulable instruction by instruction
__device__ __forceinline__ float gelu_erf1(float x) {
  const float c = __builtin_amdgcn_fmed3f(x, -4.2f, 4.2f);
  const float t = c * c;
  float q = fmaf(-9.6129670387e-10f, t, 8.3297297734e-08f);
  q = fmaf(q, t, -3.1398569575e-06f);
  q = fmaf(q, t, 6.8266010957e-05f);
  q = fmaf(q, t, -9.6075936689e-04f);
  q = fmaf(q, t, 9.3374518106e-03f);
  q = fmaf(q, t, -6.5599355124e-02f);
  q = fmaf(q, t, 3.9850871469e-01f);
  return x * fmaf(c, q, 0.5f);
}

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4_t pack8(const float (&v)[8]) {
  return u32x4_t{pack_h16x2(v[0], v[1]), pack_h16x2(v[2], v[3]), pack_h16x2(v[4], v[5]), pack_h16x2(v[6], v[7])};
}

// ---- inline-asm LDS access of the regions the LDS-DMA also writes or that cross waves (hipcc would order every compiler-visible LDS
//      access against the DMA with vmcnt(0)); results are retired by the counted CS_LGKM waits of the callers ----
template <int OFF>
__device__ __forceinline__ void lds_read1(unsigned addr, h16x8_t& w) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(w) : "v"(addr), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void lds_read_f4(unsigned addr, f32x4_t& w) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(w) : "v"(addr), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void lds_write16(unsigned addr, u32x4_t v) {
  asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}

// One LDS-DMA piece: 64 lanes x 16 B from `src + OFF` (this lane's address) to the wave-uniform LDS address `dst + OFF`
// (the instruction's immediate offset applies to both addresses)
template <int OFF>
__device__ __forceinline__ void dma_piece(const char* src, unsigned dst) {
  __builtin_amdgcn_global_load_lds(CS_GLOBAL_PTR(src), (__attribute__((address_space(3))) void*)(size_t)dst, 16, OFF, 0);
}

#ifdef CS_PANEL_ABLATE
// diagnostic builds only: per (block < 64, wave) six s_memtime stamps + s_memrealtime at both ends (tools/panel_ablate.py)
__device__ unsigned long long g_panel_dbg[64 * 8 * 16];
#define CS_STAMP(k) do { if (blockIdx.x < 64 && lane == 0) { \
    __builtin_amdgcn_sched_barrier(0); g_panel_dbg[(blockIdx.x * 8 + wv) * 16 + (k)] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#define CS_STAMP_RT(k) do { if (blockIdx.x < 64 && lane == 0) g_panel_dbg[(blockIdx.x * 8 + wv) * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define CS_STAMP(k) do { } while (0)
#define CS_STAMP_RT(k) do { } while (0)
#endif

// ABL: timing-only ablations (tools/panel_ablate.py builds them with -DCS_PANEL_ABLATE; results are wrong by design):
//   1 no GELU arithmetic, 2 no weight LDS-DMA after the prologue, 16 no s_barrier per chunk
template <bool OUTPROJ, int ABL = 0>
__global__ __launch_bounds__(512, 2) void cs_panel_kernel(CsPanelParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool is_a = wv < 4;
  const int pair = wv & 3;
  const int j = lane & 31, h = lane >> 5;
  const int row0 = blockIdx.x * PANEL_ROWS + pair * 32;
  const size_t row = (size_t)min(row0 + j, p.M - 1);
  constexpr int NOUT = OUTPROJ ? OUT_CHUNKS : 0;
  // Chunks below SOLO are copied by the A waves alone (6 weight pieces each; during the out-projection also the pair's residual-row tile,
  // 4 pieces, and its two attention-output fragments): the B waves then issue no vector-memory instruction before the MLP phase.
  // From SOLO on the B waves copy everything (6 pieces each, one per MFMA gap of the second half of a half tick): the A wave's
  // instruction stream (MFMA + GELU) is the longer one of a pair.
  constexpr int SOLO = NOUT + 3;
  const unsigned lds0 = (unsigned)(size_t)CS_LDS_PTR(smem);
  const unsigned lane16 = lane * 16;
  // per-lane 32-bit byte offsets (the 64-bit addresses are formed where they are used, from the kernel arguments: no address pair stays live)
  const unsigned xoff = (unsigned)row * (PC * 4) + 64 * h;  // this lane's 16 floats of tile 0 of its residual row (M * 1536 < 2^32: cs_panel_check)
  const unsigned ooff = (unsigned)row * (PC * 2) + 16 * h;  // this lane's 8 halves of k-step 0 of its attention-output row

  int c_next = 0;    // next chunk to make current
  int slot_cur = 0;  // ring slot of chunk c_next
  auto ring_issue = [&](auto ISA_, int c, int slot) {
    constexpr bool ISA = decltype(ISA_)::value;
    const unsigned dst = lds0 + LDS_RING + slot * CHUNK;
    const char* s = reinterpret_cast<const char*>(p.img) + (size_t)c * CHUNK + lane16;
    if (c < SOLO) {
      if constexpr (ISA) {
        if constexpr (!(ABL & 2)) {
          const char* s6 = s + pair * 6 * FRAG;
          const unsigned d6 = dst + pair * 6 * FRAG;
          dma_piece<0>(s6, d6); dma_piece<FRAG>(s6, d6); dma_piece<2 * FRAG>(s6, d6); dma_piece<3 * FRAG>(s6, d6);
          dma_piece<0>(s6 + 4 * FRAG, d6 + 4 * FRAG); dma_piece<FRAG>(s6 + 4 * FRAG, d6 + 4 * FRAG);
        }
        if (OUTPROJ && c < NOUT) {
          // residual-row tile c in the accumulator layout (piece q = registers 4q .. 4q+3 of every lane) and the B fragments of k-steps 2c, 2c+1
          const unsigned dx = lds0 + LDS_R + (slot * 4 + pair) * 4 * FRAG;
          const char* sx = reinterpret_cast<const char*>(p.x) + (size_t)(xoff + c * 128);
          const char* osrc = reinterpret_cast<const char*>(p.attn_o) + (size_t)ooff;
          dma_piece<0>(sx, dx);
          // (the immediate offset applies to both sides: source quarter q is 16 q bytes on, its LDS piece 1024 q)
          dma_piece<0>(sx + 16, dx + FRAG); dma_piece<0>(sx + 32, dx + 2 * FRAG); dma_piece<0>(sx + 48, dx + 3 * FRAG);
          const unsigned dof = lds0 + LDS_HB + (slot * 4 + pair) * 2 * FRAG;
          dma_piece<0>(osrc + c * 64, dof);
          dma_piece<0>(osrc + c * 64 + 32, dof + FRAG);
        }
      }
    } else if constexpr (!(ABL & 2)) {
      // (a transition outside the half-tick loops: the wave's whole share at once)
      const char* s6 = s + pair * 6 * FRAG;
      const unsigned d6 = dst + pair * 6 * FRAG;
      sfor<6>([&](auto K_) {
        constexpr int K = decltype(K_)::value;
        if constexpr (ISA == (K >= 6 - NA_PIECES)) dma_piece<(K & 3) * FRAG>(s6 + (K & 4) * FRAG, d6 + (K & 4) * FRAG);
      });
    }
  };
  // transition into chunk c_next: this wave's LDS reads of the current chunk are complete (its slot is refilled right after the barrier) and
  // its own LDS-DMA pieces of chunk c_next have landed; barrier (the same holds for every wave); then the chunk two ahead of c_next goes into
  // the slot of the chunk before c_next.  Returns this lane's LDS address of fragment 0 of chunk c_next.
  int pend_c = 0, pend_slot = 0;  // chunk / slot whose weight pieces a deferred transition left to issue_piece()
  // piece K of this wave's share of chunk pend_c (MLP phase only: A waves 2 pieces, B waves 4), issued between MFMAs instead of in a
  // burst behind the barrier: an LDS-DMA costs its wave 60+ cycles of issue, during which the partner keeps the matrix pipe busy
  auto issue_piece = [&](auto ISA_, auto K_) {
    constexpr bool ISA = decltype(ISA_)::value;
    constexpr int K = decltype(K_)::value;
    if constexpr (ABL & 2) return;
    constexpr int PIECE = ISA ? 6 - NA_PIECES + K : K;  // index among the pair's 6 pieces
    const unsigned dst = lds0 + LDS_RING + pend_slot * CHUNK + (pair * 6 + (PIECE & 4)) * FRAG;
    const char* s = reinterpret_cast<const char*>(p.img) + ((size_t)pend_c * CHUNK + (pair * 6 + (PIECE & 4)) * FRAG) + lane16;
    dma_piece<(PIECE & 3) * FRAG>(s, dst);
  };
  unsigned long long tw_drain = 0, tw_vm = 0, tw_bar = 0;  // (ABL & 64: where a transition's time goes, summed over the launch)
  auto transition = [&](auto ISA_, bool defer = false) -> unsigned {
    constexpr bool ISA = decltype(ISA_)::value;
    CS_SB();
    unsigned long long ta = 0, tb = 0, tc = 0, td = 0;
    if constexpr (ABL & 64) { asm volatile("s_memtime %0" : "=s"(ta)::"memory"); }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if constexpr (ABL & 64) { asm volatile("s_memtime %0" : "=s"(tb)::"memory"); }
    if constexpr (ISA) {
      // pieces of chunk c_next + 1 that may still be in flight
      if (c_next + 1 < NOUT) CS_VMCNT(12);
      else if (c_next + 1 < SOLO) CS_VMCNT(6);
      else CS_VMCNT(NA_PIECES);
    } else {
      CS_VMCNT(6 - NA_PIECES);
    }
    if constexpr (ABL & 64) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tc)::"memory"); }
    if constexpr (!(ABL & 16)) __builtin_amdgcn_s_barrier();
    if constexpr (ABL & 64) {
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(td)::"memory");
      tw_drain += tb - ta; tw_vm += tc - tb; tw_bar += td - tc;
    }
    asm volatile("" ::: "memory");
    int s2 = slot_cur + 2; s2 = s2 >= NSLOT ? s2 - NSLOT : s2;
    if (defer) { pend_c = c_next + 2; pend_slot = s2; }   // (only from SOLO on: the caller issues the pieces one by one)
    else ring_issue(ISA_, c_next + 2, s2);  // the image ends with PAD_CHUNKS, so the stream needs no tail case
    const unsigned base = lds0 + LDS_RING + slot_cur * CHUNK + lane16;
    ++c_next;
    slot_cur = slot_cur + 1 >= NSLOT ? 0 : slot_cur + 1;
    CS_SB();  // (register-only MFMAs must not be scheduled above the waits of this statement)
    return base;
  };
  using TA = std::true_type;
  using TB = std::false_type;
  h16x8_t w[6];  // rolling pool of weight fragments: fragment k of a chunk lives in w[k % 6], five reads ahead of its MFMA

  if (is_a) {
    // =====================================================================================================================
    // A wave: loader of the out-projection phase, then fc1 + GELU
    // =====================================================================================================================
    CS_STAMP_RT(8); CS_STAMP(0);
    // bias vectors -> LDS once per workgroup: fc1 (LN2 beta folded in), out-projection and fc2 (LayerScale folded in)
    for (int i = tid; i < PF / 4; i += 256)
      reinterpret_cast<f32x4_t*>(smem + LDS_B1)[i] = reinterpret_cast<const f32x4_t*>(p.b1)[i];
    if (tid < PC / 4) {
      reinterpret_cast<f32x4_t*>(smem + LDS_BV)[tid] = OUTPROJ ? reinterpret_cast<const f32x4_t*>(p.bo)[tid] : f32x4_t{0.f, 0.f, 0.f, 0.f};
      reinterpret_cast<f32x4_t*>(smem + LDS_BV + PC * 4)[tid] = reinterpret_cast<const f32x4_t*>(p.b2)[tid];
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    ring_issue(TA{}, 0, 0);
    ring_issue(TA{}, 1, 1);
    unsigned cur = transition(TA{});                       // chunk 0
    CS_STAMP(1);
    for (int c = 0; c < NOUT; ++c) cur = transition(TA{});  // the B waves multiply chunks 0 .. NOUT-1; `cur` ends at chunk NOUT
    CS_STAMP(2);
    // ---- norm2(x) from the partner: two halves of 12 fragments through the R region ----
    h16x8_t xf[KS];
    const unsigned r_addr = lds0 + LDS_R + pair * 12 * FRAG + lane16;
    __builtin_amdgcn_s_barrier();  // H1: first half written
    sfor<12>([&](auto F_) { lds_read1<decltype(F_)::value * FRAG>(r_addr, xf[decltype(F_)::value]); });
    CS_LGKM(0);
    __builtin_amdgcn_s_barrier();  // H2: first half read
    __builtin_amdgcn_s_barrier();  // H3: second half written
    sfor<12>([&](auto F_) { lds_read1<decltype(F_)::value * FRAG>(r_addr, xf[12 + decltype(F_)::value]); });
    CS_LGKM(0);
    CS_STAMP(3);

    // ---- fc1 ticks.  Tick t: acc(t & 1) = b1 + W1[slice t] . xf (two half ticks = two chunks of 12 fragments); the GELU of slice t-1 (the
    //      other accumulator) rides in the gaps between the MFMAs, one value per gap in 8 of the 12 gaps of a half tick, and leaves as two B
    //      fragments in hb slot (t-1) & 1 (the partner reads them LAG half ticks after the slice's first half tick) ----
    f32x16_t acE, acO;  // even / odd slices
    const unsigned bias_addr = lds0 + LDS_B1 + 64 * h;
    const unsigned hb_addr = lds0 + LDS_HB + pair * 2 * FRAG + lane16;
    float gv[8];
    f32x4_t bb[4];
    auto read_bias = [&](int t) {  // this lane's 16 hidden units of slice t: 32 t + 16 h + r
      const unsigned a = bias_addr + t * 128;
      lds_read_f4<0>(a, bb[0]); lds_read_f4<16>(a, bb[1]); lds_read_f4<32>(a, bb[2]); lds_read_f4<48>(a, bb[3]);
    };
    // One half tick: per fragment m [counted wait for it; MFMA; read of fragment m + 6 (of the next chunk from m = 6 on); a GELU value];
    // the transition to the next chunk sits before fragment 6, when all 12 fragments of this chunk are in registers or behind it in the
    // LDS queue.  P = 0: k-steps 0..11, GELU values 0..7 -> hb fragment 0; P = 1: k-steps 12..23, values 8..15 -> hb fragment 1, and the
    // next slice's bias.
    auto half_tick = [&](auto P_, auto FC1_, f32x16_t& acc, const f32x16_t& act, int hb_slot, int t_next) {
      constexpr int P = decltype(P_)::value;
      constexpr bool FC1 = decltype(FC1_)::value;
      sfor<12>([&](auto M_) {
        constexpr int M = decltype(M_)::value;
        if constexpr (M == 6) {
          cur = transition(TA{}, true);
          if constexpr (P == 1 && FC1) read_bias(t_next);  // (never leave an inline-asm read without a consumer: hipcc would reuse its
                                                           //  destination registers at once, and the LDS data would land on top of the new owner)
        }
        if constexpr (M >= 12 - NA_PIECES) issue_piece(TA{}, IC<M - (12 - NA_PIECES)>{});
        if constexpr (FC1) {
          CS_LGKM(5);
          if constexpr (P == 0 && M == 0) {
            f32x16_t b16;
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
              for (int i = 0; i < 4; ++i) b16[4 * q + i] = bb[q][i];
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[0], xf[0], b16, 0, 0, 0);
          } else {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[M % 6], xf[12 * P + M], acc, 0, 0, 0);
          }
          if constexpr (M < 6) lds_read1<(M + 6) * FRAG>(cur, w[M % 6]);
          else lds_read1<(M - 6) * FRAG>(cur, w[M % 6]);
        }
        constexpr int G = (M % 3 == 2) ? -1 : (M / 3) * 2 + (M % 3);  // gaps 0,1,3,4,6,7,9,10 carry GELU values 0..7 of this half
        if constexpr (G >= 0) gv[G] = (ABL & 1) ? act[8 * P + G] : gelu_erf1(act[8 * P + G]);
        if constexpr (M == 11) lds_write16<P * FRAG>(hb_addr + hb_slot * (4 * 2 * FRAG), pack8(gv));
        CS_SB();
      });
    };
    read_bias(0);  // (before the fragments: the counted waits retire LDS reads in order)
    sfor<6>([&](auto F_) { lds_read1<decltype(F_)::value * FRAG>(cur, w[decltype(F_)::value]); });
#pragma unroll
    for (int i = 0; i < 16; ++i) acO[i] = 0.f;  // tick 0 "activates" this (slice -1: never read by the partner)
    for (int t = 0; t < NSL; t += 2) {
      half_tick(IC<0>{}, std::true_type{}, acE, acO, 1, 0);
      half_tick(IC<1>{}, std::true_type{}, acE, acO, 1, t + 1);
      half_tick(IC<0>{}, std::true_type{}, acO, acE, 0, 0);
      half_tick(IC<1>{}, std::true_type{}, acO, acE, 0, min(t + 2, NSL - 1));
    }
    CS_STAMP(4);
    // tick NSL: only the GELU of the last slice (odd)
    CS_LGKM(0);
    half_tick(IC<0>{}, std::false_type{}, acE, acO, 1, 0);
    half_tick(IC<1>{}, std::false_type{}, acE, acO, 1, NSL - 1);
    // the B waves' last LAG - 2 half ticks
    for (int g = 0; g < LAG - 3; ++g) cur = transition(TA{});
    CS_VMCNT(0);  // the padding chunks' LDS-DMA has landed before the workgroup can end
    CS_STAMP(5);
    // ---- epilogue, A side: the partner stages its rows in LDS (three rounds: residual halves, normalised rows), this wave writes them to
    //      memory as whole 128-byte lines: a store instruction covers 4 rows x 256 contiguous bytes (16 lanes x 16 B per row) ----
    {
      const unsigned st_addr = lds0 + pair * ST_PAIR + (lane >> 4) * ST_ROW + (lane & 15) * 16;
      const int r_in = lane >> 4;
      auto drain = [&](char* gbase, size_t row_bytes, int col_bytes) {  // gbase: row0's first byte of this round's 768-byte column range
        (void)col_bytes;
#pragma unroll
        for (int rg = 0; rg < 8; ++rg) {
          f32x4_t v[3];
          lds_read_f4<0>(st_addr + rg * 4 * ST_ROW, v[0]);
          lds_read_f4<256>(st_addr + rg * 4 * ST_ROW, v[1]);
          lds_read_f4<512>(st_addr + rg * 4 * ST_ROW, v[2]);
          CS_LGKM(0);
          const int r = rg * 4 + r_in;
          if (row0 + r < p.M) {
            char* g = gbase + (size_t)r * row_bytes + (lane & 15) * 16;
            *reinterpret_cast<f32x4_t*>(g) = v[0];
            *reinterpret_cast<f32x4_t*>(g + 256) = v[1];
            *reinterpret_cast<f32x4_t*>(g + 512) = v[2];
          }
        }
      };
      char* xg = reinterpret_cast<char*>(p.x + (size_t)row0 * PC);
      __builtin_amdgcn_s_barrier();  // E0: every wave's LDS-DMA has landed (the staging area lies over the ring)
      __builtin_amdgcn_s_barrier();  // E1: residual columns 0..191 staged
      drain(xg, PC * 4, 768);
      __builtin_amdgcn_s_barrier();  // E2: read
      __builtin_amdgcn_s_barrier();  // E3: residual columns 192..383 staged
      drain(xg + 768, PC * 4, 768);
      __builtin_amdgcn_s_barrier();  // E4: read
      if (p.u_out) {
        __builtin_amdgcn_s_barrier();  // E5: normalised rows staged
        drain(reinterpret_cast<char*>(p.u_out + (size_t)row0 * PC), PC * 2, 768);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    CS_STAMP_RT(9);
#ifdef CS_PANEL_ABLATE
    if (blockIdx.x < 64 && lane == 0) { unsigned long long* d = g_panel_dbg + (blockIdx.x * 8 + wv) * 16; d[10] = tw_drain; d[11] = tw_vm; d[12] = tw_bar; }
#endif
    return;
  }

  // =======================================================================================================================
  // B wave: residual rows; out-projection, LayerNorm hand-off, fc2, epilogue.  acc2[T][r] = x[row j][32 T + 16 h + r]
  // =======================================================================================================================
  CS_STAMP_RT(8); CS_STAMP(0);
  f32x16_t acc2[NT];
  const bool row_ok = row0 + j < p.M;
  const unsigned bv_addr = lds0 + LDS_BV + 64 * h;
  unsigned cur = transition(TB{});  // chunk 0 (and: the bias vectors are in LDS)
  CS_STAMP(1);
  auto add_bias = [&](auto T_, auto INIT_, int which) {  // acc2[T] (+)= bias[32 T + 16 h + r]
    constexpr int T = decltype(T_)::value;
    f32x4_t b4[4];
    const unsigned a = bv_addr + which * (PC * 4);
    lds_read_f4<T * 128>(a, b4[0]); lds_read_f4<T * 128 + 16>(a, b4[1]); lds_read_f4<T * 128 + 32>(a, b4[2]); lds_read_f4<T * 128 + 48>(a, b4[3]);
    CS_LGKM(0);
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc2[T][4 * q + i] = (decltype(INIT_)::value ? 0.f : acc2[T][4 * q + i]) + b4[q][i];
  };
  if constexpr (OUTPROJ) {
    // ---- attention output projection.  Chunk c = k-steps 2c, 2c+1 x 12 output tiles (fragment 12 ksl + T).  Its B fragments (this lane's
    //      16 B of its attention-output row per k-step) and residual-row tile c arrive in LDS with the chunk (copied by the partner). ----
    sfor<NT>([&](auto T_) { add_bias(T_, std::true_type{}, 0); });
    const unsigned xs_addr = lds0 + LDS_R + pair * 4 * FRAG + lane16;
    const unsigned of_addr = lds0 + LDS_HB + pair * 2 * FRAG + lane16;
    h16x8_t of[2];
    lds_read1<0>(of_addr, of[0]);
    lds_read1<FRAG>(of_addr, of[1]);
    sfor<6>([&](auto F_) { lds_read1<decltype(F_)::value * FRAG>(cur, w[decltype(F_)::value]); });
    int slot_x = 0;  // ring slot of the current chunk (its residual tile / attention-output fragments share the index)
    sfor<OUT_CHUNKS>([&](auto C_) {
      constexpr int C = decltype(C_)::value;
      f32x4_t xs[2];
      auto add_x = [&](auto Q0_) {  // registers 4 Q0 .. 4 Q0 + 7 of tile C
        constexpr int Q0 = decltype(Q0_)::value;
        CS_LGKM(0);
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc2[C][4 * (Q0 + q) + i] += xs[q][i];
        CS_SB();
      };
      sfor<24>([&](auto M_) {
        constexpr int M = decltype(M_)::value;
        // the chunk's residual-row tile C, two quarters at a time (its LDS slot is refilled after the transition below)
        if constexpr (M == 12) { const unsigned a = xs_addr + slot_x * (4 * 4 * FRAG); lds_read_f4<0>(a, xs[0]); lds_read_f4<FRAG>(a, xs[1]); }
        if constexpr (M == 14) add_x(IC<0>{});
        if constexpr (M == 15) { const unsigned a = xs_addr + slot_x * (4 * 4 * FRAG); lds_read_f4<2 * FRAG>(a, xs[0]); lds_read_f4<3 * FRAG>(a, xs[1]); }
        if constexpr (M == 17) add_x(IC<2>{});
        if constexpr (M == 18) {
          cur = transition(TB{});
          slot_x = slot_x + 1 >= NSLOT ? 0 : slot_x + 1;
          if constexpr (C + 1 < OUT_CHUNKS) lds_read1<0>(of_addr + slot_x * (4 * 2 * FRAG), of[0]);  // k-step 0's fragment is free from M = 12 on
        }
        if constexpr (M == 0 && C > 0) lds_read1<FRAG>(of_addr + slot_x * (4 * 2 * FRAG), of[1]);  // (needed from M = 12 on)
        CS_LGKM(5);
        constexpr int T = M % 12;
        acc2[T] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[M % 6], of[M / 12], acc2[T], 0, 0, 0);
        if constexpr (M < 18) lds_read1<(M + 6) * FRAG>(cur, w[M % 6]);
        else lds_read1<(M - 18) * FRAG>(cur, w[M % 6]);  // (after the last chunk: the A half of chunk NOUT, unused)
        CS_SB();
      });
    });
    CS_LGKM(0);
  } else {
    // no out-projection (tests): x straight from memory
    const float* xr = p.x + row * PC + 16 * h;
#pragma unroll
    for (int T = 0; T < NT; ++T)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4_t v = *reinterpret_cast<const f32x4_t*>(xr + 32 * T + 4 * q);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc2[T][4 * q + i] = v[i];
      }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  CS_STAMP(2);

  // ---- LayerNorm statistics of the rows held in acc2 (two-pass, fp32, in registers; a row lives in lanes j and j + 32) ----
  auto row_stats = [&](float& mean, float& rstd) {
    float s = 0.f;
#pragma unroll
    for (int T = 0; T < NT; ++T)
#pragma unroll
      for (int r = 0; r < 16; r += 4) s += (acc2[T][r] + acc2[T][r + 1]) + (acc2[T][r + 2] + acc2[T][r + 3]);
    s += __shfl_xor(s, 32, 64);
    mean = s * (1.0f / PC);
    float q = 0.f;
#pragma unroll
    for (int T = 0; T < NT; ++T)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float d = acc2[T][r] - mean;
        q = fmaf(d, d, q);
      }
    q += __shfl_xor(q, 32, 64);
    rstd = 1.0f / sqrtf(q * (1.0f / PC) + p.eps);
  };
  {
    // norm2 -> the partner's fc1 B fragments: k-step 2T + s is registers 8s .. 8s+7 of tile T
    float mean, rstd;
    row_stats(mean, rstd);
    const float nb = -mean * rstd;
    const unsigned r_addr = lds0 + LDS_R + pair * 12 * FRAG + lane16;
    auto put = [&](auto T_, auto S_) {
      constexpr int T = decltype(T_)::value, S = decltype(S_)::value;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = fmaf(acc2[T][8 * S + e], rstd, nb);  // (not (x - mean) * rstd: the 192 differences of the variance pass would be kept alive)
      lds_write16<((2 * T + S) % 12) * FRAG>(r_addr, pack8(v));
    };
    sfor<6>([&](auto T_) { put(T_, IC<0>{}); put(T_, IC<1>{}); });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // H1
    __builtin_amdgcn_s_barrier();  // H2
    sfor<6>([&](auto T_) { put(IC<6 + decltype(T_)::value>{}, IC<0>{}); put(IC<6 + decltype(T_)::value>{}, IC<1>{}); });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // H3
  }
  // the fc2 accumulators start at residual + bias: the residual add and the bias add are free
  sfor<NT>([&](auto T_) { add_bias(T_, std::false_type{}, 1); });
  CS_STAMP(3);

  // ---- fc2, LAG half ticks behind fc1.  Half tick g >= LAG: fragments 12..23 of its chunk = tile f of k-step p = (g - LAG) & 1 of hidden
  //      slice (g - LAG) / 2, whose activations are in hb slot ((g - LAG) / 2) & 1 ----
  for (int g = 0; g < LAG - 1; ++g) cur = transition(TB{});  // chunks NOUT+1 .. NOUT+LAG-1 become current
  const unsigned hb_addr = lds0 + LDS_HB + pair * 2 * FRAG + lane16;
  h16x8_t hb[2];
  // Half p of a slice multiplies k-step p (activation fragment hb[p]) into all 12 tiles, so the other fragment register is free to be
  // refilled meanwhile: after the transition inside half 1 comes fragment 0 of the NEXT slice (the partner wrote it before that barrier),
  // after the one inside half 0 fragment 1 of the current slice.
  auto hb_slot_addr = [&](int g_half) { return hb_addr + (((g_half - LAG) >> 1) & 1) * (4 * 2 * FRAG); };
  auto fc2_half = [&](auto P_, bool more) {
    constexpr int P = decltype(P_)::value;
    sfor<12>([&](auto M_) {
      constexpr int M = decltype(M_)::value;
      if constexpr (M == 6) {
        if (more) {
          cur = transition(TB{}, true);
          const int g_new = c_next - 1 - NOUT;  // the half tick that just became current
          if constexpr (P == 1) lds_read1<0>(hb_slot_addr(g_new), hb[0]);
          else lds_read1<FRAG>(hb_slot_addr(g_new), hb[1]);
        }
      }
      if constexpr (M >= 6 && M < 12 - NA_PIECES) {
        if (more) issue_piece(TB{}, IC<M - 6>{});
      }
      CS_LGKM(5);
      acc2[M] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[M % 6], hb[P], acc2[M], 0, 0, 0);
      if constexpr (M < 6) lds_read1<(12 + M + 6) * FRAG>(cur, w[M % 6]);
      else lds_read1<(12 + M - 6) * FRAG>(cur, w[M % 6]);  // (after the last half tick: padding)
      CS_SB();
    });
  };
  cur = transition(TB{});  // chunk NOUT + LAG: slice 0, half 0
  lds_read1<0>(hb_addr, hb[0]);
  sfor<6>([&](auto F_) { lds_read1<(12 + decltype(F_)::value) * FRAG>(cur, w[decltype(F_)::value]); });
  for (int t = 0; t < NSL - 1; ++t) {
    fc2_half(IC<0>{}, true);
    fc2_half(IC<1>{}, true);
  }
  fc2_half(IC<0>{}, true);
  fc2_half(IC<1>{}, false);
  CS_LGKM(0);
  CS_VMCNT(0);  // the padding chunks' LDS-DMA has landed before the workgroup can end
  CS_STAMP(4);

  // ---- epilogue, B side: rows go to memory through LDS and the partner (a lane holds 64 B of a row per tile: stored directly, every store
  //      instruction would touch 32 rows) ----
  {
    const unsigned st_addr = lds0 + pair * ST_PAIR + j * ST_ROW + 64 * h;
    auto stage_x = [&](auto HALF_) {
      constexpr int HALF = decltype(HALF_)::value;
      sfor<6>([&](auto T_) {
        constexpr int T = 6 * HALF + decltype(T_)::value;
        sfor<4>([&](auto Q_) {
          constexpr int Q = decltype(Q_)::value;
          lds_write16<decltype(T_)::value * 128 + 16 * Q>(st_addr, __builtin_bit_cast(u32x4_t, f32x4_t{acc2[T][4 * Q], acc2[T][4 * Q + 1], acc2[T][4 * Q + 2], acc2[T][4 * Q + 3]}));
        });
      });
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    __builtin_amdgcn_s_barrier();  // E0
    stage_x(IC<0>{});
    __builtin_amdgcn_s_barrier();  // E1
    float mean = 0.f, rstd = 0.f;
    if (p.u_out) row_stats(mean, rstd);  // (under the partner's stores)
    __builtin_amdgcn_s_barrier();  // E2
    stage_x(IC<1>{});
    __builtin_amdgcn_s_barrier();  // E3
    __builtin_amdgcn_s_barrier();  // E4
    if (p.u_out) {
      const float nb = -mean * rstd;
      const unsigned su = lds0 + pair * ST_PAIR + j * ST_ROW + 32 * h;
      sfor<NT>([&](auto T_) {
        constexpr int T = decltype(T_)::value;
        sfor<2>([&](auto Q_) {
          constexpr int Q = decltype(Q_)::value;
          float v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaf(acc2[T][8 * Q + e], rstd, nb);
          lds_write16<T * 64 + 16 * Q>(su, pack8(v));
        });
      });
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // E5
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  CS_STAMP(5); CS_STAMP_RT(9);
#ifdef CS_PANEL_ABLATE
  if (blockIdx.x < 64 && lane == 0) { unsigned long long* d = g_panel_dbg + (blockIdx.x * 8 + wv) * 16; d[10] = tw_drain; d[11] = tw_vm; d[12] = tw_bar; }
#endif
}

// ---- weight image.  One thread per 16-byte fragment element (8 f16): [chunk][fragment 0..23][lane 0..63].  Lane (i = lane & 31,
//      h = lane >> 5) of an A-operand fragment holds MFMA row i of its 32-feature tile and 8 contraction indices of its 16-wide k-step.
//      MFMA row i of a D tile is register r = (i & 3) + 4 (i >> 3) of lane half (i >> 2) & 1, and the kernel keeps feature
//      32 T + 16 h + r there, so row i carries output feature  perm(i) = 16 ((i >> 2) & 1) + (i & 3) + 4 (i >> 3)  of its tile.
//      Contraction index of element e:  natural  16 ks + 8 h + e                 (out-projection: B fragments come from memory)
//                                       tiled    32 (ks >> 1) + 16 h + 8 (ks & 1) + e   (fc1 / fc2: B fragments are accumulator tiles)
//      Chunks: [12 out-projection chunks c: fragment 12 ksl + T = (k-step 2c + ksl, tile T)]
//              [101 MLP half ticks g: fragments 0..11  = fc1 slice g/2, k-step 12 (g&1) + f                                  (g < 96)
//                                     fragments 12..23 = fc2 slice (g-5)/2, tile f', k-step (g-5) & 1                         (g >= 5)]
//              [2 padding chunks] ----
__device__ __forceinline__ int panel_perm(int i) { return 16 * ((i >> 2) & 1) + (i & 3) + 4 * (i >> 3); }
__global__ __launch_bounds__(256) void cs_panel_pack_kernel(const float* __restrict__ wo, const float* __restrict__ ls1,
                                                            const float* __restrict__ w1, const float* __restrict__ g2,
                                                            const float* __restrict__ w2, const float* __restrict__ ls2,
                                                            h16_t* __restrict__ img) {
  const int nout = wo ? OUT_CHUNKS : 0;
  const int nch = nout + MLP_CHUNKS;
  const int gi = blockIdx.x * blockDim.x + threadIdx.x;
  if (gi >= (nch + PAD_CHUNKS) * (CHUNK / 16)) return;
  const int c = gi / (CHUNK / 16);
  const int within = gi - c * (CHUNK / 16);
  const int f = within >> 6, lane = within & 63;
  const int i = lane & 31, h = lane >> 5;
  float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c < nout) {
    const int ks = 2 * c + f / 12, T = f % 12;
    const int rowi = 32 * T + panel_perm(i);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = wo[(size_t)rowi * PC + 16 * ks + 8 * h + e] * (ls1 ? ls1[rowi] : 1.f);
  } else if (c < nch) {
    const int g = c - nout;
    if (f < 12) {
      if (g < 2 * NSL) {
        const int t = g >> 1, ks = 12 * (g & 1) + f;
        const int rowi = 32 * t + panel_perm(i);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int col = 32 * (ks >> 1) + 16 * h + 8 * (ks & 1) + e;
          v[e] = w1[(size_t)rowi * PC + col] * (g2 ? g2[col] : 1.f);
        }
      }
    } else if (g >= LAG) {
      const int tt = (g - LAG) >> 1, fp = f - 12;
      const int T = fp, s = (g - LAG) & 1;
      const int rowi = 32 * T + panel_perm(i);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int col = 32 * tt + 16 * h + 8 * s + e;
        v[e] = w2[(size_t)rowi * PF + col] * (ls2 ? ls2[rowi] : 1.f);
      }
    }
  }
  const uint4 o = {pack_h16x2(v[0], v[1]), pack_h16x2(v[2], v[3]), pack_h16x2(v[4], v[5]), pack_h16x2(v[6], v[7])};
  reinterpret_cast<uint4*>(img)[gi] = o;
}

}  // namespace

extern "C" {

#ifdef CS_PANEL_ABLATE
int cs_panel_debug_read(unsigned long long* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_panel_dbg), sizeof(g_panel_dbg)); }
#endif

int cs_panel_supported(int C, int mlp_ratio) { return C == PC && mlp_ratio * C == PF; }
size_t cs_panel_image_bytes(int with_outproj) { return (size_t)((with_outproj ? OUT_CHUNKS : 0) + MLP_CHUNKS + PAD_CHUNKS) * CHUNK; }

hipError_t cs_panel_pack_launch(const float* wo, const float* ls1, const float* w1, const float* g2, const float* w2, const float* ls2,
                                h16_t* img, hipStream_t st) {
  const int total = ((wo ? OUT_CHUNKS : 0) + MLP_CHUNKS + PAD_CHUNKS) * (CHUNK / 16);
  hipLaunchKernelGGL(cs_panel_pack_kernel, dim3((total + 255) / 256), dim3(256), 0, st, wo, ls1, w1, g2, w2, ls2, img);
  return hipGetLastError();
}

const char* cs_panel_check(const CsPanelParams* p) {
  if (!p->x || !p->img || !p->b1 || !p->b2) return "panel: null operand";
  if (p->attn_o && !p->bo) return "panel: the out-projection needs its bias";
  if (p->M <= 0) return "panel: empty shape";
  if ((long long)p->M * PC * 4 >= (1ll << 32)) return "panel: too many rows for 32-bit byte offsets";
  if (((uintptr_t)p->x | (uintptr_t)p->img | (uintptr_t)p->b1 | (uintptr_t)p->b2 | (uintptr_t)p->attn_o | (uintptr_t)p->u_out | (uintptr_t)p->bo) & 15)
    return "panel: operands must be 16-byte aligned";
  return nullptr;
}

hipError_t cs_panel_launch(const CsPanelParams* p, hipStream_t st) {
  static bool attr_done[16][2] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return hipErrorInvalidDevice;
  const int v = p->attn_o ? 1 : 0;
  if (!attr_done[dev][v]) {
    const void* fn = v ? reinterpret_cast<const void*>(cs_panel_kernel<true>) : reinterpret_cast<const void*>(cs_panel_kernel<false>);
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) return e;
    attr_done[dev][v] = true;
  }
  const int grid = (p->M + PANEL_ROWS - 1) / PANEL_ROWS;
#ifdef CS_PANEL_ABLATE
  if (const char* e = getenv("CS_PANEL_ABL")) {
    const int abl = atoi(e);
#define CS_ABL_CASE(N) if (abl == N) { \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(cs_panel_kernel<true, N>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES); \
      hipLaunchKernelGGL((cs_panel_kernel<true, N>), dim3(grid), dim3(512), LDS_BYTES, st, *p); return hipGetLastError(); }
    CS_ABL_CASE(1) CS_ABL_CASE(2) CS_ABL_CASE(3) CS_ABL_CASE(16) CS_ABL_CASE(18) CS_ABL_CASE(64)
#undef CS_ABL_CASE
  }
#endif
  if (v) hipLaunchKernelGGL(cs_panel_kernel<true>, dim3(grid), dim3(512), LDS_BYTES, st, *p);
  else hipLaunchKernelGGL(cs_panel_kernel<false>, dim3(grid), dim3(512), LDS_BYTES, st, *p);
  return hipGetLastError();
}

}  // extern "C"
