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
#include "panel_shared.h"
#include <atomic>
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
constexpr int CHUNK = 24 * FRAG;      // out-projection unit: two k-steps x 12 tiles
constexpr int TICK = 48 * FRAG;       // MLP unit: one tick = [24 fragments of an fc1 slice (A waves) | 24 of an fc2 slice (B waves)]
constexpr int OUT_CHUNKS = KS / 2;    // 12 out-projection units
constexpr int LAGT = 2;               // ticks the fc2 side runs behind the fc1 side
constexpr int NTICK = NSL + LAGT;     // 50 MLP units
constexpr int PAD_TICKS = 1;          // fetched by the last transition, never read
constexpr int PANEL_ROWS = 128;
// LDS map.  Out-projection phase: 3 ring slots of 24 KiB, the residual-row tiles and attention-output fragments that travel with them.
// MLP phase: 2 ring slots of 48 KiB (one barrier per tick) over the ring and the start of the then-idle tile region, activation slots.
constexpr int LDS_RING = 0;
constexpr int LDS_X = 3 * CHUNK;                     // 72 K: 3 slots x 4 pairs x one 6-KiB record [4 KiB residual-row tile | 2 attention-output fragments]
constexpr int XREC = 6 * FRAG;                       //       (one record, one address register on the B side: r4)
constexpr int LDS_OF = LDS_X + 3 * 4 * 4 * FRAG;     // 120 K (the MLP phase's hand-off areas start here; the out-projection records end at 144 K)
constexpr int LDS_XF = 2 * TICK;                     // 96 K: norm2(x) hand-off, 4 pairs x 12 fragments at a time (48 KiB; ends at 144 K)
// 120 K, MLP phase: the pairs' hand-off areas of 6 KiB each (one base register, everything else immediates):
//   [activation fragment of fc2 k-step 0, slot 0 | slot 1 (1 KiB each: written by the A wave) | the slice's other 8 pre-activations per lane as
//    fp32, slot 0 | slot 1 (2 KiB each: the B wave activates them)]
constexpr int LDS_HB = LDS_OF;
constexpr int HB_PAIR = 6 * FRAG;
constexpr int HB_RAW = 2 * FRAG;
constexpr int LDS_B1 = LDS_OF + 3 * 4 * 2 * FRAG;    // 144 K: fc1 bias, fp32
constexpr int LDS_BV = LDS_B1 + PF * 4;              // bo | b2, fp32
constexpr int LDS_BYTES = LDS_BV + 2 * PC * 4;       // 153 KiB
// epilogue staging (over the ring, idle by then): per pair 32 rows x 768 B (half a residual row, or a whole normalised f16 row), rows
// padded to 784 B so that the B waves' 16-byte writes of 16 different rows and the A waves' row-contiguous reads are both conflict-free
constexpr int ST_ROW = 784;
constexpr int ST_PAIR = 32 * ST_ROW;

#ifdef CS_PANEL_ABLATE
// diagnostic builds only: per (block < 64, wave) six s_memtime stamps + s_memrealtime at both ends (tools/panel_ablate.py)
__device__ unsigned long long g_panel_dbg[64 * 8 * 16];
#define CS_STAMP(k) do { if (blockIdx.x < 64 && lane == 0) { \
    __builtin_amdgcn_sched_barrier(0); g_panel_dbg[(blockIdx.x * 8 + wv) * 16 + (k)] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#define CS_STAMP_RT(k) do { if (blockIdx.x < 64 && lane == 0) g_panel_dbg[(blockIdx.x * 8 + wv) * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
// arrival / release times at the MLP phase's unit boundaries v = 40 .. 55 (ticks 20 .. 27) of blocks 0 .. 3: who is late at the barriers?
__device__ unsigned long long g_panel_bar[4 * 8 * 16 * 2];
#define CS_BAR_STAMP(v, k) do { if (blockIdx.x < 4 && (v) >= 40 && (v) < 56 && (threadIdx.x & 63) == 0) { \
    __builtin_amdgcn_sched_barrier(0); g_panel_bar[((blockIdx.x * 8 + wv) * 16 + ((v) - 40)) * 2 + (k)] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define CS_STAMP(k) do { } while (0)
#define CS_STAMP_RT(k) do { } while (0)
#define CS_BAR_STAMP(v, k) do { } while (0)
#endif

// ABL: timing-only ablations (tools/panel_ablate.py builds them with -DCS_PANEL_ABLATE; results are wrong by design):
//   1 no GELU arithmetic, 2 no weight LDS-DMA after the first two MLP units, 16 no s_barrier per unit, 64 transition-time accounting
template <bool OUTPROJ, int ABL = 0, bool BF = false>
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
  constexpr size_t IMG_MLP = (size_t)NOUT * CHUNK;  // byte offset of MLP unit 0 in the image
  const unsigned lds0 = (unsigned)(size_t)CS_LDS_PTR(smem);
  const unsigned lane16 = lane * 16;
  // per-lane 32-bit byte offsets (the 64-bit addresses are formed where they are used, from the kernel arguments: no address pair stays live)
  const unsigned xoff = (unsigned)row * (PC * 4) + 64 * h;  // this lane's 16 floats of tile 0 of its residual row (M * 1536 < 2^32: cs_panel_check)
  const unsigned ooff = (unsigned)row * (PC * 2) + 16 * h;  // this lane's 8 halves of k-step 0 of its attention-output row

  // Units: u < NOUT an out-projection chunk (24 KiB, ring slot u % 3, fetched two units ahead); u = NOUT + t MLP tick t (48 KiB, ring slot
  // t % 2, fetched one unit ahead).  The A waves copy the out-projection units (with their pair's residual-row tile and attention-output
  // fragments) and MLP ticks 0 and 1 -- the B waves issue no vector-memory instruction before the MLP phase --, the B waves every later
  // tick, one 1-KiB piece per MFMA gap.
  int u_next = 0;  // next unit to make current
  auto out_unit_issue = [&](int c) {  // A waves: out-projection chunk c with its tile and fragments
    const int slot = c % 3;
    const unsigned dst = lds0 + LDS_RING + slot * CHUNK + pair * 6 * FRAG;
    const char* s6 = reinterpret_cast<const char*>(p.img) + (size_t)c * CHUNK + pair * 6 * FRAG + lane16;
    dma_piece<0>(s6, dst); dma_piece<FRAG>(s6, dst); dma_piece<2 * FRAG>(s6, dst); dma_piece<3 * FRAG>(s6, dst);
    dma_piece<0>(s6 + 4 * FRAG, dst + 4 * FRAG); dma_piece<FRAG>(s6 + 4 * FRAG, dst + 4 * FRAG);
    // residual-row tile c in the accumulator layout (piece q = registers 4q .. 4q+3 of every lane) and the B fragments of k-steps 2c, 2c+1
    const unsigned dx = lds0 + LDS_X + (slot * 4 + pair) * XREC;
    const char* sx = reinterpret_cast<const char*>(p.x) + (size_t)(xoff + c * 128);
    dma_piece<0>(sx, dx); dma_piece<0>(sx + 16, dx + FRAG); dma_piece<0>(sx + 32, dx + 2 * FRAG); dma_piece<0>(sx + 48, dx + 3 * FRAG);
    const unsigned dof = dx + 4 * FRAG;
    const char* so = reinterpret_cast<const char*>(p.attn_o) + (size_t)(ooff + c * 64);
    dma_piece<0>(so, dof); dma_piece<0>(so + 32, dof + FRAG);
  };
  // MLP weight stream (r4): the image is cut into HALF-tick units v = 2 t + hf of 24 KiB, [12 fragments of the A waves | 12 of the B waves]
  // (fc1 slice t, k-steps 12 hf .. 12 hf + 11 | fc2 slice t - 2, k-step hf, tiles 0 .. 11); unit v lives at ring offset (v & 3) * 24 KiB, i.e.
  // tick t still occupies the 48-KiB half (t & 1) of the ring.  Every A wave copies six 1-KiB pieces of a unit (pair * 6 + K).
  int pend_u = 0;          // unit whose pieces issue_piece() copies
  bool pend = false;       // ... if any
  auto issue_piece = [&](auto K_) {
    constexpr int K = decltype(K_)::value;
    const unsigned dst = lds0 + LDS_RING + (pend_u & 3) * CHUNK + (pair * 6 + (K & ~3)) * FRAG;
    const char* s = reinterpret_cast<const char*>(p.img) + (IMG_MLP + (size_t)pend_u * CHUNK + (pair * 6 + (K & ~3)) * FRAG) + lane16;
    dma_piece<(K & 3) * FRAG>(s, dst);
  };
  auto tick_issue_all = [&](int t) {  // a wave's whole share of tick t at once (outside the tick loops): unit 2t first, then 2t + 1
    sfor<12>([&](auto K_) {
      constexpr int K = decltype(K_)::value % 6, HF = decltype(K_)::value / 6;
      const unsigned dst = lds0 + LDS_RING + (t & 1) * TICK + HF * CHUNK + (pair * 6 + (K & ~3)) * FRAG;
      const char* s = reinterpret_cast<const char*>(p.img) + (IMG_MLP + (size_t)t * TICK + HF * CHUNK + (pair * 6 + (K & ~3)) * FRAG) + lane16;
      dma_piece<(K & 3) * FRAG>(s, dst);
    });
  };
  unsigned long long tw_drain = 0, tw_vm = 0, tw_bar = 0;  // (ABL & 64: where a transition's time goes, summed over the launch)
  // transition into unit u_next: this wave's LDS reads of the current unit are complete (its slot may be refilled right after the barrier)
  // and its own LDS-DMA pieces of unit u_next have landed; barrier (the same holds for every wave); then the next fetch is started (or, with
  // `defer`, left to issue_piece() between the caller's MFMAs).  Returns this lane's LDS address of fragment 0 of unit u_next.
  auto transition = [&](auto ISA_) -> unsigned {
    constexpr bool ISA = decltype(ISA_)::value;
    CS_SB();
    unsigned long long ta = 0, tb = 0, tc = 0, td = 0;
    if constexpr (ABL & 64) { asm volatile("s_memtime %0" : "=s"(ta)::"memory"); }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if constexpr (ABL & 64) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tb)::"memory"); }
    if constexpr (ISA) {
      if (u_next + 1 < NOUT) CS_VMCNT(12);  // the out-projection unit after u_next may still be in flight
      else CS_VMCNT(0);
    } else {
      CS_VMCNT(0);
    }
    if constexpr (ABL & 64) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tc)::"memory"); }
    if constexpr (!(ABL & 16)) __builtin_amdgcn_s_barrier();
    if constexpr (ABL & 64) {
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(td)::"memory");
      tw_drain += tb - ta; tw_vm += tc - tb; tw_bar += td - tc;
    }
    asm volatile("" ::: "memory");
    if (u_next + 2 < NOUT) {
      if constexpr (ISA && OUTPROJ) out_unit_issue(u_next + 2);
    } else if (u_next + 1 >= NOUT) {
      const int t = u_next + 1 - NOUT;  // the tick to fetch now (the image ends with PAD_TICKS: no tail case)
      // MLP ticks 0 and 1 (units 0..3) are fetched here, whole, by the A waves; every later unit at the MLP phase's own unit boundaries
      if constexpr (ISA) {
        if (t < 2) tick_issue_all(t);
      }
    }
    const unsigned base = lds0 + LDS_RING + lane16 + (u_next < NOUT ? (u_next % 3) * CHUNK : ((u_next - NOUT) & 1) * TICK);
    ++u_next;
    CS_SB();  // (register-only MFMAs must not be scheduled above the waits of this statement)
    return base;
  };
  using TA = std::true_type;
  using TB = std::false_type;
  h16x8_t w[6];  // rolling pool of weight fragments: fragment k of a unit lives in w[k % 6], five reads ahead of its MFMA

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
    if constexpr (OUTPROJ) { out_unit_issue(0); out_unit_issue(1); }
    else tick_issue_all(0);
    unsigned cur = transition(TA{});              // unit 0
    CS_STAMP(1);
    for (int c = 0; c < NOUT; ++c) cur = transition(TA{});  // the B waves multiply units 0 .. NOUT-1; `cur` ends at MLP tick 0
    CS_STAMP(2);
    // ---- norm2(x) from the partner: two halves of 12 fragments through the hand-off region ----
    h16x8_t xf[KS];
    const unsigned r_addr = lds0 + LDS_XF + pair * 12 * FRAG + lane16;
    __builtin_amdgcn_s_barrier();  // H1: first half written
    sfor<12>([&](auto F_) { lds_read1<decltype(F_)::value * FRAG>(r_addr, xf[decltype(F_)::value]); });
    CS_LGKM(0);
    __builtin_amdgcn_s_barrier();  // H2: first half read
    __builtin_amdgcn_s_barrier();  // H3: second half written
    sfor<12>([&](auto F_) { lds_read1<decltype(F_)::value * FRAG>(r_addr, xf[12 + decltype(F_)::value]); });
    CS_LGKM(0);
    CS_STAMP(3);

    // ---- fc1 ticks.  Tick t: acc(t & 1) = b1 + W1[slice t] . xf (24 fragments of the tick's unit); the GELU of slice t-1 (the other
    //      accumulator) rides in the gaps between the MFMAs, one value per gap in the first 16 gaps, and leaves as two B fragments in hb slot
    //      (t-1) & 1 before the tick's transition (gap 18): the partner multiplies them from tick t+1 on ----
    f32x16_t acE, acO;  // even / odd slices
    const unsigned bias_addr = lds0 + LDS_B1 + 64 * h;
    f32x4_t bb[4];
    auto read_bias = [&](int t) {  // this lane's 16 hidden units of slice t: 32 t + 16 h + r
      const unsigned a = bias_addr + t * 128;
      lds_read_f4<0>(a, bb[0]); lds_read_f4<16>(a, bb[1]); lds_read_f4<32>(a, bb[2]); lds_read_f4<48>(a, bb[3]);
    };
    // One tick: per fragment m [counted wait for it; MFMA; read of fragment m + 6 (of the next unit from m = 18 on); a GELU value]; the
    // transition to the next unit sits before fragment 18, when all 24 fragments of this unit are in registers or behind it in the LDS queue.
    // hb_write = false for tick 0: its "previous slice" does not exist, and hb slot 1 lies inside the norm2 hand-off region (LDS_XF), which
    // the sibling pairs' A waves may still be reading behind H3 -- nothing may be written there before the first tick's transition
    // GELU split (r4): the VALU work of a slice is shared by the two waves of the pair -- a single wave issues a vector instruction every 4-8
    // cycles, and with all 16 values per lane on the A wave its instruction stream (24 MFMAs + ~190 VALU) set the tick (2 600 cycles against
    // 1 536 of MFMA) while the B wave idled at the barrier.  The A wave now activates registers 0..7 of the previous slice (= the B fragment of
    // fc2 k-step 0, one half value per MFMA gap) and hands registers 8..15 over as they are (fp32, 2 KiB per pair) at the start of the tick;
    // the B wave activates those in the gaps of its first 12 MFMAs (which use k-step 0) into the fragment of k-step 1.  Same arithmetic on the
    // same values: bit-identical to the one-wave form.
    // Unit boundaries of the MLP phase (r4).  Round 3 had ONE barrier per tick with an LDS drain in front of it (the slot of the tick just
    // read was refilled right behind the barrier): all eight waves stopped issuing MFMAs, emptied their prefetch queues, met, and refilled
    // them -- the matrix pipe idled a quarter of every tick.  Now a unit is HALF a tick and is released LATE: the barrier in front of the first
    // read of unit v + 1 (gap 6 / gap 18) only says "unit v + 1 has landed and everybody is done with unit v - 1" -- which the counted LDS waits
    // of the gaps in between already guarantee -- so nobody drains anything, the reads in flight across the barrier target units v and v + 1,
    // and the loader refills the slot of unit v - 1 with unit v + 3 (one 1-KiB piece per gap, 6 per A wave) two boundaries ahead of its use.
    const unsigned hb_base = lds0 + LDS_HB + pair * HB_PAIR + lane16;
    const unsigned ring = lds0 + LDS_RING + lane16;   // + (t & 1) * TICK: this lane's address of tick t's fragment 0
    unsigned xa[4] = {0u, 0u, 0u, 0u}, xb[4];  // (fp16 mode) the previous slice as packed halves: this wave's pairs, the partner's
    PkGelu pg;
    const PkGeluK kk = pk_gelu_consts();
    auto boundary = [&](auto FC1_, int v) {  // in front of the first read of unit v + 1
      CS_SB();
      if constexpr (!decltype(FC1_)::value) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }  // (last tick: no counted waits retire its hand-off writes)
      CS_BAR_STAMP(v, 0);
      CS_VMCNT(6);  // this wave's pieces of unit v + 1 have landed (those of unit v + 2, issued one boundary ago, may be in flight)
      if constexpr (!(ABL & 16)) __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      CS_BAR_STAMP(v, 1);
      pend_u = v + 3;
      pend = v >= 1 && v + 3 < 2 * NTICK && !(ABL & 2);
      CS_SB();
    };
    auto tick = [&](auto FC1_, f32x16_t& acc, const f32x16_t& act, int hb_slot, int t, int t_next, bool hb_write) {
      constexpr bool FC1 = decltype(FC1_)::value;
      sfor<24>([&](auto M_) {
        constexpr int M = decltype(M_)::value;
        if constexpr (M == 6) boundary(FC1_, 2 * t);
        if constexpr (M == 18) {
          boundary(FC1_, 2 * t + 1);
          cur = ring + ((t + 1) & 1) * TICK;
          if constexpr (FC1) read_bias(t_next);  // (never leave an inline-asm read without a consumer: hipcc would reuse its destination
                                                 //  registers at once, and the LDS data would land on top of the new owner)
        }
        if constexpr ((M >= 6 && M < 12) || M >= 18) {
          if (pend) issue_piece(IC<(M >= 18 ? M - 18 : M - 6)>{});
        }
        // The raw half of the previous slice leaves in gap 3, NOT in gap 0: `act` was completed by the last MFMA of the previous tick, and an
        // inline-asm consumer gets none of the wait states hipcc inserts between an MFMA and a reader of its result (16 passes = 64 cycles
        // for 32x32x16).  In gap 0 the ds_write read the accumulator before MFMAs 22 / 23 had landed (r4: every even slice lost k-steps 22 and
        // 23 in registers 8..15, non-deterministically).  Three MFMAs of this tick (>= 32 cycles of matrix pipe each, issued in order behind
        // that MFMA) lie in between now; the MFMA-free last tick waits explicitly.
        // The previous slice is rounded to half here -- compiler-visible conversions, so hipcc places the MFMA -> VALU wait states itself; gap 1
        // is two MFMAs behind the instruction that completed `act` (the MFMA-free last tick waits explicitly).  The partner's half leaves in
        // gap 5: fp16 mode as 4 packed words (1 KiB per pair), bf16 mode as the fp32 accumulators (2 KiB; its relu must keep fp32's range).
        if constexpr (M == 1) {
          if constexpr (!FC1) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
          for (int i = 0; i < 4; ++i) xa[i] = pack_h16x2(act[2 * i], act[2 * i + 1]);
        }
        if constexpr (!BF) {
          if constexpr (M == 4) {
#pragma unroll
            for (int i = 0; i < 4; ++i) xb[i] = pack_h16x2(act[8 + 2 * i], act[9 + 2 * i]);
          }
          if constexpr (M == 5) {
            if (hb_write) lds_write16<HB_RAW>(hb_base + hb_slot * (2 * FRAG), u32x4_t{xb[0], xb[1], xb[2], xb[3]});
          }
        } else {
          // (an inline-asm consumer of an MFMA result gets none of hipcc's wait states: five MFMAs of this tick lie between the instruction
          //  that completed `act` and these stores, r4's finding at gap 0; the compiler-visible conversions of gap 1 carry the explicit ones)
          if constexpr (M == 5) {
            if (hb_write) {
              lds_write16<HB_RAW>(hb_base + hb_slot * (2 * FRAG), __builtin_bit_cast(u32x4_t, f32x4_t{act[8], act[9], act[10], act[11]}));
              lds_write16<HB_RAW + FRAG>(hb_base + hb_slot * (2 * FRAG), __builtin_bit_cast(u32x4_t, f32x4_t{act[12], act[13], act[14], act[15]}));
            }
          }
        }
        if constexpr (FC1) {
          CS_LGKM(5);
          if constexpr (M == 0) {
            f32x16_t b16;
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
              for (int i = 0; i < 4; ++i) b16[4 * q + i] = bb[q][i];
            if constexpr (ABL & 4) acc = b16;
            else acc = mfma_32x32x16<BF>(w[0], xf[0], b16);
          } else if constexpr (!(ABL & 4)) {
            acc = mfma_32x32x16<BF>(w[M % 6], xf[M], acc);
          }
          constexpr int F = (M + 6) % 24;  // the fragment read now: of this tick up to gap 17, of the next one (cur has moved) from gap 18 on
          if constexpr (!(ABL & 32)) lds_read1<((F / 12) * 24 + F % 12) * FRAG>(cur, w[M % 6]);
        }
        // GELU of registers 0..7 of the previous slice: 16 half values over gaps 0..11 (two in gaps 6..9); the fragment leaves in gap 11, so that
        // the counted wait of gap 17 has retired the write before the barrier of gap 18 tells the partner to read it
        // packed-half GELU of this wave's four pairs: twelve blocks in gaps 2..11 (two in gaps 2 and 7); the pairs turn into the B fragment in place
        if constexpr (M >= 2 && M < 12 && !(ABL & 1)) {
          constexpr int B0 = M < 3 ? 0 : (M < 8 ? M - 1 : M);  // first block of this gap: 0, 2, 3, 4, 5, 6, 8, 9, 10, 11
          const float a8[8] = {act[0], act[1], act[2], act[3], act[4], act[5], act[6], act[7]};
          pk_gelu_op<B0, BF>(pg, xa, kk, a8);
          if constexpr (M == 2 || M == 7) pk_gelu_op<B0 + 1, BF>(pg, xa, kk, a8);
        }
        if constexpr (M == 11) { if (hb_write) lds_write16<0>(hb_base + hb_slot * FRAG, u32x4_t{xa[0], xa[1], xa[2], xa[3]}); }
        CS_SB();
      });
    };
    read_bias(0);  // (before the fragments: the counted waits retire LDS reads in order)
    sfor<6>([&](auto F_) { lds_read1<decltype(F_)::value * FRAG>(cur, w[decltype(F_)::value]); });
#pragma unroll
    for (int i = 0; i < 16; ++i) acO[i] = 0.f;  // tick 0 "activates" this (slice -1: never read by the partner)
    for (int t = 0; t < NSL; t += 2) {
      tick(std::true_type{}, acE, acO, 1, t, t + 1, t > 0);
      tick(std::true_type{}, acO, acE, 0, t + 1, min(t + 2, NSL - 1), true);
    }
    CS_STAMP(4);
    // tick NSL: only the GELU of the last slice (odd), and the boundaries the B waves' tick NSL needs
    CS_LGKM(0);
    tick(std::false_type{}, acE, acO, 1, NSL, 0, true);
    CS_VMCNT(0);  // every LDS-DMA has landed before the workgroup can end ...
    if constexpr (!(ABL & 16)) __builtin_amdgcn_s_barrier();  // ... and the B waves' last boundary (tick NSL + 1, gap 6: unit 2 NTICK - 1)
    CS_STAMP(5);
    // ---- epilogue, A side: the partner stages its rows in LDS (three rounds: residual halves, normalised rows), this wave writes them to
    //      memory as whole 128-byte lines: a store instruction covers 4 rows x 256 contiguous bytes (16 lanes x 16 B per row) ----
    {
      const unsigned st_addr = lds0 + pair * ST_PAIR + (lane >> 4) * ST_ROW + (lane & 15) * 16;
      const int r_in = lane >> 4;
      auto drain = [&](char* gbase, size_t row_bytes) {  // gbase: row0's first byte of this round's 768-byte column range
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
      drain(xg, PC * 4);
      __builtin_amdgcn_s_barrier();  // E2: read
      __builtin_amdgcn_s_barrier();  // E3: residual columns 192..383 staged
      drain(xg + 768, PC * 4);
      __builtin_amdgcn_s_barrier();  // E4: read
      if (p.u_out) {
        __builtin_amdgcn_s_barrier();  // E5: normalised rows staged
        drain(reinterpret_cast<char*>(p.u_out + (size_t)row0 * PC), PC * 2);
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
  const unsigned bv_addr = lds0 + LDS_BV + 64 * (fresh_lane() >> 5);
  unsigned cur = transition(TB{});  // unit 0 (and: the bias vectors are in LDS)
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
    // ---- attention output projection.  Unit c = k-steps 2c, 2c+1 x 12 output tiles (fragment 12 ksl + T).  Its B fragments (this lane's
    //      16 B of its attention-output row per k-step) and residual-row tile c arrive in LDS with the unit (copied by the partner). ----
    sfor<NT>([&](auto T_) { add_bias(T_, std::true_type{}, 0); });
    const unsigned xs_addr = lds0 + LDS_X + pair * XREC + lane16;  // this pair's record of slot 0 (+ slot * 4 * XREC)
    h16x8_t of[2];
    lds_read1<4 * FRAG>(xs_addr, of[0]);
    lds_read1<5 * FRAG>(xs_addr, of[1]);
    sfor<6>([&](auto F_) { lds_read1<decltype(F_)::value * FRAG>(cur, w[decltype(F_)::value]); });
    int slot_x = 0;  // ring slot of the current unit (its residual tile / attention-output fragments share the index)
    sfor<OUT_CHUNKS>([&](auto C_) {
      constexpr int C = decltype(C_)::value;
      f32x4_t xs;
      sfor<24>([&](auto M_) {
        constexpr int M = decltype(M_)::value;
        // the unit's residual-row tile C, ONE quarter (4 registers) at a time: quarter q is read at the start of gap 12 + q and added in
        // gap 13 + q (its LDS slot is refilled after the transition below).  The only LDS read issued after it is weight fragment
        // M + 5, so lgkmcnt(1) retires it; the phase is bound by the A waves' memory stream, not by this wave's LDS latency.
        // (Two quarters at a time with their 8 + 16 transient registers pushed this loop over 256 VGPRs: 80 spilled dwords, r2.)
        if constexpr (M >= 13 && M <= 16) {
          constexpr int Q = M - 13;
          CS_LGKM(1);
#pragma unroll
          for (int i = 0; i < 4; ++i) acc2[C][4 * Q + i] += xs[i];
          CS_SB();
        }
        if constexpr (M >= 12 && M <= 15) lds_read_f4<(M - 12) * FRAG>(xs_addr + slot_x * (4 * XREC), xs);
        if constexpr (M == 18) {
          cur = transition(TB{});
          slot_x = slot_x + 1 >= 3 ? 0 : slot_x + 1;
          if constexpr (C + 1 < OUT_CHUNKS) lds_read1<4 * FRAG>(xs_addr + slot_x * (4 * XREC), of[0]);  // k-step 0's fragment is free from M = 12 on
        }
        if constexpr (M == 0 && C > 0) lds_read1<5 * FRAG>(xs_addr + slot_x * (4 * XREC), of[1]);  // (needed from M = 12 on)
        CS_LGKM(5);
        constexpr int T = M % 12;
        acc2[T] = mfma_32x32x16<BF>(w[M % 6], of[M / 12], acc2[T]);
        if constexpr (M < 18) lds_read1<(M + 6) * FRAG>(cur, w[M % 6]);
        else lds_read1<(M - 18) * FRAG>(cur, w[M % 6]);  // (after the last unit: the A half of MLP tick 0, unused)
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
    s = add_other_half(s);
    mean = s * (1.0f / PC);
    float q = 0.f;
#pragma unroll
    for (int T = 0; T < NT; ++T)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float d = acc2[T][r] - mean;
        q = fmaf(d, d, q);
      }
    q = add_other_half(q);
    rstd = 1.0f / sqrtf(q * (1.0f / PC) + p.eps);
  };
  {
    // norm2 -> the partner's fc1 B fragments: k-step 2T + s is registers 8s .. 8s+7 of tile T
    float mean, rstd;
    row_stats(mean, rstd);
    const float nb = -mean * rstd;
    const unsigned r_addr = lds0 + LDS_XF + pair * 12 * FRAG + lane16;
    auto put = [&](auto T_, auto S_) {
      constexpr int T = decltype(T_)::value, S = decltype(S_)::value;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = fmaf(acc2[T][8 * S + e], rstd, nb);  // (not (x - mean) * rstd: the 192 differences of the variance pass would be kept alive)
      lds_write16<((2 * T + S) % 12) * FRAG>(r_addr, pack8<BF>(v));
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

  // ---- fc2, LAGT ticks behind fc1.  Tick t >= LAGT: fragments 24 .. 47 of its unit = (k-step f / 12, tile f % 12) of hidden slice t - LAGT,
  //      whose activations are in hb slot (t - LAGT) & 1.  A k-step uses ONE activation fragment for all 12 tiles, so the other fragment
  //      register is free to be refilled meanwhile: behind the tick's transition (gap 18) comes fragment 0 of the NEXT slice (the partner
  //      wrote it before that barrier), in gap 0 fragment 1 of the current slice. ----
  // the A waves' first LAGT ticks: two unit boundaries each (nothing to multiply yet)
  for (int g = 0; g < 2 * LAGT; ++g) { if constexpr (!(ABL & 16)) __builtin_amdgcn_s_barrier(); }
  const unsigned hb_base = lds0 + LDS_HB + pair * HB_PAIR + lane16;
  int cur_step = TICK;  // this lane's address of tick T's fragment 0 is lds0 + LDS_RING + lane16 + (T & 1) * TICK: `cur` alternates (no base register kept)
  h16x8_t hb0;       // B fragment of k-step 0 of the current slice (activated by the partner)
  u32x4_t hb1;       // B fragment of k-step 1: activated here, in the gaps of the k-step-0 MFMAs, from the partner's raw fp32 values
  f32x4_t rw[2];     // those raw values (registers 8..15 of the partner's fc1 accumulator of the slice)
  u32x4_t xr;        // (fp16 mode) those values as 4 packed pairs, rounded by the partner
  unsigned hx[4];
  PkGelu pg;
  const PkGeluK kk = pk_gelu_consts();
  // fc2 tick T = slice + LAGT (same parity): unit boundaries in front of gaps 6 and 18 as on the A side, but a B wave has nothing to wait for
  // except the barrier itself (it issues no LDS-DMA, and its own reads of the released unit were retired by the counted waits long before)
  auto fc2_tick = [&](int slice, bool more) {
    sfor<24>([&](auto M_) {
      constexpr int M = decltype(M_)::value;
      if constexpr (M == 6) { CS_SB(); CS_BAR_STAMP(2 * (slice + LAGT), 0); if constexpr (!(ABL & 16)) __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); CS_BAR_STAMP(2 * (slice + LAGT), 1); CS_SB(); }
      if constexpr (M == 18) {
        if (more) {
          CS_SB();
          CS_BAR_STAMP(2 * (slice + LAGT) + 1, 0);
          if constexpr (!(ABL & 16)) __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
          CS_BAR_STAMP(2 * (slice + LAGT) + 1, 1);
          cur += cur_step;
          cur_step = -cur_step;
          lds_read1<0>(hb_base + ((slice + 1) & 1) * FRAG, hb0);  // free from gap 12 on; the partner wrote it before this barrier
          if constexpr (BF) {
            lds_read_f4<HB_RAW>(hb_base + ((slice + 1) & 1) * (2 * FRAG), rw[0]);
            lds_read_f4<HB_RAW + FRAG>(hb_base + ((slice + 1) & 1) * (2 * FRAG), rw[1]);
          } else {
            lds_read_u4<HB_RAW>(hb_base + ((slice + 1) & 1) * (2 * FRAG), xr);
          }
          CS_SB();
        }
      }
      CS_LGKM(5);
      if constexpr (ABL & 8) { if constexpr (M == 12) asm volatile("" ::"v"(hb1)); }
      else if constexpr (M < 12) acc2[M % 12] = mfma_32x32x16<BF>(w[M % 6], hb0, acc2[M % 12]);
      else acc2[M % 12] = mfma_32x32x16<BF>(w[M % 6], __builtin_bit_cast(h16x8_t, hb1), acc2[M % 12]);
      constexpr int F = (M + 6) % 24;  // (after the last tick: re-reads of this tick's first fragments, unused)
      if constexpr (!(ABL & 32)) lds_read1<((F / 12) * 24 + 12 + F % 12) * FRAG>(cur, w[M % 6]);
      // the slice's second fragment: 8 values = 16 half values over gaps 0..11 (two in gaps 6..9), packed pairwise
      // packed-half GELU of the partner's four pairs: one block per gap in gaps 0..11, in place (fp16 mode: the partner rounded them; bf16 mode:
      // its fp32 accumulators arrive and are rounded here for the correction term, their relu stays fp32)
      if constexpr (M == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if constexpr (BF) hx[i] = pack_h16x2(rw[i >> 1][2 * (i & 1)], rw[i >> 1][2 * (i & 1) + 1]);
          else hx[i] = xr[i];
        }
      }
      if constexpr (M < 12 && !(ABL & 1)) {
        const float a8[8] = {rw[0][0], rw[0][1], rw[0][2], rw[0][3], rw[1][0], rw[1][1], rw[1][2], rw[1][3]};
        pk_gelu_op<M, BF>(pg, hx, kk, a8);
      }
      if constexpr (M == 11) hb1 = u32x4_t{hx[0], hx[1], hx[2], hx[3]};
      CS_SB();
    });
  };
  cur = lds0 + LDS_RING + lane16;  // tick LAGT (even): slice 0
  lds_read1<0>(hb_base, hb0);
  if constexpr (BF) {
    lds_read_f4<HB_RAW>(hb_base, rw[0]);
    lds_read_f4<HB_RAW + FRAG>(hb_base, rw[1]);
  } else {
    lds_read_u4<HB_RAW>(hb_base, xr);
  }
  sfor<6>([&](auto F_) { lds_read1<(12 + decltype(F_)::value) * FRAG>(cur, w[decltype(F_)::value]); });
  for (int sl = 0; sl < NSL - 1; ++sl) fc2_tick(sl, true);
  fc2_tick(NSL - 1, false);
  CS_LGKM(0);
  CS_VMCNT(0);  // every LDS-DMA has landed before the workgroup can end
  CS_STAMP(4);

  // ---- epilogue, B side: rows go to memory through LDS and the partner (a lane holds 64 B of a row per tile: stored directly, every store
  //      instruction would touch 32 rows) ----
  {
    const unsigned le = fresh_lane();
    const unsigned st_addr = (unsigned)(size_t)CS_LDS_PTR(smem) + pair * ST_PAIR + (le & 31) * ST_ROW + 64 * (le >> 5);
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
      const unsigned su = (unsigned)(size_t)CS_LDS_PTR(smem) + pair * ST_PAIR + (le & 31) * ST_ROW + 32 * (le >> 5);
      sfor<NT>([&](auto T_) {
        constexpr int T = decltype(T_)::value;
        sfor<2>([&](auto Q_) {
          constexpr int Q = decltype(Q_)::value;
          float v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaf(acc2[T][8 * Q + e], rstd, nb);
          lds_write16<T * 64 + 16 * Q>(su, pack8<BF>(v));
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

// ---- weight image.  One thread per 16-byte fragment element (8 f16): [unit][fragment][lane 0..63].  Lane (i = lane & 31, h = lane >> 5) of
//      an A-operand fragment holds MFMA row i of its 32-feature tile and 8 contraction indices of its 16-wide k-step.
//      MFMA row i of a D tile is register r = (i & 3) + 4 (i >> 3) of lane half (i >> 2) & 1, and the kernel keeps feature
//      32 T + 16 h + r there, so row i carries output feature  perm(i) = 16 ((i >> 2) & 1) + (i & 3) + 4 (i >> 3)  of its tile.
//      Contraction index of element e:  natural  16 ks + 8 h + e                 (out-projection: B fragments come from memory)
//                                       tiled    32 (ks >> 1) + 16 h + 8 (ks & 1) + e   (fc1 / fc2: B fragments are accumulator tiles)
//      Units: [12 out-projection chunks c of 24 fragments: fragment 12 ksl + T = (k-step 2c + ksl, tile T)]
//             [50 MLP ticks t of 48 fragments: fragments 0..23  = fc1 slice t, k-step f                              (t < 48)
//                                              fragments 24..47 = fc2 slice t - 2, k-step f' / 12, tile f' % 12       (t >= 2)]
//             [1 padding tick] ----
__device__ __forceinline__ int panel_perm(int i) { return 16 * ((i >> 2) & 1) + (i & 3) + 4 * (i >> 3); }
template <bool BF>
__global__ __launch_bounds__(256) void cs_panel_pack_kernel(const float* __restrict__ wo, const float* __restrict__ ls1,
                                                            const float* __restrict__ w1, const float* __restrict__ g2,
                                                            const float* __restrict__ w2, const float* __restrict__ ls2,
                                                            h16_t* __restrict__ img) {
  const int nout = wo ? OUT_CHUNKS : 0;
  const long long n_out16 = (long long)nout * (CHUNK / 16);
  const long long total16 = n_out16 + (long long)(NTICK + PAD_TICKS) * (TICK / 16);
  const long long gi = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gi >= total16) return;
  float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (gi < n_out16) {
    const int c = (int)(gi / (CHUNK / 16));
    const int within = (int)(gi - (long long)c * (CHUNK / 16));
    const int f = within >> 6, lane = within & 63, i = lane & 31, h = lane >> 5;
    const int ks = 2 * c + f / 12, T = f % 12;
    const int rowi = 32 * T + panel_perm(i);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = wo[(size_t)rowi * PC + 16 * ks + 8 * h + e] * (ls1 ? ls1[rowi] : 1.f);
  } else {
    const long long g2i = gi - n_out16;
    const int t = (int)(g2i / (TICK / 16));
    const int within = (int)(g2i - (long long)t * (TICK / 16));
    // position in the tick: [half 0: 12 A | 12 B][half 1: 12 A | 12 B]  ->  f = role * 24 + (fragment of that role, 0..23)
    const int fpos = within >> 6, lane = within & 63, i = lane & 31, h = lane >> 5;
    const int f = ((fpos % 24) / 12) * 24 + (fpos / 24) * 12 + fpos % 12;
    if (t < NTICK) {
      if (f < 24) {
        if (t < NSL) {
          const int ks = f;
          const int rowi = 32 * t + panel_perm(i);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int col = 32 * (ks >> 1) + 16 * h + 8 * (ks & 1) + e;
            v[e] = w1[(size_t)rowi * PC + col] * (g2 ? g2[col] : 1.f);
          }
        }
      } else if (t >= LAGT) {
        const int tt = t - LAGT, fp = f - 24;
        const int s = fp / 12, T = fp % 12;
        const int rowi = 32 * T + panel_perm(i);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int col = 32 * tt + 16 * h + 8 * s + e;
          v[e] = w2[(size_t)rowi * PF + col] * (ls2 ? ls2[rowi] : 1.f);
        }
      }
    }
  }
  const uint4 o = {pack_o16x2<BF>(v[0], v[1]), pack_o16x2<BF>(v[2], v[3]), pack_o16x2<BF>(v[4], v[5]), pack_o16x2<BF>(v[6], v[7])};
  reinterpret_cast<uint4*>(img)[gi] = o;
}

template <bool OUTPROJ, bool BF>
hipError_t panel_launch_t(const CsPanelParams* p, hipStream_t st) {
  static std::atomic<bool> attr_done[16];  // (zero-initialised; hipFuncSetAttribute is idempotent, a racing second caller only repeats it)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return hipErrorInvalidDevice;
  if (!attr_done[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(cs_panel_kernel<OUTPROJ, 0, BF>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) return e;
    attr_done[dev] = true;
  }
  const int grid = (p->M + PANEL_ROWS - 1) / PANEL_ROWS;
  hipLaunchKernelGGL((cs_panel_kernel<OUTPROJ, 0, BF>), dim3(grid), dim3(512), LDS_BYTES, st, *p);
  return hipGetLastError();
}

}  // namespace

extern "C" {

#ifdef CS_PANEL_ABLATE
int cs_panel_debug_read(unsigned long long* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_panel_dbg), sizeof(g_panel_dbg)); }
int cs_panel_bar_read(unsigned long long* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_panel_bar), sizeof(g_panel_bar)); }
#endif

int cs_panel_supported(int C, int mlp_ratio) { return C == PC && mlp_ratio * C == PF; }
size_t cs_panel8_image_bytes(int with_outproj) { return (size_t)(with_outproj ? OUT_CHUNKS : 0) * CHUNK + (size_t)(NTICK + PAD_TICKS) * TICK; }

hipError_t cs_panel_pack_launch(const float* wo, const float* ls1, const float* w1, const float* g2, const float* w2, const float* ls2,
                                h16_t* img, int bf16, hipStream_t st) {
  const int total = (int)(cs_panel8_image_bytes(wo ? 1 : 0) / 16);
  if (bf16) hipLaunchKernelGGL(cs_panel_pack_kernel<true>, dim3((total + 255) / 256), dim3(256), 0, st, wo, ls1, w1, g2, w2, ls2, img);
  else hipLaunchKernelGGL(cs_panel_pack_kernel<false>, dim3((total + 255) / 256), dim3(256), 0, st, wo, ls1, w1, g2, w2, ls2, img);
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
#ifdef CS_PANEL_ABLATE
  if (const char* e = getenv("CS_PANEL_ABL")) {
    const int abl = atoi(e);
    const int grid = (p->M + PANEL_ROWS - 1) / PANEL_ROWS;
#define CS_ABL_CASE(N) if (abl == N) { \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(cs_panel_kernel<true, N>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES); \
      hipLaunchKernelGGL((cs_panel_kernel<true, N>), dim3(grid), dim3(512), LDS_BYTES, st, *p); return hipGetLastError(); }
    CS_ABL_CASE(1) CS_ABL_CASE(2) CS_ABL_CASE(3) CS_ABL_CASE(4) CS_ABL_CASE(8) CS_ABL_CASE(12) CS_ABL_CASE(16) CS_ABL_CASE(18) CS_ABL_CASE(32) CS_ABL_CASE(34) CS_ABL_CASE(50) CS_ABL_CASE(64)
#undef CS_ABL_CASE
  }
#endif
  if (p->attn_o) return p->bf16 ? panel_launch_t<true, true>(p, st) : panel_launch_t<true, false>(p, st);
  return p->bf16 ? panel_launch_t<false, true>(p, st) : panel_launch_t<false, false>(p, st);
}

}  // extern "C"
