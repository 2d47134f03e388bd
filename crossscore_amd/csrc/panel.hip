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
constexpr int LAG = 4;                // half ticks the fc2 side runs behind the fc1 side
constexpr int MLP_CHUNKS = 2 * NSL + LAG;  // 100 half ticks
constexpr int PAD_CHUNKS = 2;         // fetched by the last transitions, never read
constexpr int PANEL_ROWS = 128;
// LDS map
constexpr int LDS_RING = 0;
constexpr int LDS_R = NSLOT * CHUNK;                 // 48 KiB: norm2(x) hand-off, 12 fragments per pair at a time
constexpr int LDS_HB = LDS_R + 4 * 12 * FRAG;        // 2 slots x 4 pairs x 2 fragments
constexpr int LDS_B1 = LDS_HB + 2 * 4 * 2 * FRAG;    // fc1 bias, fp32
constexpr int LDS_BYTES = LDS_B1 + PF * 4;           // 142 KiB

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
__device__ __forceinline__ void keep_alive(const h16x8_t& w) { asm volatile("" ::"v"(w)); }  // (ablation builds: rule 17)
template <int OFF, bool SKIP = false>
__device__ __forceinline__ void lds_read3(unsigned addr, h16x8_t (&w)[3]) {
  if constexpr (SKIP) { asm volatile("" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]) : "v"(addr)); return; }
  asm volatile("ds_read_b128 %0, %3 offset:%4\n\tds_read_b128 %1, %3 offset:%5\n\tds_read_b128 %2, %3 offset:%6"
               : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]) : "v"(addr), "n"(OFF), "n"(OFF + FRAG), "n"(OFF + 2 * FRAG) : "memory");
}
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

// One LDS-DMA piece = one fragment: 64 lanes x 16 B from `src + OFF` (this lane's address) to the wave-uniform LDS address `dst + OFF`
// (the instruction's immediate offset applies to both addresses)
template <int OFF>
__device__ __forceinline__ void dma_piece(const char* src, unsigned dst) {
  __builtin_amdgcn_global_load_lds(CS_GLOBAL_PTR(src), (__attribute__((address_space(3))) void*)(size_t)dst, 16, OFF, 0);
}

// ABL: timing-only ablations (tools/panel_ablate.py builds them with -DCS_PANEL_ABLATE; results are wrong by design):
//   1 no GELU arithmetic, 2 no LDS-DMA after the prologue, 4 A waves skip their MFMAs, 8 B waves skip theirs, 16 no s_barrier per chunk,
//   32 no weight-fragment LDS reads
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
  constexpr int NOUT = OUTPROJ ? OUT_CHUNKS : 0;
  // Chunks below SOLO are copied by the A waves alone (6 pieces each): during the out-projection the B waves have ordinary global
  // loads in flight, which share the vmcnt queue with LDS-DMA.  From SOLO on: A waves 2 pieces, B waves 4.
  constexpr int SOLO = NOUT + 3;
  const unsigned lds0 = (unsigned)(size_t)CS_LDS_PTR(smem);
  const unsigned lane16 = lane * 16;
  const char* img = reinterpret_cast<const char*>(p.img) + lane16;

  int c_next = 0;    // next chunk to make current
  int slot_cur = 0;  // ring slot of chunk c_next
  auto ring_issue = [&](auto ISA_, int c, int slot) {
    constexpr bool ISA = decltype(ISA_)::value;
    const unsigned dst = lds0 + LDS_RING + slot * CHUNK;
    const char* s = img + (size_t)c * CHUNK;
    if (c < SOLO) {
      if constexpr (ISA) {
        const char* s6 = s + pair * 6 * FRAG;
        const unsigned d6 = dst + pair * 6 * FRAG;
        dma_piece<0>(s6, d6); dma_piece<FRAG>(s6, d6); dma_piece<2 * FRAG>(s6, d6); dma_piece<3 * FRAG>(s6, d6);
        dma_piece<0>(s6 + 4 * FRAG, d6 + 4 * FRAG); dma_piece<FRAG>(s6 + 4 * FRAG, d6 + 4 * FRAG);
      }
    } else if constexpr (ISA) {
      dma_piece<0>(s + pair * 2 * FRAG, dst + pair * 2 * FRAG);
      dma_piece<FRAG>(s + pair * 2 * FRAG, dst + pair * 2 * FRAG);
    } else {
      const char* s4 = s + (8 + pair * 4) * FRAG;
      const unsigned d4 = dst + (8 + pair * 4) * FRAG;
      dma_piece<0>(s4, d4); dma_piece<FRAG>(s4, d4); dma_piece<2 * FRAG>(s4, d4); dma_piece<3 * FRAG>(s4, d4);
    }
  };
  // transition into chunk c_next: this wave's LDS reads of the previous chunk are complete (its slot is about to be refilled) and its
  // own LDS-DMA pieces of chunk c_next have landed; barrier (the same holds for every wave); then the chunk two ahead goes into the slot
  // of the previous chunk.  Returns this lane's LDS address of fragment 0 of the now-current chunk.
  auto transition = [&](auto ISA_) -> unsigned {
    constexpr bool ISA = decltype(ISA_)::value;
    CS_SB();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if constexpr (ISA) {
      if (c_next <= SOLO - 2) CS_VMCNT(6); else CS_VMCNT(2);
    } else {
      CS_VMCNT(4);
    }
    if constexpr (!(ABL & 16)) __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    int s2 = slot_cur + 2; s2 = s2 >= NSLOT ? s2 - NSLOT : s2;
    if constexpr (!(ABL & 2)) ring_issue(ISA_, c_next + 2, s2);  // the image ends with PAD_CHUNKS, so the stream needs no tail case
    const unsigned base = lds0 + LDS_RING + slot_cur * CHUNK + lane16;
    ++c_next;
    slot_cur = slot_cur + 1 >= NSLOT ? 0 : slot_cur + 1;
    CS_SB();  // (rule: register-only MFMAs must not be scheduled above the waits of this statement)
    return base;
  };
  using TA = std::true_type;
  using TB = std::false_type;

  if (is_a) {
    // =====================================================================================================================
    // A wave: LDS-DMA for the out-projection phase, then fc1 + GELU
    // =====================================================================================================================
    // fc1 bias (LN2 beta folded in) -> LDS once per workgroup (the A waves are idle here)
    for (int i = tid; i < PF / 4; i += 256)
      reinterpret_cast<f32x4_t*>(smem + LDS_B1)[i] = reinterpret_cast<const f32x4_t*>(p.b1)[i];
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    ring_issue(TA{}, 0, 0);
    ring_issue(TA{}, 1, 1);
    unsigned cur = transition(TA{});                       // chunk 0
    for (int c = 0; c < NOUT; ++c) cur = transition(TA{});  // the B waves multiply chunks 0 .. NOUT-1; `cur` ends at chunk NOUT
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

    // ---- fc1 ticks.  Tick t: acc(t & 1) = b1 + W1[slice t] . xf (two half ticks = two chunks of 12 fragments); GELU of slice t-1
    //      (the other accumulator) rides between the MFMAs and leaves as two B fragments in hb slot (t-1) & 1 ----
    f32x16_t acE, acO;  // even / odd slices
    const unsigned bias_addr = lds0 + LDS_B1 + 16 * h;
    const unsigned hb_addr = lds0 + LDS_HB + pair * 2 * FRAG + lane16;
    float gv[8];
    h16x8_t wa[3], wb[3];
    auto gelu2 = [&](auto V0_, const f32x16_t& src) {  // values V0, V0+1 of the slice being activated
      constexpr int V0 = decltype(V0_)::value;
      gv[V0 & 7] = (ABL & 1) ? src[V0] : gelu_erf1(src[V0]);
      gv[(V0 + 1) & 7] = (ABL & 1) ? src[V0 + 1] : gelu_erf1(src[V0 + 1]);
    };
    auto hb_store = [&](int slot, auto S_) { lds_write16<decltype(S_)::value * FRAG>(hb_addr + slot * (4 * 2 * FRAG), pack8(gv)); };
    auto mm3 = [&](auto K0_, const h16x8_t (&w)[3], f32x16_t& acc) {
      constexpr int K0 = decltype(K0_)::value;
      if constexpr (ABL & 4) { keep_alive(w[0]); keep_alive(w[1]); keep_alive(w[2]); return; }
#pragma unroll
      for (int i = 0; i < 3; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[i], xf[K0 + i], acc, 0, 0, 0);
    };
    // One half tick: 12 fragments in 4 batches of 3; the reads of batch k+1 are in flight under the MFMAs of batch k, the transition to
    // the next chunk (and the first reads of it) happens before the last batch.  P = 0: k-steps 0..11, GELU values 0..7 (-> hb fragment
    // 0); P = 1: k-steps 12..23, values 8..15 in the first two batches, hb fragment 1 before the transition (the partner reads it after).
    auto half_tick = [&](auto P_, auto FC1_, f32x16_t& acc, const f32x16_t& act, int hb_slot) {
      constexpr int P = decltype(P_)::value;
      constexpr bool FC1 = decltype(FC1_)::value;
      if constexpr (FC1) { lds_read3<3 * FRAG, (ABL & 32) != 0>(cur, wb); CS_LGKM(3); }
      if constexpr (P == 0) { gelu2(IC<0>{}, act); gelu2(IC<2>{}, act); }
      else { gelu2(IC<8>{}, act); gelu2(IC<10>{}, act); gelu2(IC<12>{}, act); }
      if constexpr (FC1) mm3(IC<12 * P + 0>{}, wa, acc);
      CS_SB();
      if constexpr (FC1) { lds_read3<6 * FRAG, (ABL & 32) != 0>(cur, wa); CS_LGKM(3); }
      if constexpr (P == 0) { gelu2(IC<4>{}, act); }
      else { gelu2(IC<14>{}, act); }
      if constexpr (FC1) mm3(IC<12 * P + 3>{}, wb, acc);
      CS_SB();
      if constexpr (FC1) { lds_read3<9 * FRAG, (ABL & 32) != 0>(cur, wb); CS_LGKM(3); }
      if constexpr (P == 0) { gelu2(IC<6>{}, act); hb_store(hb_slot, IC<0>{}); }
      else { hb_store(hb_slot, IC<1>{}); }
      if constexpr (FC1) mm3(IC<12 * P + 6>{}, wa, acc);
      CS_SB();
      cur = transition(TA{});  // waits for every LDS operation of this wave first (batch 3's fragments, the hb store)
      if constexpr (FC1) {
        lds_read3<0, (ABL & 32) != 0>(cur, wa);
        mm3(IC<12 * P + 9>{}, wb, acc);
        CS_SB();
      }
    };
    auto bias_init = [&](f32x16_t& acc, int t) {
      f32x4_t b4[4];
      const unsigned a = bias_addr + t * 128;
      lds_read_f4<0>(a, b4[0]); lds_read_f4<32>(a, b4[1]); lds_read_f4<64>(a, b4[2]); lds_read_f4<96>(a, b4[3]);
      CS_LGKM(0);  // (also retires the first fragments of the chunk: they are older)
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[4 * q + i] = b4[q][i];
    };
    lds_read3<0, (ABL & 32) != 0>(cur, wa);
#pragma unroll
    for (int i = 0; i < 16; ++i) acO[i] = 0.f;  // tick 0 "activates" this (slice -1: never read by the partner)
    for (int t = 0; t < NSL; t += 2) {
      bias_init(acE, t);
      half_tick(IC<0>{}, std::true_type{}, acE, acO, 1);
      half_tick(IC<1>{}, std::true_type{}, acE, acO, 1);
      bias_init(acO, t + 1);
      half_tick(IC<0>{}, std::true_type{}, acO, acE, 0);
      half_tick(IC<1>{}, std::true_type{}, acO, acE, 0);
    }
    // tick NSL: only the GELU of the last slice (odd)
    CS_LGKM(0);
    half_tick(IC<0>{}, std::false_type{}, acE, acO, 1);
    half_tick(IC<1>{}, std::false_type{}, acE, acO, 1);
    // the B waves' last tick
    cur = transition(TA{});
    CS_VMCNT(0);  // the padding chunks' LDS-DMA has landed before the workgroup can end
    return;
  }

  // =======================================================================================================================
  // B wave: residual rows; out-projection, LayerNorm hand-off, fc2, epilogue
  // =======================================================================================================================
  f32x16_t acc2[NT];
  const size_t row = (size_t)min(row0 + j, p.M - 1);
  const bool row_ok = row0 + j < p.M;
  {
    const float* xr = p.x + row * PC + 4 * h;
#pragma unroll
    for (int T = 0; T < NT; ++T)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4_t v = *reinterpret_cast<const f32x4_t*>(xr + 32 * T + 8 * q);
        if constexpr (OUTPROJ) v += *reinterpret_cast<const f32x4_t*>(p.bo + 32 * T + 8 * q + 4 * h);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc2[T][4 * q + i] = v[i];
      }
  }
  h16x8_t wa[3], wb[3];
  unsigned cur = transition(TB{});  // chunk 0
  if constexpr (OUTPROJ) {
    // ---- attention output projection: chunk c = k-steps 2c, 2c+1 x 12 output tiles (fragment 12 ksl + T); the B fragment of a k-step
    //      is this lane's 16 contiguous bytes of its attention-output row ----
    const h16_t* orow = p.attn_o + row * PC + 8 * h;
    h16x8_t of0 = *reinterpret_cast<const h16x8_t*>(orow), of1 = *reinterpret_cast<const h16x8_t*>(orow + 16);
    lds_read3<0, (ABL & 32) != 0>(cur, wa);
    for (int c = 0; c < OUT_CHUNKS; ++c) {
      const int cn = min(c + 1, OUT_CHUNKS - 1);
      const h16x8_t nf0 = *reinterpret_cast<const h16x8_t*>(orow + 32 * cn), nf1 = *reinterpret_cast<const h16x8_t*>(orow + 32 * cn + 16);
      sfor<8>([&](auto B_) {
        constexpr int B = decltype(B_)::value;  // batch: fragments 3B .. 3B+2
        h16x8_t(&wc)[3] = (B & 1) ? wb : wa;
        h16x8_t(&wn)[3] = (B & 1) ? wa : wb;
        if constexpr (B < 7) {
          lds_read3<(3 * B + 3) * FRAG, (ABL & 32) != 0>(cur, wn);
          CS_LGKM(3);
        } else {
          cur = transition(TB{});
          lds_read3<0, (ABL & 32) != 0>(cur, wn);
          CS_SB();
        }
        sfor<3>([&](auto I_) {
          constexpr int F = 3 * B + decltype(I_)::value, T = F % 12;
          if constexpr (ABL & 8) keep_alive(wc[decltype(I_)::value]);
          else acc2[T] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wc[decltype(I_)::value], F < 12 ? of0 : of1, acc2[T], 0, 0, 0);
        });
        CS_SB();
      });
      of0 = nf0; of1 = nf1;
    }
    CS_LGKM(0);  // (the last transition's reads took the A half of chunk NOUT: unused)
  }

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
    // norm2 -> the partner's fc1 B fragments: k-step 2T + s is registers 8s .. 8s+7 of tile T (kappa order, see the header)
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
#pragma unroll
  for (int T = 0; T < NT; ++T)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4_t b4 = *reinterpret_cast<const f32x4_t*>(p.b2 + 32 * T + 8 * q + 4 * h);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc2[T][4 * q + i] += b4[i];
    }

  // ---- fc2, LAG half ticks behind fc1.  Half tick g >= LAG: fragments 12..23 of its chunk = (tile 6p + f/2, k-step f % 2), p = g & 1,
  //      of hidden slice (g - LAG) / 2, whose activations are in hb slot ((g - LAG) / 2) & 1 ----
  for (int g = 0; g < LAG; ++g) cur = transition(TB{});  // chunks NOUT+1 .. NOUT+LAG become current; `cur` = chunk NOUT + LAG
  const unsigned hb_addr = lds0 + LDS_HB + pair * 2 * FRAG + lane16;
  h16x8_t hb[2];
  auto fc2_half = [&](auto P_, bool more) {
    constexpr int P = decltype(P_)::value;
    if constexpr (P == 0) {  // a new slice: its activations (written by the partner before the barrier that made this chunk current)
      const unsigned a = hb_addr + (((c_next - 1 - NOUT - LAG) >> 1) & 1) * (4 * 2 * FRAG);
      lds_read1<0>(a, hb[0]);
      lds_read1<FRAG>(a, hb[1]);
    }
    sfor<4>([&](auto B_) {
      constexpr int B = decltype(B_)::value;  // batch: fragments 12 + 3B .. +2
      h16x8_t(&wc)[3] = (B & 1) ? wb : wa;
      h16x8_t(&wn)[3] = (B & 1) ? wa : wb;
      if constexpr (B < 3) {
        lds_read3<(12 + 3 * B + 3) * FRAG, (ABL & 32) != 0>(cur, wn);
        CS_LGKM(3);
      } else if (more) {
        cur = transition(TB{});
        lds_read3<12 * FRAG, (ABL & 32) != 0>(cur, wn);
        CS_SB();
      } else {
        CS_LGKM(0);
      }
      sfor<3>([&](auto I_) {
        constexpr int F = 3 * B + decltype(I_)::value, T = 6 * P + F / 2, S = F % 2;
        if constexpr (ABL & 8) keep_alive(wc[decltype(I_)::value]);
        else acc2[T] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wc[decltype(I_)::value], hb[S], acc2[T], 0, 0, 0);
      });
      CS_SB();
    });
  };
  lds_read3<12 * FRAG, (ABL & 32) != 0>(cur, wa);
  for (int t = 0; t < NSL - 1; ++t) {
    fc2_half(IC<0>{}, true);
    fc2_half(IC<1>{}, true);
  }
  fc2_half(IC<0>{}, true);
  fc2_half(IC<1>{}, false);
  CS_VMCNT(0);  // the padding chunks' LDS-DMA has landed before the workgroup can end

  // ---- epilogue: new residual rows, and the next layer's normalised rows ----
  if (row_ok) {
    float* xr = p.x + row * PC + 4 * h;
#pragma unroll
    for (int T = 0; T < NT; ++T)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<f32x4_t*>(xr + 32 * T + 8 * q) = f32x4_t{acc2[T][4 * q], acc2[T][4 * q + 1], acc2[T][4 * q + 2], acc2[T][4 * q + 3]};
  }
  if (p.u_out) {
    float mean, rstd;
    row_stats(mean, rstd);
    const float nb = -mean * rstd;
    if (row_ok) {
      h16_t* ur = p.u_out + row * PC + 4 * h;
#pragma unroll
      for (int T = 0; T < NT; ++T)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float a0 = fmaf(acc2[T][4 * q], rstd, nb), a1 = fmaf(acc2[T][4 * q + 1], rstd, nb);
          const float a2 = fmaf(acc2[T][4 * q + 2], rstd, nb), a3 = fmaf(acc2[T][4 * q + 3], rstd, nb);
          *reinterpret_cast<uint2*>(ur + 32 * T + 8 * q) = make_uint2(pack_h16x2(a0, a1), pack_h16x2(a2, a3));
        }
    }
  }
}

// ---- weight image.  One thread per 16-byte fragment element (8 f16): [chunk][fragment 0..23][lane 0..63].  Lane (i = lane & 31,
//      h = lane >> 5) of an A-operand fragment holds output feature i of its 32-feature tile and 8 contraction indices of its 16-wide
//      k-step:  natural   : 8 h + e                                   (out-projection: its B fragments are loaded from memory)
//               permuted  : kappa(h, e) = (e & 3) + 8 (e >> 2) + 4 h   (fc1 / fc2: their B fragments are accumulator tiles)
//      Chunks: [12 out-projection chunks c: fragment 12 ksl + T = (k-step 2c + ksl, tile T)]
//              [100 MLP half ticks g: fragments 0..11  = fc1 slice g/2, k-step 12 (g&1) + f                         (g < 96)
//                                     fragments 12..23 = fc2 slice (g-4)/2, tile 6 (g&1) + f'/2, k-step f' % 2       (g >= 4)]
//              [2 padding chunks] ----
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
    const int rowi = 32 * T + i;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = wo[(size_t)rowi * PC + 16 * ks + 8 * h + e] * (ls1 ? ls1[rowi] : 1.f);
  } else if (c < nch) {
    const int g = c - nout;
    if (f < 12) {
      if (g < 2 * NSL) {
        const int t = g >> 1, ks = 12 * (g & 1) + f;
        const int rowi = 32 * t + i;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int col = 16 * ks + (e & 3) + 8 * (e >> 2) + 4 * h;
          v[e] = w1[(size_t)rowi * PC + col] * (g2 ? g2[col] : 1.f);
        }
      }
    } else if (g >= LAG) {
      const int tt = (g - LAG) >> 1, fp = f - 12;
      const int T = 6 * (g & 1) + fp / 2, s = fp % 2;
      const int rowi = 32 * T + i;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int col = 32 * tt + 16 * s + (e & 3) + 8 * (e >> 2) + 4 * h;
        v[e] = w2[(size_t)rowi * PF + col] * (ls2 ? ls2[rowi] : 1.f);
      }
    }
  }
  const uint4 o = {pack_h16x2(v[0], v[1]), pack_h16x2(v[2], v[3]), pack_h16x2(v[4], v[5]), pack_h16x2(v[6], v[7])};
  reinterpret_cast<uint4*>(img)[gi] = o;
}

}  // namespace

extern "C" {

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
  if ((long long)p->M * PC >= (1ll << 31)) return "panel: too many rows for 32-bit offsets";
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
    CS_ABL_CASE(1) CS_ABL_CASE(2) CS_ABL_CASE(4) CS_ABL_CASE(8) CS_ABL_CASE(16) CS_ABL_CASE(32) CS_ABL_CASE(12) CS_ABL_CASE(3) CS_ABL_CASE(47) CS_ABL_CASE(5)
#undef CS_ABL_CASE
  }
#endif
  if (v) hipLaunchKernelGGL(cs_panel_kernel<true>, dim3(grid), dim3(512), LDS_BYTES, st, *p);
  else hipLaunchKernelGGL(cs_panel_kernel<false>, dim3(grid), dim3(512), LDS_BYTES, st, *p);
  return hipGetLastError();
}

}  // extern "C"
