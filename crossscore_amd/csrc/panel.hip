// Encoder token-panel kernel for gfx950 (ViT-S: C = 384, MLP 4C = 1536): everything a DINOv2 layer does between its attention
// and the next layer's QKV projection, in ONE launch, with the 4C hidden activations never leaving the register file.
//
//   x   += attn_o Wo'^T + bo'                       Dinov2SelfOutput + layer_scale1 + residual   (HF modeling_dinov2.py:249-252,365-370)
//   x   += GELU(LN2(x) W1'^T + b1') W2'^T + b2'     norm2 + Dinov2MLP + layer_scale2 + residual  (HF:373-378, 293-297)
//   u    = bf16((x - mean(x)) * rstd(x))            norm1 of the next layer (its gamma/beta live in that layer's packed Wqkv / bias)
//
// Replaces five launches of round 1 (out-proj GEMM, LayerNorm, fc1+GELU GEMM, fc2 GEMM, LayerNorm) and their HBM round trips:
// per 65 760-row pass the 202 MB hidden write + 202 MB re-read, two fp32 LayerNorm re-reads and one fp32 read-modify-write go.
//
// Structure ("flash-MLP"): a workgroup = 4 waves = 128 token rows, one wave per SIMD with the whole 512-register budget; a wave
// owns 32 rows for the entire kernel.  MFMA v_mfma_f32_16x16x32_bf16 with the WEIGHT fragment as the A operand and the
// activation fragment as the B operand, so D[n][m]: a lane (g = lane/16, m = lane%16) holds 4 consecutive output features
// 16t+4g.. of token row m.  Because the C/D layout of one product is the B-operand layout of the next up to a permutation of the
// contraction index (the weights are pre-packed with that permutation), the chain
//      acc2 (x, fp32, 24x2 tiles) --LN--> xf (bf16 B fragments) --fc1--> acc1 --GELU--> hb (bf16 B fragment) --fc2--> acc2
// needs no LDS traffic and no cross-lane movement at all: only weight fragments are read from LDS.
// Weights stream HBM/L2 -> LDS by global_load_lds_dwordx4 as 24-KiB "units" (one unit = 48 MFMAs per wave) from an image that
// was packed at finalize in exactly the LDS layout and consumption order (linear 1-KiB pieces: full-line requests), through a
// 4-slot ring, two units in flight, counted s_waitcnt vmcnt + one raw s_barrier per unit.
// The fc2 accumulators are INITIALISED with the residual rows (+ bias), so the residual add costs nothing and x is read once.
#include "cs_common.h"
#include <type_traits>
#include <utility>

namespace {

constexpr int PC = 384;             // hidden size
constexpr int PF = 1536;            // MLP hidden
constexpr int NT = PC / 16;         // 24 output tiles of 16 features
constexpr int KS = PC / 32;         // 12 k-steps over C
constexpr int NSL = PF / 32;        // 48 hidden slices of 32
constexpr int UNIT = 24 * 1024;     // bytes per weight unit: 24 pieces of 16 rows x 64 B
constexpr int NSLOT = 4;            // ring slots
constexpr int AHEAD = 2;            // units in flight; slot (u+AHEAD)%4 was last read two barriers ago
constexpr int DPW = 6;              // LDS-DMA instructions per wave per unit
constexpr int LDS_B1 = NSLOT * UNIT;
constexpr int LDS_BYTES = LDS_B1 + PF * 4;  // 102 KiB: one workgroup per CU (the register file admits only one anyway)
constexpr int PANEL_ROWS = 128;
constexpr int OUT_UNITS = KS;       // out-projection: 12 units of 32 output features
constexpr int MLP_UNITS = 2 * NSL;  // 96

#define CS_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

template <int... Js, class F>
__device__ __forceinline__ void static_for(std::integer_sequence<int, Js...>, F&& f) {
  (f(std::integral_constant<int, Js>{}), ...);
}

__device__ __forceinline__ float quad_sum(float v) {  // sum over the 4 lanes (g = 0..3) that share a token row
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

// exact-erf GELU, same degree-7 minimax fit of Phi as gelu_erf4 (cs_common.h), one value per call in plain fma form: in this
// kernel the VALU work shares ONE wave's issue stream with the MFMAs, so it must be schedulable instruction by instruction
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

template <bool OUTPROJ>
__global__ __launch_bounds__(256, 1) void cs_panel_kernel(CsPanelParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, m = lane & 15;
  const int row0 = blockIdx.x * PANEL_ROWS + wv * 32;
  constexpr int NU = (OUTPROJ ? OUT_UNITS : 0) + MLP_UNITS;

  // ---- weight stream: wave w copies pieces w, w+4, .. of each unit; the image is linear, so lane i moves bytes [16i, 16i+16)
  //      of its piece to the same offset of the slot (whole 1-KiB requests) ----
  const char* img = reinterpret_cast<const char*>(p.img) + wv * 1024 + lane * 16;
  auto issue = [&](int u) {
    const char* src = img + (size_t)u * UNIT;
    char* dst = smem + (u & (NSLOT - 1)) * UNIT + wv * 1024;
#pragma unroll
    for (int q = 0; q < DPW; ++q)
      __builtin_amdgcn_global_load_lds(CS_GLOBAL_PTR(src + q * 4096), CS_LDS_PTR(dst + q * 4096), 16, 0, 0);
  };
  issue(0);
  issue(1);

  // fc1 bias (LN2 beta folded in) -> LDS once per workgroup
  for (int i = tid; i < PF / 4; i += 256)
    reinterpret_cast<f32x4_t*>(smem + LDS_B1)[i] = reinterpret_cast<const f32x4_t*>(p.b1)[i];

  f32x4_t acc2[NT][2];  // residual rows / fc2 accumulators: tile nt = features 16nt.., row tile mt
  h16x8_t xf[2][KS];   // B fragments of the current GEMM input (attention output, then norm2(x))

  // ---- unit / batch machinery -----------------------------------------------------------------------------------------------
  // A unit is consumed as 4 batches of 6 weight fragments (12 MFMAs per batch).  The wave is alone on its SIMD, so LDS latency
  // is hidden by the wave itself: the reads of batch k+1 are issued before the MFMAs of batch k, and waited for with a COUNTED
  // lgkmcnt one batch (>= 192 MFMA cycles) later.  hipcc only ever emits lgkmcnt(0) around LDS-DMA kernels, so the ring reads
  // and their waits are inline asm (form (iii) of the guide: "=&v" loads, a wait-only statement, then sched_barrier(0) that keeps
  // the consuming MFMAs below the wait).  Every step is therefore its own scheduling region [12 MFMAs + a slice of the GELU
  // arithmetic], inside which hipcc interleaves VALU and MFMA.
  // The transition to the next unit (counted vmcnt, barrier, LDS-DMA of the unit two ahead) happens before the LAST batch of the
  // current unit is multiplied, so the first reads of a unit are also a batch ahead of their use.
  // per-lane address of its weight-fragment row inside a unit: row m of a 16-row x 64-B piece, 16-byte chunk g XOR-swizzled
  // exactly as the packed image is (conflict-free ds_read_b128, see cs_panel_pack_kernel)
  const unsigned lane_base = (unsigned)(size_t)CS_LDS_PTR(smem) + m * 64 + ((g ^ (((m >> 2) & 1) << 1)) << 4);
  const unsigned bias_base = (unsigned)(size_t)CS_LDS_PTR(smem) + LDS_B1 + 16 * g;
  int u_next = 0;  // next unit to make current
  auto transition = [&](auto LAST_) -> unsigned {
    CS_VMCNT(DPW * (AHEAD - 1));
    __builtin_amdgcn_s_barrier();  // every wave's pieces of unit u_next landed; everyone finished reading unit u_next-2
    asm volatile("" ::: "memory");
    issue(u_next + AHEAD);  // the image ends with AHEAD padding units, so the stream never needs a tail case
    const unsigned sl = lane_base + (u_next & (NSLOT - 1)) * UNIT;
    ++u_next;
    return sl;
  };
#define CS_SB() __builtin_amdgcn_sched_barrier(0)  /* nothing crosses (any other mask let hipcc move MFMAs over the asm waits) */
  auto read6 = [&](unsigned sl, auto B_, h16x8_t (&w)[6]) {
    constexpr int O = decltype(B_)::value * 6 * 1024;
    asm volatile("ds_read_b128 %0, %6 offset:%7\n\tds_read_b128 %1, %6 offset:%8\n\tds_read_b128 %2, %6 offset:%9\n\t"
                 "ds_read_b128 %3, %6 offset:%10\n\tds_read_b128 %4, %6 offset:%11\n\tds_read_b128 %5, %6 offset:%12"
                 : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3]), "=&v"(w[4]), "=&v"(w[5])
                 : "v"(sl), "n"(O), "n"(O + 1024), "n"(O + 2048), "n"(O + 3072), "n"(O + 4096), "n"(O + 5120)
                 : "memory");
  };
  auto read_bias = [&](int slice, f32x4_t (&bb)[2]) {
    const unsigned a = bias_base + slice * 128;
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:64" : "=&v"(bb[0]), "=&v"(bb[1]) : "v"(a) : "memory");
  };
#define CS_LGKM(n) do { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(n) : "memory"); CS_SB(); } while (0)
  // fc1-type batch B: pieces 6B..6B+5 = (jt, ks) pairs; D tile (jt, mt) += W piece x xf[mt][ks]
  auto mm_fc1 = [&](auto B_, auto INIT_, const h16x8_t (&w)[6], f32x4_t (&a0)[2], f32x4_t (&a1)[2], const f32x4_t (&bb)[2]) {
    constexpr int B = decltype(B_)::value;
    constexpr bool INIT = decltype(INIT_)::value;  // the accumulators start at bb (fc1 bias); else they keep accumulating
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int pc = 6 * B + i, jt = pc / KS, ks = pc % KS;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        f32x4_t& d = jt == 0 ? a0[mt] : a1[mt];
        d = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[i], xf[mt][ks], (INIT && ks == 0) ? bb[jt] : d, 0, 0, 0);
      }
    }
    CS_SB();
  };
  auto mm_fc2 = [&](auto B_, const h16x8_t (&w)[6], const h16x8_t (&hb)[2]) {
    constexpr int B = decltype(B_)::value;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
        acc2[6 * B + i][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[i], hb[mt], acc2[6 * B + i][mt], 0, 0, 0);
    CS_SB();
  };
  using B0 = std::integral_constant<int, 0>;
  using B1 = std::integral_constant<int, 1>;
  using B2 = std::integral_constant<int, 2>;
  using B3 = std::integral_constant<int, 3>;
  using NotLast = std::false_type;
  using Last = std::true_type;

  // ---- prologue: this wave's 32 rows of the residual stream into the fc2 accumulator layout ----
  const size_t r_mt[2] = {(size_t)min(row0 + m, p.M - 1), (size_t)min(row0 + 16 + m, p.M - 1)};
  if constexpr (OUTPROJ) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) xf[mt][ks] = *reinterpret_cast<const h16x8_t*>(p.attn_o + r_mt[mt] * PC + 32 * ks + 8 * g);
  }
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    f32x4_t b4 = {0.f, 0.f, 0.f, 0.f};
    if constexpr (OUTPROJ) b4 = *reinterpret_cast<const f32x4_t*>(p.bo + 16 * nt + 4 * g);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) acc2[nt][mt] = *reinterpret_cast<const f32x4_t*>(p.x + r_mt[mt] * PC + 16 * nt + 4 * g) + b4;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's b1 writes are in LDS before the first barrier

  // (inline-asm read results must never stay in flight across compiler-scheduled code such as the LayerNorm below: hipcc counts
  //  them as written at the end of the asm statement and copies / spills them before the data has landed)
  h16x8_t wa[6], wb[6];  // the two fragment batches in flight
  unsigned sl = transition(NotLast{});

  // ---- attention output projection: 12 units of 32 output features, accumulated straight onto the residual rows ----
  if constexpr (OUTPROJ) {
    const f32x4_t nob[2] = {};
    read6(sl, B0{}, wa);
    static_for(std::make_integer_sequence<int, OUT_UNITS>{}, [&](auto U_) {
      constexpr int U = decltype(U_)::value;
      read6(sl, B1{}, wb); CS_LGKM(6); mm_fc1(B0{}, NotLast{}, wa, acc2[2 * U], acc2[2 * U], nob);   // batches 0,1 are tile jt = 0
      read6(sl, B2{}, wa); CS_LGKM(6); mm_fc1(B1{}, NotLast{}, wb, acc2[2 * U], acc2[2 * U], nob);
      read6(sl, B3{}, wb); CS_LGKM(6); mm_fc1(B2{}, NotLast{}, wa, acc2[2 * U + 1], acc2[2 * U + 1], nob);  // batches 2,3 are tile jt = 1
      sl = transition(NotLast{});
      if constexpr (U + 1 < OUT_UNITS) { read6(sl, B0{}, wa); CS_LGKM(6); }
      else CS_LGKM(0);
      mm_fc1(B3{}, NotLast{}, wb, acc2[2 * U + 1], acc2[2 * U + 1], nob);
    });
  }

  // ---- LayerNorm statistics of the rows held in acc2 (two-pass, fp32, in registers) ----
  auto row_stats = [&](float (&mean)[2], float (&rstd)[2]) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      float s = 0.f;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) s += (acc2[nt][mt][0] + acc2[nt][mt][1]) + (acc2[nt][mt][2] + acc2[nt][mt][3]);
      mean[mt] = quad_sum(s) * (1.0f / PC);
      float q = 0.f;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const f32x4_t d = acc2[nt][mt] - mean[mt];
        q += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
      }
      rstd[mt] = 1.0f / sqrtf(quad_sum(q) * (1.0f / PC) + p.eps);
    }
  };

  // ---- norm2 -> fc1 B fragments.  k-slot j of k-step ks is feature 32ks + 4g + j (j < 4) or 32ks + 16 + 4g + j-4: the packed
  //      W1 uses the same order, so the accumulator registers ARE the fragment (no data movement) ----
  {
    float mean[2], rstd[2];
    row_stats(mean, rstd);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const f32x4_t a = (acc2[2 * ks][mt] - mean[mt]) * rstd[mt], b = (acc2[2 * ks + 1][mt] - mean[mt]) * rstd[mt];
        const uint4 pk = {pack_h16x2(a[0], a[1]), pack_h16x2(a[2], a[3]), pack_h16x2(b[0], b[1]), pack_h16x2(b[2], b[3])};
        xf[mt][ks] = __builtin_bit_cast(h16x8_t, pk);
      }
  }
  // the fc2 accumulators start at residual + bias: the residual add and the bias add are free
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const f32x4_t b4 = *reinterpret_cast<const f32x4_t*>(p.b2 + 16 * nt + 4 * g);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) acc2[nt][mt] += b4;
  }

  // ---- MLP: hidden slice i = 32 hidden units.  fc1 (one unit: 32 hidden x 384) -> acc1; GELU + bf16 pack -> hb; fc2 (one unit:
  //      384 out x 32 hidden) accumulates into acc2.  Software pipeline inside the wave: fc1(i) | GELU(i-1) | fc2(i-2) are
  //      independent, so the GELU arithmetic of slice i-1 (16 values per lane) is cut into 8 pieces of 2 values, one per step of
  //      the two units of an iteration, and fills VALU issue slots between the MFMAs of the other two. ----
  f32x4_t ac0[2][2], ac1[2][2], bb0[2], bb1[2];
  h16x8_t hb0[2], hb1[2];
  float gt[8];  // GELU results of the row tile in progress
  using Q0 = std::integral_constant<int, 0>;
  using Q1 = std::integral_constant<int, 1>;
  using Q2 = std::integral_constant<int, 2>;
  using Q3 = std::integral_constant<int, 3>;
  using QN = std::integral_constant<int, -1>;  // no GELU piece in this unit
  using QA = std::integral_constant<int, 10>;
  using QB = std::integral_constant<int, 11>;
  using QC = std::integral_constant<int, 12>;
  using QD = std::integral_constant<int, 13>;
  // piece Q (0..3) of row tile MT: values (jt = Q/2, r = 2(Q%2), +1) of ac[.][MT]; the last piece packs the B fragment
  auto gelu_piece = [&](auto Q_, auto MT_, const f32x4_t (&ac)[2][2], h16x8_t (&hb)[2]) {
    constexpr int Q = decltype(Q_)::value, MT = decltype(MT_)::value;
    if constexpr (Q >= 10) {  // a whole slice inside one unit (first and last slice): pieces 2(Q-10), 2(Q-10)+1 over both row tiles
      constexpr int H = Q - 10;
      auto two = [&](auto P0_, auto P1_, auto M_) {
        constexpr int P0 = decltype(P0_)::value, P1 = decltype(P1_)::value, MM = decltype(M_)::value;
        gt[2 * P0] = gelu_erf1(ac[P0 / 2][MM][2 * (P0 % 2)]);
        gt[2 * P0 + 1] = gelu_erf1(ac[P0 / 2][MM][2 * (P0 % 2) + 1]);
        gt[2 * P1] = gelu_erf1(ac[P1 / 2][MM][2 * (P1 % 2)]);
        gt[2 * P1 + 1] = gelu_erf1(ac[P1 / 2][MM][2 * (P1 % 2) + 1]);
        if constexpr (P1 == 3) {
          const uint4 pk = {pack_h16x2(gt[0], gt[1]), pack_h16x2(gt[2], gt[3]), pack_h16x2(gt[4], gt[5]), pack_h16x2(gt[6], gt[7])};
          hb[MM] = __builtin_bit_cast(h16x8_t, pk);
        }
      };
      if constexpr (H == 0) two(Q0{}, Q1{}, Q0{});
      if constexpr (H == 1) two(Q2{}, Q3{}, Q0{});
      if constexpr (H == 2) two(Q0{}, Q1{}, Q1{});
      if constexpr (H == 3) two(Q2{}, Q3{}, Q1{});
    } else if constexpr (Q >= 0) {
      gt[2 * Q] = gelu_erf1(ac[Q / 2][MT][2 * (Q % 2)]);
      gt[2 * Q + 1] = gelu_erf1(ac[Q / 2][MT][2 * (Q % 2) + 1]);
      if constexpr (Q == 3) {
        const uint4 pk = {pack_h16x2(gt[0], gt[1]), pack_h16x2(gt[2], gt[3]), pack_h16x2(gt[4], gt[5]), pack_h16x2(gt[6], gt[7])};
        hb[MT] = __builtin_bit_cast(h16x8_t, pk);
      }
    }
  };
  // One unit = 4 steps; NEXT = the first reads of the following unit (issued right after the transition), NCNT = how many;
  // (GQ.., GMT, GAC, GHB) = the GELU pieces that ride in this unit.
#define CS_FC1_UNIT(AC, BB, NEXT, NCNT, G0, G1, G2, G3, GMT, GAC, GHB)                                            \
  read6(sl, B1{}, wb); CS_LGKM(6); gelu_piece(G0{}, GMT{}, GAC, GHB); mm_fc1(B0{}, Last{}, wa, AC[0], AC[1], BB); \
  read6(sl, B2{}, wa); CS_LGKM(6); gelu_piece(G1{}, GMT{}, GAC, GHB); mm_fc1(B1{}, Last{}, wb, AC[0], AC[1], BB); \
  read6(sl, B3{}, wb); CS_LGKM(6); gelu_piece(G2{}, GMT{}, GAC, GHB); mm_fc1(B2{}, Last{}, wa, AC[0], AC[1], BB); \
  sl = transition(NotLast{}); NEXT; CS_LGKM(NCNT); gelu_piece(G3{}, GMT{}, GAC, GHB); mm_fc1(B3{}, Last{}, wb, AC[0], AC[1], BB);
#define CS_FC2_UNIT(HB, TRANS, NEXT, NCNT, G0, G1, G2, G3, GMT, GAC, GHB)                                        \
  read6(sl, B1{}, wb); CS_LGKM(6); gelu_piece(G0{}, GMT{}, GAC, GHB); mm_fc2(B0{}, wa, HB);                      \
  read6(sl, B2{}, wa); CS_LGKM(6); gelu_piece(G1{}, GMT{}, GAC, GHB); mm_fc2(B1{}, wb, HB);                      \
  read6(sl, B3{}, wb); CS_LGKM(6); gelu_piece(G2{}, GMT{}, GAC, GHB); mm_fc2(B2{}, wa, HB);                      \
  TRANS; NEXT; CS_LGKM(NCNT); gelu_piece(G3{}, GMT{}, GAC, GHB); mm_fc2(B3{}, wb, HB);
  // unit order of the packed stream: W1[0] W1[1] | W1[2] W2[0] | W1[3] W2[1] | ... | W1[47] W2[45] | W2[46] W2[47]
  read6(sl, B0{}, wa);
  read_bias(0, bb0);
  CS_FC1_UNIT(ac0, bb0, read6(sl, B0{}, wa); read_bias(1, bb1), 8, QN, QN, QN, QN, Q0, ac0, hb0)   // W1[0] -> ac0
  CS_FC1_UNIT(ac1, bb1, read6(sl, B0{}, wa); read_bias(2, bb0), 8, QA, QB, QC, QD, Q0, ac0, hb0)   // W1[1] -> ac1, GELU(slice 0) -> hb0
  for (int i = 2; i < NSL; i += 2) {
    CS_FC1_UNIT(ac0, bb0, read6(sl, B0{}, wa), 6, Q0, Q1, Q2, Q3, Q0, ac1, hb1)                                              // W1[i]   -> ac0
    CS_FC2_UNIT(hb0, sl = transition(NotLast{}), read6(sl, B0{}, wa); read_bias(i + 1, bb1), 8, Q0, Q1, Q2, Q3, Q1, ac1, hb1)  // W2[i-2] <- hb0
    CS_FC1_UNIT(ac1, bb1, read6(sl, B0{}, wa), 6, Q0, Q1, Q2, Q3, Q0, ac0, hb0)                                              // W1[i+1] -> ac1
    CS_FC2_UNIT(hb1, sl = transition(NotLast{}), read6(sl, B0{}, wa); read_bias(min(i + 2, NSL - 1), bb0), 8, Q0, Q1, Q2, Q3, Q1, ac0, hb0)  // W2[i-1] <- hb1
  }
  CS_FC2_UNIT(hb0, sl = transition(NotLast{}), read6(sl, B0{}, wa), 6, QA, QB, QC, QD, Q0, ac1, hb1)   // W2[46], GELU(slice 47) -> hb1
  CS_FC2_UNIT(hb1, (void)0, (void)0, 0, QN, QN, QN, QN, Q0, ac1, hb1)                                  // W2[47]
  CS_VMCNT(0);  // the padding units' LDS-DMA has landed before the workgroup can end
#undef CS_FC1_UNIT
#undef CS_FC2_UNIT

  // ---- epilogue: new residual rows, and the next layer's normalised rows ----
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int row = row0 + 16 * mt + m;
    if (row < p.M) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) *reinterpret_cast<f32x4_t*>(p.x + (size_t)row * PC + 16 * nt + 4 * g) = acc2[nt][mt];
    }
  }
  if (p.u_out) {
    float mean[2], rstd[2];
    row_stats(mean, rstd);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int row = row0 + 16 * mt + m;
      if (row < p.M) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const f32x4_t a = (acc2[nt][mt] - mean[mt]) * rstd[mt];
          *reinterpret_cast<uint2*>(p.u_out + (size_t)row * PC + 16 * nt + 4 * g) = make_uint2(pack_h16x2(a[0], a[1]), pack_h16x2(a[2], a[3]));
        }
      }
    }
  }
}

// ---- weight image.  One thread per 16-byte chunk (8 bf16).  Unit-local layout: piece p (1 KiB) = 16 rows x 64 B; physical
//      chunk c of row r holds logical k-chunk gl = c ^ 2*((r>>2)&1) (the XOR makes the 16-lane groups of ds_read_b128 hit 16
//      distinct 16-byte slots).  Element e of logical chunk gl is contraction index
//         natural : 8 gl + e                         (out-projection: its B fragments are loaded from memory in natural order)
//         permuted: 4 gl + e  (e < 4),  16 + 4 gl + e - 4  (e >= 4)     (fc1 / fc2: their B fragments are accumulator tiles)
//      inside the 32-wide k-step. ----
__global__ __launch_bounds__(256) void cs_panel_pack_kernel(const float* __restrict__ wo, const float* __restrict__ ls1,
                                                            const float* __restrict__ w1, const float* __restrict__ g2,
                                                            const float* __restrict__ w2, const float* __restrict__ ls2,
                                                            h16_t* __restrict__ img) {
  const int nu = (wo ? OUT_UNITS : 0) + MLP_UNITS;
  const int gi = blockIdx.x * blockDim.x + threadIdx.x;
  if (gi >= (nu + AHEAD) * (UNIT / 16)) return;
  if (gi >= nu * (UNIT / 16)) {  // padding units: fetched by the last transitions, never read
    reinterpret_cast<uint4*>(img)[gi] = make_uint4(0, 0, 0, 0);
    return;
  }
  int U = gi / (UNIT / 16);
  const int within = gi - U * (UNIT / 16);
  const int pc = within >> 6, r = (within >> 2) & 15, c = within & 3;
  const int gl = c ^ (((r >> 2) & 1) << 1);
  float v[8];
  if (wo && U < OUT_UNITS) {
    const int row = 32 * U + 16 * (pc / KS) + r, ks = pc % KS;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = wo[(size_t)row * PC + 32 * ks + 8 * gl + e] * (ls1 ? ls1[row] : 1.f);
  } else {
    const int k = U - (wo ? OUT_UNITS : 0);
    // stream order: W1[0] W1[1] | W1[2] W2[0] | W1[3] W2[1] | ... | W1[47] W2[45] | W2[46] W2[47]
    bool is_fc1;
    int s;
    if (k < 2) { is_fc1 = true; s = k; }
    else if (k >= MLP_UNITS - 2) { is_fc1 = false; s = NSL - (MLP_UNITS - k); }
    else { const int j = k - 2; is_fc1 = (j & 1) == 0; s = is_fc1 ? 2 + j / 2 : (j - 1) / 2; }
    if (is_fc1) {
      const int row = 32 * s + 16 * (pc / KS) + r, ks = pc % KS;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int col = 32 * ks + (e < 4 ? 4 * gl + e : 16 + 4 * gl + e - 4);
        v[e] = w1[(size_t)row * PC + col] * (g2 ? g2[col] : 1.f);
      }
    } else {
      const int row = 16 * pc + r;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int col = 32 * s + (e < 4 ? 4 * gl + e : 16 + 4 * gl + e - 4);
        v[e] = w2[(size_t)row * PF + col] * (ls2 ? ls2[row] : 1.f);
      }
    }
  }
  const uint4 o = {pack_h16x2(v[0], v[1]), pack_h16x2(v[2], v[3]), pack_h16x2(v[4], v[5]), pack_h16x2(v[6], v[7])};
  reinterpret_cast<uint4*>(img)[gi] = o;
}

}  // namespace

extern "C" {

int cs_panel_supported(int C, int mlp_ratio) { return C == PC && mlp_ratio * C == PF; }
size_t cs_panel_image_bytes(int with_outproj) { return (size_t)((with_outproj ? OUT_UNITS : 0) + MLP_UNITS + AHEAD) * UNIT; }

hipError_t cs_panel_pack_launch(const float* wo, const float* ls1, const float* w1, const float* g2, const float* w2, const float* ls2,
                                h16_t* img, hipStream_t st) {
  const int total = ((wo ? OUT_UNITS : 0) + MLP_UNITS + AHEAD) * (UNIT / 16);
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
  if (v) hipLaunchKernelGGL(cs_panel_kernel<true>, dim3(grid), dim3(256), LDS_BYTES, st, *p);
  else hipLaunchKernelGGL(cs_panel_kernel<false>, dim3(grid), dim3(256), LDS_BYTES, st, *p);
  return hipGetLastError();
}

}  // extern "C"
