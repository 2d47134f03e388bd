// Encoder token-panel kernel, second structure (round 6): FOUR waves, one per SIMD, each with the whole 512-register file.
// Same contract as panel.hip (ViT-S: C = 384, MLP 4C = 1536), in ONE launch per encoder layer, for 128-row panels of the residual stream:
//
//   x   += attn_o Wo'^T + bo'                       Dinov2SelfOutput + layer_scale1 + residual   (HF modeling_dinov2.py:249-252,365-370)
//   x   += GELU(LN2(x) W1'^T + b1') W2'^T + b2'     norm2 + Dinov2MLP + layer_scale2 + residual  (HF:373-378, 293-297)
//   u    = f16((x - mean(x)) * rstd(x))             norm1 of the next layer (its gamma/beta live in that layer's packed Wqkv / bias)
//
// Why a second structure.  panel.hip gives every SIMD a pair of waves that share 32 rows: each weight fragment read from LDS feeds ONE MFMA
// (54 fragment reads, 48 one-KiB LDS-DMA pieces and two 8-wave rendezvous per 48 MFMAs and SIMD), and the rendezvous of eight skewed waves
// cost a quarter of every tick (profiles/r05_panel_ablate.txt).  Here the output COLUMNS of the two residual products are split over the waves
// instead of the rows:
//   * wave w owns columns 96 w .. 96 w + 95 of all 128 rows: 4 row blocks x 3 column tiles = 12 accumulator tiles of 32 x 32 (192 registers),
//     which are in turn the residual rows, the out-projection accumulators and the fc2 accumulators.  Its weight fragments (Wo', W2') are
//     PRIVATE -- nobody else reads them -- so they never touch LDS: plain 1-KiB global loads straight into registers, one tick ahead, each
//     fragment feeding FOUR MFMAs (the four row blocks);
//   * fc1 is split by rows: wave w keeps norm2(x) of row block w as 24 B fragments in registers (96) and multiplies a 64-wide hidden slice per
//     tick (48 MFMAs); only W1' is shared by the four waves and streams L2 -> LDS by LDS-DMA (48 KiB per tick, two slots);
//   * the activated slice crosses the waves through LDS as ready-made B fragments (4 written, 16 read per wave and tick) behind ONE 4-wave
//     barrier per tick of 96 MFMAs per SIMD.
// Per 96 MFMAs a wave issues 64 LDS fragment reads (panel.hip: 108 per SIMD), 12 LDS-DMA pieces + 12 global loads (24), and meets one barrier
// of four waves running the same program (four rendezvous of eight).  The price: one instruction stream per SIMD, so every latency is covered
// by software pipelining inside the wave (fragment pools filled 6-8 MFMAs ahead, weights a whole phase ahead).
//
// MFMA v_mfma_f32_32x32x16_f16 / _bf16 with the WEIGHT fragment as the A operand and the activation fragment as the B operand, exactly as in
// panel.hip: D[feature][row]; lane (j = lane & 31, h = lane >> 5) holds, for token row j of the tile, features 32 T + 16 h + r in register r
// (the packed weights carry the row permutation that makes it so), and registers 8 s .. 8 s + 7 of a D tile ARE the B fragment of k-step
// 2 T + s of the next product (contraction order 32 T + 16 h + 8 s + e).
#include "cs_common.h"
#include "panel_shared.h"
#include <atomic>

namespace {

constexpr int PC = 384;               // hidden size
constexpr int PF = 1536;              // MLP hidden
constexpr int KS = PC / 16;           // 24 k-steps over C
constexpr int FRAG = 1024;            // bytes of one MFMA operand fragment (64 lanes x 16 B)
constexpr int ROWS = 128;             // rows per workgroup
constexpr int HT = 64;                // hidden columns per tick
constexpr int NTICK = PF / HT;        // 24
constexpr int PAD_TICKS = 3;          // zero ticks behind the image: the fetches of S2 run unconditionally (W2 one tick, W1 three ticks ahead)
constexpr size_t IMG_WO = (size_t)4 * KS * 3 * FRAG;  // out-projection section: [wave][k-step][column tile]   (288 KiB)
constexpr size_t IMG_W1 = 48 * FRAG;                   // per tick: W1 [k-step 0..23][hidden tile 0..1]            (48 KiB, shared: LDS-DMA)
constexpr size_t IMG_TICK = IMG_W1 + 48 * FRAG;        //           W2 [wave][fc2 k-step 0..3][column tile 0..2]   (48 KiB, private: registers)
// LDS map
constexpr int L_A = 0;                 // 96 KiB: attention-output fragments [row block][k-step] -> norm2(x) fragments -> W1 ring (2 x 48 KiB)
constexpr int L_HB = 96 * 1024;        // 32 KiB: activated slices, 2 buffers x [fc2 k-step 0..3][row block 0..3]
constexpr int L_RED = 128 * 1024;      // 4 KiB: LayerNorm partial sums [pass][wave][row]
constexpr int L_B1 = L_RED + 4096;     // 6 KiB: fc1 bias (fp32)
constexpr int LDS4_BYTES = L_B1 + PF * 4;  // 138 KiB

#ifndef CS_P4_ABL
#define CS_P4_ABL 0   // timing-only ablations of diagnostic builds (results wrong by design): 1 no GELU arithmetic, 2 no W1 fragment reads, 4 no fc1 MFMAs, 8 no counted LDS waits in S1
#endif
#ifdef CS_P4_STAMP
// diagnostic builds only (tools/panel4_phases.py): per (block < 64, wave) phase stamps and per-tick sums; read by cs_panel4_debug_read
__device__ unsigned long long g_p4_dbg[64 * 4 * 16];
__device__ unsigned long long g_p4_dbg2[64 * 4 * 8];
#define P4_NOW(v) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define P4_STAMP(k) do { unsigned long long t_; P4_NOW(t_); if (blockIdx.x < 64 && lane == 0) g_p4_dbg[(blockIdx.x * 4 + wv) * 16 + (k)] = t_; } while (0)
#ifdef CS_P4_STAMPL
#define P4_NOWL(v) P4_NOW(v)
#else
#define P4_NOWL(v) do { v = 0; } while (0)
#endif
#else
#define P4_NOW(v) do { } while (0)
#define P4_NOWL(v) do { } while (0)
#define P4_STAMP(k) do { } while (0)
#endif

__device__ __forceinline__ int panel4_perm(int i) { return 16 * ((i >> 2) & 1) + (i & 3) + 4 * (i >> 3); }

template <bool OUTPROJ, bool BF>
__global__ __launch_bounds__(256) void cs_panel4_kernel(CsPanelParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;
  const int row_base = blockIdx.x * ROWS;
  const unsigned lds0 = (unsigned)(size_t)CS_LDS_PTR(smem);
  const unsigned lane16 = lane * 16;
  const char* img = reinterpret_cast<const char*>(p.img);
  const char* img_mlp = img + (OUTPROJ ? IMG_WO : 0);

  P4_STAMP(0);

  // ---- attention-output rows -> LDS, whole 128-byte lines: six planes [128-byte segment s of a row][row 0..127][128 B].  One LDS-DMA piece =
  //      8 rows x 128 B of one plane (8 full lines; the fragment-shaped copy of round 6's first version fetched 64 separate 16-byte pieces per
  //      instruction).  Lane l of a piece writes row 8 g + l / 8, 16-byte position l % 8, and FETCHES chunk (l % 8) ^ ((row >> 1) & 7) of the
  //      segment: with the same XOR on the read side a B-fragment read (32 rows x one chunk per lane half) is conflict-free.  Wave w copies
  //      its own row block (row groups 4 w .. 4 w + 3). ----
  if constexpr (OUTPROJ) {
    const int rl = lane >> 3, cp = lane & 7;
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const int r = 32 * wv + 8 * gq + rl;                       // row of the panel
      const int c = cp ^ ((r >> 1) & 7);
      const char* so = reinterpret_cast<const char*>(p.attn_o) + (size_t)min(row_base + r, p.M - 1) * (PC * 2) + c * 16;
      const unsigned dst = lds0 + L_A + (32 * wv + 8 * gq) * 128;
      sfor<6>([&](auto S_) {
        constexpr int S = decltype(S_)::value;
        dma_piece<0>(so + S * 128, dst + S * 16384);
      });
    }
  }

  // ---- residual rows: xt[rb][ct][q] = 4 floats of x[row_base + 32 rb + j][96 w + 32 ct + 16 h + 4 q ..].  With the out-projection they are
  //      fetched BEHIND the attention-output pieces and only waited for after it (the projection needs the attention output and the weights;
  //      the 192 KiB of residual rows -- two thirds of the workgroup's input bytes, from HBM -- arrive under its 288 MFMAs) ----
  f32x16_t acc[4][3];
  {
    f32x4_t xt[4][3][4];
    auto load_xt = [&]() {
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) {
        const size_t row = (size_t)min(row_base + rb * 32 + j, p.M - 1);
        const float* xr = p.x + row * PC + 96 * wv + 16 * h;
#pragma unroll
        for (int ct = 0; ct < 3; ++ct)
#pragma unroll
          for (int q = 0; q < 4; ++q) xt[rb][ct][q] = *reinterpret_cast<const f32x4_t*>(xr + 32 * ct + 4 * q);
      }
    };
    if constexpr (!OUTPROJ || (CS_P4_ABL & 32768)) load_xt();
    if constexpr (OUTPROJ) {
      // every LDS-DMA piece of this wave has landed; then everybody's.  The residual rows are requested BEHIND this point (and behind the first
      // weight fragments): issued in front of it, their 50 MB (chip-wide, from HBM) delayed the 25 MB of attention output every workgroup waits for
      if constexpr (CS_P4_ABL & 32768) asm volatile("s_waitcnt vmcnt(48) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      P4_STAMP(1);
      // ---- attention output projection: per k-step 3 private weight fragments (global -> registers) x 4 row-block fragments (LDS) = 12 MFMAs ----
      // The weight fragments are private to the wave: buffer loads straight into registers, PD k-steps ahead (left to itself hipcc issues each
      // load right in front of its use: one L2 round trip per k-step, 23 k cycles for 288 MFMAs); the B fragments one k-step ahead.
      const __amdgpu_buffer_rsrc_t rs_wo = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(img), 0, (int)IMG_WO, 0x00020000);
      const unsigned wo_s = wv * (KS * 3 * FRAG);
      unsigned boff[4];  // this lane's byte offset of its chunk of k-step (ks % 4) inside a row-block plane
#pragma unroll
      for (int kq = 0; kq < 4; ++kq) boff[kq] = j * 128 + (((kq * 2 + h) ^ ((j >> 1) & 7)) * 16);
      const char* bpl = smem + L_A;
      constexpr int PD = 6;
      h16x8_t wq[PD][3], bq[2][4];
      auto wload = [&](auto KS_) {
        constexpr int K = decltype(KS_)::value;
#pragma unroll
        for (int ct = 0; ct < 3; ++ct)
          wq[K % PD][ct] = __builtin_bit_cast(h16x8_t, __builtin_amdgcn_raw_buffer_load_b128(rs_wo, lane16 + ct * FRAG, wo_s + K * 3 * FRAG, 0));
      };
      auto bload = [&](auto KS_) {
        constexpr int K = decltype(KS_)::value;
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) bq[K & 1][rb] = *reinterpret_cast<const h16x8_t*>(bpl + (K / 4) * 16384 + rb * 4096 + boff[K % 4]);
      };
      sfor<PD>([&](auto K_) { wload(K_); });
      bload(IC<0>{});
      CS_SB();
      if constexpr (!(CS_P4_ABL & 32768)) load_xt();
      CS_SB();
      const f32x16_t zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      sfor<KS>([&](auto K_) {
        constexpr int K = decltype(K_)::value;
        if constexpr (K + 1 < KS) bload(IC<K + 1>{});
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
          for (int ct = 0; ct < 3; ++ct) acc[rb][ct] = mfma_32x32x16<BF>(wq[K % PD][ct], bq[K & 1][rb], K == 0 ? zero : acc[rb][ct]);
        if constexpr (K + PD < KS) wload(IC<K + PD>{});
        CS_SB();
      });
      P4_STAMP(2);
      // x + (projection + bias): the residual rows have arrived meanwhile
      const float* bsrc = p.bo + 96 * wv + 16 * h;
#pragma unroll
      for (int ct = 0; ct < 3; ++ct)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4_t b4 = *reinterpret_cast<const f32x4_t*>(bsrc + 32 * ct + 4 * q);
#pragma unroll
          for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[rb][ct][4 * q + i] += xt[rb][ct][q][i] + b4[i];
        }
    } else {
      P4_STAMP(1);
      P4_STAMP(2);
#pragma unroll
      for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int ct = 0; ct < 3; ++ct)
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[rb][ct][4 * q + i] = xt[rb][ct][q][i];
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // (the fc1 bias is in LDS)
      asm volatile("" ::: "memory");
    }
  }

  // ---- LayerNorm statistics of complete rows: a row's 384 columns live in lanes j / j + 32 of all four waves.  Two passes (mean, then
  //      squared deviations), each: 48 values in the lane + the other half of the wave + the other three waves through LDS, summed in a fixed order ----
  float* red = reinterpret_cast<float*>(smem + L_RED);
  auto row_stats = [&](float (&mean)[4], float (&rstd)[4], int j, int h) {
    // per wave: mean and sum of squared deviations of its 96 columns of each row (two passes in registers, no exchange); then ONE exchange of the
    // four waves' (mean, M2) pairs and their exact combination  M2 = sum M2_w + 96 sum (mean_w - mean)^2  -- as accurate as the two-pass form
    // over the whole row (no E[x^2] - mean^2 cancellation) at one barrier instead of three.  Fixed order: a row's result does not depend on
    // where the row sits.
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
      float t = 0.f;
#pragma unroll
      for (int ct = 0; ct < 3; ++ct)
#pragma unroll
        for (int r = 0; r < 16; r += 4) t += (acc[rb][ct][r] + acc[rb][ct][r + 1]) + (acc[rb][ct][r + 2] + acc[rb][ct][r + 3]);
      const float mw = add_other_half(t) * (1.0f / 96.0f);
      float q = 0.f;
#pragma unroll
      for (int ct = 0; ct < 3; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float d = acc[rb][ct][r] - mw;
          q = fmaf(d, d, q);
        }
      q = add_other_half(q);
      if (h == 0) { red[wv * ROWS + rb * 32 + j] = mw; red[4 * ROWS + wv * ROWS + rb * 32 + j] = q; }
    }
    __syncthreads();
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
      const float* r0 = red + rb * 32 + j;
      const float m0 = r0[0], m1 = r0[ROWS], m2 = r0[2 * ROWS], m3 = r0[3 * ROWS];
      const float m = ((m0 + m1) + (m2 + m3)) * 0.25f;
      const float* q0 = r0 + 4 * ROWS;
      const float d0 = m0 - m, d1 = m1 - m, d2 = m2 - m, d3 = m3 - m;
      const float M2 = ((q0[0] + q0[ROWS]) + (q0[2 * ROWS] + q0[3 * ROWS])) + 96.0f * ((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3));
      mean[rb] = m;
      rstd[rb] = 1.0f / sqrtf(M2 * (1.0f / PC) + p.eps);
    }
    __syncthreads();  // (the scratch may be rewritten by the next call)
  };

  // ---- norm2(x) -> B fragments of fc1: k-step 2 T + s of row block rb = registers 8 s .. 8 s + 7 of tile (rb, T = 3 w + ct) ----
  h16x8_t xf[KS];
  {
    float mean[4], rstd[4];
    row_stats(mean, rstd, j, h);  // (its barriers also order the out-projection's reads of L_A before the writes below)
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
      const float nb = -mean[rb] * rstd[rb];
#pragma unroll
      for (int ct = 0; ct < 3; ++ct)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          float v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaf(acc[rb][ct][8 * s + e], rstd[rb], nb);
          *reinterpret_cast<u32x4_t*>(smem + L_A + (rb * KS + 2 * (3 * wv + ct) + s) * FRAG + lane16) = pack8<BF>(v);
        }
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) xf[ks] = *reinterpret_cast<const h16x8_t*>(smem + L_A + (wv * KS + ks) * FRAG + lane16);
    __syncthreads();  // every wave holds its fragments: L_A becomes the W1 ring
  }
  // ---- fc1 bias -> LDS (plain accesses: no LDS-DMA is in flight here; kept out of the kernel's first instructions, where its vmcnt(0)
  //      would wait for the attention-output pieces and the residual rows) ----
  for (int i = tid; i < PF / 4; i += 256)
    reinterpret_cast<f32x4_t*>(smem + L_B1)[i] = reinterpret_cast<const f32x4_t*>(p.b1)[i];
  // the fc2 accumulators start at residual + bias.  The bias enters through the matrix pipe (one MFMA per tile: b2 split hi + lo in contraction
  // slots 0 / 1 against a fragment of ones): a vector add on the accumulators here, with norm2(x) live beside them, overflowed the arch
  // VGPRs and hipcc spilled 22 accumulator registers around the prologue
  {
    const unsigned lb = fresh_lane();
    const int jb = lb & 31, hb_ = lb >> 5;
    const h16x8_t ones_b = __builtin_bit_cast(h16x8_t, u32x4_t{hb_ == 0 ? (BF ? 0x3F803F80u : 0x3C003C00u) : 0u, 0u, 0u, 0u});
#pragma unroll
    for (int ct = 0; ct < 3; ++ct) {
      const float bv = p.b2[96 * wv + 32 * ct + panel4_perm(jb)];
      const unsigned hi = BF ? (unsigned)f2bf(bv) : (unsigned)f2h(bv);
      const float rest = bv - (BF ? bf2f((h16_t)hi) : h2f((h16_t)hi));
      const unsigned lo = BF ? (unsigned)f2bf(rest) : (unsigned)f2h(rest);
      const h16x8_t bfrag = __builtin_bit_cast(h16x8_t, u32x4_t{hb_ == 0 ? (hi | (lo << 16)) : 0u, 0u, 0u, 0u});
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) acc[rb][ct] = mfma_32x32x16<BF>(bfrag, ones_b, acc[rb][ct]);
    }
  }

  // =========================================================================================================================
  // MLP: 24 ticks of 64 hidden columns, software-pipelined over three ticks (see `tick` below): per iteration fc1 of tick t + 1, the GELU of
  // tick t and fc2 of tick t - 1, one barrier.
  // =========================================================================================================================
  h16x8_t w2r[12];       // this wave's W2 fragments of the current tick: [kk][ct]
  constexpr int WPN = BF ? 6 : 8;  // (the bf16 form keeps two fc1 accumulator buffers: a smaller pool keeps it out of scratch)
  h16x8_t wp[WPN];       // rolling pool of W1 fragments: fragment M in wp[M % WPN], WPN MFMAs ahead
  f32x16_t hA[2], hB[2]; // fc1 accumulators of even / odd ticks: [hidden tile]
  unsigned xp[16];       // the previous tick's pre-activations as packed halves: xp[4 (2 ht + s) + q] = (r = 8 s + 2 q, + 1) of tile ht
  PkGelu pg;
  const PkGeluK kk_ = pk_gelu_consts();
  // lane-derived values of the MLP phase are formed from a fresh lane id (opaque to the compiler), those of the epilogue likewise: nothing
  // lane-derived has to stay in a register -- or in scratch -- across the loop, which runs at the 512-register limit.  (A kernel with a
  // private segment ran the SAME loop instructions at half the speed: tools/panel4_phases.py, EXPERIMENTS.md round 6.)
  const unsigned lm = fresh_lane();
  const unsigned lm16 = lm * 16;
  const int hm = lm >> 5;
  const unsigned ring = lds0 + L_A + lm16;
  const unsigned hbw = lds0 + L_HB + wv * FRAG + lm16;   // + (buf * 16 + 4 kkf) * FRAG: this wave's row block
  const unsigned hbr = lds0 + L_HB + lm16;               // + (buf * 16 + 4 kkf + rb) * FRAG
  const unsigned b1g = lds0 + L_B1 + 4 * panel4_perm(lm & 31);  // + (64 t + 32 ht) * 4: the bias of the hidden unit in MFMA row j of a tile
  // B fragment of the bias k-step: 1.0 in contraction slots 0 and 1 (lane half 0, elements 0 and 1), 0 elsewhere
  const h16x8_t ones = __builtin_bit_cast(h16x8_t, u32x4_t{hm == 0 ? (BF ? 0x3F803F80u : 0x3C003C00u) : 0u, 0u, 0u, 0u});

  // The MLP section of the image through a buffer descriptor: a fetch is (descriptor, this lane's 32-bit offset, a scalar offset, an immediate
  // < 4 KiB) -- no 64-bit address pair per lane has to be formed or kept (the flat forms cost 6 VGPRs and a v_add_co / v_addc pair per fetch;
  // in the bf16 form one such pair was spilled into the loop).  LDS-DMA piece K of a 12-piece run: groups of four share a scalar offset.
  const __amdgpu_buffer_rsrc_t rs_mlp = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(img_mlp), 0, (int)((NTICK + PAD_TICKS) * IMG_TICK), 0x00020000);
  auto dma_w1 = [&](auto K_, unsigned soff, unsigned dst) {  // piece K of this wave's 12: image offset soff + K KiB -> LDS dst + K KiB
    constexpr int K = decltype(K_)::value;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_mlp, (__attribute__((address_space(3))) void*)(size_t)(dst + (K & ~3) * FRAG), 16, lm16,
                                             soff + (K & ~3) * FRAG, (K & 3) * FRAG, 0);
  };
  auto w1_issue = [&](int t) {  // this wave's 12 pieces of tick t's W1 slice, all at once (prologue)
    const unsigned dst = lds0 + L_A + (t & 1) * (48 * FRAG) + wv * 12 * FRAG;
    const unsigned soff = t * (unsigned)IMG_TICK + wv * 12 * FRAG;
    sfor<12>([&](auto K_) { dma_w1(K_, soff, dst); });
  };
  // the two packed registers of GELU group g (of 8): xp[2 g], xp[2 g + 1] = values r0 .. r0 + 3 of tile g / 4, r0 = 8 ((g / 2) & 1) + 4 (g & 1)
  auto pack_group = [&](auto G_, const f32x16_t (&ha)[2]) {
    constexpr int g = decltype(G_)::value, ht = g / 4, r0 = 8 * ((g / 2) & 1) + 4 * (g & 1);
    xp[2 * g] = pack_h16x2(ha[ht][r0], ha[ht][r0 + 1]);
    xp[2 * g + 1] = pack_h16x2(ha[ht][r0 + 2], ha[ht][r0 + 3]);
  };
  // GELU block B (0..47) of the tick: group B / 6 = packed registers 2 G, 2 G + 1, block B % 6 of its six
  auto gelu_block = [&](auto B_, const f32x16_t (&ha)[2]) {
    constexpr int B = decltype(B_)::value;
    constexpr int G = B / 6, ht = G / 4, r0 = 8 * ((G / 2) & 1) + 4 * (G & 1);
    if constexpr (!(CS_P4_ABL & 1))
      pk_gelu_block<B % 6, BF>(pg, xp[2 * G], xp[2 * G + 1], kk_, ha[ht][r0], ha[ht][r0 + 1], ha[ht][r0 + 2], ha[ht][r0 + 3]);
  };
  // One iteration = 98 MFMA gaps and ONE barrier.  Iteration t, in this order:
  //   F2  fc2 of tick t - 1 from hand-off buffer (t - 1) & 1 (written before the previous barrier): k-step kk, row block rb, column tile ct; W2
  //       fragments in registers, B fragment (kk, rb) from LDS once per 3 MFMAs -- 48 gaps.  First, because it needs nothing but three LDS reads
  //       behind the barrier: the packing of the previous tick's pre-activations, the fc1 bias fragments and the first W1 fragment reads all
  //       happen under its MFMAs.  It also carries every fetch of the iteration: W1 of tick t + 2 into ring slot t & 1 (free since the previous
  //       barrier; 12 LDS-DMA pieces) and W2 of tick t into the registers just used (12 loads, consumed an iteration on) -- all issued in the
  //       first half, so the closing vmcnt(0) finds them landed;
  //   F1  fc1 of tick t + 1 into hn (W1 from ring slot (t + 1) & 1: one LDS read per MFMA, WPN ahead; the bias through one more k-step per
  //       tile) -- 50 gaps;
  //   G   GELU of tick t (its pre-activations `ha`, packed to halves at the start) -- one block of the packed-half GELU every second gap of the
  //       whole iteration; the activated slice of this wave's row block leaves as 4 B fragments into hand-off buffer t & 1 at the end.
  auto tick = [&](auto F1_, auto G_, auto F2_, f32x16_t (&hn)[2], const f32x16_t (&ha)[2], int t) {
    constexpr bool F1 = decltype(F1_)::value, G = decltype(G_)::value, F2 = decltype(F2_)::value;
    const unsigned cur = ring + ((t + 1) & 1) * (48 * FRAG);
    const unsigned hr = hbr + ((t + 1) & 1) * (16 * FRAG);
    const unsigned w1s = (unsigned)(t + 2) * (unsigned)IMG_TICK + wv * 12 * FRAG;            // scalar offsets into the MLP section
    const unsigned w1d = lds0 + L_A + (t & 1) * (48 * FRAG) + wv * 12 * FRAG;
    const unsigned w2s = (unsigned)t * (unsigned)IMG_TICK + (unsigned)IMG_W1 + wv * 12 * FRAG;
    constexpr int HBN = BF ? 3 : 4;  // (bf16 form: one register quad less, see WPN)
    h16x8_t hb[HBN];  // rolling pool of B fragments of fc2: fragment g = 4 kk + rb lives in hb[g % HBN], HBN - 1 groups ahead
    float bv[2];      // fc1 bias of this lane's MFMA row, tiles 0 / 1 of tick t + 1
    constexpr int PRE = F1 ? WPN : 0;   // LDS reads of the F1 phase issued behind the last fc2 fragment read
    unsigned bfr0[2]; // ... as register 0 of the bias k-step's weight fragment (hi | lo << 16 in lane half 0)
    unsigned b1a = b1g + (t + 1) * (HT * 4);
    auto bias_reads = [&]() {
      asm volatile("ds_read_b32 %0, %1" : "=&v"(bv[0]) : "v"(b1a) : "memory");
      asm volatile("ds_read_b32 %0, %1 offset:128" : "=&v"(bv[1]) : "v"(b1a) : "memory");
    };
    auto bias_prep = [&]() {
#pragma unroll
      for (int ht = 0; ht < 2; ++ht) {
        const unsigned hi = BF ? (unsigned)f2bf(bv[ht]) : (unsigned)f2h(bv[ht]);
        const float rest = bv[ht] - (BF ? bf2f((h16_t)hi) : h2f((h16_t)hi));
        const unsigned lo = BF ? (unsigned)f2bf(rest) : (unsigned)f2h(rest);
        bfr0[ht] = hm == 0 ? (hi | (lo << 16)) : 0u;
      }
    };
    auto f1_first_reads = [&]() { sfor<WPN>([&](auto F_) { lds_read1<decltype(F_)::value * FRAG>(cur, wp[decltype(F_)::value]); }); };
    // the bias reads lead the iteration's LDS queue (older than every fragment read: the first counted wait retires them)
    if constexpr (F1) bias_reads();
    if constexpr (F2) sfor<HBN - 1>([&](auto G_) { constexpr int Gi = decltype(G_)::value; lds_read1<Gi * FRAG>(hr, hb[Gi]); });
    if constexpr (G) pack_group(IC<0>{}, ha);
    // ---- F2 phase (with the fetches and GELU blocks 0..23) ----
    sfor<16>([&](auto G_) {
      constexpr int Gi = decltype(G_)::value;
      constexpr int kk = Gi / 4, rb = Gi % 4;
      if constexpr (F2) {
        if constexpr (Gi + HBN - 1 < 16 && !(CS_P4_ABL & 2)) lds_read1<(Gi + HBN - 1) * FRAG>(hr, hb[(Gi + HBN - 1) % HBN]);
        if constexpr (Gi + HBN - 1 == 15 && F1) f1_first_reads();   // behind the last fc2 fragment read
        constexpr int BEHIND = (Gi + HBN - 1 < 16) ? HBN - 1 + (Gi + HBN - 1 == 15 ? PRE : 0) : 15 - Gi + PRE;
        if constexpr (!(CS_P4_ABL & 8)) CS_LGKM(BEHIND);
      }
      sfor<3>([&](auto C_) {
        constexpr int ct = decltype(C_)::value;
        constexpr int M2 = Gi * 3 + ct;
        if constexpr (F2) acc[rb][ct] = mfma_32x32x16<BF>(w2r[kk * 3 + ct], hb[Gi % HBN], acc[rb][ct]);
        if constexpr (F2 || G) {
          // behind the last use of this weight fragment: refill it with tick t's (consumed by the fc2 phase of the next iteration)
          if constexpr (rb == 3 && !(CS_P4_ABL & 16))
            w2r[kk * 3 + ct] = __builtin_bit_cast(h16x8_t, __builtin_amdgcn_raw_buffer_load_b128(rs_mlp, lm16 + ((kk * 3 + ct) & 3) * FRAG, w2s + ((kk * 3 + ct) & ~3) * FRAG, 0));
        }
        if constexpr (F1 && rb != 3 && ct == 1 && !(CS_P4_ABL & 16)) dma_w1(IC<kk * 3 + rb>{}, w1s, w1d);   // 12 gaps: one W1 piece each
        if constexpr (G && M2 % 2 == 0) gelu_block(IC<M2 / 2>{}, ha);
        // the packing of groups 1..7 (before the fc1 phase overwrites the accumulators they are packed from) and the bias fragments: odd gaps
        if constexpr (G && M2 % 4 == 1 && M2 / 4 >= 1 && M2 / 4 <= 7) pack_group(IC<M2 / 4>{}, ha);
        // (only behind this phase's counted waits -- F2: in the prologue iteration, which has none, an early bias_prep would read the registers
        //  before the LDS data has landed, and hipcc merges the later, correct one with it: common subexpressions of the same asm outputs)
        if constexpr (F1 && F2 && M2 == 3) bias_prep();
        CS_SB();
      });
    });
    // ---- F1 phase (with GELU blocks 24..47) ----
    if constexpr (F1) {
      // fc1 bias through the matrix pipe: one more k-step per tile whose weight fragment carries b1 (split hi + lo: exact to 2^-22 / 2^-16) in
      // contraction slots 0 and 1 and whose activation fragment is 1 there -- the accumulators start from the inline constant 0, nothing is
      // written into them by the vector unit (the MFMA results live in accumulator registers: 32 v_accvgpr_write per tick otherwise)
      if constexpr (!F2) { f1_first_reads(); CS_LGKM(WPN); bias_prep(); }
#pragma unroll
      for (int ht = 0; ht < 2; ++ht) {
        const f32x16_t zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        hn[ht] = mfma_32x32x16<BF>(__builtin_bit_cast(h16x8_t, u32x4_t{bfr0[ht], 0u, 0u, 0u}), ones, zero);
      }
    }
    sfor<48>([&](auto M_) {
      constexpr int M = decltype(M_)::value;
      if constexpr (F1) {
        // consumption order: the two hidden tiles alternate; fragment M of the tick's W1 slice (in this order in the image) = tile ht, k-step ks.
        // Fragment M is complete when at most (the LDS reads issued behind it) are outstanding: WPN - 1 in the steady state, fewer at the tail
        constexpr int ht = M % 2, ks = M / 2;
        constexpr int BEHIND = (M + WPN <= 48) ? WPN - 1 : 47 - M;
        if constexpr (!(CS_P4_ABL & 8)) CS_LGKM(BEHIND);
        hn[ht] = mfma_32x32x16<BF>(wp[M % WPN], xf[ks], hn[ht]);
        if constexpr (M + WPN < 48 && !(CS_P4_ABL & 2)) lds_read1<(M + WPN) * FRAG>(cur, wp[M % WPN]);
      }
      if constexpr (G && M % 2 == 0) gelu_block(IC<24 + M / 2>{}, ha);
      CS_SB();
    });
    if constexpr (G) {
      const unsigned a = hbw + (t & 1) * (16 * FRAG);
      sfor<4>([&](auto K_) {
        constexpr int K = decltype(K_)::value;
        lds_write16<K * 4 * FRAG>(a, u32x4_t{xp[4 * K], xp[4 * K + 1], xp[4 * K + 2], xp[4 * K + 3]});
      });
    }
    // every fetch of the iteration was issued in its first half: the LDS-DMA pieces of the ring slot the next iteration reads have landed
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    CS_SB();
  };
  using T_ = std::true_type;
  using F_ = std::false_type;

  // prologue: W1 of tick 0; "iteration -1" = fc1 of tick 0 (and the fetch of W1 of tick 1)
  P4_STAMP(3);
  // iteration 0 runs the full body: its fc2 phase ("tick -1") multiplies zero weight fragments with a zeroed hand-off buffer -- 48 MFMAs that
  // add exact zeros -- instead of being one more instance of the body (whose register allocation spilled in the bf16 form)
  {
    const u32x4_t z4 = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int q = 0; q < 4; ++q) *reinterpret_cast<u32x4_t*>(smem + L_HB + 16 * FRAG + (q * 256 + tid) * 16) = z4;
#pragma unroll
    for (int f = 0; f < 12; ++f) w2r[f] = __builtin_bit_cast(h16x8_t, z4);
  }
  w1_issue(0);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  tick(T_{}, F_{}, F_{}, hA, hA, -1);
  P4_STAMP(4);
  // (fp16 mode: the GELU works on the packed halves alone, so fc1 of the next tick may overwrite the accumulators it was packed from; the bf16
  //  form's relu reads the fp32 pre-activations: two buffers, alternating)
  if constexpr (BF) {
    for (int t = 0; t < NTICK - 2; t += 2) {
      tick(T_{}, T_{}, T_{}, hB, hA, t);
      tick(T_{}, T_{}, T_{}, hA, hB, t + 1);
    }
    tick(T_{}, T_{}, T_{}, hB, hA, NTICK - 2);
    tick(F_{}, T_{}, T_{}, hA, hB, NTICK - 1);
  } else {
    tick(T_{}, T_{}, F_{}, hA, hA, 0);  // (no fc2 phase yet; the bf16 form runs the full body on zero operands instead: see above)
    for (int t = 1; t < NTICK - 1; ++t) tick(T_{}, T_{}, T_{}, hA, hA, t);
    tick(F_{}, T_{}, T_{}, hA, hA, NTICK - 1);
  }
  tick(F_{}, F_{}, T_{}, hA, hA, NTICK);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  P4_STAMP(5);

  // ---- epilogue: x rows back to memory, the next layer's normalised rows.  A lane holds 64 B of a row per tile: stored directly, every store
  //      instruction would touch 64 separate 16-byte pieces in 32 rows (measured: 29 k cycles for the 48 stores of a wave).  Each 32 x 32 tile
  //      goes through a wave-private LDS patch instead (rows padded to 144 B: conflict-free 16-byte writes of a lane's row, row-contiguous reads)
  //      and leaves as whole 128-byte lines, 8 rows per store instruction.  Wave-private: no barrier, the LDS queue keeps a wave's accesses in order. ----
  const unsigned le = fresh_lane();
  const int je = le & 31, he = le >> 5;
  {
    char* patch = smem + L_A + wv * (16 * 1024);                  // two patches of 32 x 144 B per wave (the ring is idle: last barrier passed)
    char* pw = patch + je * 144 + he * 64;                        // this lane's 64 bytes of row je
    const char* pr = patch + (le >> 3) * 144 + (le & 7) * 16;     // row le / 8 (+ 8 per pass), 16-byte piece le % 8
    char* xg = reinterpret_cast<char*>(p.x) + ((size_t)row_base * PC + 96 * wv) * 4 + (size_t)(le >> 3) * (PC * 4) + (le & 7) * 16;
    const int r0 = row_base + (le >> 3);
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int ct = 0; ct < 3; ++ct) {
        const int pb = ((rb * 3 + ct) & 1) * (32 * 144);
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<f32x4_t*>(pw + pb + 16 * q) = f32x4_t{acc[rb][ct][4 * q], acc[rb][ct][4 * q + 1], acc[rb][ct][4 * q + 2], acc[rb][ct][4 * q + 3]};
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
          const f32x4_t v = *reinterpret_cast<const f32x4_t*>(pr + pb + ps * 8 * 144);
          if (r0 + rb * 32 + ps * 8 < p.M) *reinterpret_cast<f32x4_t*>(xg + (size_t)(rb * 32 + ps * 8) * (PC * 4) + ct * 128) = v;
        }
      }
  }
  P4_STAMP(6);
  if (p.u_out) {
    __syncthreads();
    float mean[4], rstd[4];
    row_stats(mean, rstd, je, he);
    // normalised 16-bit rows: 64 B of a row per tile; patch rows padded to 80 B, 16 rows (of 64 B) per store instruction
    char* patch = smem + L_A + wv * (16 * 1024);
    char* pw = patch + je * 80 + he * 32;
    const char* pr = patch + (le >> 2) * 80 + (le & 3) * 16;
    char* ug = reinterpret_cast<char*>(p.u_out) + ((size_t)row_base * PC + 96 * wv) * 2 + (size_t)(le >> 2) * (PC * 2) + (le & 3) * 16;
    const int r0 = row_base + (le >> 2);
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
      const float nb = -mean[rb] * rstd[rb];
#pragma unroll
      for (int ct = 0; ct < 3; ++ct) {
        const int pb = ((rb * 3 + ct) & 1) * (32 * 80);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          float v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaf(acc[rb][ct][8 * s + e], rstd[rb], nb);
          *reinterpret_cast<u32x4_t*>(pw + pb + 16 * s) = pack8<BF>(v);
        }
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
          const u32x4_t v = *reinterpret_cast<const u32x4_t*>(pr + pb + ps * 16 * 80);
          if (r0 + rb * 32 + ps * 16 < p.M) *reinterpret_cast<u32x4_t*>(ug + (size_t)(rb * 32 + ps * 16) * (PC * 2) + ct * 64) = v;
        }
      }
    }
  }
#ifdef CS_P4_STAMP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  P4_STAMP(7);
  if (blockIdx.x < 64 && lane == 0) g_p4_dbg[(blockIdx.x * 4 + wv) * 16 + 12] = __builtin_amdgcn_s_memrealtime();
#endif
}

// ---- weight image.  One thread per 16-byte fragment element (8 halves): [fragment][lane 0..63].  Lane (i = lane & 31, h = lane >> 5) of an
//      A-operand fragment holds MFMA row i of its 32-feature tile -- output feature perm(i) of the tile, so that D register r of lane half h is
//      feature 16 h + r -- and 8 contraction indices of its 16-wide k-step:
//        natural  16 ks + 8 h + e                        (out-projection: B fragments come from memory)
//        tiled    32 (ks >> 1) + 16 h + 8 (ks & 1) + e   (fc1 / fc2: B fragments are accumulator tiles)
//      Sections: [out-projection: wave w, k-step ks, column tile ct -> output tile T = 3 w + ct]
//                [24 ticks: W1 fragment 2 ks + ht = hidden tile 2 t + ht, k-step ks | W2: wave w, fc2 k-step kk, column tile ct =
//                 output tile 3 w + ct, hidden k-step 4 t + kk] ----
template <bool BF>
__global__ __launch_bounds__(256) void cs_panel4_pack_kernel(const float* __restrict__ wo, const float* __restrict__ ls1,
                                                             const float* __restrict__ w1, const float* __restrict__ g2,
                                                             const float* __restrict__ w2, const float* __restrict__ ls2,
                                                             h16_t* __restrict__ img) {
  const long long n_out16 = wo ? (long long)(IMG_WO / 16) : 0;
  const long long total16 = n_out16 + (long long)(NTICK + PAD_TICKS) * (IMG_TICK / 16);
  const long long gi = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gi >= total16) return;
  float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int lane = (int)(gi & 63), i = lane & 31, h = lane >> 5;
  if (gi < n_out16) {
    const int f = (int)(gi >> 6);  // (w * 24 + ks) * 3 + ct
    const int ct = f % 3, ks = (f / 3) % KS, w = f / (3 * KS);
    const int rowi = 32 * (3 * w + ct) + panel4_perm(i);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = wo[(size_t)rowi * PC + 16 * ks + 8 * h + e] * (ls1 ? ls1[rowi] : 1.f);
  } else {
    const long long g2i = gi - n_out16;
    const int t = (int)(g2i / (IMG_TICK / 16));
    const int f = (int)((g2i - (long long)t * (IMG_TICK / 16)) >> 6);  // 0..95
    if (t >= NTICK) {
      // padding: zeros
    } else if (f < 48) {
      const int ht = f % 2, ks = f / 2;
      const int rowi = HT * t + 32 * ht + panel4_perm(i);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int col = 32 * (ks >> 1) + 16 * h + 8 * (ks & 1) + e;
        v[e] = w1[(size_t)rowi * PC + col] * (g2 ? g2[col] : 1.f);
      }
    } else {
      const int fp = f - 48;  // w * 12 + kk * 3 + ct
      const int ct = fp % 3, kk = (fp / 3) % 4, w = fp / 12;
      const int rowi = 32 * (3 * w + ct) + panel4_perm(i);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int col = HT * t + 32 * (kk >> 1) + 16 * h + 8 * (kk & 1) + e;
        v[e] = w2[(size_t)rowi * PF + col] * (ls2 ? ls2[rowi] : 1.f);
      }
    }
  }
  const uint4 o = {pack_o16x2<BF>(v[0], v[1]), pack_o16x2<BF>(v[2], v[3]), pack_o16x2<BF>(v[4], v[5]), pack_o16x2<BF>(v[6], v[7])};
  reinterpret_cast<uint4*>(img)[gi] = o;
}

template <bool OUTPROJ, bool BF>
hipError_t panel4_launch_t(const CsPanelParams* p, hipStream_t st) {
  static std::atomic<bool> attr_done[16];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return hipErrorInvalidDevice;
  if (!attr_done[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(cs_panel4_kernel<OUTPROJ, BF>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS4_BYTES);
    if (e != hipSuccess) return e;
    attr_done[dev] = true;
  }
  const int grid = (p->M + ROWS - 1) / ROWS;
  hipLaunchKernelGGL((cs_panel4_kernel<OUTPROJ, BF>), dim3(grid), dim3(256), LDS4_BYTES, st, *p);
  return hipGetLastError();
}

}  // namespace

extern "C" {

#ifdef CS_P4_STAMP
int cs_panel4_debug_read(unsigned long long* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_p4_dbg), sizeof(g_p4_dbg)); }
int cs_panel4_debug_read2(unsigned long long* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_p4_dbg2), sizeof(g_p4_dbg2)); }
#endif

size_t cs_panel4_image_bytes(int with_outproj) { return (with_outproj ? IMG_WO : 0) + (size_t)(NTICK + PAD_TICKS) * IMG_TICK; }

hipError_t cs_panel4_pack_launch(const float* wo, const float* ls1, const float* w1, const float* g2, const float* w2, const float* ls2,
                                 h16_t* img, int bf16, hipStream_t st) {
  const int total = (int)(cs_panel4_image_bytes(wo ? 1 : 0) / 16);
  if (bf16) hipLaunchKernelGGL(cs_panel4_pack_kernel<true>, dim3((total + 255) / 256), dim3(256), 0, st, wo, ls1, w1, g2, w2, ls2, img);
  else hipLaunchKernelGGL(cs_panel4_pack_kernel<false>, dim3((total + 255) / 256), dim3(256), 0, st, wo, ls1, w1, g2, w2, ls2, img);
  return hipGetLastError();
}

hipError_t cs_panel4_launch(const CsPanelParams* p, hipStream_t st) {
  if (p->attn_o) return p->bf16 ? panel4_launch_t<true, true>(p, st) : panel4_launch_t<true, false>(p, st);
  return p->bf16 ? panel4_launch_t<false, true>(p, st) : panel4_launch_t<false, false>(p, st);
}

}  // extern "C"
