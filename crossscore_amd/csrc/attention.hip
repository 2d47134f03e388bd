// Flash-style fused attention for gfx950: O = softmax(Q K^T / sqrt(dh)) V, no mask, fp16 in/out, fp32 softmax.
//
// One kernel serves the three attention sites of the CrossScore forward:
//   * DINOv2 encoder self-attention, dh = 64, per image over T = 1 + h*w tokens   (HF modeling_dinov2.py:153-234)
//   * decoder self-attention, 8 heads, dh = C/8 in {48, 96}                        (transformer.py:182-192)
//   * query <-> multi-reference cross-attention, Lk = N*h*w keys                   (transformer.py:195-205)
// Q/K/V are addressed through (row stride, batch stride, head*dh column offset) so the packed projections the
// GEMMs emit ([tokens][3C], [tokens][4C]) are consumed in place -- no head-split copies.
//
// Structure (wave64, MFMA 32x32x16 fp16): a workgroup = 4 waves = 128 query rows, each wave 32 rows; keys are
// streamed in 64-key tiles, K and V staged HBM -> VGPR -> LDS (issue early / write late), double buffered, one
// barrier per tile.  QK^T is computed "swapped" (S^T = K Q^T) so each lane owns one query column: row max and
// row sum are lane-local plus one cross-half exchange, and the S^T accumulator IS the B operand of the PV MFMA
// (O^T = V^T P^T) after an in-register fp16 pack -- P never touches LDS.  The softmax reference point moves lazily (see
// kTau below): on most tiles the per-score VALU work is max, exp2, sum and the fp16 pack only.  V^T fragments come from a row-major V
// image through ds_read_b64_tr_b16 (hardware transposed read); LDS row strides are chosen bank-conflict free
// (K: odd number of 16-B slots; V: 192 B so the four rows of a transposed block hit disjoint bank windows).
#include "cs_common.h"
#include <atomic>
#include <math.h>

#ifdef CS_ATTN_STAMP
// phase clocks of the tile loop (tools/attn_phases.py): per (block < 64, wave) the cycles summed over all tiles of
// 0: K reads + QK^T + max (S available)  1: exp2 / sums / pack  2: V reads + PV issue  3: tile write (waits for the
// global loads)  4: barrier;  5: loop start (10-ns ticks), 6: the loop in shader cycles, 7: the loop in 10-ns ticks
__device__ unsigned long long g_attn_dbg[64 * 4 * 8];
#define CS_TS(k) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
    ph[k] += now_ - tlast; tlast = now_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define CS_TS(k) do { } while (0)
#endif

namespace {

template <int DH>
struct AttnCfg {
  static constexpr int KS = DH / 16;                 // QK^T k-steps (MFMA K = 16)
  static constexpr int DT = (DH + 31) / 32;          // 32-wide d tiles of O^T
  static constexpr int CH = DH / 8;                  // 16-byte chunks per K/V row
  static constexpr int KROW = DH * 2 + 16;           // bytes; odd multiple of 16 -> ds_read_b128 conflict free
  static constexpr int VROW = (DH <= 16) ? 64 : (DH <= 96 ? 192 : (DH <= 128 ? 320 : 448)); // bytes; >= DT*64 and == 16 or 48 dwords (mod 64)
  static_assert(VROW >= DT * 64 && (VROW / 4) % 32 == 16, "V row pitch");
  static constexpr int KTILE = 64 * KROW;
  static constexpr int VTILE = 64 * VROW;
  static constexpr int STAGE = KTILE + VTILE;
  static constexpr int NIT = (64 * CH + 255) / 256;  // 16-byte chunks per thread per tile (K and V each)
};

template <int DH, bool BF>
__global__ __launch_bounds__(256, DH > 128 ? 1 : 2) void cs_attn_kernel(CsAttnParams p) {  // (dh = 192: 108 KB of LDS, one workgroup per CU anyway)
  using Cfg = AttnCfg<DH>;
  constexpr int KS = Cfg::KS, DT = Cfg::DT, CH = Cfg::CH, NIT = Cfg::NIT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31;   // MFMA row/col index inside a 32-tile
  const int hh = lane >> 5;  // lane half
  // XCD-aware 1-D grid: the QB query blocks of one (batch, head) re-read the same K/V, so they are placed on one XCD
  // (blocks b, b+8, .. share an XCD under round-robin dispatch; speed only): group g = gi*8 + (b%8), j = b/8 = gi*QB + qb.
  const int QB = (p.Lq + 127) / 128;
  const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
  const int gi = jj / QB, qb = jj - gi * QB;
  const int grp = gi * 8 + xcd;
  if (grp >= p.heads * p.nbatch) return;  // whole block exits together (before any barrier)
  const int head = grp % p.heads;
  const int bat = grp / p.heads;
  const int q0 = qb * 128 + wv * 32;

  const h16_t* Qb = p.Q + (size_t)bat * p.q_bs + head * DH;
  const h16_t* Kb = p.K + (size_t)bat * p.k_bs + head * DH;
  const h16_t* Vb = p.V + (size_t)bat * p.v_bs + head * DH;

  // ---- Q^T fragments (B operand of S^T = K Q^T): lane (r,hh) holds Q[q0+r][16s + 8hh .. +7] ----
  h16x8_t qf[KS];
  {
    int qr = q0 + r;
    qr = qr < p.Lq ? qr : p.Lq - 1;
    const h16_t* qp = Qb + (size_t)qr * p.ldq + 8 * hh;
#pragma unroll
    for (int s = 0; s < KS; ++s) qf[s] = *reinterpret_cast<const h16x8_t*>(qp + 16 * s);
  }

  // ---- staging maps: chunk c = tid + it*256 -> (key = c / CH, ch = c % CH); the last pass may be partial
  //      (whole waves idle: the guard is wave-uniform) ----
  constexpr bool kFullLast = (64 * CH) % 256 == 0;
  int skey[NIT], sch[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int c = tid + it * 256;
    skey[it] = c / CH;
    sch[it] = c - skey[it] * CH;
  }
  uint4 kreg[NIT], vreg[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) { kreg[it] = make_uint4(0, 0, 0, 0); vreg[it] = make_uint4(0, 0, 0, 0); }
  // global byte offsets of this thread's chunks inside a tile (32-bit: cs_attn_check bounds Lk*ld); a tile's base is wave-uniform,
  // so a full tile costs no address arithmetic per load (SGPR base + VGPR offset); only the ragged last tile clamps its rows
  unsigned kgo[NIT], vgo[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    kgo[it] = ((unsigned)skey[it] * (unsigned)p.ldk + (unsigned)sch[it] * 8u) * 2u;
    vgo[it] = ((unsigned)skey[it] * (unsigned)p.ldv + (unsigned)sch[it] * 8u) * 2u;
  }
  // K / V tiles through buffer loads.  The descriptor's range check returns zeros for keys past Lk (they are masked to -inf in S and
  // multiply p = 0 in PV), so the ragged last tile needs no clamped addresses and no second code path (cdna_hip_programming T8).  The
  // hardware check covers the VGPR offset only -- an SGPR offset is added AFTER it -- so the tile's position must not travel in soffset
  // (rows past Lk would then be read, and a NaN bit pattern behind V's last row gives 0 * NaN in O).  Each tile therefore gets
  // its own descriptor: base advanced to the tile's first row, num_records = the bytes that remain (scalar arithmetic only, nothing per lane).
  const unsigned k_bytes = (unsigned)(((size_t)(p.Lk - 1) * p.ldk + DH) * 2), v_bytes = (unsigned)(((size_t)(p.Lk - 1) * p.ldv + DH) * 2);
#define CS_ATTN_LOAD_TILE(T)                                                                       \
  {                                                                                                \
    const unsigned ks_ = (unsigned)(T) * 64u * (unsigned)p.ldk * 2u, vs_ = (unsigned)(T) * 64u * (unsigned)p.ldv * 2u; \
    const __amdgpu_buffer_rsrc_t k_rs = __builtin_amdgcn_make_buffer_rsrc(                         \
        const_cast<char*>(reinterpret_cast<const char*>(Kb)) + ks_, 0, (int)(k_bytes - ks_), 0x00020000); \
    const __amdgpu_buffer_rsrc_t v_rs = __builtin_amdgcn_make_buffer_rsrc(                         \
        const_cast<char*>(reinterpret_cast<const char*>(Vb)) + vs_, 0, (int)(v_bytes - vs_), 0x00020000); \
    _Pragma("unroll") for (int it = 0; it < NIT; ++it) {                                           \
      if (kFullLast || it + 1 < NIT || tid + it * 256 < 64 * CH) {                                 \
        kreg[it] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(k_rs, kgo[it], 0, 0)); \
        vreg[it] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(v_rs, vgo[it], 0, 0)); \
      }                                                                                            \
    }                                                                                              \
  }
#define CS_ATTN_WRITE_TILE(BUF)                                                                    \
  _Pragma("unroll") for (int it = 0; it < NIT; ++it) {                                             \
    if (kFullLast || it + 1 < NIT || tid + it * 256 < 64 * CH) {                                   \
      char* kb_ = smem + (BUF) * Cfg::STAGE;                                                       \
      *reinterpret_cast<uint4*>(kb_ + skey[it] * Cfg::KROW + sch[it] * 16) = kreg[it];             \
      *reinterpret_cast<uint4*>(kb_ + Cfg::KTILE + skey[it] * Cfg::VROW + sch[it] * 16) = vreg[it]; \
    }                                                                                              \
  }

  f32x16_t ot[DT];
#pragma unroll
  for (int d = 0; d < DT; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) ot[d][e] = 0.f;
  // Online softmax with a lazily moved reference point (base-2 domain): every lane owns one query row; m_run is the row's
  // reference, p = 2^(s - m_run).  The reference is only moved when some row of the wave exceeds it by more than kTau
  // (p <= 2^kTau = 256: exact in fp32 sums, in range for the fp16 P operand, same relative precision), so on most tiles
  // nothing is subtracted and nothing is rescaled: -m_run enters through the C operand of the first QK^T MFMA of the tile
  // (negm: 16 registers of the lane's -m_run), and the only per-score VALU work left is max, exp2, sum and the fp16 pack.
  constexpr float kTau = 8.0f;
  float m_run = 0.f;        // reference point of s*scale*log2e (set from the first tile)
  float l_run = 0.f;        // running sum over this lane's keys (other half lives in lane^32)
  f32x16_t negm;
#pragma unroll
  for (int e = 0; e < 16; ++e) negm[e] = 0.f;
  {  // Q is used pre-multiplied by scale*log2e (scale_log2e == 1: the producer already folded it into the Q projection)
    const float sc = p.scale_log2e;
    if (sc != 1.0f) {
#pragma unroll
      for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const _Float16 e = qf[s][j];  // (a scalar copy first: __builtin_bit_cast applied to the vector element itself miscompiles, r3)
          qf[s][j] = __builtin_bit_cast(_Float16, f2o<BF>(o2f<BF>(__builtin_bit_cast(h16_t, e)) * sc));
        }
    }
  }

  // per-lane LDS byte offsets
  const int koff = r * Cfg::KROW + hh * 16;                                  // + kt2*32*KROW + s*32
  const int li = lane & 15, lg = (lane >> 4) & 1;
  const int voff = (4 * hh + (li >> 2)) * Cfg::VROW + (16 * lg + 4 * (li & 3)) * 2;  // + (kt2*32+16*s2 [+8])*VROW + dt*64

  const int nt = (p.Lk + 63) / 64;
  CS_ATTN_LOAD_TILE(0)
  CS_ATTN_WRITE_TILE(0)
  __syncthreads();
#ifdef CS_ATTN_STAMP
  unsigned long long ph[5] = {0, 0, 0, 0, 0};
  const unsigned long long tbegin = __builtin_amdgcn_s_memtime(), rbegin = __builtin_amdgcn_s_memrealtime();
  unsigned long long tlast = tbegin;
#endif
  for (int t = 0; t < nt; ++t) {
    if (t + 1 < nt) { CS_ATTN_LOAD_TILE(t + 1) }
    const char* kb = smem + (t & 1) * Cfg::STAGE;
    const char* vb = kb + Cfg::KTILE;

    f32x16_t st[2];
    float ps0 = 0.f, ps1 = 0.f;  // two chains of plain v_add_f32 (packed f32 adds cost more issue cycles than they save)
    h16x8_t pf[2][2];
    auto scores = [&]() {  // S^T - m = K Q^T - m : two 32-key sub-tiles (-m_run enters through the C operand), ragged tail masked
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          h16x8_t kf = *reinterpret_cast<const h16x8_t*>(kb + koff + k2 * 32 * Cfg::KROW + s * 32);
          st[k2] = mfma_32x32x16<BF>(kf, qf[s], s == 0 ? negm : st[k2]);
        }
      }
      if (t == nt - 1 && (p.Lk & 63)) {  // keys >= Lk : wave-uniform branch
        const int kbase = t * 64 + 4 * hh;
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int key = kbase + k2 * 32 + (e & 3) + 8 * (e >> 2);
            if (key >= p.Lk) st[k2][e] = -INFINITY;
          }
      }
    };
    auto row_max = [&]() {
      float tm = st[0][0];
#pragma unroll
      for (int e = 1; e < 16; ++e) tm = fmaxf(tm, st[0][e]);
#pragma unroll
      for (int e = 0; e < 16; ++e) tm = fmaxf(tm, st[1][e]);
      return fmaxf(tm, __shfl_xor(tm, 32, 64));  // the row's maximum of this tile, relative to m_run
    };
    // moves the reference point by `delta` (st, l and O follow): the rare path
    auto move_reference = [&](float delta, bool first) {
      const float alpha = first ? 1.0f : __builtin_amdgcn_exp2f(-delta);
      m_run += delta;
#pragma unroll
      for (int e = 0; e < 16; ++e) negm[e] = -m_run;
      asm volatile("" : "+v"(negm));  // keep the 16 copies resident instead of re-materialising them per tile
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
        for (int e = 0; e < 16; ++e) st[k2][e] -= delta;
      l_run *= alpha;
#pragma unroll
      for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) ot[d][e] *= alpha;
    };
    auto exp_pack = [&]() {
      ps0 = 0.f; ps1 = 0.f;
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
          for (int j = 0; j < 8; j += 2) {
            const float pa = __builtin_amdgcn_exp2f(st[k2][8 * s2 + j]);
            const float pb = __builtin_amdgcn_exp2f(st[k2][8 * s2 + j + 1]);
            ps0 += pa;
            ps1 += pb;
            pf[k2][s2][j] = __builtin_bit_cast(_Float16, f2o<BF>(pa));
            pf[k2][s2][j + 1] = __builtin_bit_cast(_Float16, f2o<BF>(pb));
          }
        }
    };
    // (Tried, r3: no row maximum on the common pass -- form p at once and let the tile's row sum, needed anyway, tell whether the reference
    //  point was still good, redoing the tile otherwise.  18 VALU instructions less per tile, but on peaky rows the redo fires often and
    //  costs a second QK^T: 290 vs 220 us per encoder launch of 48 images.  The lazily moved reference below stays.)
    scores();
    const float tmax = row_max();
    CS_TS(0);
    // ---- move the reference (rare after the first tiles): wave-uniform branch ----
    if (t == 0 || __builtin_amdgcn_ballot_w64(tmax > kTau) != 0) move_reference(t == 0 ? tmax : fmaxf(tmax, 0.f), t == 0);
    exp_pack();
    l_run += ps0 + ps1;
    CS_TS(1);

    // ---- O^T += V^T P^T : V^T fragments via transposed LDS reads ----
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const char* vrow = vb + voff + (k2 * 32 + 16 * s2) * Cfg::VROW;
#pragma unroll
        for (int d = 0; d < DT; ++d) {
          short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) short4_t*)(vrow + d * 64));
          short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) short4_t*)(vrow + 8 * Cfg::VROW + d * 64));
          short8_t vf8 = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
          ot[d] = mfma_32x32x16<BF>(__builtin_bit_cast(h16x8_t, vf8), pf[k2][s2], ot[d]);
        }
      }

    CS_TS(2);
    if (t + 1 < nt) { CS_ATTN_WRITE_TILE((t + 1) & 1) }
    CS_TS(3);
    __syncthreads();
    CS_TS(4);
  }
#ifdef CS_ATTN_STAMP
  if (blockIdx.x % 50 == 0 && blockIdx.x / 50 < 64 && lane == 0) {  // a sample across the whole grid
    unsigned long long* d = g_attn_dbg + (blockIdx.x / 50 * 4 + wv) * 8;
    for (int k = 0; k < 5; ++k) d[k] = ph[k];
    d[5] = rbegin;
    d[6] = __builtin_amdgcn_s_memtime() - tbegin;
    d[7] = __builtin_amdgcn_s_memrealtime() - rbegin;  // 100 MHz
  }
#endif

  // ---- epilogue: O[q][head*DH + d] = O^T[d][q] / l ----
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  const int q = q0 + r;
  if (q < p.Lq) {
    h16_t* op = p.O + (size_t)bat * p.o_bs + (size_t)q * p.ldo + head * DH;
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int dd = d * 32 + 8 * g4 + 4 * hh;
        if (dd < DH) {
          const float v0 = ot[d][4 * g4 + 0] * inv, v1 = ot[d][4 * g4 + 1] * inv, v2 = ot[d][4 * g4 + 2] * inv, v3 = ot[d][4 * g4 + 3] * inv;
          uint2 o;
          o.x = pack_o16x2<BF>(v0, v1);
          o.y = pack_o16x2<BF>(v2, v3);
          *reinterpret_cast<uint2*>(op + dd) = o;
        }
      }
    if (p.lse && hh == 0) p.lse[((size_t)bat * p.heads + head) * p.Lq + q] = m_run + log2f(l_tot);
  }
}

template <int DH, bool BF>
hipError_t launch(CsAttnParams p, int batch, hipStream_t stream) {
  p.nbatch = batch;
  const int groups = p.heads * batch;
  dim3 grid(((groups + 7) / 8) * 8 * ((p.Lq + 127) / 128));
  const int lds = 2 * AttnCfg<DH>::STAGE + 64;  // +64: the last rows' transposed reads of a padded d tile may run past the image
  if (lds > 48 * 1024) {  // (dh = 128: 74 KiB)
    static std::atomic<bool> attr_done[16];  // (zero-initialised; hipFuncSetAttribute is idempotent, a racing second caller only repeats it)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return hipErrorInvalidDevice;
    if (!attr_done[dev]) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(cs_attn_kernel<DH, BF>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      if (e != hipSuccess) return e;
      attr_done[dev] = true;
    }
  }
  hipLaunchKernelGGL((cs_attn_kernel<DH, BF>), grid, dim3(256), lds, stream, p);
  return hipGetLastError();
}

}  // namespace

#ifdef CS_ATTN_STAMP
extern "C" int cs_attn_debug_read(unsigned long long* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_attn_dbg), sizeof(g_attn_dbg)); }
#endif

extern "C" const char* cs_attn_check(const CsAttnParams* p, int dh, int batch) {
  if (dh != 16 && dh != 48 && dh != 64 && dh != 96 && dh != 128 && dh != 192) return "attention: head dim must be 16, 48, 64, 96, 128 or 192";
  if (p->Lq <= 0 || p->Lk <= 0 || p->heads <= 0 || batch <= 0) return "attention: empty shape";
  if ((long long)batch * p->heads * ((p->Lq + 127) / 128) > (1ll << 30)) return "attention: grid too large";
  if (p->ldq % 8 || p->ldk % 8 || p->ldv % 8 || p->ldo % 4) return "attention: row strides must keep 16-byte rows";
  if (p->q_bs % 8 || p->k_bs % 8 || p->v_bs % 8 || p->o_bs % 4) return "attention: batch strides must keep 16-byte rows";
  if (!p->Q || !p->K || !p->V || !p->O) return "attention: null operand";
  if ((long long)p->Lk * p->ldk >= (1ll << 30) || (long long)p->Lk * p->ldv >= (1ll << 30)) return "attention: Lk * row stride must stay below 2^30 elements (32-bit tile offsets)";
  return nullptr;
}

extern "C" hipError_t cs_attn_launch(const CsAttnParams* p, int dh, int batch, hipStream_t stream) {
  switch (dh) {
    case 16: return p->bf16 ? launch<16, true>(*p, batch, stream) : launch<16, false>(*p, batch, stream);
    case 48: return p->bf16 ? launch<48, true>(*p, batch, stream) : launch<48, false>(*p, batch, stream);
    case 64: return p->bf16 ? launch<64, true>(*p, batch, stream) : launch<64, false>(*p, batch, stream);
    case 96: return p->bf16 ? launch<96, true>(*p, batch, stream) : launch<96, false>(*p, batch, stream);
    case 128: return p->bf16 ? launch<128, true>(*p, batch, stream) : launch<128, false>(*p, batch, stream);
    case 192: return p->bf16 ? launch<192, true>(*p, batch, stream) : launch<192, false>(*p, batch, stream);  // (dinov2-giant's decoder: 1536 / 8)
  }
  return hipErrorInvalidValue;
}
