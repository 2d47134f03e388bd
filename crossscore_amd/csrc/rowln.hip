// Decoder "linear + residual + LayerNorm" in ONE launch (round 4): the post-norm decoder layer of the reference closes each of its three
// sub-blocks with  x = LN(x + Linear(y))  (transformer.py:157-173: norm1(x + out_proj(sa)), norm2(x + out_proj(ca)), norm3(x + linear2(h));
// without the residual when decoder_do_short_cut is off) -- a GEMM with an fp32 residual epilogue followed by a LayerNorm launch that re-read
// its output.  Here a workgroup owns 64 COMPLETE rows (all C output columns), so the LayerNorm happens on the accumulators:
//   out_f32[m][:] = LN(resid[m][:] + A[m][:] W^T + bias; gamma, beta, eps),   out_f16 = the same rows in the 16-bit operand type.
// Shapes: K = N = C = 384 (the ViT-S decoder: out-projections and linear2, dim_feedforward = C, cross_reference.py:32); M rows (B * h*w).
// Structure: C / 96 waves, wave w owns output columns [96 w, 96 w + 96) of the 64 rows = 2 x 3 accumulators of v_mfma_f32_32x32x16 (96
// registers); the WEIGHT fragment is the A operand (32 output features x 16 k) and the activation fragment the B operand (16 k x 32 rows), so
// D[feature][row]: a lane holds 48 values of ONE row per row block and the row sums are lane-local plus one exchange with the other half of
// the wave and one with the other waves (LDS, 2 x C/96 x 64 floats).  Both operands come straight from L2 / L1 as 16-byte per-lane loads
// (every 128-byte line is used by four consecutive k-steps): the kernel moves 0.3 MB of weights per workgroup and is latency-, not
// bandwidth-bound -- 172 workgroups for the 10 952 rows of cfg-2, one round.
// Second stage (G2 > 0): the sub-block's NEXT linear in the same launch -- out2 = act2(LN rows x W2^T + bias2) for G2 groups of C output
// columns (the cross-attention's Q projection, transformer.py:195-205; linear1 + ReLU, :208-210; the head's first linear + LeakyReLU,
// cross_reference.py:45-50).  The normalised rows, rounded to the operand type as a separate GEMM would read them, replace the activation rows
// in LDS and the K loop runs again.
#include "cs_common.h"
#include <atomic>

namespace {

__device__ __forceinline__ float rl_add_other_half(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

template <int C, bool BF, int G2>  // G2: column groups of C outputs of the second stage (0: none)
__global__ __launch_bounds__(C / 96 * 64) void cs_rowln_kernel(CsRowLnParams p) {
  constexpr int NW = C / 96, NT = NW * 64;
  constexpr bool NEXT = G2 > 0;
  constexpr int BK = 64, NSL = C / BK;          // K slices of the weight stream
  constexpr int APITCH = C * 2 + 16;            // bytes; odd multiples of 16: ds_read_b128 of 32 different rows is conflict free
  constexpr int WPITCH = BK * 2 + 16;
  constexpr int WBUF = C * WPITCH;
  constexpr int ACH = C / 8, WCH = BK / 8;      // 16-byte chunks per row
  constexpr int AIT = 64 * ACH / NT, WIT = C * WCH / NT;
  static_assert(64 * ACH % NT == 0 && C * WCH % NT == 0, "staging maps");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* As = smem;                               // [64][APITCH]: the workgroup's activation rows, all of K
  char* Ws = smem + 64 * APITCH;                 // [2][C][WPITCH]: K slices of the weights, double buffered
  float* red = reinterpret_cast<float*>(Ws + 2 * WBUF);  // [2][NW][64]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;
  const int row0 = blockIdx.x * 64;
  const int col0 = wv * 96;
  // ---- staging: coalesced 16-byte loads (8 consecutive threads per 128-byte row segment), register staged (issue early, write late) ----
  // (macros, not lambdas: with the register arrays captured by reference hipcc kept them in scratch memory -- 144 scratch instructions, 37 us)
  uint4 wreg[WIT];
#define RL_LOAD_W(REG, WP, LDW, SL)                                                                                \
  _Pragma("unroll") for (int it = 0; it < WIT; ++it) {                                                         \
    const int c_ = tid + it * NT, r_ = c_ / WCH, ch_ = c_ - r_ * WCH;                                          \
    REG[it] = *reinterpret_cast<const uint4*>((WP) + (size_t)r_ * (LDW) + (SL) * BK + ch_ * 8);                \
  }
#define RL_WRITE_W(REG, BUF)                                                                                   \
  _Pragma("unroll") for (int it = 0; it < WIT; ++it) {                                                         \
    const int c_ = tid + it * NT, r_ = c_ / WCH, ch_ = c_ - r_ * WCH;                                          \
    *reinterpret_cast<uint4*>(Ws + (BUF) * WBUF + r_ * WPITCH + ch_ * 16) = REG[it];                           \
  }
  RL_LOAD_W(wreg, p.W, p.ldw, 0)
  {
    uint4 areg[AIT];
#pragma unroll
    for (int it = 0; it < AIT; ++it) {
      const int c = tid + it * NT, r = c / ACH, ch = c - r * ACH;
      areg[it] = *reinterpret_cast<const uint4*>(p.A + (size_t)min(row0 + r, p.M - 1) * p.lda + ch * 8);  // (rows past M: computed and dropped)
    }
#pragma unroll
    for (int it = 0; it < AIT; ++it) {
      const int c = tid + it * NT, r = c / ACH, ch = c - r * ACH;
      *reinterpret_cast<uint4*>(As + r * APITCH + ch * 16) = areg[it];
    }
  }
#define RL_KSLICE(SLICE)                                                                                       \
  _Pragma("unroll") for (int ks = 0; ks < BK / 16; ++ks) {                                                     \
    h16x8_t a[2], w[3];                                                                                        \
    _Pragma("unroll") for (int rb = 0; rb < 2; ++rb)                                                           \
      a[rb] = *reinterpret_cast<const h16x8_t*>(a_rd + rb * 32 * APITCH + ((SLICE) * (BK / 16) + ks) * 32);    \
    _Pragma("unroll") for (int ct = 0; ct < 3; ++ct) w[ct] = *reinterpret_cast<const h16x8_t*>(w_rd + ct * 32 * WPITCH + ks * 32); \
    _Pragma("unroll") for (int rb = 0; rb < 2; ++rb)                                                           \
      _Pragma("unroll") for (int ct = 0; ct < 3; ++ct) acc[rb][ct] = mfma_32x32x16<BF>(w[ct], a[rb], acc[rb][ct]); \
  }
  RL_WRITE_W(wreg, 0)
  __syncthreads();
  f32x16_t acc[2][3];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int ct = 0; ct < 3; ++ct)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[rb][ct][e] = 0.f;
  const char* a_rd = As + j * APITCH + h * 16;                         // + rb * 32 * APITCH + k-step * 32
  for (int sl = 0; sl < NSL; ++sl) {
    if (sl + 1 < NSL) { RL_LOAD_W(wreg, p.W, p.ldw, sl + 1) }
    const char* w_rd = Ws + (sl & 1) * WBUF + (col0 + j) * WPITCH + h * 16;  // + ct * 32 * WPITCH + k-step * 32
    RL_KSLICE(sl)
    if (sl + 1 < NSL) { RL_WRITE_W(wreg, (sl + 1) & 1) }
    __syncthreads();
  }
  // ---- epilogue.  Register e of tile ct of this lane = feature col0 + 32 ct + (e & 3) + 8 (e >> 2) + 4 h of row (row0 + 32 rb + j):
  //      four 16-byte groups per tile.  v = acc + bias (+ residual) ----
  f32x4_t gam[3][4], bet[3][4];
#pragma unroll
  for (int ct = 0; ct < 3; ++ct)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int c = col0 + 32 * ct + 8 * g + 4 * h;
      const f32x4_t b4 = *reinterpret_cast<const f32x4_t*>(p.bias + c);
      gam[ct][g] = *reinterpret_cast<const f32x4_t*>(p.gamma + c);
      bet[ct][g] = *reinterpret_cast<const f32x4_t*>(p.beta + c);
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[rb][ct][4 * g + i] += b4[i];
    }
  if (p.resid) {
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      const float* rr = p.resid + (size_t)min(row0 + 32 * rb + j, p.M - 1) * p.ldr + col0 + 4 * h;
#pragma unroll
      for (int ct = 0; ct < 3; ++ct)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4_t r4 = *reinterpret_cast<const f32x4_t*>(rr + 32 * ct + 8 * g);
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[rb][ct][4 * g + i] += r4[i];
        }
    }
  }
  // ---- LayerNorm over the complete rows, two passes in registers (mean, then the centred sum of squares), fp32 ----
  float mean[2], rstd[2];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    float s = 0.f;
#pragma unroll
    for (int ct = 0; ct < 3; ++ct)
#pragma unroll
      for (int e = 0; e < 16; e += 4) s += (acc[rb][ct][e] + acc[rb][ct][e + 1]) + (acc[rb][ct][e + 2] + acc[rb][ct][e + 3]);
    s = rl_add_other_half(s);
    if (h == 0) red[(0 * NW + wv) * 64 + 32 * rb + j] = s;
  }
  __syncthreads();
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    float s = 0.f;
#pragma unroll
    for (int w2 = 0; w2 < NW; ++w2) s += red[(0 * NW + w2) * 64 + 32 * rb + j];
    mean[rb] = s * (1.0f / C);
    float q = 0.f;
#pragma unroll
    for (int ct = 0; ct < 3; ++ct)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float d = acc[rb][ct][e] - mean[rb];
        q = fmaf(d, d, q);
      }
    q = rl_add_other_half(q);
    if (h == 0) red[(1 * NW + wv) * 64 + 32 * rb + j] = q;
  }
  __syncthreads();
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    float q = 0.f;
#pragma unroll
    for (int w2 = 0; w2 < NW; ++w2) q += red[(1 * NW + w2) * 64 + 32 * rb + j];
    rstd[rb] = 1.0f / sqrtf(q * (1.0f / C) + p.eps);
  }
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    const int row = row0 + 32 * rb + j;
    if (row < p.M) {
      float* o32 = p.out_f32 ? p.out_f32 + (size_t)row * C + col0 + 4 * h : nullptr;
      h16_t* o16 = p.out_f16 ? p.out_f16 + (size_t)row * C + col0 + 4 * h : nullptr;
#pragma unroll
      for (int ct = 0; ct < 3; ++ct)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4_t y;
#pragma unroll
          for (int i = 0; i < 4; ++i) y[i] = (acc[rb][ct][4 * g + i] - mean[rb]) * rstd[rb] * gam[ct][g][i] + bet[ct][g][i];
          if (o32) *reinterpret_cast<f32x4_t*>(o32 + 32 * ct + 8 * g) = y;
          if (o16) {
            uint2 o;
            o.x = pack_o16x2<BF>(y[0], y[1]);
            o.y = pack_o16x2<BF>(y[2], y[3]);
            *reinterpret_cast<uint2*>(o16 + 32 * ct + 8 * g) = o;
          }
        }
    }
  }
  if constexpr (NEXT) {
    // ---- second stage: the normalised rows (rounded to the operand type, exactly what a separate GEMM would read from out_f16) replace the
    //      activation rows in LDS -- every wave is past its last read of them (the K loop's closing barrier) -- and feed the sub-block's
    //      next linear: n2 / C column groups of C outputs, the same K loop per group, bias + activation, 16-bit rows out ----
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      char* arow = As + (32 * rb + j) * APITCH + (col0 + 4 * h) * 2;
#pragma unroll
      for (int ct = 0; ct < 3; ++ct)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float y[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) y[i] = (acc[rb][ct][4 * g + i] - mean[rb]) * rstd[rb] * gam[ct][g][i] + bet[ct][g][i];
          uint2 o;
          o.x = pack_o16x2<BF>(y[0], y[1]);
          o.y = pack_o16x2<BF>(y[2], y[3]);
          *reinterpret_cast<uint2*>(arow + (32 * ct + 8 * g) * 2) = o;
        }
    }
    // (the first weight slice is loaded here, not before the LayerNorm where it would travel under it: issued there, hipcc kept the staging
    //  array in scratch memory -- 84 scratch instructions)
    RL_LOAD_W(wreg, p.W2, p.ldw2, 0)
    RL_WRITE_W(wreg, 0)
    __syncthreads();
    constexpr int nsl2 = G2 * NSL;
#pragma unroll
    for (int s2 = 0; s2 < nsl2; ++s2) {
      const int grp = s2 / NSL, sl = s2 - grp * NSL;
      if (sl == 0) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
          for (int ct = 0; ct < 3; ++ct)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[rb][ct][e] = 0.f;
      }
      if (s2 + 1 < nsl2) {
        const int g2 = (s2 + 1) / NSL, sn = (s2 + 1) - g2 * NSL;
        RL_LOAD_W(wreg, p.W2 + (size_t)g2 * C * p.ldw2, p.ldw2, sn)
      }
      const char* w_rd = Ws + (s2 & 1) * WBUF + (col0 + j) * WPITCH + h * 16;
      RL_KSLICE(sl)
      if (s2 + 1 < nsl2) { RL_WRITE_W(wreg, (s2 + 1) & 1) }
      __syncthreads();
      if (sl == NSL - 1) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
          const int row = row0 + 32 * rb + j;
          if (row < p.M) {
            h16_t* o2 = p.out2 + (size_t)row * p.ld2 + grp * C + col0 + 4 * h;
            const float* b2 = p.bias2 + grp * C + col0 + 4 * h;
#pragma unroll
            for (int ct = 0; ct < 3; ++ct)
#pragma unroll
              for (int g = 0; g < 4; ++g) {
                const f32x4_t b4 = *reinterpret_cast<const f32x4_t*>(b2 + 32 * ct + 8 * g);
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                  v[i] = acc[rb][ct][4 * g + i] + b4[i];
                  if (p.act2 == 1) v[i] = fmaxf(v[i], 0.f);
                  if (p.act2 == 2) v[i] = v[i] >= 0.f ? v[i] : 0.01f * v[i];
                }
                uint2 o;
                o.x = pack_o16x2<BF>(v[0], v[1]);
                o.y = pack_o16x2<BF>(v[2], v[3]);
                *reinterpret_cast<uint2*>(o2 + 32 * ct + 8 * g) = o;
              }
          }
        }
      }
    }
  }
#undef RL_KSLICE
#undef RL_LOAD_W
#undef RL_WRITE_W
}

}  // namespace

extern "C" int cs_rowln_supported(int C) { return C == 384; }

extern "C" const char* cs_rowln_check(const CsRowLnParams* p, int C) {
  if (!cs_rowln_supported(C)) return "linear + LayerNorm: built for C = 384";
  if (!p->A || !p->W || !p->bias || !p->gamma || !p->beta || (!p->out_f32 && !p->out_f16 && !p->n2)) return "linear + LayerNorm: null operand";
  if (p->n2) {
    if ((p->n2 != C && p->n2 != 3 * C) || p->act2 < 0 || p->act2 > 2) return "linear + LayerNorm: the second stage takes C or 3 C output columns and act2 in 0..2";
    if (!p->W2 || !p->bias2 || !p->out2) return "linear + LayerNorm: null operand (second stage)";
    if (p->ldw2 % 8 || p->ldw2 < C || p->ld2 % 4 || p->ld2 < p->n2) return "linear + LayerNorm: row strides must keep 16-byte rows (second stage)";
    if (((uintptr_t)p->W2 | (uintptr_t)p->bias2) & 15 || ((uintptr_t)p->out2 & 7)) return "linear + LayerNorm: operands must be 16-byte aligned (second stage)";
  }
  if (p->M <= 0) return "linear + LayerNorm: empty shape";
  if (p->lda % 8 || p->ldw % 8 || p->lda < C || p->ldw < C || (p->resid && (p->ldr % 4 || p->ldr < C))) return "linear + LayerNorm: row strides must keep 16-byte rows";
  if (((uintptr_t)p->A | (uintptr_t)p->W | (uintptr_t)p->bias | (uintptr_t)p->gamma | (uintptr_t)p->beta | (uintptr_t)p->resid | (uintptr_t)p->out_f32 |
       (uintptr_t)p->out_f16) & 15)
    return "linear + LayerNorm: operands must be 16-byte aligned";
  return nullptr;
}

extern "C" hipError_t cs_rowln_launch(const CsRowLnParams* p, int C, int bf16, hipStream_t st) {
  if (C != 384) return hipErrorInvalidValue;
  constexpr int CC = 384, NW = CC / 96;
  constexpr int lds = 64 * (CC * 2 + 16) + 2 * CC * (64 * 2 + 16) + 2 * NW * 64 * 4;  // activation rows + two weight slices + the row-sum exchange
  static std::atomic<bool> attr_done[16];  // (zero-initialised; hipFuncSetAttribute is idempotent, a racing second caller only repeats it)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return hipErrorInvalidDevice;
  if (!attr_done[dev]) {
    const void* fns[6] = {reinterpret_cast<const void*>(cs_rowln_kernel<384, true, 0>), reinterpret_cast<const void*>(cs_rowln_kernel<384, false, 0>),
                          reinterpret_cast<const void*>(cs_rowln_kernel<384, true, 1>), reinterpret_cast<const void*>(cs_rowln_kernel<384, false, 1>),
                          reinterpret_cast<const void*>(cs_rowln_kernel<384, true, 3>), reinterpret_cast<const void*>(cs_rowln_kernel<384, false, 3>)};
    for (const void* f : fns)
      if (hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, lds); e != hipSuccess) return e;
    attr_done[dev] = true;
  }
  const dim3 grid((p->M + 63) / 64), block(NW * 64);
#define RL_GO(G) do { if (bf16) hipLaunchKernelGGL((cs_rowln_kernel<384, true, G>), grid, block, lds, st, *p);  \
                      else hipLaunchKernelGGL((cs_rowln_kernel<384, false, G>), grid, block, lds, st, *p); } while (0)
  if (p->n2 == 0) RL_GO(0);
  else if (p->n2 == CC) RL_GO(1);
  else if (p->n2 == 3 * CC) RL_GO(3);
  else return hipErrorInvalidValue;
#undef RL_GO
  return hipGetLastError();
}
