// bf16 MFMA GEMM with fused epilogues for the CrossScore hot path (gfx950).
//
//   acc[m][n] = sum_k A[m][k] * W[n][k]      A:[M][K] bf16 (activations), W:[N][K] bf16 (nn.Linear layout)
//
// Replaces the eager nn.Linear / Conv2d-patchify op groups K2,K4,K6,K7,K10,K12,K13,K15-K18 of SURVEY.md 2a
// (HF modeling_dinov2.py:148,211-213,250,293-297; torch functional.py:5785-5852; cross_reference.py:45-50).
//
// Structure: 128x128x64 tile, 4 waves (2x2), each wave 64x64 = 4x4 tiles of v_mfma_f32_16x16x32_bf16.
// Both operands are K-contiguous, so both fragments are 16-byte rows: staged HBM->LDS with
// global_load_lds_dwordx4 (no VGPR round trip) into a lane-linear image whose 16-byte chunks are
// XOR-swizzled on the SOURCE address and on the ds_read_b128 address (conflict-free, same involution),
// double buffered, one barrier per K step.  Operands are swapped in the MFMA (W is the "A" operand) so each
// lane owns 4 consecutive output columns -> 8/16-byte epilogue stores and float4 bias/scale loads.
// Block -> tile mapping is XCD aware: the blocks that re-read one A row panel run on one XCD (one L2).
#include "cs_common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int A_BYTES = BM * BK * 2;
constexpr int STAGE_BYTES = (BM + BN) * BK * 2;  // 32 KiB

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }

template <int EPI>
__global__ __launch_bounds__(256, 2) void cs_gemm_kernel(CsGemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_n = (p.N + BN - 1) / BN;
  const int tiles_m = (p.M + BM - 1) / BM;
  // blocks b, b+8, b+16.. share an XCD (round-robin dispatch); give each XCD whole A row panels.
  const int bid = blockIdx.x;
  const int xcd = bid & 7, idx = bid >> 3;
  const int tn = idx % tiles_n;
  const int tm = (idx / tiles_n) * 8 + xcd;
  if (tm >= tiles_m) return;
  const int m0 = tm * BM, n0 = tn * BN;
  const int wm = wv >> 1, wn = wv & 1;

  // ---- staging addresses: 4 wave-instructions for A, 4 for W per K step; each covers 8 rows x 128 B ----
  const int srow = lane >> 3;                    // row within the 8-row group (== row & 7)
  const int gchunk = (lane & 7) ^ srow;          // source 16-B chunk that lands in LDS chunk (lane&7)
  int offA[4], offW[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = (j * 4 + wv) * 8 + srow;
    int ga = m0 + r; ga = ga < p.M ? ga : p.M - 1;
    int gw = n0 + r; gw = gw < p.N ? gw : p.N - 1;
    offA[j] = ga * p.lda + gchunk * 8;
    offW[j] = gw * p.ldw + gchunk * 8;
  }
  auto stage = [&](int buf, int kt) {
    char* base = smem + buf * STAGE_BYTES;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int rb = j * 4 + wv;
      __builtin_amdgcn_global_load_lds(CS_GLOBAL_PTR(p.A + offA[j] + kt * BK), CS_LDS_PTR(base + rb * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(CS_GLOBAL_PTR(p.W + offW[j] + kt * BK), CS_LDS_PTR(base + A_BYTES + rb * 1024), 16, 0, 0);
    }
  };

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // fragment read addresses: row = sub*16 + (lane&15), logical chunk q = ks*4 + (lane>>4), LDS chunk = q ^ (row&7)
  const int frow = lane & 15;
  const int fq = lane >> 4;
  const int fsw = lane & 7;
  const int nk = p.K / BK;
  stage(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();  // tile kt landed (vmcnt(0) precedes the barrier) and everyone left buffer (kt+1)&1
    if (kt + 1 < nk) stage((kt + 1) & 1, kt + 1);
    const char* sa = smem + (kt & 1) * STAGE_BYTES + (wm * 64 + frow) * 128;
    const char* sw = smem + (kt & 1) * STAGE_BYTES + A_BYTES + (wn * 64 + frow) * 128;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int coff = ((ks * 4 + fq) ^ fsw) * 16;
      bf16x8_t fa[4], fw[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const bf16x8_t*>(sa + i * 16 * 128 + coff);
#pragma unroll
      for (int j = 0; j < 4; ++j) fw[j] = *reinterpret_cast<const bf16x8_t*>(sw + j * 16 * 128 + coff);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[j], fa[i], acc[i][j], 0, 0, 0);
    }
  }

  // ---- epilogue: lane owns m = .. + (lane&15), n = .. + (lane>>4)*4 + r ----
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + i * 16 + (lane & 15);
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + (lane >> 4) * 4;
      if (n >= p.N) continue;
      float v[4];
      float4 b4 = p.bias ? *reinterpret_cast<const float4*>(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
      v[0] = acc[i][j][0] + b4.x; v[1] = acc[i][j][1] + b4.y; v[2] = acc[i][j][2] + b4.z; v[3] = acc[i][j][3] + b4.w;
      if constexpr (EPI == CS_EPI_BIAS_BF16 || EPI == CS_EPI_BIAS_GELU_BF16 || EPI == CS_EPI_BIAS_RELU_BF16 ||
                    EPI == CS_EPI_BIAS_LEAKY_BF16) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if constexpr (EPI == CS_EPI_BIAS_GELU_BF16) v[r] = gelu_erf(v[r]);
          if constexpr (EPI == CS_EPI_BIAS_RELU_BF16) v[r] = fmaxf(v[r], 0.f);
          if constexpr (EPI == CS_EPI_BIAS_LEAKY_BF16) v[r] = v[r] >= 0.f ? v[r] : 0.01f * v[r];
        }
        uint2 o;
        o.x = pack_bf16x2(v[0], v[1]);
        o.y = pack_bf16x2(v[2], v[3]);
        *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.out) + (size_t)m * p.ldc + n) = o;
      } else if constexpr (EPI == CS_EPI_RESID_F32) {
        if (p.scale) {
          float4 s4 = *reinterpret_cast<const float4*>(p.scale + n);
          v[0] *= s4.x; v[1] *= s4.y; v[2] *= s4.z; v[3] *= s4.w;
        }
        if (p.resid) {
          float4 r4 = *reinterpret_cast<const float4*>(p.resid + (size_t)m * p.ldr + n);
          v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
        }
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + (size_t)m * p.ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
      } else if constexpr (EPI == CS_EPI_PATCH_F32) {
        const int img = m / p.Np;
        const int pp = m - img * p.Np;
        float4 e4 = *reinterpret_cast<const float4*>(p.pos + (size_t)(pp + 1) * p.ldc + n);
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + (size_t)(m + img + 1) * p.ldc + n) =
            make_float4(v[0] + e4.x, v[1] + e4.y, v[2] + e4.z, v[3] + e4.w);
      } else if constexpr (EPI == CS_EPI_HEAD_SCORE) {
        const int b = m / p.Np;
        const int pp = m - b * p.Np;
        const int pi = pp / p.gw, pj = pp - pi * p.gw;
        const int gh = p.Np / p.gw;
        const int Ws = p.P * p.gw;
        float* dst = reinterpret_cast<float*>(p.out) + ((size_t)b * gh * p.P + (size_t)pi * p.P) * Ws + pj * p.P;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int nn = n + r;
          const int py = nn / p.P, px = nn - py * p.P;
          float y = p.act == 0 ? 1.0f / (1.0f + __expf(-v[r])) : tanhf(v[r]);
          if (p.powp != 1.0f) y = powf(y, p.powp);
          dst[(size_t)py * Ws + px] = y;
        }
      }
    }
  }
}

template <int EPI>
hipError_t launch(const CsGemmParams& p, hipStream_t stream) {
  const int tiles_n = (p.N + BN - 1) / BN;
  const int tiles_m = (p.M + BM - 1) / BM;
  const int grid = ((tiles_m + 7) / 8) * 8 * tiles_n;
  hipLaunchKernelGGL(cs_gemm_kernel<EPI>, dim3(grid), dim3(256), 2 * STAGE_BYTES, stream, p);
  return hipGetLastError();
}

}  // namespace

// Host-side shape contract (checked here so a bad call fails loudly instead of faulting on the GPU).
extern "C" const char* cs_gemm_check(const CsGemmParams* p, int epi) {
  if (p->M <= 0 || p->N <= 0 || p->K <= 0) return "gemm: empty shape";
  if (p->K % BK) return "gemm: K must be a multiple of 64";
  if (p->N % 4 || p->ldc % 4) return "gemm: N and ldc must be multiples of 4";
  if (p->lda % 8 || p->ldw % 8) return "gemm: lda/ldw must be multiples of 8 (16-byte rows)";
  if (p->lda < p->K || p->ldw < p->K) return "gemm: lda/ldw smaller than K";
  if (!p->A || !p->W || !p->out) return "gemm: null operand";
  if (epi == CS_EPI_RESID_F32 && p->resid && p->ldr % 4) return "gemm: ldr must be a multiple of 4";
  if (epi == CS_EPI_PATCH_F32 && (!p->pos || p->Np <= 0 || p->M % p->Np)) return "gemm: bad patch epilogue params";
  if (epi == CS_EPI_HEAD_SCORE && (p->Np <= 0 || p->gw <= 0 || p->Np % p->gw || p->M % p->Np || p->N != p->P * p->P))
    return "gemm: bad head epilogue params";
  if ((long long)p->M * p->lda >= (1ll << 31) || (long long)p->N * p->ldw >= (1ll << 31)) return "gemm: operand too large for 32-bit offsets";
  return nullptr;
}

extern "C" hipError_t cs_gemm_launch(const CsGemmParams* p, int epi, hipStream_t stream) {
  switch (epi) {
    case CS_EPI_BIAS_BF16: return launch<CS_EPI_BIAS_BF16>(*p, stream);
    case CS_EPI_BIAS_GELU_BF16: return launch<CS_EPI_BIAS_GELU_BF16>(*p, stream);
    case CS_EPI_BIAS_RELU_BF16: return launch<CS_EPI_BIAS_RELU_BF16>(*p, stream);
    case CS_EPI_BIAS_LEAKY_BF16: return launch<CS_EPI_BIAS_LEAKY_BF16>(*p, stream);
    case CS_EPI_RESID_F32: return launch<CS_EPI_RESID_F32>(*p, stream);
    case CS_EPI_PATCH_F32: return launch<CS_EPI_PATCH_F32>(*p, stream);
    case CS_EPI_HEAD_SCORE: return launch<CS_EPI_HEAD_SCORE>(*p, stream);
  }
  return hipErrorInvalidValue;
}
