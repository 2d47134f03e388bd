// bf16 MFMA GEMM with fused epilogues for the CrossScore hot path (gfx950).
//
//   acc[m][n] = sum_k A[m][k] * W[n][k]      A:[M][K] bf16 (activations), W:[N][K] bf16 (nn.Linear layout)
//
// Replaces the eager nn.Linear / Conv2d-patchify op groups K2,K4,K6,K7,K10,K12,K13,K15-K18 of SURVEY.md 2a
// (HF modeling_dinov2.py:148,211-213,250,293-297; torch functional.py:5785-5852; cross_reference.py:45-50).
//
// Structure: 256 x {256,192,128} x 64 tiles, 8 waves (2 x 4), wave tile 128 x {64,48,32} of
// v_mfma_f32_16x16x32_bf16.  The first version used 128x128 tiles and ran at the L2->LDS rate (64 flop per staged
// byte needs ~39 TB/s at MFMA peak; measured ~12 TB/s): the tile is sized so that the kernel stages half the bytes.
// Both operands are K-contiguous, so both fragments are 16-byte rows: staged HBM->LDS with
// global_load_lds_dwordx4 (no VGPR round trip) into a lane-linear image whose 16-byte chunks are
// XOR-swizzled on the SOURCE address and on the ds_read_b128 address (conflict-free, same involution),
// double buffered, one barrier per K step.  Operands are swapped in the MFMA (W is the "A" operand) so each
// lane owns 4 consecutive output columns -> 8/16-byte epilogue stores and float4 bias/scale loads.
// One persistent block per CU walks tiles in an XCD-aware order (the blocks that share one A row panel run on one
// XCD / one L2) and prefetches the next tile's first K slice under the current tile's last MFMAs and epilogue:
// K is only 384-1536 here, so an un-overlapped prologue + epilogue would cost as much as the K loop.
#include "cs_common.h"

namespace {

constexpr int BM = 256, BK = 64;
constexpr int A_BYTES = BM * BK * 2;  // 32 KiB per stage

// exact-erf GELU; erf by Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, far below the bf16 output rounding):
// ~12 VALU ops per element instead of the ~40 of libm erff -- the fc1 epilogue was costing more than its K loop.
__device__ __forceinline__ float gelu_erf(float x) {
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
  float poly = 1.061405429f;
  poly = poly * t - 1.453152027f;
  poly = poly * t + 1.421413741f;
  poly = poly * t - 0.284496736f;
  poly = poly * t + 0.254829592f;
  const float e = __builtin_amdgcn_exp2f(-z * z * 1.4426950408889634f);
  const float erf_abs = 1.0f - poly * t * e;
  const float erfv = x >= 0.f ? erf_abs : -erf_abs;
  return 0.5f * x * (1.0f + erfv);
}

// RegressionLayer activation (regression_layer.py:26-62); kept out of line so the unrolled head epilogue does not spill
__device__ __noinline__ float head_activation(float v, int act, float powp) {
  float y = act == 0 ? 1.0f / (1.0f + __expf(-v)) : tanhf(v);
  if (powp != 1.0f) y = powf(y, powp);
  return y;
}

// NSUB = 16-column sub-tiles per wave; block tile = 256 x (64*NSUB); 8 waves as 2 (M) x 4 (N), wave tile 128 x 16*NSUB.
template <int EPI, int NSUB>
__global__ __launch_bounds__(512, 2) void cs_gemm_kernel(CsGemmParams p) {
  constexpr int BN = 64 * NSUB;
  constexpr int STAGE_BYTES = (BM + BN) * BK * 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_n = (p.N + BN - 1) / BN;
  const int tiles_m = (p.M + BM - 1) / BM;
  const int wm = wv >> 2, wn = wv & 3;
  const int nk = p.K / BK;

  // ---- persistent tile walk, XCD aware: blocks b, b+8, .. share an XCD (round-robin dispatch); XCD x owns the A row
  //      panels tm == x (mod 8) and its blocks walk that list n-fastest, so concurrent blocks of one L2 share A. ----
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
  const int panels_x = (tiles_m - xcd + 7) / 8;          // row panels owned by this XCD
  const int ntile_x = panels_x * tiles_n;
  auto tile_of = [&](int idx, int& m0, int& n0) {
    const int pl = idx / tiles_n;
    m0 = (pl * 8 + xcd) * BM;
    n0 = (idx - pl * tiles_n) * BN;
  };

  // staging: each wave-instruction covers 8 rows x 128 B; A has 32 row groups (4 per wave), W has 8*NSUB (NSUB per wave)
  const int srow = lane >> 3;
  const int gchunk = (lane & 7) ^ srow;  // source 16-B chunk that lands in LDS chunk (lane&7): XOR swizzle on the source
  int offA[4], offW[NSUB];
  auto set_tile = [&](int m0, int n0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int ga = m0 + (j * 8 + wv) * 8 + srow;
      ga = ga < p.M ? ga : p.M - 1;
      offA[j] = ga * p.lda + gchunk * 8;
    }
#pragma unroll
    for (int j = 0; j < NSUB; ++j) {
      int gw = n0 + (j * 8 + wv) * 8 + srow;
      gw = gw < p.N ? gw : p.N - 1;
      offW[j] = gw * p.ldw + gchunk * 8;
    }
  };
  auto stage = [&](int buf, int kt) {
    char* base = smem + buf * STAGE_BYTES;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      __builtin_amdgcn_global_load_lds(CS_GLOBAL_PTR(p.A + offA[j] + kt * BK), CS_LDS_PTR(base + (j * 8 + wv) * 1024), 16, 0, 0);
#pragma unroll
    for (int j = 0; j < NSUB; ++j)
      __builtin_amdgcn_global_load_lds(CS_GLOBAL_PTR(p.W + offW[j] + kt * BK), CS_LDS_PTR(base + A_BYTES + (j * 8 + wv) * 1024), 16, 0, 0);
  };

  const int frow = lane & 15, fq = lane >> 4, fsw = lane & 7;
  int buf = 0;
  int m0, n0;
  if (slot < ntile_x) {
    tile_of(slot, m0, n0);
    set_tile(m0, n0);
    stage(0, 0);
  }
  for (int idx = slot; idx < ntile_x; idx += slots) {
    f32x4_t acc[8][NSUB];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < NSUB; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int cm0 = m0, cn0 = n0;
    for (int kt = 0; kt < nk; ++kt) {
      __syncthreads();  // K slice kt landed (vmcnt(0) precedes the barrier); everyone has left the other buffer
      if (kt + 1 < nk) {
        stage(buf ^ 1, kt + 1);
      } else if (idx + slots < ntile_x) {  // prefetch the next tile's first K slice under this tile's last MFMAs + epilogue
        tile_of(idx + slots, m0, n0);
        set_tile(m0, n0);
        stage(buf ^ 1, 0);
      }
      const char* sa = smem + buf * STAGE_BYTES + (wm * 128 + frow) * 128;
      const char* sw = smem + buf * STAGE_BYTES + A_BYTES + (wn * 16 * NSUB + frow) * 128;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int coff = ((ks * 4 + fq) ^ fsw) * 16;
        bf16x8_t fw[NSUB];
#pragma unroll
        for (int j = 0; j < NSUB; ++j) fw[j] = *reinterpret_cast<const bf16x8_t*>(sw + j * 16 * 128 + coff);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const bf16x8_t fa = *reinterpret_cast<const bf16x8_t*>(sa + i * 16 * 128 + coff);
#pragma unroll
          for (int j = 0; j < NSUB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[j], fa, acc[i][j], 0, 0, 0);
        }
      }
      buf ^= 1;
    }

    // ---- epilogue: lane owns m = .. + (lane&15), n = .. + (lane>>4)*4 + r (operands swapped in the MFMA) ----
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int m = cm0 + wm * 128 + i * 16 + (lane & 15);
      if (m >= p.M) continue;
#pragma unroll
      for (int j = 0; j < NSUB; ++j) {
        const int n = cn0 + wn * 16 * NSUB + j * 16 + (lane >> 4) * 4;
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
            dst[(size_t)py * Ws + px] = head_activation(v[r], p.act, p.powp);
          }
        }
      }
    }
  }
}

int g_num_cus = 0;

template <int EPI, int NSUB>
hipError_t launch_n(const CsGemmParams& p, hipStream_t stream) {
  constexpr int BN = 64 * NSUB;
  constexpr int LDS = 2 * (BM + BN) * BK * 2;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(cs_gemm_kernel<EPI, NSUB>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  if (g_num_cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorUnknown;
    g_num_cus = prop.multiProcessorCount;
  }
  const int tiles_n = (p.N + BN - 1) / BN;
  const int tiles_m = (p.M + BM - 1) / BM;
  // one persistent block per CU (LDS admits one); grid is a multiple of 8 so that b%8 labels the XCD group
  int grid = (g_num_cus / 8) * 8;
  const int need = ((tiles_m + 7) / 8) * tiles_n * 8;
  if (grid > need) grid = need;
  if (grid < 8) grid = 8;
  hipLaunchKernelGGL((cs_gemm_kernel<EPI, NSUB>), dim3(grid), dim3(512), LDS, stream, p);
  return hipGetLastError();
}

template <int EPI>
hipError_t launch(const CsGemmParams& p, hipStream_t stream) {
  // column tile: 256 where it divides N, else 192, else 128 (every N of the path is a multiple of 384, 128 or the 196 head)
  if (p.N % 256 == 0) return launch_n<EPI, 4>(p, stream);
  if (p.N % 192 == 0) return launch_n<EPI, 3>(p, stream);
  if (p.N % 128 == 0 || p.N < 192) return launch_n<EPI, 2>(p, stream);
  return launch_n<EPI, 4>(p, stream);
}

}  // namespace

// Host-side shape contract (checked here so a bad call fails loudly instead of faulting on the GPU).
extern "C" const char* cs_gemm_check(const CsGemmParams* p, int epi) {
  if (p->M <= 0 || p->N <= 0 || p->K <= 0) return "gemm: empty shape";
  if (p->K % 64) return "gemm: K must be a multiple of 64";
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
