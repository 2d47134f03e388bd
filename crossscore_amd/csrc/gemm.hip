// bf16 MFMA GEMM with fused epilogues for the CrossScore hot path (gfx950).
//
//   acc[m][n] = sum_k A[m][k] * W[n][k]      A:[M][K] bf16 (activations), W:[N][K] bf16 (nn.Linear layout)
//
// Replaces the eager nn.Linear / Conv2d-patchify op groups K2,K4,K6,K7,K10,K12,K13,K15-K18 of SURVEY.md 2a
// (HF modeling_dinov2.py:148,211-213,250,293-297; torch functional.py:5785-5852; cross_reference.py:45-50).
//
// Structure: 128 x {256,192,128} x 32 tiles, 4 waves (1 x 4), wave tile 128 x {64,48,32} of
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
#include <stdlib.h>

namespace {

constexpr int BK = 32;

// WM = wave rows: block tile = (128*WM) x (64*NSUB), 4*WM waves
template <int NSUB, int WM> struct GemmCfg {
  static constexpr int BM = 128 * WM;
  static constexpr int A_BYTES = BM * BK * 2;
  static constexpr int WAVES = 4 * WM;
  static constexpr int BN = 64 * NSUB;
  static constexpr int STAGE_BYTES = (BM + BN) * BK * 2;           // 24 / 20 / 16 KiB
#ifdef CS_NS_OVERRIDE
  static constexpr int NS = CS_NS_OVERRIDE;
#else
  static constexpr int NS = (WM == 2 || NSUB == 4) ? 2 : 3;        // ring slots (deeper rings measured slower: the LDS-DMA
                                                                   // path is throughput bound, a fuller queue only blocks issue)
#endif
  static constexpr int BLOCKS_PER_CU = WM == 2 ? 1 : 2;
  static constexpr int D = NS - 1;                                 // K slices kept in flight
  static constexpr int WL = (4 * NSUB + WAVES - 1) / WAVES;        // W LDS-DMA instructions per wave per slice (BN=192, 8 waves: 2, 4 duplicated)
  static constexpr int LPS = 2 + WL;                               // LDS-DMA instructions per wave per slice
  static constexpr int RING = NS * STAGE_BYTES;
  static constexpr int WN = 16 * NSUB;                             // wave tile columns
  static constexpr int PROW_F = WN * 4 + 16;                       // epilogue patch row (fp32), padded: conflict-free b128 writes
  static constexpr int PROW_H = WN * 2 + 16;                       // epilogue patch row (bf16)
  static constexpr int PATCH = 16 * PROW_F;                        // one 16-row patch per wave
  static constexpr int LDS = RING + WAVES * PATCH;
};

#define CS_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

#ifdef CS_ABLATE  // timing-only debug builds (tools/gemm_ablate.py): bit0 no epilogue, bit1 no MFMA, bit2 no LDS-DMA
#define CS_ABL(bit) (p.ablate & (bit))
#else
#define CS_ABL(bit) 0
#endif

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

// NSUB = 16-column sub-tiles per wave; block tile = 128 x (64*NSUB); 4 waves side by side in N, wave tile 128 x 16*NSUB.
// K is walked in 32-deep slices through an NS-slot LDS ring filled by LDS-DMA; D = NS-1 slices stay in flight across
// barriers (counted s_waitcnt vmcnt, raw s_barrier), also across tile boundaries.  TWO such blocks are resident per CU
// (one wave of each per SIMD): with K of only 384-1536 the epilogue (activation, transpose, stores) costs as much issue
// time as the K loop, and it can only hide under MFMAs that belong to ANOTHER block in a different phase.
template <int EPI, int NSUB, int WM>
__global__ __launch_bounds__(256 * WM, 2) void cs_gemm_kernel(CsGemmParams p) {
  using Cfg = GemmCfg<NSUB, WM>;
  constexpr int BN = Cfg::BN, NS = Cfg::NS, D = Cfg::D, LPS = Cfg::LPS, STAGE_BYTES = Cfg::STAGE_BYTES;
  constexpr int BM = Cfg::BM, A_BYTES = Cfg::A_BYTES, WAVES = Cfg::WAVES;
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
  const int panels_x = (tiles_m - xcd + 7) / 8;  // row panels owned by this XCD
  const int ntile_x = panels_x * tiles_n;
  const int my_tiles = slot < ntile_x ? (ntile_x - slot + slots - 1) / slots : 0;
  const int total = my_tiles * nk;  // K slices this block consumes
  auto tile_of = [&](int idx, int& m0, int& n0) {
    const int pl = idx / tiles_n;
    m0 = (pl * 8 + xcd) * BM;
    n0 = (idx - pl * tiles_n) * BN;
  };

  // ---- LDS-DMA staging: one wave-instruction = 16 rows x 64 B.  Lane i writes LDS chunk (i&3) of row (i>>2); it
  //      fetches source chunk (i&3) ^ 2*((row>>2)&1), and ds_read_b128 applies the same XOR: conflict free. ----
  const int srow = lane >> 2;
  const int gchunk = (lane & 3) ^ (((srow >> 2) & 1) << 1);
  int offA[2], offW[Cfg::WL];
  auto set_tile = [&](int m0, int n0) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int ga = m0 + (j * WAVES + wv) * 16 + srow;  // A tile: 8*WM groups of 16 rows, 2 per wave
      ga = ga < p.M ? ga : p.M - 1;
      offA[j] = ga * p.lda + gchunk * 8;
    }
#pragma unroll
    for (int j = 0; j < Cfg::WL; ++j) {
      int grp = j * WAVES + wv;                    // W tile: 4*NSUB groups of 16 rows
      if (grp >= 4 * NSUB) grp = wv;               // 192-wide tile on 8 waves: waves 4-7 re-issue their first group (uniform LPS)
      int gw = n0 + grp * 16 + srow;
      gw = gw < p.N ? gw : p.N - 1;
      offW[j] = gw * p.ldw + gchunk * 8;
    }
  };
  auto stage = [&](int ring, int kt) {
    char* base = smem + ring * STAGE_BYTES;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_global_load_lds(CS_GLOBAL_PTR(p.A + offA[j] + kt * BK), CS_LDS_PTR(base + (j * WAVES + wv) * 1024), 16, 0, 0);
#pragma unroll
    for (int j = 0; j < Cfg::WL; ++j) {
      int grp = j * WAVES + wv;
      if (grp >= 4 * NSUB) grp = wv;
      __builtin_amdgcn_global_load_lds(CS_GLOBAL_PTR(p.W + offW[j] + kt * BK), CS_LDS_PTR(base + A_BYTES + grp * 1024), 16, 0, 0);
    }
  };

  // stagger: blocks that share a CU run the same program on equal tiles and would stay in phase (all in the K loop, then
  // all in the epilogue); delaying every other resident block once desynchronises them so one block's epilogue
  // (VALU/LDS/stores) runs under the other's MFMAs
  if (p.stagger_ticks > 0) {
    const bool late = p.stagger_mode == 0 ? (slot >= (slots >> 1)) : (slot & 1);
    if (late) {
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)p.stagger_ticks) __builtin_amdgcn_s_sleep(32);
    }
  }

  // load cursor (runs D slices ahead of the compute cursor, across tile boundaries)
  int issued = 0, l_idx = slot, l_kt = 0, l_ring = 0;
  if (total > 0) {
    int m0, n0;
    tile_of(l_idx, m0, n0);
    set_tile(m0, n0);
  }
  auto issue_one = [&]() {
    stage(l_ring, l_kt);
    ++issued;
    l_ring = l_ring + 1 == NS ? 0 : l_ring + 1;
    if (++l_kt == nk) {
      l_kt = 0;
      l_idx += slots;
      if (l_idx < ntile_x) {
        int m0, n0;
        tile_of(l_idx, m0, n0);
        set_tile(m0, n0);
      }
    }
  };
  for (int s0 = 0; s0 < D; ++s0)
    if (issued < total) issue_one();

  const int frow = lane & 15;
  const int coff = ((lane >> 4) ^ (((frow >> 2) & 1) << 1)) * 16;
  int it = 0, c_ring = 0;
  for (int idx = slot; idx < ntile_x; idx += slots) {
    f32x4_t acc[8][NSUB];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < NSUB; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    int cm0, cn0;
    tile_of(idx, cm0, cn0);
    for (int kt = 0; kt < nk; ++kt) {
      // slice `it` must have landed: all but the `ahead` younger slices of this wave's LDS-DMA queue are done
      const int ahead = min(total - 1 - it, D - 1);
      if (ahead >= D - 1) CS_VMCNT((D - 1) * LPS);
      else if (D > 3 && ahead == 2) CS_VMCNT(2 * LPS);
      else if (D > 2 && ahead == 1) CS_VMCNT(1 * LPS);
      else CS_VMCNT(0);
      __builtin_amdgcn_s_barrier();  // every wave's part of slice `it` landed; everyone has left ring slot (it-1)%NS
      asm volatile("" ::: "memory");
      if (issued < total && !CS_ABL(4)) issue_one();  // refill the slot just vacated
      else if (issued < total) ++issued;
      const char* sa = smem + c_ring * STAGE_BYTES + (wm * 128 + frow) * 64 + coff;
      const char* sw = smem + c_ring * STAGE_BYTES + A_BYTES + (wn * 16 * NSUB + frow) * 64 + coff;
      if (!CS_ABL(2)) {
        // all 8 + NSUB fragment reads of the slice are issued before the first MFMA (the compiler then waits with counted
        // lgkmcnt per MFMA group): one LDS latency per slice instead of one per 8 MFMAs
        bf16x8_t fw[NSUB], fa[8];
#pragma unroll
        for (int j = 0; j < NSUB; ++j) fw[j] = *reinterpret_cast<const bf16x8_t*>(sw + j * 16 * 64);
#pragma unroll
        for (int i = 0; i < 8; ++i) fa[i] = *reinterpret_cast<const bf16x8_t*>(sa + i * 16 * 64);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < NSUB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[j], fa[i], acc[i][j], 0, 0, 0);
        // pin the interleave: W fragments + three A fragments up front, then one A-fragment read per MFMA group, always two
        // groups ahead of its use, so each counted lgkmcnt waits on a read issued ~2*NSUB MFMAs earlier
        __builtin_amdgcn_sched_group_barrier(0x100, NSUB + 3, 0);
#pragma unroll
        for (int i = 0; i < 5; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, NSUB, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 3 * NSUB, 0);
      }
      ++it;
      c_ring = c_ring + 1 == NS ? 0 : c_ring + 1;
    }

    // ---- epilogue.  The MFMA leaves each lane with 4 consecutive columns of ONE row, so direct stores would write
    //      32-byte pieces of 16 different rows per instruction (measured: the store tail, not the K loop, set the kernel
    //      time).  Each wave therefore transposes its 16-row sub-tiles through a private LDS patch and stores whole row
    //      segments (full 128-byte lines) with 16-byte accesses; the residual / position addends are read the same way.
    if (CS_ABL(1)) {
      if (acc[0][0][0] == 12345.678f) reinterpret_cast<float*>(p.out)[0] = acc[7][NSUB - 1][3];  // keep acc live
    } else if constexpr (EPI == CS_EPI_HEAD_SCORE) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int m = cm0 + wm * 128 + i * 16 + (lane & 15);
        if (m >= p.M) continue;
        const int b = m / p.Np;
        const int pp = m - b * p.Np;
        const int pi = pp / p.gw, pj = pp - pi * p.gw;
        const int gh = p.Np / p.gw;
        const int Ws = p.P * p.gw;
        float* dst = reinterpret_cast<float*>(p.out) + ((size_t)b * gh * p.P + (size_t)pi * p.P) * Ws + pj * p.P;
#pragma unroll
        for (int j = 0; j < NSUB; ++j) {
          const int n = cn0 + wn * Cfg::WN + j * 16 + (lane >> 4) * 4;
          if (n >= p.N) continue;
          const float4 b4 = p.bias ? *reinterpret_cast<const float4*>(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
          const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int nn = n + r;
            const int py = nn / p.P, px = nn - py * p.P;
            dst[(size_t)py * Ws + px] = head_activation(acc[i][j][r] + bb[r], p.act, p.powp);
          }
        }
      }
    } else {
      constexpr bool kHalf = EPI == CS_EPI_BIAS_BF16 || EPI == CS_EPI_BIAS_GELU_BF16 || EPI == CS_EPI_BIAS_RELU_BF16 ||
                             EPI == CS_EPI_BIAS_LEAKY_BF16;
      char* patch = smem + Cfg::RING + wv * Cfg::PATCH;
      const int ncol0 = cn0 + wn * Cfg::WN;
      float4 bias4[NSUB], scale4[NSUB];
#pragma unroll
      for (int j = 0; j < NSUB; ++j) {
        const int n = min(ncol0 + j * 16 + (lane >> 4) * 4, p.N - 4);
        bias4[j] = p.bias ? *reinterpret_cast<const float4*>(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
        scale4[j] = (EPI == CS_EPI_RESID_F32 && p.scale) ? *reinterpret_cast<const float4*>(p.scale + n) : make_float4(1.f, 1.f, 1.f, 1.f);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        // 1) activation in the accumulator layout, then into the patch: row = lane&15, columns j*16 + (lane>>4)*4 .. +3
#pragma unroll
        for (int j = 0; j < NSUB; ++j) {
          float v[4];
          v[0] = acc[i][j][0] + bias4[j].x; v[1] = acc[i][j][1] + bias4[j].y;
          v[2] = acc[i][j][2] + bias4[j].z; v[3] = acc[i][j][3] + bias4[j].w;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if constexpr (EPI == CS_EPI_BIAS_GELU_BF16) v[r] = gelu_erf(v[r]);
            if constexpr (EPI == CS_EPI_BIAS_RELU_BF16) v[r] = fmaxf(v[r], 0.f);
            if constexpr (EPI == CS_EPI_BIAS_LEAKY_BF16) v[r] = v[r] >= 0.f ? v[r] : 0.01f * v[r];
          }
          if constexpr (kHalf) {
            uint2 o;
            o.x = pack_bf16x2(v[0], v[1]);
            o.y = pack_bf16x2(v[2], v[3]);
            *reinterpret_cast<uint2*>(patch + (lane & 15) * Cfg::PROW_H + (j * 16 + (lane >> 4) * 4) * 2) = o;
          } else {
            if constexpr (EPI == CS_EPI_RESID_F32) { v[0] *= scale4[j].x; v[1] *= scale4[j].y; v[2] *= scale4[j].z; v[3] *= scale4[j].w; }
            *reinterpret_cast<float4*>(patch + (lane & 15) * Cfg::PROW_F + (j * 16 + (lane >> 4) * 4) * 4) = make_float4(v[0], v[1], v[2], v[3]);
          }
        }
        // 2) whole row segments out of the patch: 16 bytes per lane, LPR lanes per row
        const int mrow0 = cm0 + wm * 128 + i * 16;
        if constexpr (kHalf) {
          constexpr int CPR = Cfg::WN * 2 / 16;            // 16-byte chunks per row: 8 / 6 / 4
          constexpr int LPR = NSUB == 2 ? 4 : 8;           // lanes assigned per row
          constexpr int RPI = 64 / LPR;                    // rows per instruction
#pragma unroll
          for (int rr = 0; rr < 16; rr += RPI) {
            const int row = rr + lane / LPR, ch = lane % LPR;
            const int m = mrow0 + row, n = ncol0 + ch * 8;
            if (ch < CPR) {
              const uint4 o = *reinterpret_cast<const uint4*>(patch + row * Cfg::PROW_H + ch * 16);
              if (m < p.M && n < p.N) *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.out) + (size_t)m * p.ldc + n) = o;
            }
          }
        } else {
          constexpr int CPR = Cfg::WN * 4 / 16;            // 16 / 12 / 8
          constexpr int LPR = NSUB == 2 ? 8 : 16;
          constexpr int RPI = 64 / LPR;
#pragma unroll
          for (int rr = 0; rr < 16; rr += RPI) {
            const int row = rr + lane / LPR, ch = lane % LPR;
            const int m = mrow0 + row, n = ncol0 + ch * 4;
            if (ch < CPR) {
              float4 o = *reinterpret_cast<const float4*>(patch + row * Cfg::PROW_F + ch * 16);
              if (m < p.M && n < p.N) {
                if constexpr (EPI == CS_EPI_RESID_F32) {
                  if (p.resid) {
                    const float4 r4 = *reinterpret_cast<const float4*>(p.resid + (size_t)m * p.ldr + n);
                    o.x += r4.x; o.y += r4.y; o.z += r4.z; o.w += r4.w;
                  }
                  *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + (size_t)m * p.ldc + n) = o;
                } else {  // CS_EPI_PATCH_F32: token row m of image img lands at row m + img + 1 (CLS rows interleaved)
                  const int img = m / p.Np;
                  const int pp = m - img * p.Np;
                  const float4 e4 = *reinterpret_cast<const float4*>(p.pos + (size_t)(pp + 1) * p.ldc + n);
                  o.x += e4.x; o.y += e4.y; o.z += e4.z; o.w += e4.w;
                  *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + (size_t)(m + img + 1) * p.ldc + n) = o;
                }
              }
            }
          }
        }
      }
    }
  }
}

int g_num_cus = 0;

template <int EPI, int NSUB, int WM>
hipError_t launch_n(const CsGemmParams& p, hipStream_t stream) {
  constexpr int BN = 64 * NSUB, BM = 128 * WM;
  constexpr int LDS = GemmCfg<NSUB, WM>::LDS;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(cs_gemm_kernel<EPI, NSUB, WM>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
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
  // two persistent blocks per CU (LDS and registers admit two); grid is a multiple of 8 so that b%8 labels the XCD group
  int grid = (GemmCfg<NSUB, WM>::BLOCKS_PER_CU * g_num_cus / 8) * 8;
#ifdef CS_ABLATE
  if (const char* e = getenv("CS_GEMM_GRID")) grid = atoi(e);
#endif
  const int need = ((tiles_m + 7) / 8) * tiles_n * 8;
  if (grid > need) grid = need;
  if (grid < 8) grid = 8;
  hipLaunchKernelGGL((cs_gemm_kernel<EPI, NSUB, WM>), dim3(grid), dim3(256 * WM), LDS, stream, p);
  return hipGetLastError();
}

template <int EPI>
hipError_t launch(const CsGemmParams& p, hipStream_t stream) {
  // column tile: 256 where it divides N, else 192, else 128 (every N of the path is a multiple of 384, 128 or the 196 head)
  // 192 first: its ring has 3 slots inside the 80 KiB a block may use with two blocks per CU (256-wide has only 2)
#ifdef CS_ABLATE
  // 256-row tiles (8 waves, one block per CU) stage ~30 % fewer bytes per FLOP; measured (tools/gemm_ablate.py, QKV shape):
  // LDS-DMA stream alone 50 vs 60 us, whole kernel 91 vs 83 us -- with one block per CU the epilogue no longer hides.
  // Kept for experiments only.
  const bool tall = p.tall > 0 && p.M >= 4096;
  if (tall && p.N % 192 == 0) return launch_n<EPI, 3, 2>(p, stream);
  if (tall && p.N % 256 == 0) return launch_n<EPI, 4, 2>(p, stream);
#endif
  if (p.N % 192 == 0) return launch_n<EPI, 3, 1>(p, stream);
  if (p.N % 256 == 0) return launch_n<EPI, 4, 1>(p, stream);
  if (p.N % 128 == 0 || p.N < 192) return launch_n<EPI, 2, 1>(p, stream);
  return launch_n<EPI, 4, 1>(p, stream);
}

}  // namespace

// Host-side shape contract (checked here so a bad call fails loudly instead of faulting on the GPU).
extern "C" const char* cs_gemm_check(const CsGemmParams* p, int epi) {
  if (p->M <= 0 || p->N <= 0 || p->K <= 0) return "gemm: empty shape";
  if (p->K % 64) return "gemm: K must be a multiple of 64";
  if (p->N % 4 || p->ldc % 4) return "gemm: N and ldc must be multiples of 4";
  if (epi <= CS_EPI_BIAS_LEAKY_BF16 && (p->N % 8 || p->ldc % 8)) return "gemm: bf16 outputs need N and ldc multiples of 8 (16-byte row stores)";
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

extern "C" hipError_t cs_gemm_launch(const CsGemmParams* p0, int epi, hipStream_t stream) {
  CsGemmParams pp = *p0;
#ifdef CS_ABLATE
  if (const char* e = getenv("CS_GEMM_ABLATE")) pp.ablate = atoi(e);
  if (const char* e = getenv("CS_GEMM_STAGGER")) pp.stagger_ticks = atoi(e);
  if (const char* e = getenv("CS_GEMM_STAGGER_MODE")) pp.stagger_mode = atoi(e);
  if (const char* e = getenv("CS_GEMM_TALL")) pp.tall = atoi(e);
#endif
  const CsGemmParams* p = &pp;
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
