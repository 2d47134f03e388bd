// Shared device/host helpers for the CrossScore gfx950 kernels.  gfx950 (CDNA4) only: wave64, MFMA bf16,
// 160 KiB LDS per CU.  No portability layer on purpose.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;  // raw bf16 bits in memory

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef short short8_t __attribute__((ext_vector_type(8)));
typedef short short4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

#define CS_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define CS_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

// fp32 -> bf16 round-to-nearest-even.  A plain cast lowers to v_cvt_pk_bf16_f32 on gfx950 and keeps NaNs.
__device__ __forceinline__ bf16_t f2bf(float x) {
  __bf16 b = (__bf16)x;
  return __builtin_bit_cast(bf16_t, b);
}
__device__ __forceinline__ float bf2f(bf16_t b) { return __uint_as_float(((uint32_t)b) << 16); }
// two values in one v_cvt_pk_bf16_f32 (same round-to-nearest-even as f2bf)
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  const bf16x2_t v = __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_t);
  return __builtin_bit_cast(uint32_t, v);
}

// wave64 butterfly reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- kernel parameter blocks (plain structs; launchers live in the matching .hip files) -------------
enum CsEpilogue {
  CS_EPI_BIAS_BF16 = 0,        // out_bf16[m][n] = acc + bias[n]
  CS_EPI_BIAS_GELU_BF16 = 1,   // exact erf GELU
  CS_EPI_BIAS_RELU_BF16 = 2,
  CS_EPI_BIAS_LEAKY_BF16 = 3,  // slope 0.01
  CS_EPI_RESID_F32 = 4,        // out_f32[m][n] = (resid? resid[m][n]:0) + (scale? scale[n]:1)*(acc+bias[n])
  CS_EPI_PATCH_F32 = 5,        // out_f32[(m + m/Np + 1)][n] = acc + bias[n] + pos[(m%Np+1)][n]
  CS_EPI_HEAD_SCORE = 6,       // score[b][P*i+py][P*j+px] = act(acc + bias[n]), m=b*Np+i*gw+j, n=py*P+px
  // LayerNorm folded into the consuming projection (no separate LN pass over the fp32 residual stream):
  //   LN(x) W^T + b = rstd[m] * (bf16(x) W'^T - mu[m] * s[n]) + c[n],  W' = W*gamma (per input column), s[n] = sum_k W'[n][k],
  //   c[n] = b[n] + sum_k beta[k] W[n][k];  mu / rstd come from per-row partial sums the PRODUCING epilogue wrote.
  CS_EPI_LN_BF16 = 7,          // out_bf16 = rstd*(acc - mu*s) + c
  CS_EPI_LN_GELU_BF16 = 8,     // ... then GELU
  CS_EPI_RESID_F32_LN = 9,     // CS_EPI_RESID_F32 + bf16 copy of the new rows + their partial (sum, sum of squares)
};

struct CsGemmParams {
  const bf16_t* A;    // [M][lda] bf16, K contiguous
  const bf16_t* W;    // [N][ldw] bf16, K contiguous (nn.Linear layout)
  int lda, ldw;
  int M, N, K;        // K % 64 == 0
  const float* bias;  // [N] or null
  const float* scale; // [N] or null
  const float* resid; // [M][ldr] fp32 or null
  int ldr;
  void* out;          // bf16 or fp32, see epilogue
  int ldc;
  // CS_EPI_PATCH_F32 / CS_EPI_HEAD_SCORE extras
  const float* pos;   // [(1+Np)][ldc] position table (patch)
  int bpc;            // persistent blocks per CU to launch (0 = 2, the number that is resident)
  const float* pmean; // patch epilogue: [M][4] per-row channel means removed by im2col (nullptr: none), added back as pmean . wsum
  const float* wsum;  // [3][ldc] fp32 sums of the patch weights over each channel's P*P taps
  int Np;             // patches per image
  int gw;             // patch-grid width
  int P;              // patch size (head)
  int act;            // 0 sigmoid, 1 tanh
  float powp;         // 1 -> identity
  // LayerNorm fold (see CS_EPI_LN_*): producer side (RESID_F32_LN, PATCH_F32) ...
  bf16_t* out_bf16;    // [rows][ldc] bf16 copy of the fp32 rows written (the next GEMM's A operand), or null
  float* stats_out;    // [rows][stats_sp][2] partial (sum, sumsq) per row: slot = column_tile*4 + wave, or null
  int stats_sp;
  // ... consumer side (LN_BF16, LN_GELU_BF16); `bias` carries c[n]
  const float* ln_part;  // [M][ln_sp][2] partial sums of the A rows (fp32 values before bf16 rounding)
  int ln_sp;             // 4, 8 or 16
  const float* col_s;    // [N] s[n]
  float ln_eps;
  int ablate;         // debug timing builds only (CS_ABLATE); 0 in the product
};

struct CsAttnParams {
  const bf16_t* Q; const bf16_t* K; const bf16_t* V; bf16_t* O;
  int ldq, ldk, ldv, ldo;                              // row strides (elements)
  long long q_bs, k_bs, v_bs, o_bs;                    // batch strides (elements)
  int Lq, Lk, heads;
  int nbatch;                                          // filled by the launcher
  float scale_log2e;                                   // (1/sqrt(dh)) * log2(e)
  float* lse;                                          // optional [batch][heads][Lq]: m*ln2-scaled log-sum-exp (base 2)
  int o_split;                                         // 0, or C: O rows are [hi | lo | hi] (3C wide, split-bf16 operand of the out-projection)
};
