// Shared device/host helpers for the CrossScore gfx950 kernels.  gfx950 (CDNA4) only: wave64, MFMA with IEEE half (fp16)
// operands and fp32 accumulation (v_mfma_f32_*_f16 runs at the bf16 rate; 11 significant bits instead of 8),
// 160 KiB LDS per CU.  No portability layer on purpose.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t h16_t;  // raw fp16 (IEEE binary16) bits in memory

typedef _Float16 h16x8_t __attribute__((ext_vector_type(8)));
typedef short short8_t __attribute__((ext_vector_type(8)));
typedef short short4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

#define CS_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define CS_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

// fp32 -> fp16 round-to-nearest-even (v_cvt_f16_f32 / v_cvt_pk_f16_f32 on gfx950; NaN stays NaN, |x| > 65504 becomes inf: the
// range every 16-bit activation of this path has to stay in is stated in DESIGN.md 2 and tested with scaled-up weights).
__device__ __forceinline__ h16_t f2h(float x) {
  _Float16 b = (_Float16)x;
  return __builtin_bit_cast(h16_t, b);
}
__device__ __forceinline__ float h2f(h16_t b) { return (float)__builtin_bit_cast(_Float16, b); }
// two values in one v_cvt_pk_f16_f32 (same round-to-nearest-even as f2h)
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 h16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_h16x2(float lo, float hi) {
  const h16x2_t v = __builtin_convertvector(f32x2_t{lo, hi}, h16x2_t);
  return __builtin_bit_cast(uint32_t, v);
}

// ---- operand type of the MFMA kernels: IEEE half (default) or bfloat16 (cs_config.operand_dtype).  Both are 16 bits in memory (h16_t holds
//      the raw bits either way), both MFMA forms take the same cycles; half carries 11 significant bits and is finite to 65504, bfloat16
//      8 bits with fp32's range.  Kernels with MFMAs take the choice as a template parameter, the memory-bound ones as an argument. ----
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
// round-to-nearest-even (v_cvt_pk_bf16_f32 on gfx950; a NaN stays a NaN)
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  const bf16x2_t v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ h16_t f2bf(float x) { return __builtin_bit_cast(h16_t, (__bf16)x); }
__device__ __forceinline__ float bf2f(h16_t b) { return __uint_as_float((uint32_t)b << 16); }
template <bool BF> __device__ __forceinline__ uint32_t pack_o16x2(float lo, float hi) {
  if constexpr (BF) return pack_bf16x2(lo, hi);
  else return pack_h16x2(lo, hi);
}
template <bool BF> __device__ __forceinline__ h16_t f2o(float x) {
  if constexpr (BF) return f2bf(x);
  else return f2h(x);
}
template <bool BF> __device__ __forceinline__ float o2f(h16_t b) {
  if constexpr (BF) return bf2f(b);
  else return h2f(b);
}
// run-time forms for the memory-bound kernels (a wave-uniform select)
__device__ __forceinline__ uint32_t pack_o16x2(float lo, float hi, int bf) { return bf ? pack_bf16x2(lo, hi) : pack_h16x2(lo, hi); }
__device__ __forceinline__ h16_t f2o(float x, int bf) { return bf ? f2bf(x) : f2h(x); }
__device__ __forceinline__ float o2f(h16_t b, int bf) { return bf ? bf2f(b) : h2f(b); }
// fragments are carried as h16x8_t (raw bits) in both modes
template <bool BF> __device__ __forceinline__ f32x4_t mfma_16x16x32(const h16x8_t& a, const h16x8_t& b, const f32x4_t& c) {
  if constexpr (BF) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
template <bool BF> __device__ __forceinline__ f32x16_t mfma_32x32x16(const h16x8_t& a, const h16x8_t& b, const f32x16_t& c) {
  if constexpr (BF) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
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

// GELU(x) = x * Phi(x) with the exact (erf) Phi of HF ACT2FN["gelu"] (HF modeling_dinov2.py:293-297) approximated by
// Phi(x) ~ 0.5 + x * P(x^2), x clamped to +-4.2; P is a degree-7 minimax fit of the GELU error:
// max |GELU_fit - GELU_erf| = 6.7e-5 (2.1e-4 as evaluated in fp32 Horner form, near |x| = 4 where y ~ x; an fp16 half-ulp is
// 4.9e-4 at |y| = 1 and 2e-3 at 4).  12 plain VALU ops, no
// transcendentals: the fc1 epilogue was spending more issue slots on erf (v_exp + v_rcp) than its K loop on MFMAs
// (PMC: SQ_ACTIVE_INST_VALU 51 % vs MFMA pipe busy 26 % with the Abramowitz-Stegun erf).
// Four values per call as two packed-fp32 Horner chains (v_pk_fma_f32) issued alternately from one asm block: a lone chain pays
// a dependent-issue bubble on every packed fma, and the compiler's scheduler serialises the chains again whatever the source
// order.  The fitted Phi stays inside [-1.2e-6, 1 + 1.2e-6] on the clamped range, so it is not clamped again.
__device__ __forceinline__ unsigned long long gelu_c(float c) { return (unsigned long long)__float_as_uint(c); }
__device__ __forceinline__ void gelu_erf4(float (&v)[4]) {
  const f32x2_t xa = {v[0], v[1]}, xb = {v[2], v[3]};
  const f32x2_t ca = {__builtin_amdgcn_fmed3f(v[0], -4.2f, 4.2f), __builtin_amdgcn_fmed3f(v[1], -4.2f, 4.2f)};
  const f32x2_t cb = {__builtin_amdgcn_fmed3f(v[2], -4.2f, 4.2f), __builtin_amdgcn_fmed3f(v[3], -4.2f, 4.2f)};
  const f32x2_t ta = ca * ca, tb = cb * cb;
  const f32x2_t c1 = {8.3297297734e-08f, 8.3297297734e-08f};
  f32x2_t qa, qb;
  // q = c0*t + c1, then q = q*t + c_k: the scalar operand is a register pair whose low half is broadcast (op_sel_hi 0)
  // (leading / trailing s_nop: the packed-fma result hazard against the compiler's own neighbouring instructions)
  asm("s_nop 0\n\t"
      "v_pk_fma_f32 %0, %2, %4, %5 op_sel_hi:[1,0,0]\n\t"
      "v_pk_fma_f32 %1, %3, %4, %5 op_sel_hi:[1,0,0]\n\t"
      "v_pk_fma_f32 %0, %0, %2, %6 op_sel_hi:[1,1,0]\n\t"
      "v_pk_fma_f32 %1, %1, %3, %6 op_sel_hi:[1,1,0]\n\t"
      "v_pk_fma_f32 %0, %0, %2, %7 op_sel_hi:[1,1,0]\n\t"
      "v_pk_fma_f32 %1, %1, %3, %7 op_sel_hi:[1,1,0]\n\t"
      "v_pk_fma_f32 %0, %0, %2, %8 op_sel_hi:[1,1,0]\n\t"
      "v_pk_fma_f32 %1, %1, %3, %8 op_sel_hi:[1,1,0]\n\t"
      "v_pk_fma_f32 %0, %0, %2, %9 op_sel_hi:[1,1,0]\n\t"
      "v_pk_fma_f32 %1, %1, %3, %9 op_sel_hi:[1,1,0]\n\t"
      "v_pk_fma_f32 %0, %0, %2, %10 op_sel_hi:[1,1,0]\n\t"
      "v_pk_fma_f32 %1, %1, %3, %10 op_sel_hi:[1,1,0]\n\t"
      "v_pk_fma_f32 %0, %0, %2, %11 op_sel_hi:[1,1,0]\n\t"
      "v_pk_fma_f32 %1, %1, %3, %11 op_sel_hi:[1,1,0]\n\t"
      "s_nop 0"
      : "=&v"(qa), "=&v"(qb)
      : "v"(ta), "v"(tb), "s"(gelu_c(-9.6129670387e-10f)), "v"(c1), "s"(gelu_c(-3.1398569575e-06f)), "s"(gelu_c(6.8266010957e-05f)),
        "s"(gelu_c(-9.6075936689e-04f)), "s"(gelu_c(9.3374518106e-03f)), "s"(gelu_c(-6.5599355124e-02f)), "s"(gelu_c(3.9850871469e-01f)));
  const f32x2_t ya = xa * __builtin_elementwise_fma(ca, qa, f32x2_t{0.5f, 0.5f});
  const f32x2_t yb = xb * __builtin_elementwise_fma(cb, qb, f32x2_t{0.5f, 0.5f});
  v[0] = ya[0]; v[1] = ya[1]; v[2] = yb[0]; v[3] = yb[1];
}

// ---- kernel parameter blocks (plain structs; launchers live in the matching .hip files) -------------
enum CsEpilogue {
  CS_EPI_BIAS_F16 = 0,        // out_f16[m][n] = acc + bias[n]
  CS_EPI_BIAS_GELU_F16 = 1,   // erf-GELU, degree-7 minimax fit of Phi (gelu_erf4 below: max abs error 2.1e-4 in fp32)
  CS_EPI_BIAS_RELU_F16 = 2,
  CS_EPI_BIAS_LEAKY_F16 = 3,  // slope 0.01
  CS_EPI_RESID_F32 = 4,        // out_f32[m][n] = (resid? resid[m][n]:0) + (scale? scale[n]:1)*(acc+bias[n])
  CS_EPI_PATCH_F32 = 5,        // out_f32[(m + m/Np + 1)][n] = acc + bias[n] + pos[(m%Np+1)][n]
  CS_EPI_HEAD_SCORE = 6,       // score[b][P*i+py][P*j+px] = act(acc + bias[n]), m=b*Np+i*gw+j, n=py*P+px
  // LayerNorm folded into the consuming projection (no separate LN pass over the fp32 residual stream):
  //   LN(x) W^T + b = rstd[m] * (fp16(x) W'^T - mu[m] * s[n]) + c[n],  W' = W*gamma (per input column), s[n] = sum_k W'[n][k],
  //   c[n] = b[n] + sum_k beta[k] W[n][k];  mu / rstd come from per-row partial sums the PRODUCING epilogue wrote.
  CS_EPI_LN_F16 = 7,          // out_f16 = rstd*(acc - mu*s) + c
  CS_EPI_LN_GELU_F16 = 8,     // ... then GELU
  CS_EPI_RESID_F32_LN = 9,     // CS_EPI_RESID_F32 + fp16 copy of the new rows + their partial (sum, sum of squares)
};

struct CsGemmParams {
  const h16_t* A;    // [M][lda] fp16, K contiguous
  const h16_t* W;    // [N][ldw] fp16, K contiguous (nn.Linear layout)
  int lda, ldw;
  int M, N, K;        // K % 64 == 0
  const float* bias;  // [N] or null
  const float* scale; // [N] or null
  const float* resid; // [M][ldr] fp32 or null
  int ldr;
  void* out;          // fp16 or fp32, see epilogue
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
  // CS_EPI_HEAD_SCORE: the per-image mean of the score map in the same launch (score_summariser.py:180-192); all three or none.
  float* mean_part;     // [M][4 x column tiles] fp32 scratch: per patch row, the sum of one wave's columns (any contents on entry)
  unsigned* mean_cnt;   // [M / Np] arrival counters, ZERO on entry; the wave that completes an image's count sums its partials in a fixed
                        // order, writes the mean and puts the counter back to zero
  float* mean_out;      // [M / Np] fp32
  // LayerNorm fold (see CS_EPI_LN_*): producer side (RESID_F32_LN, PATCH_F32) ...
  h16_t* out_f16;    // [rows][ldc] fp16 copy of the fp32 rows written (the next GEMM's A operand), or null
  float* stats_out;    // [rows][stats_sp][2] partial (sum, sumsq) per row: slot = column_tile*4 + wave, or null
  int stats_sp;        //   128-row kernel (gemm.hip): 4 x its column tiles (192 / 128 wide); 256-tile kernel (gemm256.hip, LN = 2): N / 64
  // ... consumer side (LN_BF16, LN_GELU_BF16); `bias` carries c[n]
  const float* ln_part;  // 128-row kernel: [M][ln_sp][2] partial sums of the A rows (fp32 values before fp16 rounding), ln_sp in {4, 8, 16};
  int ln_sp;             // 256-tile kernel (LN = 1): ln_sp == 1 and ln_part = FINALISED rows [ceil(M / 256) * 256][2] = (mean, rstd) from
                         // cs_ln_finalize_launch -- whole 256-row tiles are fetched, so the buffer must hold the padded row count
  const float* col_s;    // [N] s[n]
  float ln_eps;
  int ablate;         // debug timing builds only (CS_ABLATE); 0 in the product
  int bf16;           // operand type of A, W and of a 16-bit output: 0 IEEE half, 1 bfloat16
};

struct CsAttnParams {
  const h16_t* Q; const h16_t* K; const h16_t* V; h16_t* O;
  int ldq, ldk, ldv, ldo;                              // row strides (elements)
  long long q_bs, k_bs, v_bs, o_bs;                    // batch strides (elements)
  int Lq, Lk, heads;
  int nbatch;                                          // filled by the launcher
  float scale_log2e;                                   // (1/sqrt(dh)) * log2(e)
  float* lse;                                          // optional [batch][heads][Lq]: m*ln2-scaled log-sum-exp (base 2)
  int bf16;                                            // operand type of Q, K, V, P and O: 0 IEEE half, 1 bfloat16
};

// One-pass input stage (SURVEY.md 8f-4 as worded: uint8 in, tokens out).  Filter tables of one resize geometry (preprocess.hip) and the per-image
// descriptor the patch kernel reads (patch.hip); images of one launch may differ in source size and geometry, not in the H x W window they produce.
struct CsU8Tables {
  const int *xmin, *xsize, *ymin, *ysize;  // per resized column / row: first source tap, number of taps
  const float *wx, *wy;                    // [rs_w][taps_x], [rs_h][taps_y] normalised triangle weights
  int taps_x, taps_y;
};
struct CsU8Desc {
  const uint8_t* data;  // device HWC RGB; null = the all-zero placeholder image (nvs_dataset.py:459-470: zeros BEFORE T.Normalize)
  CsU8Tables t;
  int row_bytes, crop_y, crop_x, pad_;
};

// Encoder "token panel" kernel (panel.hip): one launch per DINOv2 layer does, for 128-row panels of the residual stream,
//   x += attn_o Wo'^T + bo'                 (attention output projection, LayerScale folded; HF modeling_dinov2.py:249-252,365-370)
//   x += GELU(LN2(x) W1'^T + b1') W2'^T + b2'   (norm2 + MLP + LayerScale; HF:373-378, 293-297)
//   u_out = fp16((x - mean) * rstd)         (norm1 of the NEXT layer without gamma/beta: they are folded into its QKV projection)
// The weights arrive as one pre-packed stream of 24-KiB "units" in the exact LDS image / consumption order (cs_panel_pack_*).
struct CsPanelParams {
  float* x;                // [M][C] fp32 residual stream, updated in place
  const h16_t* attn_o;    // [M][C] fp16 attention output, or null: no out-projection (x is taken as is)
  const h16_t* img;       // unit stream: [12 Wo units (if attn_o)] [96 MLP units]
  const float* bo;         // [C] out-projection bias (LayerScale folded), used when attn_o
  const float* b1;         // [4C] fc1 bias with LN2 beta folded in
  const float* b2;         // [C] fc2 bias (LayerScale folded)
  h16_t* u_out;           // [M][C] or null
  int M;
  float eps;               // LayerNorm eps (1e-6 in DINOv2)
  int bf16;                // operand type of attn_o, img, the hidden slices and u_out: 0 IEEE half, 1 bfloat16
};

// Decoder linear + residual + LayerNorm in one launch (rowln.hip): out = LN(resid + A W^T + bias), K = N = C
struct CsRowLnParams {
  const h16_t* A; int lda;      // [M][lda] 16-bit activations, K = C contiguous
  const h16_t* W; int ldw;      // [C][ldw] 16-bit weights (nn.Linear layout)
  const float* bias;            // [C]
  const float* resid; int ldr;  // [M][ldr] fp32 or null (no short cut)
  const float* gamma; const float* beta; float eps;
  float* out_f32; h16_t* out_f16;  // [M][C] each (out_f32 may alias resid: a lane reads exactly the elements it writes)
  int M;
  // optional second stage: out2 = act2(LN rows (16-bit) x W2^T + bias2), n2 a multiple of C (the sub-block's NEXT linear: the cross-attention's
  // Q projection, linear1 + ReLU, the next layer's packed QKV projection, the head's first linear + LeakyReLU); n2 = 0: none
  const h16_t* W2; int ldw2;    // [n2][ldw2]
  const float* bias2;           // [n2]
  h16_t* out2; int ld2;         // [M][ld2]
  int n2, act2;                 // act2: 0 none, 1 ReLU, 2 LeakyReLU(0.01)
};
