// Helpers shared by the two token-panel kernels (panel.hip: 8 waves in role-split pairs; panel4.hip: 4 waves, one per SIMD): counted-wait
// macros, compile-time loops, the packed-half GELU, inline-asm LDS access and LDS-DMA pieces.  Internal linkage (included inside each
// translation unit's anonymous namespace).
#pragma once
#include "cs_common.h"
#include <type_traits>
#include <utility>

namespace {

constexpr int PANEL_FRAG = 1024;  // bytes of one MFMA operand fragment (64 lanes x 16 B)

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

// ---- packed-half GELU (r5; both operand modes).  The fc1 accumulators are rounded to half FIRST -- the tensor the reference's own
//      `16-mixed` run hands to its GELU (config/default_predict.yaml:25; HF modeling_dinov2.py:293-297 under autocast) -- and two values
//      travel through every instruction (v_pk_*_f16): 11 instructions per PAIR behind the conversion instead of 11.5 per value.
//      A plain Horner form of Phi in x^2 cannot be evaluated in half precision (terms of magnitude 6 cancel to 0.1 at |x| = 4: 1e-2 of error),
//      so the polynomial runs in a variable in which GELU's correction term is a well-conditioned bump:
//          GELU(x) = relu(x) - |x| Phi(-|x|),    d = clamp(1 - |x| / 4, 0, 1),   z = d^2 - 1/2,   -|x| Phi(-|x|) ~ P6(z)
//      (|x| >= 4: z = -1/2, P6 = -1.3e-4, GELU = relu to 1.3e-4; the reflection about |x| = 4 is smooth to 1e-5, so even powers of d suffice;
//      sum |c_k| 2^-k = 0.76 against a bump of height 0.17: next to no cancellation).  Fit 8.2e-5 (minimax on [0, 4]); as evaluated in half arithmetic, coefficients rounded to half:
//      rms 2.6e-4 for x ~ N(0, 1) against 2.1e-4 of exact-GELU-of-the-rounded-input-rounded (the reference's own arithmetic) and 1.4e-4 of a
//      lone output rounding; max 2.1e-3 at |y| = 2.3 (one ulp there); score-map MAE 0.994e-4 against 0.977e-4 with an exact GELU in the oracle's
//      fp16-operand emulation (tools/gelu_pk16_fit.py writes the coefficients and these numbers).
//      relu, |x| and the clamp are exact; +-inf and values beyond half range behave as relu; the clamp modifier turns a NaN into 0.
//      bf16 operand mode (fp32's range is the point of that mode): only the bounded correction term P6(z) is computed in halves (the rounded
//      input saturates to d = 0 beyond 65504, where the term is -1.3e-4 anyway); relu(x) stays fp32, the sum is formed in fp32
//      (v_fma_mix_f32 reads the half) and rounded once to bf16: 15 instructions per pair instead of 23. ----
struct PkGeluK { unsigned nk, c5, c4, c3, c2, c1, c0, vc6; };
__device__ __forceinline__ PkGeluK pk_gelu_consts() {  // seven SGPRs and one VGPR for the whole MLP phase (VOP3P reads one SGPR per instruction)
  PkGeluK k;
  asm volatile("s_mov_b32 %0, 0xb400b400" : "=s"(k.nk));   // -1/4
  asm volatile("s_mov_b32 %0, 0xbd76bd76" : "=s"(k.c5));   // -1.365234375
  asm volatile("s_mov_b32 %0, 0xb8a3b8a3" : "=s"(k.c4));   // -0.57958984375
  asm volatile("s_mov_b32 %0, 0x3e8a3e8a" : "=s"(k.c3));   //  1.634765625
  asm volatile("s_mov_b32 %0, 0x394a394a" : "=s"(k.c2));   //  0.6611328125
  asm volatile("s_mov_b32 %0, 0xb52cb52c" : "=s"(k.c1));   // -0.3232421875
  asm volatile("s_mov_b32 %0, 0xb086b086" : "=s"(k.c0));   // -0.141357421875
  asm volatile("v_mov_b32 %0, 0x3a443a44" : "=v"(k.vc6));  //  0.783203125
  return k;
}
// Two pairs travel together through six asm blocks of (up to) four instructions: [stage s of pair 0, stage s of pair 1, stage s + 1 of
// pair 0, stage s + 1 of pair 1] -- a dependent instruction never follows its producer directly, and hipcc (which assumes a partial-dword
// write behind every inline-asm result and puts an s_nop between two asm statements that hand a register on) sees one statement per MFMA gap.
//   stages: 0 t = |x|   1 t = clamp(1 - t/4)   2 t = t t - 1/2   3 q = c6 t + c5   4..8 q = q t + c4..c0   9 r = max(x, 0)   10 x = q + r
struct PkGelu { unsigned t0, t1, q0, q1, r0, r1; };
// (a: the two pairs' four pre-activations in fp32 -- bf16 mode only)
template <int B, bool BF>
__device__ __forceinline__ void pk_gelu_block(PkGelu& g, unsigned& x0, unsigned& x1, const PkGeluK& k, float a0, float a1, float a2, float a3) {
  if constexpr (B == 0)
    asm("v_pk_max_f16 %0, %2, %2 neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_max_f16 %1, %3, %3 neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_fma_f16 %0, %0, %4, 1.0 op_sel_hi:[1,1,0] clamp\n\t"
        "v_pk_fma_f16 %1, %1, %4, 1.0 op_sel_hi:[1,1,0] clamp"
        : "=&v"(g.t0), "=&v"(g.t1) : "v"(x0), "v"(x1), "s"(k.nk));
  else if constexpr (B == 1)
    asm("v_pk_fma_f16 %0, %0, %0, -0.5 op_sel_hi:[1,1,0]\n\t"
        "v_pk_fma_f16 %1, %1, %1, -0.5 op_sel_hi:[1,1,0]\n\t"
        "v_pk_fma_f16 %2, %0, %4, %5\n\t"
        "v_pk_fma_f16 %3, %1, %4, %5"
        : "+v"(g.t0), "+v"(g.t1), "=&v"(g.q0), "=&v"(g.q1) : "v"(k.vc6), "s"(k.c5));
  else if constexpr (B == 2)
    asm("v_pk_fma_f16 %0, %0, %2, %4\n\t"
        "v_pk_fma_f16 %1, %1, %3, %4\n\t"
        "v_pk_fma_f16 %0, %0, %2, %5\n\t"
        "v_pk_fma_f16 %1, %1, %3, %5"
        : "+v"(g.q0), "+v"(g.q1) : "v"(g.t0), "v"(g.t1), "s"(k.c4), "s"(k.c3));
  else if constexpr (B == 3)
    asm("v_pk_fma_f16 %0, %0, %2, %4\n\t"
        "v_pk_fma_f16 %1, %1, %3, %4\n\t"
        "v_pk_fma_f16 %0, %0, %2, %5\n\t"
        "v_pk_fma_f16 %1, %1, %3, %5"
        : "+v"(g.q0), "+v"(g.q1) : "v"(g.t0), "v"(g.t1), "s"(k.c2), "s"(k.c1));
  else if constexpr (B == 4 && BF)
    asm("v_pk_fma_f16 %0, %0, %2, %4\n\t"
        "v_pk_fma_f16 %1, %1, %3, %4"
        : "+v"(g.q0), "+v"(g.q1) : "v"(g.t0), "v"(g.t1), "s"(k.c0));
  else if constexpr (B == 5 && BF) {
    float y0, y1, y2, y3;
    asm("v_max_f32 %2, 0, %6\n\t"
        "v_max_f32 %3, 0, %7\n\t"
        "v_max_f32 %4, 0, %8\n\t"
        "v_max_f32 %5, 0, %9\n\t"
        "v_fma_mix_f32 %2, %10, 1.0, %2 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %3, %10, 1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %4, %11, 1.0, %4 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %5, %11, 1.0, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_cvt_pk_bf16_f32 %0, %2, %3\n\t"
        "v_cvt_pk_bf16_f32 %1, %4, %5"
        : "=&v"(x0), "=&v"(x1), "=&v"(y0), "=&v"(y1), "=&v"(y2), "=&v"(y3)
        : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(g.q0), "v"(g.q1));
  } else if constexpr (B == 4)
    asm("v_pk_fma_f16 %0, %0, %4, %6\n\t"
        "v_pk_fma_f16 %1, %1, %5, %6\n\t"
        "v_pk_max_f16 %2, %7, 0 op_sel_hi:[1,0]\n\t"
        "v_pk_max_f16 %3, %8, 0 op_sel_hi:[1,0]"
        : "+v"(g.q0), "+v"(g.q1), "=&v"(g.r0), "=&v"(g.r1) : "v"(g.t0), "v"(g.t1), "s"(k.c0), "v"(x0), "v"(x1));
  else
    asm("v_pk_add_f16 %0, %2, %4\n\t"
        "v_pk_add_f16 %1, %3, %5"
        : "=&v"(x0), "=&v"(x1) : "v"(g.q0), "v"(g.q1), "v"(g.r0), "v"(g.r1));  // (early clobber: x0 must not land on q1 / r1)
}
// block N of the twelve that activate four pairs: pairs 0 and 1 in blocks 0..5, pairs 2 and 3 in blocks 6..11
template <int N, bool BF>
__device__ __forceinline__ void pk_gelu_op(PkGelu& g, unsigned (&x)[4], const PkGeluK& k, const float (&a)[8]) {
  if constexpr (N >= 0 && N < 12) {
    constexpr int P = 2 * (N / 6);
    pk_gelu_block<N % 6, BF>(g, x[P], x[P + 1], k, a[2 * P], a[2 * P + 1], a[2 * P + 2], a[2 * P + 3]);
  }
}

// the lane id from scratch (opaque to the compiler): lane-derived addresses of a late phase are formed from it where they are used, so that
// nothing lane-derived has to stay in a register (or be spilled) across the MFMA loops, whose B-wave side runs at the 256-register limit
__device__ __forceinline__ unsigned fresh_lane() {
  unsigned l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}
// x + (the same lane's value in the other half of the wave), without an index register (v_permlane32_swap exchanges the upper half of its
// first operand with the lower half of its second: afterwards one register holds the lower halves' values twice, the other the upper ones')
__device__ __forceinline__ float add_other_half(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
template <bool BF>
__device__ __forceinline__ u32x4_t pack8(const float (&v)[8]) {
  return u32x4_t{pack_o16x2<BF>(v[0], v[1]), pack_o16x2<BF>(v[2], v[3]), pack_o16x2<BF>(v[4], v[5]), pack_o16x2<BF>(v[6], v[7])};
}

// ---- inline-asm LDS access of the regions the LDS-DMA also writes or that cross waves (hipcc would order every compiler-visible LDS
//      access against the DMA with vmcnt(0)); results are retired by the counted CS_LGKM waits of the callers ----
template <int OFF>
__device__ __forceinline__ void lds_read1(unsigned addr, h16x8_t& w) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(w) : "v"(addr), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void lds_read_f4(unsigned addr, f32x4_t& w) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(w) : "v"(addr), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void lds_read_u4(unsigned addr, u32x4_t& w) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(w) : "v"(addr), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void lds_write16(unsigned addr, u32x4_t v) {
  asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}

// One LDS-DMA piece: 64 lanes x 16 B from `src + OFF` (this lane's address) to the wave-uniform LDS address `dst + OFF`
// (the instruction's immediate offset applies to both addresses)
template <int OFF>
__device__ __forceinline__ void dma_piece(const char* src, unsigned dst) {
  __builtin_amdgcn_global_load_lds(CS_GLOBAL_PTR(src), (__attribute__((address_space(3))) void*)(size_t)dst, 16, OFF, 0);
}


}  // namespace
