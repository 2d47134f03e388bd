// Memory-bound stages of the CrossScore forward (gfx950): im2col for the patchify GEMM, LayerNorm, the fused
// "final LayerNorm + drop CLS + split query/refs + add multi-view PE" stage, CLS row init, the two frozen
// position-table resizes (done once per (h,w)), the optional attention-weight materialisation and per-image
// score means.  All are HBM-bound: 16-byte accesses, one wave per token row, no LDS round trips.
#include "cs_common.h"
#include <atomic>
#include <math.h>

namespace {

// -------------------------------------------------------------------------------------------------------
// im2col: x fp32 (I,3,H,W) -> A fp16 [I*Np][Kp], k = ch*P*P + dy*P + dx (conv weight.reshape(C,588) order,
// HF modeling_dinov2.py:139-149); columns 588..Kp-1 are zero so the GEMM K is a multiple of 64.
// One thread per 8-element output chunk.
// -------------------------------------------------------------------------------------------------------
// Image g = img0 + img of the (B*(1+N)) image batch is the query of item b = g/(1+N) when g%(1+N) == 0, else
// reference view g%(1+N)-1 of item b: the torch.cat of core.py:134-138 is folded into the addressing.
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ xq, const float* __restrict__ xr, int N, int img0,
                                                      h16_t* __restrict__ out, int I, int H, int W, int gh, int gw, int P, int Kp, int bf) {
  const int chunks = Kp / 8;
  const long long total = (long long)I * gh * gw * chunks;
  const int KK = 3 * P * P;
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(g % chunks);
    const long long m = g / chunks;
    const int pj = (int)(m % gw);
    const int pi = (int)((m / gw) % gh);
    const int img = (int)(m / ((long long)gw * gh));
    const int g_img = img0 + img;
    const int bb = g_img / (1 + N), vv = g_img - bb * (1 + N);
    const float* x = vv == 0 ? xq + (size_t)bb * 3 * H * W : xr + ((size_t)bb * N + (vv - 1)) * 3 * H * W;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = c * 8 + e;
      float val = 0.f;
      if (k < KK) {
        const int ch = k / (P * P);
        const int rem = k - ch * P * P;
        const int dy = rem / P, dx = rem - dy * P;
        val = x[((size_t)ch * H + (pi * P + dy)) * W + pj * P + dx];
      }
      v[e] = val;
    }
    uint4 o;
    o.x = pack_o16x2(v[0], v[1], bf); o.y = pack_o16x2(v[2], v[3], bf);
    o.z = pack_o16x2(v[4], v[5], bf); o.w = pack_o16x2(v[6], v[7], bf);
    *reinterpret_cast<uint4*>(out + (size_t)m * Kp + c * 8) = o;
  }
}

// Same result, coalesced on both sides: one block per (image, patch row).  The 3*P image rows of the patch row are read with
// consecutive lanes on consecutive pixels, scattered into an LDS image of the gw output rows ([gw][Kp] fp16, pad columns zero),
// which is then streamed out as one contiguous piece.  (The gather kernel above reads 56-byte runs: 2.3 TB/s; this one 4+.)
// With `pmean`, each patch's per-channel mean (fp32, fixed summation order) is removed before the fp16 rounding and stored as
// pmean[row][ch]; the patch GEMM adds mean * sum(W) back in fp32 (gemm.hip patch_dc).
template <int P>
__global__ __launch_bounds__(256) void im2col_rows_kernel(const float* __restrict__ xq, const float* __restrict__ xr, int N, int img0,
                                                           h16_t* __restrict__ out, int H, int W, int gh, int gw, int Kp,
                                                           float* __restrict__ pmean, int bf) {
  extern __shared__ __attribute__((aligned(16))) char im_smem[];
  h16_t* tile = reinterpret_cast<h16_t*>(im_smem);                                   // [gw][Kp]
  const int Wu = gw * P;
  float* stage = reinterpret_cast<float*>(im_smem + (size_t)gw * Kp * sizeof(h16_t));  // [P][Wu] one channel (centring only)
  float* mean = stage + (size_t)P * Wu;                                                // [gw]
  float* part = mean + gw;                                                             // [gw][P] row sums
  const int img = blockIdx.x / gh, pi = blockIdx.x - img * gh;
  const int g_img = img0 + img;
  const int bb = g_img / (1 + N), vv = g_img - bb * (1 + N);
  const float* x = vv == 0 ? xq + (size_t)bb * 3 * H * W : xr + ((size_t)bb * N + (vv - 1)) * 3 * H * W;
  constexpr int KK = 3 * P * P;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  for (int i = tid; i < gw * (Kp - KK); i += 256) {  // zero padding columns
    const int pj = i / (Kp - KK);
    tile[pj * Kp + KK + (i - pj * (Kp - KK))] = 0;
  }
  if (pmean == nullptr) {
    for (int r = wv; r < 3 * P; r += 4) {  // image row (ch, dy) of this patch row, one wave per row
      const int ch = r / P, dy = r - ch * P;
      const float* src = x + ((size_t)ch * H + (pi * P + dy)) * W;
      for (int xx = lane; xx < Wu; xx += 64) {
        const int pj = xx / P, dx = xx - pj * P;
        tile[pj * Kp + ch * P * P + dy * P + dx] = f2o(src[xx], bf);
      }
    }
  } else {
    for (int ch = 0; ch < 3; ++ch) {
      for (int dy = wv; dy < P; dy += 4) {
        const float* src = x + ((size_t)ch * H + (pi * P + dy)) * W;
        for (int xx = lane; xx < Wu; xx += 64) stage[dy * Wu + xx] = src[xx];
      }
      __syncthreads();
      // fixed summation order (row sums, then the P row sums of a patch): the mean, and every rounding after it, is reproducible
      for (int i = tid; i < gw * P; i += 256) {
        const int pj = i / P, dy = i - pj * P;
        float sacc = 0.f;
        for (int dx = 0; dx < P; ++dx) sacc += stage[dy * Wu + pj * P + dx];
        part[i] = sacc;
      }
      __syncthreads();
      if (tid < gw) {
        float sacc = 0.f;
        for (int dy = 0; dy < P; ++dy) sacc += part[tid * P + dy];
        const float mu = sacc / (float)(P * P);
        mean[tid] = mu;
        pmean[(((size_t)img * gh + pi) * gw + tid) * 4 + ch] = mu;
        if (ch == 0) pmean[(((size_t)img * gh + pi) * gw + tid) * 4 + 3] = 0.f;
      }
      __syncthreads();
      for (int dy = wv; dy < P; dy += 4)
        for (int xx = lane; xx < Wu; xx += 64) {
          const int pj = xx / P, dx = xx - pj * P;
          tile[pj * Kp + ch * P * P + dy * P + dx] = f2o(stage[dy * Wu + xx] - mean[pj], bf);
        }
      __syncthreads();
    }
  }
  __syncthreads();
  uint4* dst = reinterpret_cast<uint4*>(out + ((size_t)img * gh + pi) * gw * Kp);
  const uint4* srcv = reinterpret_cast<const uint4*>(tile);
  for (int i = tid; i < gw * Kp / 8; i += 256) dst[i] = srcv[i];
}

// wsum[ch][n] = sum over the P*P taps of channel ch of the fp32 patch weight W[n][ch*P*P + t]   (one thread per (ch, n))
__global__ void patch_wsum_kernel(const float* __restrict__ w, int C, int PP, float* __restrict__ wsum) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 3 * C) return;
  const int ch = i / C, n = i - ch * C;
  float s = 0.f;
  for (int t = 0; t < PP; ++t) s += w[(size_t)n * 3 * PP + ch * PP + t];
  wsum[(size_t)ch * C + n] = s;
}

// -------------------------------------------------------------------------------------------------------
// LayerNorm over C (biased variance, two-pass in registers): one wave per row, float4 per lane.
// -------------------------------------------------------------------------------------------------------
// LN_MAXV float4 per lane: 4 for C <= 1024 (every BASELINE backbone), 8 for C <= 2048 (dinov2-giant's 1536); the launchers pick
template <int LN_MAXV> struct LnRow {
  float4 v[LN_MAXV];
};

template <int LN_MAXV> __device__ __forceinline__ void ln_load(const float* row, int C4, int lane, LnRow<LN_MAXV>& r) {
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = lane + i * 64;
    r.v[i] = c < C4 ? reinterpret_cast<const float4*>(row)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}
template <int LN_MAXV> __device__ __forceinline__ void ln_normalise(LnRow<LN_MAXV>& r, int C, int C4, int lane, const float* g, const float* b, float eps) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) s += (r.v[i].x + r.v[i].y) + (r.v[i].z + r.v[i].w);
  const float mu = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    if (lane + i * 64 < C4) {
      const float a = r.v[i].x - mu, bb = r.v[i].y - mu, c = r.v[i].z - mu, d = r.v[i].w - mu;
      q += (a * a + bb * bb) + (c * c + d * d);
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = lane + i * 64;
    if (c < C4) {
      const float4 g4 = reinterpret_cast<const float4*>(g)[c];
      const float4 b4 = reinterpret_cast<const float4*>(b)[c];
      r.v[i].x = (r.v[i].x - mu) * rstd * g4.x + b4.x;
      r.v[i].y = (r.v[i].y - mu) * rstd * g4.y + b4.y;
      r.v[i].z = (r.v[i].z - mu) * rstd * g4.z + b4.z;
      r.v[i].w = (r.v[i].w - mu) * rstd * g4.w + b4.w;
    }
  }
}
template <int LN_MAXV> __device__ __forceinline__ void ln_store(const LnRow<LN_MAXV>& r, int C4, int lane, float* of32, h16_t* obf, int bf) {
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = lane + i * 64;
    if (c < C4) {
      if (of32) reinterpret_cast<float4*>(of32)[c] = r.v[i];
      if (obf) {
        uint2 o;
        o.x = pack_o16x2(r.v[i].x, r.v[i].y, bf);
        o.y = pack_o16x2(r.v[i].z, r.v[i].w, bf);
        reinterpret_cast<uint2*>(obf)[c] = o;
      }
    }
  }
}

template <int LN_MAXV>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, int M, int C, const float* __restrict__ g,
                                                         const float* __restrict__ b, float eps, float* of32, h16_t* obf, int bf) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int C4 = C / 4;
  LnRow<LN_MAXV> r;
  ln_load(x + (size_t)row * C, C4, lane, r);
  ln_normalise(r, C, C4, lane, g, b, eps);
  ln_store(r, C4, lane, of32 ? of32 + (size_t)row * C : nullptr, obf ? obf + (size_t)row * C : nullptr, bf);
}

// -------------------------------------------------------------------------------------------------------
// Final encoder LayerNorm (HF:465-470) fused with CLS drop + query/ref split (core.py:142-153) + multi-view PE
// add (positional_encoding.py:42-75).  x: [I*T][C] fp32, image = b*(1+N)+v.  Row (b,v,p):
//   v == 0 : q_f32[b*Np+p], q_f16[b*Np+p]      (decoder residual stream + GEMM operand)
//   v >= 1 : mem_f16[b*N*Np + (v-1)*Np + p]    (cross-attention memory, GEMM operand only)
// -------------------------------------------------------------------------------------------------------
template <int LN_MAXV>
__global__ __launch_bounds__(256) void final_ln_split_kernel(const float* __restrict__ x, int I, int img0, int Np, int C, int N,
                                                              const float* __restrict__ g, const float* __restrict__ b, float eps,
                                                              const float* __restrict__ pe, float* q_f32, h16_t* q_bf,
                                                              h16_t* mem_bf, int bf) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (long long)I * Np) return;
  // (r5, measured and not adopted: ordering the waves so that those adding the same PE rows run close together -- position-major, or tiles of
  //  8 positions x all images -- brings the kernel's fetch traffic from 180 to 110-118 MB, i.e. to the algorithmic 103: the 77 MB on top are
  //  re-fetches of the 2.1-MB PE table, which does not survive 100 MB of streamed rows in a 4-MB L2.  They are served by the Infinity Cache,
  //  not by HBM, and the reordered kernel is SLOWER, 45-47 us against 41: its waves then read and write 1.5-KB rows of different images.)
  const int img = (int)(row / Np);
  const int pp = (int)(row - (long long)img * Np);
  // N < 0: every image is a reference view (reference-token cache): row goes to mem_bf[(img0+img)*Np + pp]
  const int bb = N < 0 ? 0 : (img0 + img) / (1 + N), v = N < 0 ? 1 + img0 + img : (img0 + img) - bb * (1 + N);
  const int C4 = C / 4;
  LnRow<LN_MAXV> r;
  ln_load(x + ((size_t)img * (Np + 1) + 1 + pp) * C, C4, lane, r);
  ln_normalise(r, C, C4, lane, g, b, eps);
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = lane + i * 64;
    if (c < C4) {
      const float4 e = reinterpret_cast<const float4*>(pe + (size_t)pp * C)[c];
      r.v[i].x += e.x; r.v[i].y += e.y; r.v[i].z += e.z; r.v[i].w += e.w;
    }
  }
  if (v == 0) {
    const size_t o = ((size_t)bb * Np + pp) * C;
    ln_store(r, C4, lane, q_f32 + o, q_bf + o, bf);
  } else {
    const size_t o = (((size_t)bb * (N < 0 ? 0 : N) + (v - 1)) * Np + pp) * C;
    ln_store(r, C4, lane, nullptr, mem_bf + o, bf);
  }
}

// CLS rows of the residual stream: x[img*T][c] = cls[c] + pos[0][c]  (HF:108-112).  One wave per image.  With the LayerNorm
// fold the row also gets its fp16 copy and its (sum, sumsq) in partial slot 0 (the other slots are zeroed).
__global__ __launch_bounds__(64) void cls_rows_kernel(float* x, int I, int T, int C, const float* cls, const float* pos, h16_t* xb,
                                                      float* stats, int sp, int bf) {
  const int img = blockIdx.x, lane = threadIdx.x;
  const size_t row = (size_t)img * T;
  float s1 = 0.f, s2 = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float v = cls[c] + pos[c];
    x[row * C + c] = v;
    if (xb) xb[row * C + c] = f2o(v, bf);
    s1 += v;
    s2 += v * v;
  }
  if (stats) {
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    for (int k = lane; k < 2 * sp; k += 64) stats[row * sp * 2 + k] = k == 0 ? s1 : (k == 1 ? s2 : 0.f);
  }
}

// -------------------------------------------------------------------------------------------------------
// Frozen position tables, computed once per (h,w).
// Encoder: bicubic (A=-0.75), align_corners=False, src=(dst+0.5)*in/out-0.5 un-clamped, border-clamped taps
// (HF:57-95 -> aten upsample_bicubic2d).  pos: [(1+G*G)][C] -> out [(1+gh*gw)][C], row 0 copied.
// -------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float cubic1(float x, float A) { return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }
__device__ __forceinline__ float cubic2(float x, float A) { return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }

// `grow` = 0: source coordinate (dst + 0.5) * G / gh - 0.5, what F.interpolate(size=(h, w)) of transformers >= 4.4x computes;
// `grow` = 0.1: (dst + 0.5) * G / (gh + 0.1) - 0.5, what interpolate_pos_encoding of the reference's pinned transformers 4.33.3 /
// torch 2.1.2 computes (scale_factor = ((h + 0.1) / G, (w + 0.1) / G), environment.yaml:293,340).
__global__ void pos_bicubic_kernel(const float* __restrict__ pos, int G, int C, int gh, int gw, float grow, float* __restrict__ out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)(1 + gh * gw) * C;
  if (i >= total) return;
  const int c = (int)(i % C);
  const int t = (int)(i / C);
  if (t == 0) { out[i] = pos[c]; return; }
  const int oy = (t - 1) / gw, ox = (t - 1) - oy * gw;
  const float A = -0.75f;
  const float sy = (oy + 0.5f) * ((float)G / ((float)gh + grow)) - 0.5f;
  const float sx = (ox + 0.5f) * ((float)G / ((float)gw + grow)) - 0.5f;
  const float fy = floorf(sy), fx = floorf(sx);
  const float ty = sy - fy, tx = sx - fx;
  const int iy = (int)fy, ix = (int)fx;
  const float wy[4] = {cubic2(ty + 1.f, A), cubic1(ty, A), cubic1(1.f - ty, A), cubic2(2.f - ty, A)};
  const float wx[4] = {cubic2(tx + 1.f, A), cubic1(tx, A), cubic1(1.f - tx, A), cubic2(2.f - tx, A)};
  float acc = 0.f;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int yy = min(max(iy - 1 + a, 0), G - 1);
    float rowacc = 0.f;
#pragma unroll
    for (int bq = 0; bq < 4; ++bq) {
      const int xx = min(max(ix - 1 + bq, 0), G - 1);
      rowacc += pos[(size_t)(1 + yy * G + xx) * C + c] * wx[bq];
    }
    acc += rowacc * wy[a];
  }
  out[i] = acc;
}

// Multi-view PE: bilinear, align_corners=True: src = dst*(in-1)/(out-1) (positional_encoding.py:61-69).
// PE: [pe_h][pe_w][C] -> out [gh*gw][C]
__global__ void pe_bilinear_kernel(const float* __restrict__ pe, int ph, int pw, int C, int gh, int gw, float* __restrict__ out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)gh * gw * C;
  if (i >= total) return;
  const int c = (int)(i % C);
  const int t = (int)(i / C);
  const int oy = t / gw, ox = t - oy * gw;
  const float scy = gh > 1 ? (float)(ph - 1) / (float)(gh - 1) : 0.f;
  const float scx = gw > 1 ? (float)(pw - 1) / (float)(gw - 1) : 0.f;
  const float sy = oy * scy, sx = ox * scx;
  const int y0 = (int)sy, x0 = (int)sx;
  const int y1 = min(y0 + 1, ph - 1), x1 = min(x0 + 1, pw - 1);
  const float ly = sy - (float)y0, lx = sx - (float)x0;
  const float v00 = pe[((size_t)y0 * pw + x0) * C + c], v01 = pe[((size_t)y0 * pw + x1) * C + c];
  const float v10 = pe[((size_t)y1 * pw + x0) * C + c], v11 = pe[((size_t)y1 * pw + x1) * C + c];
  const float top = v00 * (1.f - lx) + v01 * lx;
  const float bot = v10 * (1.f - lx) + v11 * lx;
  out[i] = top * (1.f - ly) + bot * ly;
}

// The same resize with mode='bicubic' (model.pos_enc.multi_view.interpolate_mode; positional_encoding.py:61-69 passes the mode through with
// align_corners=True): src = dst*(in-1)/(out-1), four taps per axis with border-clamped indices, A = -0.75 (aten upsample_bicubic2d).
__global__ void pe_bicubic_ac_kernel(const float* __restrict__ pe, int ph, int pw, int C, int gh, int gw, float* __restrict__ out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)gh * gw * C;
  if (i >= total) return;
  const int c = (int)(i % C);
  const int t = (int)(i / C);
  const int oy = t / gw, ox = t - oy * gw;
  const float A = -0.75f;
  const float scy = gh > 1 ? (float)(ph - 1) / (float)(gh - 1) : 0.f;
  const float scx = gw > 1 ? (float)(pw - 1) / (float)(gw - 1) : 0.f;
  const float sy = oy * scy, sx = ox * scx;
  const float fy = floorf(sy), fx = floorf(sx);
  const float ty = sy - fy, tx = sx - fx;
  const int iy = (int)fy, ix = (int)fx;
  const float wy[4] = {cubic2(ty + 1.f, A), cubic1(ty, A), cubic1(1.f - ty, A), cubic2(2.f - ty, A)};
  const float wx[4] = {cubic2(tx + 1.f, A), cubic1(tx, A), cubic1(1.f - tx, A), cubic2(2.f - tx, A)};
  float acc = 0.f;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int yy = min(max(iy - 1 + a, 0), ph - 1);
    float rowacc = 0.f;
#pragma unroll
    for (int bq = 0; bq < 4; ++bq) {
      const int xx = min(max(ix - 1 + bq, 0), pw - 1);
      rowacc += pe[((size_t)yy * pw + xx) * C + c] * wx[bq];
    }
    acc += rowacc * wy[a];
  }
  out[i] = acc;
}

// fp32 -> fp16 weight packing (K-contiguous rows; optional zero padding of K to ldo; optional per-output-row scale:
// LayerScale lambda folded into the projection, HF modeling_dinov2.py:277-278, so the GEMM epilogue has no scale operand)
__global__ void pack_f16_kernel(const float* __restrict__ w, int rows, int K, h16_t* __restrict__ out, int ldo,
                                 const float* __restrict__ row_scale, const float* __restrict__ col_scale, int bf) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)rows * ldo) return;
  const int k = (int)(i % ldo);
  const long long rr = i / ldo;
  const float sc = (row_scale ? row_scale[rr] : 1.0f) * ((col_scale && k < K) ? col_scale[k] : 1.0f);
  out[i] = k < K ? f2o(w[rr * K + k] * sc, bf) : (h16_t)0;
}

// LayerNorm fold constants of one projection (see CS_EPI_LN_* in cs_common.h), one wave per output row n:
//   s[n] = sum_k float(Wp[n][k])            over the PACKED fp16 weights W' = W*gamma (what the MFMA really multiplies)
//   c[n] = bias[n] + sum_k beta[k] W[n][k]   in fp32 from the original weights
__global__ __launch_bounds__(256) void ln_fold_consts_kernel(const h16_t* __restrict__ wp, int ldp, const float* __restrict__ w,
                                                             const float* __restrict__ beta, const float* __restrict__ bias, int N,
                                                             int K, float* __restrict__ s_out, float* __restrict__ c_out, int bf) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  float s = 0.f, c = 0.f;
  for (int k = lane; k < K; k += 64) {
    if (wp) s += o2f(wp[(size_t)n * ldp + k], bf);  // (the packed weights' own type: bf16 handles sum bf16 values)
    c += beta[k] * w[(size_t)n * K + k];
  }
  s = wave_sum(s);
  c = wave_sum(c);
  if (lane == 0) { if (s_out) s_out[n] = s; c_out[n] = c + (bias ? bias[n] : 0.f); }
}

__global__ void vec_mul_kernel(const float* a, const float* b, float* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = a[i] * b[i];
}

// -------------------------------------------------------------------------------------------------------
// Optional attention-weight map (cross_reference.py:91-93 / torch functional.py:6576-6612): probabilities of
// ONE head of the last decoder layer's cross-attention, P[b][q][k] = exp2(s*scale*log2e - lse2[b][head][q]).
// lse2 comes from the fused attention kernel, so this pass only recomputes q.k for one head (fp32 dot on the
// fp16 operands the fused kernel used) and streams the fp32 matrix out: HBM-write bound (B*Lq*Lk*4 bytes).
// One wave per (q, 64 keys): lane = key; q row broadcast from registers.
// -------------------------------------------------------------------------------------------------------
template <int DH>
__global__ __launch_bounds__(256) void attn_weights_kernel(CsAttnParams p, int head, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int bat = blockIdx.z;
  const int q = blockIdx.y;
  const int kbase = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;
  if (kbase >= p.Lk) return;
  const int key = kbase + lane;
  const h16_t* qp = p.Q + (size_t)bat * p.q_bs + (size_t)q * p.ldq + head * DH;
  const float lse = p.lse[((size_t)bat * p.heads + head) * p.Lq + q];
  if (key < p.Lk) {
    const h16_t* kp = p.K + (size_t)bat * p.k_bs + (size_t)key * p.ldk + head * DH;
    // the same operand values as cs_attn_kernel: Q pre-multiplied by scale_log2e (rounded to fp16 here when the producer did not fold it)
    const float sc = p.scale_log2e;
    const bool raw = sc != 1.0f;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < DH / 8; ++c) {
      const uint4 kv = *reinterpret_cast<const uint4*>(kp + c * 8);
      const uint4 qv = *reinterpret_cast<const uint4*>(qp + c * 8);
      const uint32_t kw[4] = {kv.x, kv.y, kv.z, kv.w}, qw[4] = {qv.x, qv.y, qv.z, qv.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int bf = p.bf16;
        float q0 = o2f((h16_t)(qw[e] & 0xffffu), bf), q1 = o2f((h16_t)(qw[e] >> 16), bf);
        if (raw) { q0 = o2f(f2o(q0 * sc, bf), bf); q1 = o2f(f2o(q1 * sc, bf), bf); }
        s += o2f((h16_t)(kw[e] & 0xffffu), bf) * q0;
        s += o2f((h16_t)(kw[e] >> 16), bf) * q1;
      }
    }
    out[((size_t)bat * p.Lq + q) * p.Lk + key] = __builtin_amdgcn_exp2f(s - lse);
  }
}

// One wave per workgroup that does nothing for `ticks` of the 100 MHz wall clock: the probes of streams_overlap (api.hip).  Launched
// with dynamic LDS it also limits how many workgroups a CU holds, which is what makes the dispatch of a large grid last.
__global__ void spin_kernel(unsigned long long ticks) {
  extern __shared__ char spin_lds[];
  if (ticks == ~0ull) spin_lds[threadIdx.x] = 0;  // never true: keeps the allocation
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

// The gate of Dinov2SwiGLUFFN (HF modeling_dinov2.py:311-315: x1, x2 = weights_in(u).chunk(2, -1); hidden = silu(x1) * x2), in place on the 16-bit
// rows the weights_in projection wrote: x[m][j] = silu(x[m][j]) * x[m][F + j], j < F, rows `ld` (>= 2 F) apart; fp32 arithmetic, one rounding.
// 8 values per thread (16-byte accesses; F % 8 == 0).
__global__ __launch_bounds__(256) void silu_mul_kernel(h16_t* __restrict__ x, long long n8, int F8, int ld, int bf) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n8) return;
  const long long m = i / F8;
  const int j = (int)(i - m * F8) * 8;
  h16_t* row = x + m * ld;
  const uint4 a = *reinterpret_cast<const uint4*>(row + j), b = *reinterpret_cast<const uint4*>(row + F8 * 8 + j);
  const unsigned aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
  unsigned ow[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float r[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const float x1 = o2f((h16_t)(aw[k] >> (16 * e)), bf), x2 = o2f((h16_t)(bw[k] >> (16 * e)), bf);
      r[e] = x1 / (1.0f + __expf(-x1)) * x2;
    }
    ow[k] = pack_o16x2(r[0], r[1], bf);
  }
  *reinterpret_cast<uint4*>(row + j) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
}

// number of non-finite values of the score map, added to a device counter (cs_nonfinite_count): an fp16 operand that overflowed upstream
// (|x| > 65504 -> inf -> NaN in the next LayerNorm / softmax) reaches every pixel of its image as NaN, so the output is where it shows
__global__ __launch_bounds__(256) void score_check_kernel(const float* __restrict__ score, size_t n, unsigned* __restrict__ counter) {
  unsigned bad = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float v = score[i];
    bad += !(fabsf(v) <= 3.0e38f);  // NaN and +-inf
  }
  for (int o = 32; o >= 1; o >>= 1) bad += __shfl_xor(bad, o, 64);
  if ((threadIdx.x & 63) == 0 && bad) atomicAdd(counter, bad);
}

}  // namespace

// Row statistics of the LayerNorm folded into the 256-tile GEMM's epilogues (gemm256.hip): the producing epilogue left `sp` partial (sum, sum of
// squares) pairs per row, one per 64-column wave slice; a row's (mean, rstd) = biased variance, eps inside the root (HF modeling_dinov2.py:361-380
// LayerNorm).  One thread per row; rows [M, rows_padded) (the consumer fetches whole 256-row tiles) get (0, 0).
__global__ __launch_bounds__(256) void ln_finalize_kernel(const float* __restrict__ part, int M, int rows_padded, int sp, float inv_c, float eps,
                                                          float* __restrict__ stat) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= rows_padded) return;
  float2 o = make_float2(0.f, 0.f);
  if (m < M) {
    const float2* pp = reinterpret_cast<const float2*>(part) + (size_t)m * sp;
    float a = 0.f, b = 0.f;
    for (int k = 0; k < sp; ++k) { const float2 v = pp[k]; a += v.x; b += v.y; }
    const float mu = a * inv_c;
    o = make_float2(mu, 1.0f / sqrtf(fmaxf(b * inv_c - mu * mu, 0.f) + eps));
  }
  reinterpret_cast<float2*>(stat)[m] = o;
}

extern "C" {

hipError_t cs_ln_finalize_launch(const float* part, int M, int rows_padded, int sp, int C, float eps, float* stat, hipStream_t st) {
  hipLaunchKernelGGL(ln_finalize_kernel, dim3((rows_padded + 255) / 256), dim3(256), 0, st, part, M, rows_padded, sp, 1.0f / (float)C, eps, stat);
  return hipGetLastError();
}

hipError_t cs_silu_mul_launch(h16_t* x, int M, int F, int ld, int bf, hipStream_t st) {
  const long long n8 = (long long)M * (F / 8);
  hipLaunchKernelGGL(silu_mul_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, st, x, n8, F / 8, ld, bf);
  return hipGetLastError();
}

hipError_t cs_score_check_launch(const float* score, size_t n, unsigned* counter, hipStream_t st) {
  const unsigned grid = (unsigned)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
  hipLaunchKernelGGL(score_check_kernel, dim3(grid), dim3(256), 0, st, score, n, counter);
  return hipGetLastError();
}

hipError_t cs_patch_wsum_launch(const float* w, int C, int P, float* wsum, hipStream_t st) {
  hipLaunchKernelGGL(patch_wsum_kernel, dim3((3 * C + 255) / 256), dim3(256), 0, st, w, C, P * P, wsum);
  return hipGetLastError();
}

// pmean: [I*gh*gw][4] fp32 or nullptr.  With pmean the patches are mean-centred; when the gather fallback has to be used the
// means are written as zeros (nothing removed, nothing to add back).
hipError_t cs_im2col_launch(const float* xq, const float* xr, int N, int img0, h16_t* out, int I, int H, int W, int P, int Kp,
                            float* pmean, int bf, hipStream_t st) {
  const int gh = H / P, gw = W / P;
  const size_t lds = (size_t)gw * Kp * sizeof(h16_t) + (pmean ? ((size_t)P * gw * P + gw + (size_t)gw * P) * sizeof(float) : 0);
  if (P == 14 && Kp >= 3 * P * P && lds <= 156 * 1024 && (long long)I * gh < (1ll << 31) && gw <= 256) {
    static std::atomic<bool> attr_done[16];  // (zero-initialised; hipFuncSetAttribute is idempotent, a racing second caller only repeats it)  // per device
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return hipErrorInvalidDevice;
    if (!attr_done[dev]) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(im2col_rows_kernel<14>), hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
      if (e != hipSuccess) return e;
      attr_done[dev] = true;
    }
    hipLaunchKernelGGL(im2col_rows_kernel<14>, dim3(I * gh), dim3(256), lds, st, xq, xr, N, img0, out, H, W, gh, gw, Kp, pmean, bf);
    return hipGetLastError();
  }
  if (pmean) {
    hipError_t e = hipMemsetAsync(pmean, 0, (size_t)I * gh * gw * 4 * sizeof(float), st);
    if (e != hipSuccess) return e;
  }
  const long long total = (long long)I * gh * gw * (Kp / 8);
  const int grid = (int)((total + 255) / 256 < 65536 * 4 ? (total + 255) / 256 : 65536 * 4);
  hipLaunchKernelGGL(im2col_kernel, dim3(grid), dim3(256), 0, st, xq, xr, N, img0, out, I, H, W, gh, gw, P, Kp, bf);
  return hipGetLastError();
}



hipError_t cs_layernorm_launch(const float* x, int M, int C, const float* g, const float* b, float eps, float* of32,
                               h16_t* obf, int bf, hipStream_t st) {
  if (C % 4 || C > 2048) return hipErrorInvalidValue;
  if (C <= 1024) hipLaunchKernelGGL(layernorm_kernel<4>, dim3((M + 3) / 4), dim3(256), 0, st, x, M, C, g, b, eps, of32, obf, bf);
  else hipLaunchKernelGGL(layernorm_kernel<8>, dim3((M + 3) / 4), dim3(256), 0, st, x, M, C, g, b, eps, of32, obf, bf);
  return hipGetLastError();
}

hipError_t cs_final_ln_split_launch(const float* x, int I, int img0, int Np, int C, int N, const float* g, const float* b, float eps,
                                    const float* pe, float* q_f32, h16_t* q_bf, h16_t* mem_bf, int bf, hipStream_t st) {
  const long long rows = (long long)I * Np;
  if (C % 4 || C > 2048) return hipErrorInvalidValue;
  if (C <= 1024)
    hipLaunchKernelGGL(final_ln_split_kernel<4>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, x, I, img0, Np, C, N, g, b, eps, pe, q_f32, q_bf, mem_bf, bf);
  else
    hipLaunchKernelGGL(final_ln_split_kernel<8>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, x, I, img0, Np, C, N, g, b, eps, pe, q_f32, q_bf, mem_bf, bf);
  return hipGetLastError();
}

hipError_t cs_cls_rows_launch(float* x, int I, int T, int C, const float* cls, const float* pos, h16_t* xb, float* stats, int sp,
                              int bf, hipStream_t st) {
  hipLaunchKernelGGL(cls_rows_kernel, dim3(I), dim3(64), 0, st, x, I, T, C, cls, pos, xb, stats, sp, bf);
  return hipGetLastError();
}

hipError_t cs_ln_fold_consts_launch(const h16_t* wp, int ldp, const float* w, const float* beta, const float* bias, int N, int K,
                                    float* s_out, float* c_out, int bf, hipStream_t st) {
  hipLaunchKernelGGL(ln_fold_consts_kernel, dim3((N + 3) / 4), dim3(256), 0, st, wp, ldp, w, beta, bias, N, K, s_out, c_out, bf);
  return hipGetLastError();
}

hipError_t cs_pos_bicubic_launch(const float* pos, int G, int C, int gh, int gw, float grow, float* out, hipStream_t st) {
  const long long total = (long long)(1 + gh * gw) * C;
  hipLaunchKernelGGL(pos_bicubic_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, pos, G, C, gh, gw, grow, out);
  return hipGetLastError();
}

hipError_t cs_pe_bilinear_launch(const float* pe, int ph, int pw, int C, int gh, int gw, float* out, hipStream_t st) {
  const long long total = (long long)gh * gw * C;
  hipLaunchKernelGGL(pe_bilinear_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, pe, ph, pw, C, gh, gw, out);
  return hipGetLastError();
}

// mode: 0 bilinear, 1 bicubic (both align_corners=True)
hipError_t cs_pe_interp_launch(const float* pe, int ph, int pw, int C, int gh, int gw, int mode, float* out, hipStream_t st) {
  if (mode == 0) return cs_pe_bilinear_launch(pe, ph, pw, C, gh, gw, out, st);
  if (mode != 1) return hipErrorInvalidValue;
  const long long total = (long long)gh * gw * C;
  hipLaunchKernelGGL(pe_bicubic_ac_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, pe, ph, pw, C, gh, gw, out);
  return hipGetLastError();
}

hipError_t cs_pack_f16_launch(const float* w, int rows, int K, h16_t* out, int ldo, const float* row_scale, const float* col_scale,
                               int bf, hipStream_t st) {
  const long long total = (long long)rows * ldo;
  hipLaunchKernelGGL(pack_f16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, w, rows, K, out, ldo, row_scale,
                     col_scale, bf);
  return hipGetLastError();
}

hipError_t cs_vec_mul_launch(const float* a, const float* b, float* out, int n, hipStream_t st) {
  hipLaunchKernelGGL(vec_mul_kernel, dim3((n + 255) / 256), dim3(256), 0, st, a, b, out, n);
  return hipGetLastError();
}

hipError_t cs_attn_weights_launch(const CsAttnParams* p, int dh, int batch, int head, float* out, hipStream_t st) {
  if (p->Lq <= 0 || p->Lq > 65535 || batch <= 0 || batch > 65535) return hipErrorInvalidValue;  // grid.y / grid.z limits
  dim3 grid((p->Lk + 255) / 256, p->Lq, batch);
  switch (dh) {
    case 16: hipLaunchKernelGGL(attn_weights_kernel<16>, grid, dim3(256), 0, st, *p, head, out); break;
    case 48: hipLaunchKernelGGL(attn_weights_kernel<48>, grid, dim3(256), 0, st, *p, head, out); break;
    case 64: hipLaunchKernelGGL(attn_weights_kernel<64>, grid, dim3(256), 0, st, *p, head, out); break;
    case 96: hipLaunchKernelGGL(attn_weights_kernel<96>, grid, dim3(256), 0, st, *p, head, out); break;
    case 128: hipLaunchKernelGGL(attn_weights_kernel<128>, grid, dim3(256), 0, st, *p, head, out); break;
    case 192: hipLaunchKernelGGL(attn_weights_kernel<192>, grid, dim3(256), 0, st, *p, head, out); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t cs_spin_launch(unsigned long long ticks, int blocks, int lds_bytes, hipStream_t st) {
  static std::atomic<bool> attr_done[16];  // (zero-initialised; hipFuncSetAttribute is idempotent, a racing second caller only repeats it)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return hipErrorInvalidDevice;
  if (lds_bytes > 48 * 1024 && !attr_done[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(spin_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    if (e != hipSuccess) return e;
    attr_done[dev] = true;
  }
  hipLaunchKernelGGL(spin_kernel, dim3(blocks), dim3(64), lds_bytes, st, ticks);
  return hipGetLastError();
}


}  // extern "C"
