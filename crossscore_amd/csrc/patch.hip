// Patch embedding in one launch (HF modeling_dinov2.py:141-149, Dinov2PatchEmbeddings: Conv2d(3, C, 14, stride 14)
// + flatten + position embedding): the image strip of one row of patches goes HBM -> registers -> fp16 A tile in LDS (im2col never
// touches memory), is multiplied by the fragment-ordered patch weights straight out of L2, and lands as fp32 token rows with bias,
// position embedding and the mean term added.  Replaces im2col_rows_kernel + the CS_EPI_PATCH_F32 GEMM (124.7 + 118 us per 48 images
// of 518 x 518, with an fp16 [M][640] matrix written and read in between: 168 MB per 48 images).
//
// Mean-centred form, as before (elementwise.hip im2col_rows_kernel / gemm.hip patch_dc): every patch's per-channel mean is removed before
// the 16-bit rounding (a smooth patch is mostly its mean) and comes back in fp32 as mean_ch * sum_taps W[n][ch].  The mean is summed in the
// same fixed order (14 pixels of a row, then the 14 rows), so the A operand has the bits the two-kernel path produced.
//
// One workgroup (4 waves) = one run of np <= 48 consecutive patches of one patch row (the whole row of 37 at 518 px; wider rows are cut
// into equal runs).  Phases:
//   A  thread t owns the 14-pixel row segments t, t + 256, ..: (channel, dy, patch); 7 x 8-byte loads each, all issued before the
//      first use; row sums -> LDS -> per-(patch, channel) mean -> centred fp16 pairs into the A tile [np][648] (1296-byte rows: an odd
//      number of 16-byte slots, so the fragment reads below are conflict free)
//   B  wave w computes columns 96 w .. 96 w + 95 of a 384-column pass: 19 k-steps of 32; A fragments by ds_read_b128, W fragments by
//      one contiguous 1-KiB load per (k-step, 16-column tile) from a fragment-ordered copy of the weights (cs_patch_pack_launch), read one
//      k-step ahead; v_mfma_f32_16x16x32 with W as the first operand: a lane holds 4 consecutive columns of one patch row
//   C  per 16-row tile the wave turns its accumulators into row segments through a private LDS patch and stores 384-byte pieces of
//      token rows (+ bias + position embedding + mean term)
// C = 384 n (ViT-S: one pass, ViT-B: two); rows past np of the last 16-row tile read whatever follows the A tile and are never stored.
#include "cs_common.h"
#include <atomic>
#include <type_traits>

namespace {

template <int V> using IC = std::integral_constant<int, V>;
typedef float f32x4u_t __attribute__((ext_vector_type(4), aligned(8)));  // a 16-byte load from an 8-byte-aligned address
constexpr int PF_P = 14;
constexpr int PF_KK = 3 * PF_P * PF_P;   // 588
constexpr int PF_KS = 19;                // k-steps of 32 (608 >= 588; columns 588..607 are zero on both sides)
constexpr int PF_PITCH = 1296;           // A tile row pitch in bytes (648 halves)
constexpr int PF_MAXNP = 48;             // patches per workgroup
constexpr int PF_NCOL = 384;             // columns per pass (4 waves x 96)
constexpr int PF_NT = 6;                 // 16-column tiles per wave
constexpr int PF_PPITCH = 400;           // bytes per row of a wave's 16 x 96 fp32 epilogue patch (100 dwords: conflict-free both ways)
constexpr int PF_PATCHB = 16 * PF_PPITCH;
constexpr int PF_MAXSEG = 8;             // row segments per thread (48 * 42 / 256 = 7.9)

__host__ __device__ constexpr int pf_patch0(int npmax) { return (npmax <= 37 ? 37 : PF_MAXNP) * PF_PITCH; }
__host__ __device__ constexpr int pf_lds(int npmax) { return pf_patch0(npmax) + 4 * PF_PATCHB + PF_MAXNP * 16; }

// fragment-ordered weights: piece (pass * 4 + wave, k-step, tile j, lane) = 8 consecutive k of row n = 384 pass + 96 wave + 16 j + (lane & 15)
// starting at k = 32 step + 8 (lane >> 4); fp32 -> 16 bit with the rounding of cs_pack_f16_launch
__global__ void patch_pack_kernel(const float* __restrict__ w, int C, h16_t* __restrict__ out, int bf) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int total = (C / 96) * PF_KS * PF_NT * 64;
  if (i >= total) return;
  const int lane = i & 63;
  int r = i >> 6;
  const int j = r % PF_NT; r /= PF_NT;
  const int s = r % PF_KS;
  const int pw = r / PF_KS;
  const int n = pw * 96 + 16 * j + (lane & 15);
  const int k0 = 32 * s + 8 * (lane >> 4);
  uint32_t q[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int k = k0 + 2 * e;
    const float a = k < PF_KK ? w[(size_t)n * PF_KK + k] : 0.f;
    const float b = k + 1 < PF_KK ? w[(size_t)n * PF_KK + k + 1] : 0.f;
    q[e] = pack_o16x2(a, b, bf);
  }
  *reinterpret_cast<uint4*>(out + (size_t)i * 8) = make_uint4(q[0], q[1], q[2], q[3]);
}

struct PatchParams {
  const float* xq; const float* xr;  // query images (B,3,H,W), reference images (B,N,3,H,W)
  int N, img0;                       // reference views per item, first image of this chunk (image g = item g / (1 + N), view g % (1 + N))
  int H, W, gh, gw, C;
  int nsx;                           // runs per patch row
  const h16_t* wfrag; const float* bias; const float* pos; const float* wsum;  // pos ((1 + gh gw), C); wsum (3, C)
  float* x;                          // (images of the chunk, 1 + gh gw, C) fp32 token rows; CLS rows are not written here
  // U8 form (SURVEY.md 8f-4 as worded: uint8 in, tokens out): the strip comes from the decoded images through the input stage's own arithmetic
  const CsU8Desc* u8;                // [nq query images][nq * N reference images]
  int nq;
  int lut_off;                       // byte offset of the 256-entry x / 255 table in LDS (behind the largest horizontal-pass buffer)
  float mean[3], stdv[3];
};

template <bool BF, bool U8>
__global__ __launch_bounds__(256, 2) void cs_patch_fused_kernel(PatchParams p) {
  extern __shared__ __attribute__((aligned(16))) char pf_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef CS_PF_EMPTY  // timing only: what the launch itself costs (same registers, same LDS, no work)
  if (p.H > 0) return;
#endif
  // block -> (patch row, run, image), image fastest (r5): the workgroups that add the same slice of the position table (the patch row's
  // 57 KB at 518 px) run next to each other, so the slice is fetched once per XCD instead of once per image (the table is 2.1 MB, a chunk
  // of 48 images streams 260 MB through the L2s in between: PMC traffic 349 -> 286 MB per 48 images, same time)
  int b = blockIdx.x;
  const int n_img = gridDim.x / (p.gh * p.nsx);
  const int img = b % n_img; b /= n_img;
  const int sx = b % p.nsx;
  const int pi = b / p.nsx;
  const int base = p.gw / p.nsx, rem = p.gw - base * p.nsx;
  const int np = base + (sx < rem ? 1 : 0);              // patches of this run
  const int p0 = sx * base + (sx < rem ? sx : rem);      // first patch (column index in the patch row)
  const int npmax = base + (rem ? 1 : 0);
  char* At = pf_smem;
  float* part = reinterpret_cast<float*>(pf_smem);       // row sums [42][np]: lives in the A tile's space until the means are known
  char* patch = pf_smem + pf_patch0(npmax) + wv * PF_PATCHB;
  float* mean_s = reinterpret_cast<float*>(pf_smem + pf_patch0(npmax) + 4 * PF_PATCHB);  // [48][4]

  const int g_img = p.img0 + img;
  const int bb = g_img / (1 + p.N), vv = g_img - bb * (1 + p.N);
  [[maybe_unused]] const float* x = nullptr;
  if constexpr (!U8) x = vv == 0 ? p.xq + (size_t)bb * 3 * p.H * p.W : p.xr + ((size_t)bb * p.N + (vv - 1)) * 3 * p.H * p.W;

  // ---------------- phase A: strip -> centred 16-bit A tile ----------------
  const int nseg = np * 3 * PF_P;
  float v[PF_MAXSEG][PF_P];
  int srow[PF_MAXSEG], spj[PF_MAXSEG];
#pragma unroll
  for (int i = 0; i < PF_MAXSEG; ++i) {
    const int sg = tid + 256 * i;
    const int row = sg / np;  // ch * 14 + dy
    srow[i] = row; spj[i] = sg - row * np;
  }
  if constexpr (U8) {
    // The strip's pixels straight from the decoded uint8 image, by the operations of the two-launch input stage in their order (preprocess.hip:
    // x / 255 -> width pass of the antialiased triangle filter -> height pass -> (v - mean) / std; an image that needs no resize has the identity
    // table).  Per channel: the source rows the strip's 14 pixel rows reach (first tap of the first .. last tap of the last) go through the width
    // pass into LDS, [source row][strip column] fp32, then every thread forms its 14-pixel segments by the height pass.  The buffer lives where the
    // A tile and the epilogue patches will be: nothing of them exists before all segments are in registers.
    const CsU8Desc& d = p.u8[vv == 0 ? bb : p.nq + bb * p.N + (vv - 1)];
    float* tmp = reinterpret_cast<float*>(pf_smem);
    float* lut = reinterpret_cast<float*>(pf_smem + p.lut_off);
    lut[tid] = (float)tid / 255.0f;  // np.float32(img) / 255.0 (utils/io/images.py:14-29), one IEEE division per value instead of per tap
    const int ry0 = pi * PF_P + d.crop_y, X0 = p0 * PF_P + d.crop_x, wpx = np * PF_P;
    const int R0 = d.t.ymin[ry0];
    int R1 = R0;
    for (int dy = 0; dy < PF_P; ++dy) R1 = max(R1, d.t.ymin[ry0 + dy] + d.t.ysize[ry0 + dy]);
#pragma unroll 1
    for (int ch = 0; ch < 3; ++ch) {
      __syncthreads();  // the table is there / the previous channel's height pass is done
      // a lane owns strip columns lane, lane + 64, ..: the column's first tap, tap count and weights are fetched once and serve every source row;
      // the rows go round the four waves.  Same products, same order as u8_resize_w_kernel: a = p0 w0, then fma tap by tap.
      for (int xc = lane; xc < wpx; xc += 64) {
        const int rx = X0 + xc;
        int n = 0;
        const uint8_t* col = nullptr;
        const float* w = nullptr;
        float wr[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (d.data) {
          n = d.t.xsize[rx];
          col = d.data + (size_t)d.t.xmin[rx] * 3 + ch;
          w = d.t.wx + (size_t)rx * d.t.taps_x;
#pragma unroll
          for (int j = 0; j < 8; ++j) if (j < n) wr[j] = w[j];
        }
        for (int r = wv; r < R1 - R0; r += 4) {
          float a = 0.f;
          if (n > 0) {
            const uint8_t* src = col + (size_t)(R0 + r) * d.row_bytes;
            a = lut[src[0]] * wr[0];
            if (n <= 8) {
#pragma unroll
              for (int j = 1; j < 8; ++j) if (j < n) a = __builtin_fmaf(lut[src[3 * j]], wr[j], a);
            } else {
              for (int j = 1; j < n; ++j) a = __builtin_fmaf(lut[src[3 * j]], w[j], a);
            }
          }
          tmp[r * wpx + xc] = a;
        }
      }
      __syncthreads();
      const float mu = p.mean[ch], sd = p.stdv[ch];
#pragma unroll
      for (int i = 0; i < PF_MAXSEG; ++i) {
        const int dy = srow[i] - ch * PF_P;
        if (tid + 256 * i < nseg && dy >= 0 && dy < PF_P) {
          const int ry = ry0 + dy, n = d.t.ysize[ry];
          const float* w = d.t.wy + (size_t)ry * d.t.taps_y;
          const float* col = tmp + (d.t.ymin[ry] - R0) * wpx + spj[i] * PF_P;
          float acc[PF_P];
#pragma unroll
          for (int e = 0; e < PF_P; ++e) acc[e] = 0.f;
          for (int j = 0; j < n; ++j) {
            const float wj = w[j];
#pragma unroll
            for (int e = 0; e < PF_P; ++e) acc[e] = j == 0 ? col[j * wpx + e] * wj : __builtin_fmaf(col[j * wpx + e], wj, acc[e]);
          }
#pragma unroll
          for (int e = 0; e < PF_P; ++e) v[i][e] = (acc[e] - mu) / sd;
        }
      }
    }
    __syncthreads();  // the row sums below overwrite the buffer
  }
#pragma unroll
  for (int i = 0; i < PF_MAXSEG; ++i) {
    const int sg = tid + 256 * i;
    const int row = srow[i];
    if (!U8 && sg < nseg) {
      const int ch = row / PF_P, dy = row - ch * PF_P;
      const float* src = x + ((size_t)ch * p.H + (pi * PF_P + dy)) * p.W + (p0 + spj[i]) * PF_P;
      // 56 bytes at an 8-byte-aligned address: 3 x 16 + 8 (global loads need dword alignment only; 7 x 8 bytes cost the address
      // unit of the CU 75 % more instructions, and at a 56-byte lane stride that unit is what phase A waits for)
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        const f32x4u_t t4 = *reinterpret_cast<const f32x4u_t*>(src + 4 * e);
        v[i][4 * e] = t4[0]; v[i][4 * e + 1] = t4[1]; v[i][4 * e + 2] = t4[2]; v[i][4 * e + 3] = t4[3];
      }
      const f32x2_t t2 = *reinterpret_cast<const f32x2_t*>(src + 12);
      v[i][12] = t2[0]; v[i][13] = t2[1];
#ifdef CS_PF_NOLOAD  // timing only (tools/patch_ab.py): the strip loads become one load per segment
#pragma unroll
      for (int e = 1; e < PF_P; ++e) v[i][e] = v[i][0] + (float)e;
#endif
    }
  }
#pragma unroll
  for (int i = 0; i < PF_MAXSEG; ++i) {
    if (tid + 256 * i < nseg) {
      float sacc = 0.f;  // fixed order: the 14 pixels of the row, left to right
#pragma unroll
      for (int e = 0; e < PF_P; ++e) sacc += v[i][e];
      part[tid + 256 * i] = sacc;
    }
  }
  __syncthreads();
  if (tid < np * 3) {
    const int ch = tid / np, pj = tid - ch * np;
    float sacc = 0.f;    // .. then the 14 rows, top to bottom
#pragma unroll
    for (int dy = 0; dy < PF_P; ++dy) sacc += part[(ch * PF_P + dy) * np + pj];
    mean_s[pj * 4 + ch] = sacc / (float)(PF_P * PF_P);
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < PF_MAXSEG; ++i) {
    if (tid + 256 * i < nseg) {
      const int row = srow[i], pj = spj[i];
      const int ch = row / PF_P;
      const float mu = mean_s[pj * 4 + ch];
      uint32_t* dst = reinterpret_cast<uint32_t*>(At + pj * PF_PITCH + row * (PF_P * 2));  // column ch * 196 + dy * 14 = 14 row
#pragma unroll
      for (int e = 0; e < 7; ++e) dst[e] = pack_o16x2<BF>(v[i][2 * e] - mu, v[i][2 * e + 1] - mu);
    }
  }
  for (int i = tid; i < np * 10; i += 256) {  // columns 588 .. 607 of the real rows
    const int r = i / 10;
    *reinterpret_cast<uint32_t*>(At + r * PF_PITCH + PF_KK * 2 + (i - r * 10) * 4) = 0u;
  }
  __syncthreads();

  // ---------------- phases B, C per 384-column pass ----------------
  const int fr = lane & 15, cq = lane >> 4;
  const int mtiles = (np + 15) >> 4;
  const char* a_rd = At + fr * PF_PITCH + cq * 16;
  const int c4 = lane % 24, rs = lane / 24;  // epilogue: lane -> 16-byte column group, row parity (lanes 48..63 idle)
  const int T = 1 + p.gh * p.gw;
  for (int pass = 0; pass < p.C / PF_NCOL; ++pass) {
    const uint4* wsrc = reinterpret_cast<const uint4*>(p.wfrag) + (size_t)(pass * 4 + wv) * (PF_KS * PF_NT * 64) + lane;
    f32x4_t acc[3][PF_NT];
#pragma unroll
    for (int mt = 0; mt < 3; ++mt)
#pragma unroll
      for (int j = 0; j < PF_NT; ++j) acc[mt][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    uint4 wf[4][PF_NT];  // W fragments three k-steps ahead (an L2 round trip is several k-steps of MFMA time), four register sets
    auto wload = [&](auto B_, int s) {
      constexpr int B = decltype(B_)::value;
#pragma unroll
      for (int j = 0; j < PF_NT; ++j) wf[B][j] = wsrc[(s * PF_NT + j) * 64];
    };
    auto kstep = [&](auto B_, int s) {
      constexpr int B = decltype(B_)::value;
      h16x8_t af[3];
#pragma unroll
      for (int mt = 0; mt < 3; ++mt)
        if (mt < mtiles) af[mt] = *reinterpret_cast<const h16x8_t*>(a_rd + mt * 16 * PF_PITCH + s * 64);
#pragma unroll
      for (int mt = 0; mt < 3; ++mt)
        if (mt < mtiles) {
#pragma unroll
          for (int j = 0; j < PF_NT; ++j) acc[mt][j] = mfma_16x16x32<BF>(__builtin_bit_cast(h16x8_t, wf[B][j]), af[mt], acc[mt][j]);
        }
    };
    static_assert(PF_KS == 19, "k loop below: 4 x 4 steps + 3");
#ifndef CS_PF_NOMMA  // (timing-only ablation: no W stream, no MFMAs)
    wload(IC<0>{}, 0); wload(IC<1>{}, 1); wload(IC<2>{}, 2);
#pragma unroll 1
    for (int s = 0; s < 16; s += 4) {
      wload(IC<3>{}, s + 3); kstep(IC<0>{}, s);
      wload(IC<0>{}, s + 4); kstep(IC<1>{}, s + 1);
      wload(IC<1>{}, s + 5); kstep(IC<2>{}, s + 2);
      wload(IC<2>{}, s + 6); kstep(IC<3>{}, s + 3);
    }
    kstep(IC<0>{}, 16); kstep(IC<1>{}, 17); kstep(IC<2>{}, 18);
#endif
    // ---- phase C ----
    int tok0 = 1 + pi * p.gw + p0;            // first token row of the run
    asm volatile("" : "+s"(tok0));            // (opaque: hoisted out of the pass loop, the 24 row addresses below were spilled around the k loop)
    const int n = pass * PF_NCOL + wv * 96 + c4 * 4;
    f32x4_t b4 = {0.f, 0.f, 0.f, 0.f}, w0 = b4, w1 = b4, w2 = b4;
    if (lane < 48) {
      b4 = *reinterpret_cast<const f32x4_t*>(p.bias + n);
      w0 = *reinterpret_cast<const f32x4_t*>(p.wsum + n);
      w1 = *reinterpret_cast<const f32x4_t*>(p.wsum + (size_t)p.C + n);
      w2 = *reinterpret_cast<const f32x4_t*>(p.wsum + 2 * (size_t)p.C + n);
    }
#pragma unroll
    for (int mt = 0; mt < 3; ++mt) {
      if (mt < mtiles) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (wave-private patch: the previous tile's reads are done before it is overwritten)
#pragma unroll
        for (int j = 0; j < PF_NT; ++j) *reinterpret_cast<f32x4_t*>(patch + fr * PF_PPITCH + (16 * j + 4 * cq) * 4) = acc[mt][j];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // LDS operations of one wave execute in order; this orders the compiler
        if (lane < 48) {
#pragma unroll
          for (int it = 0; it < 8; ++it) {
            const int rl = 2 * it + rs, m = mt * 16 + rl;
            if (m < np) {
              const int tok = tok0 + m;
              f32x4_t r = *reinterpret_cast<const f32x4_t*>(patch + rl * PF_PPITCH + c4 * 16) + b4;
#ifndef CS_PF_NOPOS  // (timing-only ablation)
              r += *reinterpret_cast<const f32x4_t*>(p.pos + (size_t)tok * p.C + n);
#endif
              const f32x4_t mu = *reinterpret_cast<const f32x4_t*>(mean_s + m * 4);
              f32x4_t dc = mu[0] * w0;
              dc += mu[1] * w1;
              dc += mu[2] * w2;
              r += dc;
#ifdef CS_PF_NOSTORE  // (timing-only ablation: one lane in 2^20 stores)
              if (r[0] == 123.456f)
#endif
              *reinterpret_cast<f32x4_t*>(p.x + ((size_t)img * T + tok) * p.C + n) = r;
            }
          }
        }
      }
    }
  }
}

int g_pf_enabled = 1;

}  // namespace

extern "C" void cs_debug_patch_fused_enable(int on) { g_pf_enabled = on; }

// halves in the fragment-ordered weight copy
size_t cs_patch_pack_elems(int C) { return (size_t)(C / 96) * PF_KS * PF_NT * 64 * 8; }

hipError_t cs_patch_pack_launch(const float* w, int C, h16_t* out, int bf, hipStream_t st) {
  const int total = (C / 96) * PF_KS * PF_NT * 64;
  hipLaunchKernelGGL(patch_pack_kernel, dim3((total + 255) / 256), dim3(256), 0, st, w, C, out, bf);
  return hipGetLastError();
}

// shapes the one-launch form takes (everything else: im2col + GEMM): 14-pixel patches, 384-column passes, even row pitch (8-byte loads)
int cs_patch_fused_supported(int H, int W, int P, int C) {
  return g_pf_enabled && P == PF_P && C % PF_NCOL == 0 && C > 0 && W % 2 == 0 && H >= P && W >= P;
}

namespace {

constexpr int PF_U8_TMP = 64 * 1024;               // largest horizontal-pass buffer of the U8 form (one channel: source rows x strip columns, fp32)
constexpr int PF_U8_LDS = PF_U8_TMP + 1024;        // + the x / 255 table

template <bool BF, bool U8>
hipError_t pf_launch(const PatchParams& p, int lds, long long blocks, hipStream_t st) {
  if (blocks <= 0 || blocks >= (1ll << 31)) return hipErrorInvalidValue;
  static std::atomic<bool> attr_done[16];  // (zero-initialised; hipFuncSetAttribute is idempotent, a racing second caller only repeats it)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return hipErrorInvalidDevice;
  if (!attr_done[dev]) {
    const int cap = U8 && PF_U8_LDS > pf_lds(PF_MAXNP) ? PF_U8_LDS : pf_lds(PF_MAXNP);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(cs_patch_fused_kernel<BF, U8>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    if (e != hipSuccess) return e;
    attr_done[dev] = true;
  }
  hipLaunchKernelGGL((cs_patch_fused_kernel<BF, U8>), dim3((unsigned)blocks), dim3(256), lds, st, p);
  return hipGetLastError();
}

}  // namespace

hipError_t cs_patch_fused_launch(const float* xq, const float* xr, int N, int img0, int I, int H, int W, int C, const h16_t* wfrag,
                                 const float* bias, const float* pos, const float* wsum, float* x, int bf, hipStream_t st) {
  PatchParams p{};
  p.xq = xq; p.xr = xr; p.N = N; p.img0 = img0; p.H = H; p.W = W; p.gh = H / PF_P; p.gw = W / PF_P; p.C = C;
  p.nsx = (p.gw + PF_MAXNP - 1) / PF_MAXNP;
  p.wfrag = wfrag; p.bias = bias; p.pos = pos; p.wsum = wsum; p.x = x;
  const int npmax = (p.gw + p.nsx - 1) / p.nsx;
  const long long blocks = (long long)I * p.gh * p.nsx;
  return bf ? pf_launch<true, false>(p, pf_lds(npmax), blocks, st) : pf_launch<false, false>(p, pf_lds(npmax), blocks, st);
}

// runs of a patch row the U8 form needs so that `row_span` source rows of a run fit the horizontal-pass buffer; 0: not even one patch per run does
int cs_patch_u8_runs(int W, int row_span) {
  const int gw = W / PF_P;
  for (int nsx = (gw + PF_MAXNP - 1) / PF_MAXNP; nsx <= gw; ++nsx) {
    const int npmax = (gw + nsx - 1) / nsx;
    if ((long long)row_span * npmax * PF_P * 4 <= PF_U8_TMP) return nsx;
  }
  return 0;
}

// The same launch from decoded uint8 images: `descs` (device) = nq query descriptors, then nq * N reference descriptors; row_span = the largest number
// of source rows a patch row of any of them reaches (cs_preprocess_tables).  hipErrorInvalidValue when the geometry does not fit (cs_patch_u8_runs).
hipError_t cs_patch_fused_u8_launch(const CsU8Desc* descs, int nq, int N, int img0, int I, int H, int W, int C, int row_span, const float* mean3,
                                    const float* std3, const h16_t* wfrag, const float* bias, const float* pos, const float* wsum, float* x, int bf,
                                    hipStream_t st) {
  PatchParams p{};
  p.u8 = descs; p.nq = nq; p.N = N; p.img0 = img0; p.H = H; p.W = W; p.gh = H / PF_P; p.gw = W / PF_P; p.C = C;
  p.nsx = cs_patch_u8_runs(W, row_span);
  if (p.nsx <= 0 || row_span <= 0) return hipErrorInvalidValue;
  for (int c = 0; c < 3; ++c) { p.mean[c] = mean3[c]; p.stdv[c] = std3[c]; }
  p.wfrag = wfrag; p.bias = bias; p.pos = pos; p.wsum = wsum; p.x = x;
  const int npmax = (p.gw + p.nsx - 1) / p.nsx;
  p.lut_off = ((row_span * npmax * PF_P * 4 + 255) / 256) * 256;
  const int lds = p.lut_off + 1024 > pf_lds(npmax) ? p.lut_off + 1024 : pf_lds(npmax);
  const long long blocks = (long long)I * p.gh * p.nsx;
  return bf ? pf_launch<true, true>(p, lds, blocks, st) : pf_launch<false, true>(p, lds, blocks, st);
}
