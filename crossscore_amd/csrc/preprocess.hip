// Input stage and output stage (8f-2, end of file) of the predict path on the GPU.  Input: uint8 HWC image -> the fp32, ImageNet-normalised CHW tensor that
// cs_forward consumes.  Replaces the reference's CPU transforms
//   image_read: np.float32(img) / 255.0                                           utils/io/images.py:14-29
//   T.Resize(short side, BILINEAR, antialias=True) on the float CHW tensor        task/predict.py:87-93, nvs_dataset.py:218-225
//   deterministic (top-left) crop / integer-patch crop                            dataloading/transformation/crop.py:8-25, nvs_dataset.py:227-241
//   T.Normalize(mean, std)                                                        task/predict.py:68-74
// in that order, with the same fp32 operations: (u8 / 255) -> separable antialiased triangle filter, width pass then height pass
// (ATen UpSampleKernel.cpp, _compute_indices_min_size_weights_aa + separable_upsample_generic_Nd_kernel_impl) -> (v - mean) / std.
// Divisions are IEEE (hipcc's default correctly-rounded fp32 divide), so without a resize the result is bit-identical to the
// reference's tensor.  HBM-bound byte work: one thread per output pixel, coalesced along x; H2D traffic drops 4x (uint8 in).
#include "cs_common.h"
#include <math.h>
#include <mutex>
#include <vector>

namespace {

// ---- per-axis filter tables (host): for output index i the taps start at xmin[i], xsize[i] of them, weights w[i][0..xsize) ----
struct AxisTable {
  int in = 0, out = 0, taps = 0;
  std::vector<int> xmin, xsize;
  std::vector<float> w;  // [out][taps], zero padded
};

float tri(float x) {  // HelperInterpLinear::aa_filter
  x = fabsf(x);
  return x < 1.0f ? 1.0f - x : 0.0f;
}

void build_axis(int in, int out, AxisTable& t) {
  t.in = in;
  t.out = out;
  const float scale = (float)in / (float)out;  // area_pixel_compute_scale, align_corners = False, no explicit scale factor
  const float support = scale >= 1.0f ? scale : 1.0f;  // interp_size / 2 = 1 for the linear filter
  t.taps = (int)ceilf(support) * 2 + 1;
  t.xmin.assign(out, 0);
  t.xsize.assign(out, 0);
  t.w.assign((size_t)out * t.taps, 0.f);
  const float invscale = scale >= 1.0f ? 1.0f / scale : 1.0f;
  for (int i = 0; i < out; ++i) {
    const float center = (float)((double)scale * ((double)i + 0.5));
    long long lo = (long long)((double)center - (double)support + 0.5);
    if (lo < 0) lo = 0;
    long long hi = (long long)((double)center + (double)support + 0.5);
    if (hi > in) hi = in;
    const int n = (int)(hi - lo);
    float total = 0.f;
    float* wr = &t.w[(size_t)i * t.taps];
    for (int j = 0; j < n && j < t.taps; ++j) {
      const float wv = tri((float)(((double)j + (double)lo - (double)center + 0.5) * (double)invscale));
      wr[j] = wv;
      total += wv;
    }
    if (total != 0.f)
      for (int j = 0; j < n && j < t.taps; ++j) wr[j] /= total;
    t.xmin[i] = (int)lo;
    t.xsize[i] = n < t.taps ? n : t.taps;
  }
}

// Device copies of the two filter tables of one (device, in_h, in_w, rs_h, rs_w).  An entry is written once, on the stream of the
// call that creates it, and never overwritten: kernels queued on any stream keep reading valid tables, so a change of image size
// needs no synchronisation.  A directory of images has a handful of sizes; past 64 entries the cache is dropped behind a device
// synchronisation.
struct TableEntry {
  int dev = 0, in_h = 0, in_w = 0, rs_h = 0, rs_w = 0;
  int taps_x = 0, taps_y = 0;
  int* d_int = nullptr;    // xmin[rs_w] xsize[rs_w] ymin[rs_h] ysize[rs_h]
  float* d_w = nullptr;    // wx[rs_w][taps_x] wy[rs_h][taps_y]
  std::vector<int> h_ymin, h_yend;  // host copy of the row table (first source row, one past the last) for cs_preprocess_tables' row spans
};
std::vector<TableEntry> g_tabs;
std::recursive_mutex g_tabs_mu;  // forwards of several handles may come from several host threads (pipeline.py)
unsigned g_tabs_gen = 0;  // counts the times the cache was dropped: device pointers handed out under an older count are gone

// the entry of (device, in_h, in_w, rs_h, rs_w), built and uploaded on first use; the caller holds g_tabs_mu
hipError_t table_entry(int in_h, int in_w, int rs_h, int rs_w, const TableEntry** out) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  for (const TableEntry& t : g_tabs)
    if (t.dev == dev && t.in_h == in_h && t.in_w == in_w && t.rs_h == rs_h && t.rs_w == rs_w) { *out = &t; return hipSuccess; }
  if (g_tabs.size() >= 64) {
    if ((e = hipDeviceSynchronize()) != hipSuccess) return e;
    for (TableEntry& t : g_tabs) { (void)hipFree(t.d_int); (void)hipFree(t.d_w); }
    g_tabs.clear();
    ++g_tabs_gen;
  }
  AxisTable tx, ty;
  build_axis(in_w, rs_w, tx);
  build_axis(in_h, rs_h, ty);
  TableEntry t;
  t.dev = dev; t.in_h = in_h; t.in_w = in_w; t.rs_h = rs_h; t.rs_w = rs_w; t.taps_x = tx.taps; t.taps_y = ty.taps;
  t.h_ymin = ty.xmin;
  t.h_yend.resize(rs_h);
  for (int i = 0; i < rs_h; ++i) t.h_yend[i] = ty.xmin[i] + ty.xsize[i];
  std::vector<int> h_int;
  h_int.insert(h_int.end(), tx.xmin.begin(), tx.xmin.end());
  h_int.insert(h_int.end(), tx.xsize.begin(), tx.xsize.end());
  h_int.insert(h_int.end(), ty.xmin.begin(), ty.xmin.end());
  h_int.insert(h_int.end(), ty.xsize.begin(), ty.xsize.end());
  std::vector<float> h_w;
  h_w.insert(h_w.end(), tx.w.begin(), tx.w.end());
  h_w.insert(h_w.end(), ty.w.begin(), ty.w.end());
  if ((e = hipMalloc(&t.d_int, h_int.size() * sizeof(int))) != hipSuccess) return e;
  if ((e = hipMalloc(&t.d_w, h_w.size() * sizeof(float))) != hipSuccess) { (void)hipFree(t.d_int); return e; }
  // fresh buffers nobody reads yet; hipMemcpy returns once the (pageable) host vectors have been consumed
  if ((e = hipMemcpy(t.d_int, h_int.data(), h_int.size() * sizeof(int), hipMemcpyHostToDevice)) != hipSuccess) return e;
  if ((e = hipMemcpy(t.d_w, h_w.data(), h_w.size() * sizeof(float), hipMemcpyHostToDevice)) != hipSuccess) return e;
  g_tabs.push_back(std::move(t));
  *out = &g_tabs.back();
  return hipSuccess;
}

__global__ void u8_norm_kernel(const uint8_t* __restrict__ img, int row_bytes, int crop_y, int crop_x, int oh, int ow,
                               float m0, float m1, float m2, float s0, float s1, float s2, float* __restrict__ out) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  if (x >= ow) return;
  const uint8_t* px = img + (size_t)(y + crop_y) * row_bytes + (size_t)(x + crop_x) * 3;
  const size_t plane = (size_t)oh * ow, o = (size_t)y * ow + x;
  out[o] = ((float)px[0] / 255.0f - m0) / s0;
  out[plane + o] = ((float)px[1] / 255.0f - m1) / s1;
  out[2 * plane + o] = ((float)px[2] / 255.0f - m2) / s2;
}

// width pass: tmp[y][x'][c] = sum_j wx[x'][j] * (u8[y][xmin[x'] + j][c] / 255)
__global__ void u8_resize_w_kernel(const uint8_t* __restrict__ img, int row_bytes, int in_h, int rs_w, int taps,
                                   const int* __restrict__ xmin, const int* __restrict__ xsize, const float* __restrict__ wx,
                                   float* __restrict__ tmp) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  if (x >= rs_w) return;
  const uint8_t* row = img + (size_t)y * row_bytes + (size_t)xmin[x] * 3;
  const float* w = wx + (size_t)x * taps;
  const int n = xsize[x];
  float a0 = 0.f, a1 = 0.f, a2 = 0.f;
  for (int j = 0; j < n; ++j) {
    const float wj = w[j];
    const float v0 = (float)row[3 * j] / 255.0f, v1 = (float)row[3 * j + 1] / 255.0f, v2 = (float)row[3 * j + 2] / 255.0f;
    // (explicit fma: the one-pass form in patch.hip repeats these operations and must round the same way whatever the compiler contracts)
    if (j == 0) { a0 = v0 * wj; a1 = v1 * wj; a2 = v2 * wj; }
    else { a0 = __builtin_fmaf(v0, wj, a0); a1 = __builtin_fmaf(v1, wj, a1); a2 = __builtin_fmaf(v2, wj, a2); }
  }
  float* o = tmp + ((size_t)y * rs_w + x) * 3;
  o[0] = a0; o[1] = a1; o[2] = a2;
}

// height pass + crop + normalise + HWC -> CHW
__global__ void resize_h_norm_kernel(const float* __restrict__ tmp, int rs_w, int taps, const int* __restrict__ ymin,
                                     const int* __restrict__ ysize, const float* __restrict__ wy, int crop_y, int crop_x, int oh,
                                     int ow, float m0, float m1, float m2, float s0, float s1, float s2, float* __restrict__ out) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  if (x >= ow) return;
  const int ry = y + crop_y, rx = x + crop_x;
  const float* w = wy + (size_t)ry * taps;
  const int n = ysize[ry];
  const float* col = tmp + ((size_t)ymin[ry] * rs_w + rx) * 3;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f;
  for (int j = 0; j < n; ++j) {
    const float wj = w[j];
    const float* p = col + (size_t)j * rs_w * 3;
    if (j == 0) { a0 = p[0] * wj; a1 = p[1] * wj; a2 = p[2] * wj; }
    else { a0 = __builtin_fmaf(p[0], wj, a0); a1 = __builtin_fmaf(p[1], wj, a1); a2 = __builtin_fmaf(p[2], wj, a2); }
  }
  const size_t plane = (size_t)oh * ow, o = (size_t)y * ow + x;
  out[o] = (a0 - m0) / s0;
  out[plane + o] = (a1 - m1) / s1;
  out[2 * plane + o] = (a2 - m2) / s2;
}

// ---- output stage: score map -> the integer images the reference's writers store ----
// gray: metric_map_write (utils/io/images.py:49-63): m*65535 for the [0,1] intrinsic range, (m+1)*32767 for [-1,1], truncated
// (numpy astype(int32)), stored as 16-bit PNG samples.  Same fp32 operations -> identical integers.
__global__ void score_gray16_kernel(const float* __restrict__ score, size_t n, int signed_range, uint16_t* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float m = score[i];
  const float v = signed_range ? (m + 1.0f) * 32767.0f : m * 65535.0f;
  const int q = (int)v;  // truncation toward zero, like astype(int32)
  out[i] = (uint16_t)(q < 0 ? 0 : (q > 65535 ? 65535 : q));
}

// rgb: gray2rgb (utils/misc/image.py:37-52) = matplotlib Normalize(vmin, vmax) in fp32, colormap lookup with N = 256 entries
// (index = trunc(x*256), x == 1 -> 255, below 0 -> first, above -> last entry), u8() truncation folded into the 256x3 byte table.
__global__ void score_rgb_kernel(const float* __restrict__ score, size_t n, float vmin, float vmax, const uint8_t* __restrict__ lut,
                                 uint8_t* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float x = score[i] - vmin;
  x = x / (vmax - vmin);
  x = x * 256.0f;
  int idx;
  if (x == 256.0f) idx = 255;
  else if (!(x >= 0.0f)) idx = 0;      // under (and NaN -> the "bad" colour is not reproduced: first entry)
  else if (x >= 256.0f) idx = 255;     // over
  else idx = (int)x;
  out[3 * i] = lut[3 * idx];
  out[3 * i + 1] = lut[3 * idx + 1];
  out[3 * i + 2] = lut[3 * idx + 2];
}

}  // namespace

extern "C" hipError_t cs_score_gray16_launch(const float* score, size_t n, int signed_range, uint16_t* out, hipStream_t stream) {
  hipLaunchKernelGGL(score_gray16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, score, n, signed_range, out);
  return hipGetLastError();
}
extern "C" hipError_t cs_score_rgb_launch(const float* score, size_t n, float vmin, float vmax, const uint8_t* lut, uint8_t* out,
                                          hipStream_t stream) {
  hipLaunchKernelGGL(score_rgb_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, score, n, vmin, vmax, lut, out);
  return hipGetLastError();
}

// Returns hipSuccess or the failing HIP error.  `scratch` holds in_h * rs_w * 3 floats when a resize is requested.
extern "C" hipError_t cs_preprocess_launch(const uint8_t* img, int in_h, int in_w, int row_bytes, int rs_h, int rs_w, int crop_y,
                                           int crop_x, int oh, int ow, const float* mean, const float* stdv, float* out,
                                           float* scratch, hipStream_t stream) {
  const dim3 blk(256);
  if (rs_h == in_h && rs_w == in_w) {
    hipLaunchKernelGGL(u8_norm_kernel, dim3((ow + 255) / 256, oh), blk, 0, stream, img, row_bytes, crop_y, crop_x, oh, ow, mean[0],
                       mean[1], mean[2], stdv[0], stdv[1], stdv[2], out);
    return hipGetLastError();
  }
  std::lock_guard<std::recursive_mutex> lock(g_tabs_mu);
  const TableEntry* hit = nullptr;
  hipError_t e = table_entry(in_h, in_w, rs_h, rs_w, &hit);
  if (e != hipSuccess) return e;
  const TableEntry& T = *hit;
  const int* xmin = T.d_int;
  const int* xsize = T.d_int + rs_w;
  const int* ymin = T.d_int + 2 * rs_w;
  const int* ysize = ymin + rs_h;
  const float* wx = T.d_w;
  const float* wy = T.d_w + (size_t)rs_w * T.taps_x;
  hipLaunchKernelGGL(u8_resize_w_kernel, dim3((rs_w + 255) / 256, in_h), blk, 0, stream, img, row_bytes, in_h, rs_w, T.taps_x, xmin,
                     xsize, wx, scratch);
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(resize_h_norm_kernel, dim3((ow + 255) / 256, oh), blk, 0, stream, scratch, rs_w, T.taps_y, ymin, ysize, wy, crop_y,
                     crop_x, oh, ow, mean[0], mean[1], mean[2], stdv[0], stdv[1], stdv[2], out);
  return hipGetLastError();
}

// The filter tables of one resize geometry for the one-pass input stage (patch.hip, cs_patch_fused_u8_launch): device pointers that stay valid until the
// cache is dropped (65th geometry, behind a device synchronisation), and the largest number of source rows any row of `gh` P-pixel patches starting
// at resized row crop_y needs (first tap of its first pixel row .. last tap of its last): what the kernel has to hold in LDS per channel.
// *generation = the cache's drop count when the pointers were handed out: a caller that gathers tables of several geometries compares it across its
// calls and starts over when it moved (the drop waits for the device, so work already queued is never affected).
extern "C" hipError_t cs_preprocess_tables(int in_h, int in_w, int rs_h, int rs_w, int crop_y, int gh, int P, CsU8Tables* out, int* row_span,
                                           unsigned* generation) {
  std::lock_guard<std::recursive_mutex> lock(g_tabs_mu);
  const TableEntry* hit = nullptr;
  hipError_t e = table_entry(in_h, in_w, rs_h, rs_w, &hit);
  if (e != hipSuccess) return e;
  const TableEntry& T = *hit;
  out->xmin = T.d_int; out->xsize = T.d_int + rs_w; out->ymin = T.d_int + 2 * rs_w; out->ysize = out->ymin + rs_h;
  out->wx = T.d_w; out->wy = T.d_w + (size_t)rs_w * T.taps_x;
  out->taps_x = T.taps_x; out->taps_y = T.taps_y;
  int span = 0;
  for (int pi = 0; pi < gh; ++pi) {
    const int r0 = crop_y + pi * P, r1 = r0 + P - 1;
    if (r0 < 0 || r1 >= rs_h) return hipErrorInvalidValue;
    int hi = 0;
    for (int r = r0; r <= r1; ++r) hi = T.h_yend[r] > hi ? T.h_yend[r] : hi;
    span = hi - T.h_ymin[r0] > span ? hi - T.h_ymin[r0] : span;
  }
  *row_span = span;
  if (generation) *generation = g_tabs_gen;
  return hipSuccess;
}

// A forward that carries table pointers in its descriptors holds this from its first cs_preprocess_tables call until its last launch is queued: the
// cache is only ever dropped under the same mutex, behind a device synchronisation, so no other thread's drop can fall between this call's
// gathering and its launches (its own drop, while gathering, shows in the generation count).
extern "C" void cs_preprocess_tables_hold(int on) {
  if (on) g_tabs_mu.lock();
  else g_tabs_mu.unlock();
}
