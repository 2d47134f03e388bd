// C ABI + forward orchestration of the CrossScore gfx950 path (see include/crossscore_hip.h).
// Host-side only: owns packed weights + workspace, validates shapes, enqueues the HIP kernels of gemm.hip,
// attention.hip and elementwise.hip on the caller's stream.  Restates the control flow of
// CrossScoreNet.forward / get_featmaps (task/core.py:58-161), CrossReferenceNet.forward
// (model/cross_reference.py:52-94) and the post-norm decoder layer (transformer.py:157-173).
#include "../../include/crossscore_hip.h"
#include "cs_common.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <chrono>
#include <map>
#include <string>
#include <vector>

// patch.hip (C++ linkage)
size_t cs_patch_pack_elems(int C);
hipError_t cs_patch_pack_launch(const float* w, int C, h16_t* out, int bf, hipStream_t st);
int cs_patch_fused_supported(int H, int W, int P, int C);
hipError_t cs_patch_fused_u8_launch(const CsU8Desc* descs, int nq, int N, int img0, int I, int H, int W, int C, int row_span, const float* mean3,
                                    const float* std3, const h16_t* wfrag, const float* bias, const float* pos, const float* wsum, float* x, int bf,
                                    hipStream_t st);
int cs_patch_u8_runs(int W, int row_span);
hipError_t cs_patch_fused_launch(const float* xq, const float* xr, int N, int img0, int I, int H, int W, int C, const h16_t* wfrag,
                                 const float* bias, const float* pos, const float* wsum, float* x, int bf, hipStream_t st);

extern "C" {
const char* cs_gemm_check(const CsGemmParams* p, int epi);
hipError_t cs_gemm_launch(const CsGemmParams* p, int epi, hipStream_t stream);
const char* cs_attn_check(const CsAttnParams* p, int dh, int batch);
hipError_t cs_attn_launch(const CsAttnParams* p, int dh, int batch, hipStream_t stream);
hipError_t cs_preprocess_tables(int in_h, int in_w, int rs_h, int rs_w, int crop_y, int gh, int P, CsU8Tables* out, int* row_span, unsigned* generation);
void cs_preprocess_tables_hold(int on);
hipError_t cs_im2col_launch(const float* q, const float* refs, int N, int img0, h16_t* out, int I, int H, int W, int P, int Kp,
                            float* pmean, int bf, hipStream_t st);
hipError_t cs_patch_wsum_launch(const float* w, int C, int P, float* wsum, hipStream_t st);
hipError_t cs_ln_finalize_launch(const float* part, int M, int rows_padded, int sp, int C, float eps, float* stat, hipStream_t st);
hipError_t cs_layernorm_launch(const float* x, int M, int C, const float* g, const float* b, float eps, float* of32, h16_t* obf,
                               int bf, hipStream_t st);
hipError_t cs_final_ln_split_launch(const float* x, int I, int img0, int Np, int C, int N, const float* g, const float* b, float eps,
                                    const float* pe, float* q_f32, h16_t* q_bf, h16_t* mem_bf, int bf, hipStream_t st);
hipError_t cs_cls_rows_launch(float* x, int I, int T, int C, const float* cls, const float* pos, h16_t* xb, float* stats, int sp,
                              int bf, hipStream_t st);
hipError_t cs_ln_fold_consts_launch(const h16_t* wp, int ldp, const float* w, const float* beta, const float* bias, int N, int K,
                                    float* s_out, float* c_out, int bf, hipStream_t st);
int cs_gemm_column_tiles(int N);
hipError_t cs_pos_bicubic_launch(const float* pos, int G, int C, int gh, int gw, float grow, float* out, hipStream_t st);
hipError_t cs_pe_bilinear_launch(const float* pe, int ph, int pw, int C, int gh, int gw, float* out, hipStream_t st);
hipError_t cs_pe_interp_launch(const float* pe, int ph, int pw, int C, int gh, int gw, int mode, float* out, hipStream_t st);
hipError_t cs_pack_f16_launch(const float* w, int rows, int K, h16_t* out, int ldo, const float* row_scale, const float* col_scale,
                               int bf, hipStream_t st);
hipError_t cs_score_check_launch(const float* score, size_t n, unsigned* counter, hipStream_t st);
hipError_t cs_silu_mul_launch(h16_t* x, int M, int F, int ld, int bf, hipStream_t st);
hipError_t cs_vec_mul_launch(const float* a, const float* b, float* out, int n, hipStream_t st);
hipError_t cs_spin_launch(unsigned long long ticks, int blocks, int lds_bytes, hipStream_t st);
constexpr float LOG2E = 1.4426950408889634f;
hipError_t cs_attn_weights_launch(const CsAttnParams* p, int dh, int batch, int head, float* out, hipStream_t st);
hipError_t cs_score_gray16_launch(const float* score, size_t n, int signed_range, uint16_t* out, hipStream_t stream);
hipError_t cs_score_rgb_launch(const float* score, size_t n, float vmin, float vmax, const uint8_t* lut, uint8_t* out, hipStream_t stream);
int cs_panel_supported(int C, int mlp_ratio);
size_t cs_panel8_image_bytes(int with_outproj);
hipError_t cs_panel_pack_launch(const float* wo, const float* ls1, const float* w1, const float* g2, const float* w2, const float* ls2,
                                h16_t* img, int bf16, hipStream_t st);
const char* cs_panel_check(const CsPanelParams* p);
int cs_rowln_supported(int C);
const char* cs_rowln_check(const CsRowLnParams* p, int C);
hipError_t cs_rowln_launch(const CsRowLnParams* p, int C, int bf16, hipStream_t st);
hipError_t cs_panel_launch(const CsPanelParams* p, hipStream_t st);
// panel4.hip: the four-wave form of the same kernel (its own weight image)
size_t cs_panel4_image_bytes(int with_outproj);
hipError_t cs_panel4_pack_launch(const float* wo, const float* ls1, const float* w1, const float* g2, const float* w2, const float* ls2,
                                 h16_t* img, int bf16, hipStream_t st);
hipError_t cs_panel4_launch(const CsPanelParams* p, hipStream_t st);
hipError_t cs_preprocess_launch(const uint8_t* img, int in_h, int in_w, int row_bytes, int rs_h, int rs_w, int crop_y, int crop_x, int oh,
                                int ow, const float* mean, const float* stdv, float* out, float* scratch, hipStream_t stream);
}

namespace {

thread_local std::string g_err;
int g_debug_stream_log = 0;  // cs_debug_stream_probe_log: one stderr line per lane-stream candidate of the overlap probe
int g_panel_impl = 0;  // cs_debug_panel_impl: which token-panel kernel new handles and the cs_op_panel_* entry points use: 0 = panel.hip (8 waves), 1 = panel4.hip (4 waves)
int g_rowln_off = 0;  // cs_debug_rowln_enable(0): the decoder goes back to GEMM + LayerNorm launches (A/B runs and tests; process-wide)
int g_rowln_no_next = 0;  // cs_debug_rowln_enable(2): linear + LayerNorm in one launch, the following linear as a GEMM of its own (round 4's first form)
int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}
#define HIPCHK(expr)                                                                       \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess) return fail(CS_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

struct Tensor {
  std::vector<int64_t> shape;
  float* d = nullptr;
  size_t numel = 0;
};

struct EncLayer {
  float *ln1g, *ln1b, *ln2g, *ln2b, *bqkv, *bo, *b1, *b2, *ls1, *ls2;
  h16_t *Wqkv, *Wo, *W1, *W2;
  // LayerNorm fold (CS_EPI_LN_*): Wqkv / W1 above are then the gamma-scaled versions and these hold s[n], c[n]
  float *s_qkv, *c_qkv, *s_1, *c_1;
  h16_t* panel_img;  // token-panel kernel (panel.hip): packed unit stream [Wo | W1 / W2 interleaved]; Wqkv / c_qkv / c_1 are then the LN-folded ones
};
struct DecLayer {
  float *sa_bin, *sa_bo, *ca_bq, *ca_bo, *l1b, *l2b, *n1g, *n1b, *n2g, *n2b, *n3g, *n3b;
  h16_t *sa_Win, *sa_Wo, *ca_Wq, *ca_Wo, *l1W, *l2W;
};

struct ProfRec { hipEvent_t a, b; int family; double flops; double bytes; };
constexpr int CS_MAX_LANES = 4;

}  // namespace

struct cs_model {
  cs_config cfg{};
  std::vector<std::string> names;
  std::map<std::string, Tensor> w;
  bool finalized = false;
  int Kp = 0;  // padded patch K
  int qkv_n = 0;  // columns of the encoder's packed QKV projection: 3C, or 3C padded to whole 256-column tiles (zero rows) when that lets the
                  // large-tile GEMM take it (ViT-S: 1152 -> 1280; measured 47.5 -> 36.9 us per 24-image chunk, r4); attention reads with this stride
  // launch census of the last forward (cs_forward_stats): kernel launches by kernel, and the host time the call spent enqueueing them
  std::map<std::string, int> census;
  double host_enqueue_ms = 0.0;
  int panel_impl = 0;   // which panel kernel the images of this handle were packed for (g_panel_impl at cs_finalize)
  bool panel = false;   // encoder layers run as QKV GEMM + attention + ONE token-panel kernel (panel.hip; hidden == 384 only)
  float *ones = nullptr, *zeros = nullptr;  // [C]: layer 0's norm1 without gamma/beta (they are folded into its QKV projection)
  bool lnfold = false;  // encoder LayerNorms folded into the QKV / fc1 projections (no separate LN pass)
  bool fold256 = false; // the same fold on the 256-tile GEMM (gemm256.hip LN = 1 / 2; r5): the default of the wide backbones (hidden 768 / 1024) for chunks of >= 256 rows
  int ln_sp = 0;        // partial-sum slots per row the producing epilogues write (4 per column tile)
  std::vector<void*> owned;  // device allocations of packed weights
  // packed
  h16_t* Wpatch = nullptr; float* bpatch = nullptr;
  h16_t* Wpatch_frag = nullptr;  // fragment-ordered copy for the one-launch patch embedding (patch.hip); null when C is not 384 n or P != 14
  float* wsum = nullptr;  // [3][C] fp32 sums of the patch weights per channel (mean-centred patch embedding)
  std::vector<EncLayer> enc;
  std::vector<DecLayer> dec;
  h16_t* Wkv_all = nullptr; float* bkv_all = nullptr;
  h16_t *Wh0 = nullptr, *Wh2 = nullptr; float *bh0 = nullptr, *bh2 = nullptr;
  float *lnfg = nullptr, *lnfb = nullptr, *cls = nullptr, *pos = nullptr, *pe = nullptr;
  // per-(gh,gw,square) tables: built once per shape and kept (a shape change never overwrites a table that queued work may read)
  struct Tables { int gh, gw, sq; float *pos_tab, *pe_tab; bool pos_owned, pe_owned; };
  std::vector<Tables> tables;
  float *pos_tab = nullptr, *pe_tab = nullptr;  // the current shape's (point into `tables` or at the parameters)
  // workspace; a workspace that had to grow is retired behind an event and freed once that event has completed
  char* ws = nullptr; size_t ws_bytes = 0;
  struct Retired { void* p; hipEvent_t ev; };
  std::vector<Retired> retired;
  // lanes: internal streams that run independent image chunks / batch groups concurrently (forked from and joined to
  // the caller's stream with events), so one kernel's tail and the memory-bound stages overlap another's MFMA work
  hipStream_t lane_st[CS_MAX_LANES] = {};
  // one-pass input stage: pinned host copies of the per-image descriptors of the last U8_SLOTS forwards (the upload is asynchronous; a slot is
  // reused only behind the event recorded after its copy)
  static constexpr int U8_SLOTS = 4;
  struct U8Slot { CsU8Desc* host = nullptr; size_t cap = 0; hipEvent_t ev = nullptr; bool used = false; };
  U8Slot u8_slot[U8_SLOTS];
  int u8_next = 0;
  std::vector<hipStream_t> lane_st_old;  // given back by cs_redraw_lane_streams; destroyed once the next forward has drawn their replacements
  int lanes_now = 0;  // cs_set_lanes: lanes of the next forwards (0 = as configured)
  hipStream_t last_stream = nullptr; hipEvent_t ev_done = nullptr;  // ordering of calls that arrive on different streams
  hipEvent_t ev_kv0 = nullptr, ev_kv1 = nullptr;                    // decoder: K/V projection on a side stream
  hipEvent_t ev_fork = nullptr, ev_join[CS_MAX_LANES] = {}, ev_stag[CS_MAX_LANES] = {};
  unsigned* nonfinite = nullptr;  // device counter: non-finite score-map values seen since the last cs_nonfinite_count
  // profiling
  bool prof = false;
  std::vector<ProfRec> recs;
  // debug taps (cs_debug_capture / cs_debug_read): copies of intermediate tensors of the last forward, for the stage-level parity tests
  bool capture = false;
  struct Tap { void* d = nullptr; size_t bytes = 0; int dtype = 0; int ndim = 0; int64_t shape[4] = {0, 0, 0, 0}; };
  std::map<std::string, Tap> taps;
};

namespace {

std::vector<std::string> expected_names(const cs_config& c) {
  std::vector<std::string> n;
  n.push_back("img_mean_std");
  const std::string e = "backbone.embeddings.";
  n.push_back(e + "cls_token"); n.push_back(e + "mask_token"); n.push_back(e + "position_embeddings");
  n.push_back(e + "patch_embeddings.projection.weight"); n.push_back(e + "patch_embeddings.projection.bias");
  for (int l = 0; l < c.enc_layers; ++l) {
    const std::string p = "backbone.encoder.layer." + std::to_string(l) + ".";
    for (const char* s : {"norm1.weight", "norm1.bias", "attention.attention.query.weight", "attention.attention.query.bias",
                          "attention.attention.key.weight", "attention.attention.key.bias", "attention.attention.value.weight",
                          "attention.attention.value.bias", "attention.output.dense.weight", "attention.output.dense.bias",
                          "layer_scale1.lambda1", "norm2.weight", "norm2.bias", "layer_scale2.lambda1"})
      n.push_back(p + s);
    if (c.swiglu) for (const char* s : {"mlp.weights_in.weight", "mlp.weights_in.bias", "mlp.weights_out.weight", "mlp.weights_out.bias"}) n.push_back(p + s);
    else for (const char* s : {"mlp.fc1.weight", "mlp.fc1.bias", "mlp.fc2.weight", "mlp.fc2.bias"}) n.push_back(p + s);
  }
  n.push_back("backbone.layernorm.weight"); n.push_back("backbone.layernorm.bias");
  n.push_back("pos_enc_fn.PE");
  for (int l = 0; l < c.dec_layers; ++l) {
    const std::string p = "ref_cross.attn.layers." + std::to_string(l) + ".";
    std::vector<std::string> blocks;
    if (c.do_self_attn) blocks.push_back("self_attn");
    blocks.push_back("multihead_attn");
    for (auto& b : blocks)
      for (const char* s : {".in_proj_weight", ".in_proj_bias", ".out_proj.weight", ".out_proj.bias"}) n.push_back(p + b + s);
    for (const char* s : {"linear1.weight", "linear1.bias", "linear2.weight", "linear2.bias", "norm1.weight", "norm1.bias",
                          "norm2.weight", "norm2.bias", "norm3.weight", "norm3.bias"})
      n.push_back(p + s);
  }
  for (const char* s : {"ref_cross.head.0.weight", "ref_cross.head.0.bias", "ref_cross.head.2.weight", "ref_cross.head.2.bias"})
    n.push_back(s);
  return n;
}

// hidden features of the encoder's MLP: mlp_ratio * hidden, or the SwiGLU form's (int(hidden * mlp_ratio * 2 / 3) + 7) / 8 * 8 (HF modeling_dinov2.py:303-305)
int ffn_hidden(const cs_config& c) {
  const int f = c.mlp_ratio * c.hidden;
  return c.swiglu ? ((int)((double)f * 2 / 3) + 7) / 8 * 8 : f;
}

bool supported_dh(int dh) { return dh == 16 || dh == 48 || dh == 64 || dh == 96 || dh == 128 || dh == 192; }

struct Arena {  // carve 256-byte aligned pieces out of the workspace
  char* base; size_t off = 0;
  template <typename T> T* take(size_t n) {
    T* p = reinterpret_cast<T*>(base + off);
    off += (n * sizeof(T) + 255) & ~size_t(255);
    return p;
  }
};

struct Plan {
  int B, N, H, W, gh, gw, Np, T, I, Ic, C, lanes;
  size_t total;
  // encoder chunk buffers, one set per lane
  float* x[CS_MAX_LANES]; h16_t* u[CS_MAX_LANES]; h16_t* r1[CS_MAX_LANES];
  h16_t* ob[CS_MAX_LANES]; float* stats[CS_MAX_LANES];  // LayerNorm fold: attention output, per-row partial sums
  float* lnstat[CS_MAX_LANES];                           // fold256: finalised (mean, rstd) per row, whole 256-row tiles
  float* pmean[CS_MAX_LANES];                            // per-patch channel means removed by im2col
  // decoder
  float *xq, *y, *lse; h16_t *q_bf, *mem_bf, *kv, *dqkv, *dq, *dob, *dhid;
  float* mean_part; unsigned* mean_cnt;  // the head launch's per-image mean (CsGemmParams::mean_*)
  CsU8Desc* u8desc;                      // one-pass input stage: B query descriptors, then B * N_enc reference descriptors
};

Plan make_plan(const cs_model* m, int B, int N, int N_enc, int H, int W, char* base) {
  Plan p{};
  const cs_config& c = m->cfg;
  p.B = B; p.N = N; p.H = H; p.W = W; p.C = c.hidden;
  p.gh = H / c.patch; p.gw = W / c.patch; p.Np = p.gh * p.gw; p.T = p.Np + 1; p.I = B * (1 + N_enc);
  p.lanes = m->prof ? 1 : (c.lanes <= 0 ? 2 : std::min(c.lanes, CS_MAX_LANES));  // profiling times kernels in isolation
  if (m->lanes_now > 0) p.lanes = std::min(p.lanes, m->lanes_now);               // cs_set_lanes (the workspace holds the configured number)
  // cfg-2: 2 lanes x 24 images measured best (7.51 vs 7.70 ms with 12).  ViT-B: with the 256-row-tile GEMM (gemm256.hip) a chunk has to hold
  // many row tiles per CU: cfg-4 449 q/s at 48 or 16 images per chunk, 419 at 6, 407 at 12; cfg-3 247 at 44, 248 at 11 (tools/lanes_sweep_b.py, r3)
  int ic = c.enc_chunk_images > 0 ? c.enc_chunk_images : (c.hidden <= 384 ? 24 : 48);
  // balanced chunks: a multiple of the lane count, near-equal sizes
  if (c.enc_chunk_images > 0) {
    p.Ic = std::min(ic, p.I);  // explicit: used verbatim (a shorter remainder chunk runs first)
  } else {
    // whole batch items per chunk (so a lane can decode what it just encoded), about `ic` images, balanced over the lanes
    const int per_item = 1 + N_enc;
    int items = std::max(1, ic / per_item);
    int passes = (B + items - 1) / items;
    if (B >= p.lanes) passes = ((passes + p.lanes - 1) / p.lanes) * p.lanes;
    passes = std::min(passes, B);
    items = (B + passes - 1) / passes;
    p.Ic = items * per_item;
  }
  const size_t C = c.hidden, Mc = (size_t)p.Ic * p.T, M = (size_t)B * p.Np, Mk = (size_t)B * N * p.Np;
  Arena a{base};
  const int nsets = c.lanes <= 0 ? 2 : std::min(c.lanes, CS_MAX_LANES);  // independent of the profiling mode
  for (int l = 0; l < nsets; ++l) {
    p.x[l] = a.take<float>(Mc * C);
    p.u[l] = a.take<h16_t>(Mc * C);
    p.r1[l] = a.take<h16_t>(std::max(Mc * (size_t)(c.swiglu ? 2 * ffn_hidden(c) : ffn_hidden(c)), std::max(Mc * (size_t)m->qkv_n, (size_t)p.Ic * p.Np * m->Kp)));
    p.ob[l] = a.take<h16_t>(m->lnfold || m->fold256 ? Mc * C : 0);
    p.pmean[l] = a.take<float>((size_t)p.Ic * p.Np * 4);
    const size_t Mpad = (Mc + 255) / 256 * 256;
    p.stats[l] = a.take<float>(m->lnfold ? Mc * (size_t)m->ln_sp * 2 : (m->fold256 ? Mpad * (C / 64) * 2 : 0));
    p.lnstat[l] = a.take<float>(m->fold256 ? Mpad * 2 : 0);
  }
  p.xq = a.take<float>(M * C);
  p.y = a.take<float>(M * C);
  p.q_bf = a.take<h16_t>(M * C);
  p.mem_bf = a.take<h16_t>(Mk * C);
  p.kv = a.take<h16_t>(Mk * 2 * C * c.dec_layers);
  p.dqkv = a.take<h16_t>(M * 3 * C);
  p.dq = a.take<h16_t>(M * C);
  p.dob = a.take<h16_t>(M * C);
  p.dhid = a.take<h16_t>(M * C);
  p.lse = a.take<float>((size_t)B * c.dec_heads * p.Np);
  p.mean_part = a.take<float>(M * 4 * (size_t)cs_gemm_column_tiles(c.patch * c.patch));
  p.mean_cnt = a.take<unsigned>((size_t)B);
  p.u8desc = a.take<CsU8Desc>((size_t)p.I);
  p.total = a.off;
  return p;
}

// launch helpers that record profiling events when enabled
// the optional second stage of Launcher::rowln: the sub-block's following linear, out (M, n) = act(LN rows x W^T + b)
struct NextLinear { const h16_t* W = nullptr; const float* b = nullptr; h16_t* out = nullptr; int n = 0, act = 0; };

extern "C" int cs_gemm256_supported(const CsGemmParams* p, int epi);  // gemm256.hip: the shapes cs_gemm_launch routes to the 256-tile kernel

struct Launcher {
  cs_model* m; hipStream_t st; int rc = 0;
  int bpc = 0;  // GEMM blocks per CU hint (CsGemmParams::bpc)
  void begin(int family, double flops, double bytes = 0) {
    if (!m->prof) return;
    ProfRec r{}; r.family = family; r.flops = flops; r.bytes = bytes;
    hipEventCreate(&r.a); hipEventCreate(&r.b);
    hipEventRecord(r.a, st);
    m->recs.push_back(r);
  }
  void end() { if (m->prof) hipEventRecord(m->recs.back().b, st); }
  bool gemm(CsGemmParams g, int epi, double k_real = 0) {
    if (rc) return false;
    g.bpc = bpc;
    g.bf16 = m->cfg.operand_dtype;  // before the check: its "bf16 with a LayerNorm-folded epilogue" guard reads it
    if (const char* e = cs_gemm_check(&g, epi)) { rc = fail(CS_ERR_BAD_ARG, "%s", e); return false; }
    // algorithmic HBM bytes of one launch: A and W once (fp16), bias, the output once, the residual / position addend once
    const double mn = (double)g.M * g.N;
    const bool f32out = epi == CS_EPI_RESID_F32 || epi == CS_EPI_RESID_F32_LN || epi == CS_EPI_PATCH_F32 || epi == CS_EPI_HEAD_SCORE;
    double bytes = 2.0 * g.M * g.K + 2.0 * g.N * g.K + 4.0 * g.N + mn * (f32out ? 4.0 : 2.0);
    if ((epi == CS_EPI_RESID_F32 || epi == CS_EPI_RESID_F32_LN) && g.resid) bytes += 4.0 * mn;
    if (epi == CS_EPI_PATCH_F32) bytes += 4.0 * g.Np * g.N;
    begin(epi, 2.0 * g.M * g.N * (k_real > 0 ? k_real : g.K), bytes);
    m->census[cs_gemm256_supported(&g, epi) ? "gemm256" : "gemm128"]++;
    hipError_t e = cs_gemm_launch(&g, epi, st);
    end();
    if (e != hipSuccess) { rc = fail(CS_ERR_HIP, "gemm launch: %s", hipGetErrorString(e)); return false; }
    return true;
  }
  bool attn(CsAttnParams a, int dh, int batch) {
    if (rc) return false;
    a.bf16 = m->cfg.operand_dtype;
    if (const char* e = cs_attn_check(&a, dh, batch)) { rc = fail(CS_ERR_BAD_ARG, "%s", e); return false; }
    // Q and O once, K and V once per (batch, head): 2 bytes each
    begin(16 + dh / 16, 4.0 * batch * a.heads * (double)a.Lq * a.Lk * dh, 2.0 * batch * a.heads * dh * (2.0 * a.Lq + 2.0 * a.Lk));
    m->census["attn" + std::to_string(dh)]++;
    hipError_t e = cs_attn_launch(&a, dh, batch, st);
    end();
    if (e != hipSuccess) { rc = fail(CS_ERR_HIP, "attention launch: %s", hipGetErrorString(e)); return false; }
    return true;
  }
  bool panel(CsPanelParams q) {
    if (rc) return false;
    q.bf16 = m->cfg.operand_dtype;
    if (const char* e = cs_panel_check(&q)) { rc = fail(CS_ERR_BAD_ARG, "%s", e); return false; }
    const double M = q.M, C = m->cfg.hidden, F = (double)m->cfg.mlp_ratio * C;
    // algorithmic bytes: x read + written (fp32), attention output read, u written (fp16), the weight stream once
    begin(40, 2.0 * M * C * C * (q.attn_o ? 1 : 0) + 4.0 * M * C * F,
          M * C * (8.0 + (q.attn_o ? 2.0 : 0.0) + (q.u_out ? 2.0 : 0.0)) + (double)(2 * (q.attn_o ? 1 : 0) + 16) * C * C);
    m->census[m->panel_impl ? "panel4" : "panel"]++;
    hipError_t e = m->panel_impl ? cs_panel4_launch(&q, st) : cs_panel_launch(&q, st);
    end();
    if (e != hipSuccess) { rc = fail(CS_ERR_HIP, "panel launch: %s", hipGetErrorString(e)); return false; }
    return true;
  }
  // out = LN(resid + A W^T + bias): the decoder's out-projection / linear2 + residual + LayerNorm in one launch (rowln.hip; C = 384)
  bool rowln(const h16_t* A, const h16_t* W, const float* bias, const float* resid, const float* gamma, const float* beta, float eps,
             float* out_f32, h16_t* out_f16, int M, NextLinear next = NextLinear{}) {
    if (rc) return false;
    const int C = m->cfg.hidden;
    CsRowLnParams q{};
    q.A = A; q.lda = C; q.W = W; q.ldw = C; q.bias = bias; q.resid = resid; q.ldr = C; q.gamma = gamma; q.beta = beta; q.eps = eps;
    q.out_f32 = out_f32; q.out_f16 = out_f16; q.M = M;
    q.W2 = next.W; q.ldw2 = C; q.bias2 = next.b; q.out2 = next.out; q.ld2 = next.n; q.n2 = next.n; q.act2 = next.act;
    if (const char* e = cs_rowln_check(&q, C)) { rc = fail(CS_ERR_BAD_ARG, "%s", e); return false; }
    // algorithmic bytes: A and W once, the residual rows in, the normalised rows out (fp32, and 16-bit where asked); second stage: W2 in, rows out
    begin(42, 2.0 * M * C * (double)(C + next.n),
          2.0 * M * C + 2.0 * C * C + (resid ? 4.0 : 0.0) * M * C + (out_f32 ? 4.0 : 0.0) * M * C + (out_f16 ? 2.0 : 0.0) * M * C + 2.0 * next.n * C + 2.0 * M * next.n);
    m->census["rowln"]++;
    hipError_t e = cs_rowln_launch(&q, C, m->cfg.operand_dtype, st);
    end();
    if (e != hipSuccess) { rc = fail(CS_ERR_HIP, "linear + LayerNorm launch: %s", hipGetErrorString(e)); return false; }
    return true;
  }
  bool misc(hipError_t e, const char* what) {
    m->census[what]++;
    if (rc) return false;
    if (e != hipSuccess) { rc = fail(CS_ERR_HIP, "%s launch: %s", what, hipGetErrorString(e)); return false; }
    return true;
  }
};

CsGemmParams gp(const h16_t* A, int lda, const h16_t* W, int ldw, int M, int N, int K, const float* bias, void* out, int ldc) {
  CsGemmParams g{};
  g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.M = M; g.N = N; g.K = K; g.bias = bias; g.out = out; g.ldc = ldc;
  g.powp = 1.f;
  return g;
}

int ensure_tables(cs_model* m, int gh, int gw, bool square, hipStream_t st) {
  const cs_config& c = m->cfg;
  const int sq = square ? 1 : 0;
  for (auto& t : m->tables)
    if (t.gh == gh && t.gw == gw && t.sq == sq) { m->pos_tab = t.pos_tab; m->pe_tab = t.pe_tab; return 0; }
  // first forward of this patch grid: allocate and fill its tables on the caller's stream (stream-ordered with the kernels that
  // read them).  Tables of other grids stay as they are -- work queued on any stream may still read them -- so there is nothing to
  // wait for; only past 16 distinct grids are the oldest dropped, behind a device synchronisation.
  if (m->tables.size() >= 16) {
    HIPCHK(hipDeviceSynchronize());
    for (auto& t : m->tables) { if (t.pos_owned) hipFree(t.pos_tab); if (t.pe_owned) hipFree(t.pe_tab); }
    m->tables.clear();
  }
  cs_model::Tables t{gh, gw, sq, nullptr, nullptr, false, false};
  const int Np = gh * gw, C = c.hidden;
  if (Np == c.pos_grid * c.pos_grid && square) {  // HF:71 -- parameter used as is
    t.pos_tab = m->pos;
  } else {
    HIPCHK(hipMalloc(&t.pos_tab, (size_t)(1 + Np) * C * sizeof(float)));
    t.pos_owned = true;
    HIPCHK(cs_pos_bicubic_launch(m->pos, c.pos_grid, C, gh, gw, c.pos_interp_legacy ? 0.1f : 0.0f, t.pos_tab, st));
  }
  if (gh == c.pe_h && gw == c.pe_w) {  // positional_encoding.py:51-56
    t.pe_tab = m->pe;
  } else {
    HIPCHK(hipMalloc(&t.pe_tab, (size_t)Np * C * sizeof(float)));
    t.pe_owned = true;
    HIPCHK(cs_pe_interp_launch(m->pe, c.pe_h, c.pe_w, C, gh, gw, c.pe_interp_mode, t.pe_tab, st));
  }
  m->tables.push_back(t);
  m->pos_tab = t.pos_tab; m->pe_tab = t.pe_tab;
  return 0;
}

// Frees retired workspaces whose last use has completed (never blocks).
void reap_retired(cs_model* m, bool all) {
  for (size_t i = 0; i < m->retired.size();) {
    if (all || hipEventQuery(m->retired[i].ev) == hipSuccess) {
      hipFree(m->retired[i].p); hipEventDestroy(m->retired[i].ev);
      m->retired.erase(m->retired.begin() + i);
    } else {
      ++i;
    }
  }
}

// Debug tap: copies `bytes` of `src` into the tap `name` at byte offset `off` on stream `st` (stream-ordered behind the kernel that wrote src).
// The tap buffer holds `total` bytes and is (re)allocated here when its size changes: capture mode is for tests, not for timed runs.
int tap_buffer(cs_model* m, const std::string& name, size_t total, int dtype, std::initializer_list<int64_t> shape, void** out) {
  cs_model::Tap& t = m->taps[name];
  if (t.bytes != total) {
    if (t.d) { HIPCHK(hipDeviceSynchronize()); hipFree(t.d); t.d = nullptr; t.bytes = 0; }
    HIPCHK(hipMalloc(&t.d, total));
    t.bytes = total;
  }
  t.dtype = dtype; t.ndim = (int)shape.size();
  int k = 0;
  for (int64_t v : shape) t.shape[k++] = v;
  *out = t.d;
  return 0;
}
int tap_copy(cs_model* m, const std::string& name, const void* src, size_t off, size_t bytes, size_t total, int dtype,
             std::initializer_list<int64_t> shape, hipStream_t st) {
  if (!m->capture) return 0;
  void* d = nullptr;
  if (int r = tap_buffer(m, name, total, dtype, shape, &d)) return r;
  if (off + bytes > total) return fail(CS_ERR_STATE, "debug tap %s: copy out of range", name.c_str());
  HIPCHK(hipMemcpyAsync(static_cast<char*>(d) + off, src, bytes, hipMemcpyDeviceToDevice, st));
  return 0;
}

}  // namespace

extern "C" {

const char* cs_last_error(void) { return g_err.c_str(); }

cs_handle cs_create(const cs_config* cfg) {
  if (!cfg) { fail(CS_ERR_BAD_ARG, "cs_create: null config"); return nullptr; }
  const cs_config& c = *cfg;
  if (c.hidden <= 0 || c.hidden % 64 || c.hidden > 1536) { fail(CS_ERR_UNSUPPORTED, "hidden=%d must be a multiple of 64 and <= 1536", c.hidden); return nullptr; }
  if (c.enc_layers <= 0 || c.enc_heads <= 0 || c.hidden % c.enc_heads) { fail(CS_ERR_BAD_ARG, "bad encoder layers/heads"); return nullptr; }
  if (!supported_dh(c.hidden / c.enc_heads)) { fail(CS_ERR_UNSUPPORTED, "encoder head dim %d not in {16,48,64,96,128,192}", c.hidden / c.enc_heads); return nullptr; }
  if (c.dec_heads <= 0 || c.hidden % c.dec_heads || !supported_dh(c.hidden / c.dec_heads)) { fail(CS_ERR_UNSUPPORTED, "decoder head dim %d not in {16,48,64,96,128,192}", c.dec_heads > 0 ? c.hidden / c.dec_heads : 0); return nullptr; }
  if (c.dec_layers <= 0 || c.patch <= 0 || (c.patch * c.patch) % 4 || c.pos_grid <= 0 || c.pe_h <= 0 || c.pe_w <= 0 || c.mlp_ratio <= 0) { fail(CS_ERR_BAD_ARG, "bad config"); return nullptr; }
  if (c.act != 0 && c.act != 1) { fail(CS_ERR_BAD_ARG, "act must be 0 (sigmoid) or 1 (tanh)"); return nullptr; }
  if (c.act == 1 && c.pow_p != 1.0f) { fail(CS_ERR_BAD_ARG, "power factor applies only to the sigmoid range"); return nullptr; }
  if (c.pe_interp_mode != 0 && c.pe_interp_mode != 1) { fail(CS_ERR_BAD_ARG, "pe_interp_mode must be 0 (bilinear) or 1 (bicubic)"); return nullptr; }
  if (c.operand_dtype != 0 && c.operand_dtype != 1) { fail(CS_ERR_BAD_ARG, "operand_dtype must be 0 (fp16) or 1 (bf16)"); return nullptr; }
  if (c.operand_dtype == 1 && c.ln_fold == 1) { fail(CS_ERR_UNSUPPORTED, "the LayerNorm-folded epilogues (ln_fold = 1) are built for fp16 operands only"); return nullptr; }
  if (c.swiglu != 0 && c.swiglu != 1) { fail(CS_ERR_BAD_ARG, "swiglu must be 0 or 1"); return nullptr; }
  if (c.swiglu && ffn_hidden(c) % 64) { fail(CS_ERR_UNSUPPORTED, "SwiGLU hidden features %d must be a multiple of 64 (the GEMM's K)", ffn_hidden(c)); return nullptr; }
  if (c.swiglu && c.ln_fold == 1) { fail(CS_ERR_UNSUPPORTED, "ln_fold = 1 is built for the GELU MLP only"); return nullptr; }
  cs_model* m = new cs_model();
  m->cfg = c;
  m->names = expected_names(c);
  m->Kp = ((3 * c.patch * c.patch + 63) / 64) * 64;
  // the 256 x 256 x 64-tile GEMM (gemm256.hip) needs whole 256-column tiles and K >= 384 in multiples of 128
  m->qkv_n = (c.hidden >= 384 && c.hidden % 128 == 0) ? ((3 * c.hidden + 255) / 256) * 256 : 3 * c.hidden;
  m->ln_sp = 4 * cs_gemm_column_tiles(c.hidden);
  // opt-in: measured slower than separate LayerNorm kernels on cfg-2 (kernel-time sum 10.2 vs 9.5 ms: the folded consumers run
  // at the 256-register limit and their tile-switch loads drain the LDS-DMA queue)
  m->lnfold = c.ln_fold == 1 && (m->ln_sp == 4 || m->ln_sp == 8 || m->ln_sp == 16);
  // default for ViT-S: out-proj + norm2 + MLP + next norm1 in one launch per layer, the 4C hidden never leaves the registers
  m->panel = c.enc_fused != 1 && !m->lnfold && !c.swiglu && cs_panel_supported(c.hidden, c.mlp_ratio);
  m->panel_impl = g_panel_impl;
  // wide backbones (r5): the encoder's LayerNorms ride in the 256-tile GEMM's epilogues; ln_fold = 2 keeps the separate LayerNorm launches
  m->fold256 = c.ln_fold == 0 && !m->panel && !c.swiglu && c.hidden >= 512 && c.hidden % 256 == 0 && (c.mlp_ratio * c.hidden) % 256 == 0;
  return m;
}

void cs_destroy(cs_handle h) {
  if (!h) return;
  // waits explicitly for THIS handle's work (its last forward's completion event and its lane streams; a forward that failed half-way may have
  // left lane work that no event covers).  The hipFree calls below synchronise the device on top of that in today's runtime -- which is also
  // what covers caller-stream work of a half-failed forward -- so siblings of a pipeline do stall while one handle is torn down; nothing here
  // relies on them not stalling, and nothing of this handle is freed before its own streams have drained.
  if (h->ev_done) hipEventSynchronize(h->ev_done);
  for (int l = 0; l < CS_MAX_LANES; ++l) if (h->lane_st[l]) hipStreamSynchronize(h->lane_st[l]);
  for (auto& r : h->retired) hipEventSynchronize(r.ev);
  for (auto& kv : h->w) if (kv.second.d) hipFree(kv.second.d);
  for (void* p : h->owned) hipFree(p);
  for (auto& t : h->tables) { if (t.pos_owned) hipFree(t.pos_tab); if (t.pe_owned) hipFree(t.pe_tab); }
  reap_retired(h, true);
  if (h->ws) hipFree(h->ws);
  if (h->ev_done) hipEventDestroy(h->ev_done);
  if (h->ev_kv0) hipEventDestroy(h->ev_kv0);
  if (h->ev_kv1) hipEventDestroy(h->ev_kv1);
  for (int l = 0; l < CS_MAX_LANES; ++l) {
    if (h->lane_st[l]) hipStreamDestroy(h->lane_st[l]);
    if (h->ev_join[l]) hipEventDestroy(h->ev_join[l]);
    if (h->ev_stag[l]) hipEventDestroy(h->ev_stag[l]);
  }
  for (hipStream_t o : h->lane_st_old) hipStreamDestroy(o);
  if (h->ev_fork) hipEventDestroy(h->ev_fork);
  if (h->nonfinite) hipFree(h->nonfinite);
  for (auto& u : h->u8_slot) { if (u.host) hipHostFree(u.host); if (u.ev) hipEventDestroy(u.ev); }
  for (auto& r : h->recs) { hipEventDestroy(r.a); hipEventDestroy(r.b); }
  for (auto& kv : h->taps) if (kv.second.d) hipFree(kv.second.d);
  delete h;
}

int cs_num_weights(cs_handle h) { return h ? (int)h->names.size() : 0; }
const char* cs_weight_name(cs_handle h, int i) { return (h && i >= 0 && i < (int)h->names.size()) ? h->names[i].c_str() : nullptr; }

// dtype: CS_DTYPE_F32 / F16 / BF16 of the SOURCE buffer; the handle keeps fp32 copies (exact for all three) and packs its fp16
// operand images from those in cs_finalize
int cs_set_weight_typed(cs_handle h, const char* name, const void* data, int is_device, int dtype, int ndim, const int64_t* shape) {
  if (!h || !name || !data || ndim < 1 || !shape) return fail(CS_ERR_BAD_ARG, "cs_set_weight: null argument");
  if (dtype != CS_DTYPE_F32 && dtype != CS_DTYPE_F16 && dtype != CS_DTYPE_BF16) return fail(CS_ERR_BAD_ARG, "cs_set_weight: unknown dtype %d", dtype);
  if (h->finalized) return fail(CS_ERR_STATE, "cs_set_weight after cs_finalize");
  if (std::find(h->names.begin(), h->names.end(), name) == h->names.end()) return fail(CS_ERR_BAD_ARG, "unexpected key %s", name);
  Tensor t;
  t.numel = 1;
  for (int i = 0; i < ndim; ++i) { if (shape[i] <= 0) return fail(CS_ERR_BAD_ARG, "%s: bad shape", name); t.shape.push_back(shape[i]); t.numel *= (size_t)shape[i]; }
  auto it = h->w.find(name);
  if (it != h->w.end() && it->second.d) { hipFree(it->second.d); it->second.d = nullptr; }
  HIPCHK(hipMalloc(&t.d, t.numel * sizeof(float)));
  if (dtype == CS_DTYPE_F32) {
    HIPCHK(hipMemcpy(t.d, data, t.numel * sizeof(float), is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
  } else {
    // 16-bit sources are widened on the host (checkpoint loading is set-up work; a device source is read back first)
    std::vector<uint16_t> src(t.numel);
    HIPCHK(hipMemcpy(src.data(), data, t.numel * sizeof(uint16_t), is_device ? hipMemcpyDeviceToHost : hipMemcpyHostToHost));
    std::vector<float> wide(t.numel);
    for (size_t i = 0; i < t.numel; ++i) {
      if (dtype == CS_DTYPE_BF16) {
        const uint32_t u = (uint32_t)src[i] << 16;
        std::memcpy(&wide[i], &u, 4);
      } else {
        wide[i] = (float)__builtin_bit_cast(_Float16, src[i]);  // host side: IEEE half -> float
      }
    }
    HIPCHK(hipMemcpy(t.d, wide.data(), t.numel * sizeof(float), hipMemcpyHostToDevice));
  }
  h->w[name] = t;
  return 0;
}

int cs_set_weight(cs_handle h, const char* name, const float* data, int is_device, int ndim, const int64_t* shape) {
  return cs_set_weight_typed(h, name, data, is_device, CS_DTYPE_F32, ndim, shape);
}

int cs_finalize(cs_handle h) {
  if (!h) return fail(CS_ERR_BAD_ARG, "null handle");
  if (h->finalized) return 0;
  const cs_config& c = h->cfg;
  const int64_t C = c.hidden, P = c.patch, F = ffn_hidden(c), F1 = c.swiglu ? 2 * F : F;  // F1: rows of the first projection (x1 | x2 for SwiGLU)
  auto need = [&](const std::string& n, std::vector<int64_t> shp) -> Tensor* {
    auto it = h->w.find(n);
    if (it == h->w.end()) { fail(CS_ERR_STATE, "missing key %s", n.c_str()); return nullptr; }
    if (it->second.shape != shp) {
      std::string got, want;
      for (auto v : it->second.shape) got += std::to_string(v) + ",";
      for (auto v : shp) want += std::to_string(v) + ",";
      fail(CS_ERR_BAD_ARG, "size mismatch for %s: got (%s) expected (%s)", n.c_str(), got.c_str(), want.c_str());
      return nullptr;
    }
    return &it->second;
  };
  hipStream_t st = nullptr;
  auto pack = [&](const float* src, int rows, int K, int ldo, h16_t* dst, const float* row_scale = nullptr,
                  const float* col_scale = nullptr) -> int {
    HIPCHK(cs_pack_f16_launch(src, rows, K, dst, ldo, row_scale, col_scale, c.operand_dtype, st));
    return 0;
  };
  auto alloc_bf = [&](size_t n) -> h16_t* { void* p = nullptr; if (hipMalloc(&p, n * sizeof(h16_t)) != hipSuccess) return nullptr; h->owned.push_back(p); return (h16_t*)p; };
  auto alloc_f = [&](size_t n) -> float* { void* p = nullptr; if (hipMalloc(&p, n * sizeof(float)) != hipSuccess) return nullptr; h->owned.push_back(p); return (float*)p; };
#define NEED(var, name, ...) Tensor* var = need(name, {__VA_ARGS__}); if (!var) return CS_ERR_STATE;
#define ALLOC_BF(var, n) h16_t* var = alloc_bf(n); if (!var) return fail(CS_ERR_HIP, "hipMalloc failed");
#define ALLOC_F(var, n) float* var = alloc_f(n); if (!var) return fail(CS_ERR_HIP, "hipMalloc failed");
#define D2D(dst, src, n) HIPCHK(hipMemcpy(dst, src, (n) * sizeof(float), hipMemcpyDeviceToDevice))

  const std::string e = "backbone.embeddings.";
  NEED(t_ms, "img_mean_std", 6) (void)t_ms;
  NEED(t_cls, e + "cls_token", 1, 1, C)
  NEED(t_mask, e + "mask_token", 1, C) (void)t_mask;
  NEED(t_pos, e + "position_embeddings", 1, (int64_t)c.pos_grid * c.pos_grid + 1, C)
  NEED(t_pw, e + "patch_embeddings.projection.weight", C, 3, P, P)
  NEED(t_pb, e + "patch_embeddings.projection.bias", C)
  h->cls = t_cls->d; h->pos = t_pos->d; h->bpatch = t_pb->d;
  { ALLOC_BF(wp, (size_t)C * h->Kp) if (int r = pack(t_pw->d, (int)C, (int)(3 * P * P), h->Kp, wp)) return r; h->Wpatch = wp; }
  { ALLOC_F(ws, (size_t)3 * C) HIPCHK(cs_patch_wsum_launch(t_pw->d, (int)C, (int)P, ws, st)); h->wsum = ws; }
  if (P == 14 && C % 384 == 0) {
    ALLOC_BF(wf, cs_patch_pack_elems((int)C))
    HIPCHK(cs_patch_pack_launch(t_pw->d, (int)C, wf, c.operand_dtype, st));
    h->Wpatch_frag = wf;
  }
  // softmax scale folded into the Q projections: the attention kernel takes Q pre-multiplied by log2(e)/sqrt(dh) (scale_log2e = 1),
  // so the factor is applied to the fp32 weights and biases before their single fp16 rounding instead of to fp16 Q values
  ALLOC_F(qs_enc, (size_t)C) ALLOC_F(qs_dec, (size_t)C)
  {
    std::vector<float> hv((size_t)C, LOG2E / std::sqrt((float)(C / c.enc_heads)));
    HIPCHK(hipMemcpy(qs_enc, hv.data(), (size_t)C * sizeof(float), hipMemcpyHostToDevice));
    std::fill(hv.begin(), hv.end(), LOG2E / std::sqrt((float)(C / c.dec_heads)));
    HIPCHK(hipMemcpy(qs_dec, hv.data(), (size_t)C * sizeof(float), hipMemcpyHostToDevice));
  }
  h->enc.resize(c.enc_layers);
  for (int l = 0; l < c.enc_layers; ++l) {
    const std::string p = "backbone.encoder.layer." + std::to_string(l) + ".";
    EncLayer& L = h->enc[l];
    NEED(n1w, p + "norm1.weight", C) NEED(n1b, p + "norm1.bias", C) NEED(n2w, p + "norm2.weight", C) NEED(n2b, p + "norm2.bias", C)
    NEED(qw, p + "attention.attention.query.weight", C, C) NEED(qb, p + "attention.attention.query.bias", C)
    NEED(kw, p + "attention.attention.key.weight", C, C) NEED(kb, p + "attention.attention.key.bias", C)
    NEED(vw, p + "attention.attention.value.weight", C, C) NEED(vb, p + "attention.attention.value.bias", C)
    NEED(ow, p + "attention.output.dense.weight", C, C) NEED(ob, p + "attention.output.dense.bias", C)
    NEED(l1, p + "layer_scale1.lambda1", C) NEED(l2, p + "layer_scale2.lambda1", C)
    NEED(f1w, p + (c.swiglu ? "mlp.weights_in.weight" : "mlp.fc1.weight"), F1, C) NEED(f1b, p + (c.swiglu ? "mlp.weights_in.bias" : "mlp.fc1.bias"), F1)
    NEED(f2w, p + (c.swiglu ? "mlp.weights_out.weight" : "mlp.fc2.weight"), C, F) NEED(f2b, p + (c.swiglu ? "mlp.weights_out.bias" : "mlp.fc2.bias"), C)
    L.ln1g = n1w->d; L.ln1b = n1b->d; L.ln2g = n2w->d; L.ln2b = n2b->d; L.b1 = f1b->d;
    // LayerScale (x += lambda * (a Wo^T + bo), HF:367-370,376-378) is folded into the projection: rows of Wo / W2 and the
    // biases are scaled by lambda once, so the GEMM epilogue is a plain residual add
    ALLOC_F(bo_s, (size_t)C) ALLOC_F(b2_s, (size_t)C)
    HIPCHK(cs_vec_mul_launch(ob->d, l1->d, bo_s, (int)C, st));
    HIPCHK(cs_vec_mul_launch(f2b->d, l2->d, b2_s, (int)C, st));
    L.bo = bo_s; L.b2 = b2_s; L.ls1 = nullptr; L.ls2 = nullptr;
    const size_t NQ = (size_t)h->qkv_n;  // >= 3C: the padding rows of the weights and the padding entries of the bias / fold vectors are zero
    ALLOC_BF(wqkv, NQ * C) ALLOC_F(bqkv, NQ)
    if (NQ > 3 * C) { HIPCHK(hipMemset(wqkv + 3 * C * C, 0, (NQ - 3 * C) * C * sizeof(h16_t))); HIPCHK(hipMemset(bqkv + 3 * C, 0, (NQ - 3 * C) * sizeof(float))); }
    const bool fold = h->lnfold || h->panel || h->fold256;
    const float* g1 = fold ? n1w->d : nullptr;  // LayerNorm gamma folded into the columns of the consuming projection
    const float* g2 = fold ? n2w->d : nullptr;
    if (int r = pack(qw->d, (int)C, (int)C, (int)C, wqkv, qs_enc, g1)) return r;
    if (int r = pack(kw->d, (int)C, (int)C, (int)C, wqkv + C * C, nullptr, g1)) return r;
    if (int r = pack(vw->d, (int)C, (int)C, (int)C, wqkv + 2 * C * C, nullptr, g1)) return r;
    HIPCHK(cs_vec_mul_launch(qb->d, qs_enc, bqkv, (int)C, st));
    D2D(bqkv + C, kb->d, C); D2D(bqkv + 2 * C, vb->d, C);
    ALLOC_BF(wo, (size_t)C * C) ALLOC_BF(w1, (size_t)F1 * C) ALLOC_BF(w2, (size_t)C * F)
    if (int r = pack(ow->d, (int)C, (int)C, (int)C, wo, l1->d)) return r;
    if (int r = pack(f1w->d, (int)F1, (int)C, (int)C, w1, nullptr, g2)) return r;
    if (int r = pack(f2w->d, (int)C, (int)F, (int)F, w2, l2->d)) return r;
    L.Wqkv = wqkv; L.bqkv = bqkv; L.Wo = wo; L.W1 = w1; L.W2 = w2;
    L.s_qkv = L.c_qkv = L.s_1 = L.c_1 = nullptr;
    L.panel_img = nullptr;
    if (h->panel) {
      ALLOC_BF(img, (h->panel_impl ? cs_panel4_image_bytes(1) : cs_panel8_image_bytes(1)) / sizeof(h16_t))
      if (h->panel_impl) HIPCHK(cs_panel4_pack_launch(ow->d, l1->d, f1w->d, n2w->d, f2w->d, l2->d, img, c.operand_dtype, st));
      else HIPCHK(cs_panel_pack_launch(ow->d, l1->d, f1w->d, n2w->d, f2w->d, l2->d, img, c.operand_dtype, st));
      L.panel_img = img;
    }
    if (fold) {
      ALLOC_F(sq, NQ) ALLOC_F(cq, NQ) ALLOC_F(s1v, (size_t)F) ALLOC_F(c1v, (size_t)F)
      if (NQ > 3 * C) { HIPCHK(hipMemset(sq + 3 * C, 0, (NQ - 3 * C) * sizeof(float))); HIPCHK(hipMemset(cq + 3 * C, 0, (NQ - 3 * C) * sizeof(float))); }
      const float* wsrc[3] = {qw->d, kw->d, vw->d};
      const float* bsrc[3] = {qb->d, kb->d, vb->d};
      for (int part = 0; part < 3; ++part)
        HIPCHK(cs_ln_fold_consts_launch(wqkv + (size_t)part * C * C, (int)C, wsrc[part], n1b->d, bsrc[part], (int)C, (int)C,
                                        sq + part * C, cq + part * C, c.operand_dtype, st));
      HIPCHK(cs_vec_mul_launch(cq, qs_enc, cq, (int)C, st));  // c = b + W beta of the Q rows carries the softmax scale too
      HIPCHK(cs_ln_fold_consts_launch(w1, (int)C, f1w->d, n2b->d, f1b->d, (int)F, (int)C, s1v, c1v, c.operand_dtype, st));
      L.s_qkv = sq; L.c_qkv = cq; L.s_1 = s1v; L.c_1 = c1v;
    }
  }
  { NEED(g, "backbone.layernorm.weight", C) NEED(b, "backbone.layernorm.bias", C) h->lnfg = g->d; h->lnfb = b->d; }
  if (h->panel || h->fold256) {
    ALLOC_F(on, (size_t)C) ALLOC_F(ze, (size_t)C)
    std::vector<float> hv((size_t)C, 1.0f);
    HIPCHK(hipMemcpy(on, hv.data(), (size_t)C * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(hipMemset(ze, 0, (size_t)C * sizeof(float)));
    h->ones = on; h->zeros = ze;
  }
  { NEED(pe, "pos_enc_fn.PE", 1, c.pe_h, c.pe_w, C) h->pe = pe->d; }
  h->dec.resize(c.dec_layers);
  { ALLOC_BF(wkv, (size_t)c.dec_layers * 2 * C * C) ALLOC_F(bkv, (size_t)c.dec_layers * 2 * C) h->Wkv_all = wkv; h->bkv_all = bkv; }
  for (int l = 0; l < c.dec_layers; ++l) {
    const std::string p = "ref_cross.attn.layers." + std::to_string(l) + ".";
    DecLayer& L = h->dec[l];
    std::memset(&L, 0, sizeof L);
    if (c.do_self_attn) {
      NEED(iw, p + "self_attn.in_proj_weight", 3 * C, C) NEED(ib, p + "self_attn.in_proj_bias", 3 * C)
      NEED(ow, p + "self_attn.out_proj.weight", C, C) NEED(ob, p + "self_attn.out_proj.bias", C)
      ALLOC_BF(w, (size_t)3 * C * C) ALLOC_F(bin, (size_t)3 * C)
      if (int r = pack(iw->d, (int)C, (int)C, (int)C, w, qs_dec)) return r;
      if (int r = pack(iw->d + C * C, (int)(2 * C), (int)C, (int)C, w + C * C)) return r;
      HIPCHK(cs_vec_mul_launch(ib->d, qs_dec, bin, (int)C, st));
      D2D(bin + C, ib->d + C, 2 * C);
      ALLOC_BF(wo, (size_t)C * C) if (int r = pack(ow->d, (int)C, (int)C, (int)C, wo)) return r;
      L.sa_Win = w; L.sa_bin = bin; L.sa_Wo = wo; L.sa_bo = ob->d;
    }
    NEED(iw, p + "multihead_attn.in_proj_weight", 3 * C, C) NEED(ib, p + "multihead_attn.in_proj_bias", 3 * C)
    NEED(ow, p + "multihead_attn.out_proj.weight", C, C) NEED(ob, p + "multihead_attn.out_proj.bias", C)
    ALLOC_BF(wq, (size_t)C * C) ALLOC_F(bq, (size_t)C)
    if (int r = pack(iw->d, (int)C, (int)C, (int)C, wq, qs_dec)) return r;
    HIPCHK(cs_vec_mul_launch(ib->d, qs_dec, bq, (int)C, st));
    // rows [C:3C) = [Wk;Wv] of this layer -> rows [l*2C, (l+1)*2C) of the fused KV projection (same memory for both layers)
    if (int r = pack(iw->d + C * C, (int)(2 * C), (int)C, (int)C, h->Wkv_all + (size_t)l * 2 * C * C)) return r;
    D2D(h->bkv_all + (size_t)l * 2 * C, ib->d + C, 2 * C);
    ALLOC_BF(wo, (size_t)C * C) if (int r = pack(ow->d, (int)C, (int)C, (int)C, wo)) return r;
    L.ca_Wq = wq; L.ca_bq = bq; L.ca_Wo = wo; L.ca_bo = ob->d;
    NEED(l1w, p + "linear1.weight", C, C) NEED(l1b, p + "linear1.bias", C) NEED(l2w, p + "linear2.weight", C, C) NEED(l2b, p + "linear2.bias", C)
    ALLOC_BF(w1, (size_t)C * C) if (int r = pack(l1w->d, (int)C, (int)C, (int)C, w1)) return r;
    ALLOC_BF(w2, (size_t)C * C) if (int r = pack(l2w->d, (int)C, (int)C, (int)C, w2)) return r;
    L.l1W = w1; L.l1b = l1b->d; L.l2W = w2; L.l2b = l2b->d;
    NEED(n1g, p + "norm1.weight", C) NEED(n1b, p + "norm1.bias", C) NEED(n2g, p + "norm2.weight", C) NEED(n2b, p + "norm2.bias", C)
    NEED(n3g, p + "norm3.weight", C) NEED(n3b, p + "norm3.bias", C)
    L.n1g = n1g->d; L.n1b = n1b->d; L.n2g = n2g->d; L.n2b = n2b->d; L.n3g = n3g->d; L.n3b = n3b->d;
  }
  {
    const int64_t PP = P * P;
    NEED(h0w, "ref_cross.head.0.weight", C, C) NEED(h0b, "ref_cross.head.0.bias", C)
    NEED(h2w, "ref_cross.head.2.weight", PP, C) NEED(h2b, "ref_cross.head.2.bias", PP)
    ALLOC_BF(w0, (size_t)C * C) if (int r = pack(h0w->d, (int)C, (int)C, (int)C, w0)) return r;
    ALLOC_BF(w2, (size_t)PP * C) if (int r = pack(h2w->d, (int)PP, (int)C, (int)C, w2)) return r;
    h->Wh0 = w0; h->bh0 = h0b->d; h->Wh2 = w2; h->bh2 = h2b->d;
  }
  if (!h->nonfinite) {
    HIPCHK(hipMalloc(&h->nonfinite, 16));
    HIPCHK(hipMemset(h->nonfinite, 0, 16));
  }
  HIPCHK(hipDeviceSynchronize());
  // fp32 copies of the big matrices are no longer needed
  for (auto& kv : h->w) {
    const std::string& n = kv.first;
    const bool big = kv.second.shape.size() >= 2 && n.find("weight") != std::string::npos && n.find("norm") == std::string::npos;
    if (big && kv.second.d) { hipFree(kv.second.d); kv.second.d = nullptr; }
  }
  h->finalized = true;
  return 0;
}

size_t cs_workspace_bytes(cs_handle h, int B, int N, int H, int W) {
  if (!h || B <= 0 || N <= 0 || H < h->cfg.patch || W < h->cfg.patch) return 0;
  return make_plan(h, B, N, N, H, W, nullptr).total;
}

// Do kernels queued on streams a and b run side by side?  The HIP runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues,
// and hardware queues onto the pipes of the compute micro-engine.  Two streams on one queue serialise outright; two queues on one
// pipe are dispatched one kernel at a time, so a grid that does not fit the chip at once holds back the other stream's kernel until
// its last round (a two-lane forward then runs at the one-lane time although tiny kernels on the two streams overlap).  Probe, both
// released by one event: on `a` a grid of 4 workgroups per CU that fit two to a CU (64 KiB of LDS each) and idle 60 us each, i.e.
// two rounds; on `b` one wave that idles 2 us.  `b` finishes within a few microseconds when the two dispatch side by side and
// after >= 60 us when it has to wait for a's second round.  Waits for both streams (set-up only, ~0.4 ms).
static int streams_overlap(hipStream_t a, hipStream_t b, bool* yes) {
  int dev = 0, cus = 0;
  HIPCHK(hipGetDevice(&dev));
  HIPCHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  hipEvent_t e0 = nullptr, eb = nullptr, ea = nullptr;
  HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&eb)); HIPCHK(hipEventCreate(&ea));
  int rc = 0;
  float best = 1e30f;
  for (int rep = 0; rep < 3 && !rc; ++rep) {  // the first round also absorbs the kernel's load
    hipError_t e = hipEventRecord(e0, a);
    if (e == hipSuccess) e = hipStreamWaitEvent(b, e0, 0);
    if (e == hipSuccess) e = cs_spin_launch(6000, 4 * cus, 64 * 1024, a);
    if (e == hipSuccess) e = cs_spin_launch(200, 1, 0, b);
    if (e == hipSuccess) e = hipEventRecord(eb, b);
    if (e == hipSuccess) e = hipEventRecord(ea, a);
    if (e == hipSuccess) e = hipEventSynchronize(ea);
    if (e == hipSuccess) e = hipEventSynchronize(eb);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, eb);
    if (e != hipSuccess) { rc = fail(CS_ERR_HIP, "stream overlap probe: %s", hipGetErrorString(e)); break; }
    if (rep > 0 && ms < best) best = ms;
  }
  hipEventDestroy(e0); hipEventDestroy(eb); hipEventDestroy(ea);
  if (!rc) *yes = best < 0.040f;
  return rc;
}

// mode 0: full forward (query + reference images); mode 1: query images + cached reference tokens (`ref_tokens`, fp16
// [B][N][Np][C]); mode 2: encode `B` images as references into `tokens_out` (fp16 [B][Np][C]), no decoder.
// the images of a forward as decoded uint8 (cs_forward_u8 and its siblings): host arrays of cs_u8_image
struct U8In { const cs_u8_image* query; const cs_u8_image* refs; const float* mean3; const float* std3; };

static int forward_body(cs_handle h, int mode, const float* query, const float* refs, const h16_t* ref_tokens, h16_t* tokens_out,
                        int B, int N, int H, int W, float* score_out, float* attn_out, int head_id, float* mean_out,
                        cs_stream stream, const U8In* u8 = nullptr) {
  if (!h) return fail(CS_ERR_BAD_ARG, "null handle");
  if (!h->finalized) return fail(CS_ERR_STATE, "cs_forward before cs_finalize");
  const cs_config& c = h->cfg;
  const bool have_q = u8 ? u8->query != nullptr : query != nullptr, have_r = u8 ? u8->refs != nullptr : refs != nullptr;
  if (mode == 0 && (!have_q || !have_r || !score_out)) return fail(CS_ERR_BAD_ARG, "null tensor (ref_cross_imgs is required when do_reference_cross)");
  if (mode == 1 && (!have_q || !ref_tokens || !score_out)) return fail(CS_ERR_BAD_ARG, "null tensor");
  if (mode == 2 && (!have_q || !tokens_out)) return fail(CS_ERR_BAD_ARG, "null tensor");
  if (u8 && (!u8->mean3 || !u8->std3 || !(u8->std3[0] > 0.f) || !(u8->std3[1] > 0.f) || !(u8->std3[2] > 0.f)))
    return fail(CS_ERR_BAD_ARG, "uint8 input: mean / std missing or std not positive");
  if (B <= 0 || (mode != 2 && N <= 0)) return fail(CS_ERR_BAD_ARG, "empty batch or no reference views");
  if (H < c.patch || W < c.patch) return fail(CS_ERR_BAD_ARG, "image smaller than one patch");
  if (attn_out && (head_id < 0 || head_id >= c.dec_heads)) return fail(CS_ERR_BAD_ARG, "need_attn_weights_head_id %d out of range", head_id);
  hipStream_t st = (hipStream_t)stream;
  const int C = c.hidden, P = c.patch;
  const int N_enc = mode == 0 ? N : 0;          // reference views that go through the encoder with their query
  const int N_plan = mode == 2 ? 0 : N;
  {
    const long long Np = (long long)(H / P) * (W / P);
    if ((long long)B * N * Np * 2 * C * c.dec_layers >= (1ll << 31)) return fail(CS_ERR_UNSUPPORTED, "batch too large for 32-bit offsets; split the batch");
    if (attn_out && Np > 65535) return fail(CS_ERR_UNSUPPORTED, "need_attn_weights with more than 65535 patches per image is not built");
  }
  const size_t need = make_plan(h, B, N_plan, N_enc, H, W, nullptr).total;
  reap_retired(h, false);
  if (need > h->ws_bytes) {
    // grow: the old workspace may still be in use by work queued earlier (on this or another stream), so it is retired behind an
    // event recorded on this call's stream (which is ordered after every earlier call) and freed by a later call once that event has completed -- no wait here
    if (h->ws) {
      cs_model::Retired r{h->ws, nullptr};
      HIPCHK(hipEventCreateWithFlags(&r.ev, hipEventDisableTiming));
      HIPCHK(hipEventRecord(r.ev, st));  // st already waits for the previous call's stream (forward_impl)
      h->retired.push_back(r);
    }
    h->ws = nullptr; h->ws_bytes = 0;
    HIPCHK(hipMalloc(&h->ws, need));
    h->ws_bytes = need;
  }
  Plan p = make_plan(h, B, N_plan, N_enc, H, W, h->ws);
  if (int r = ensure_tables(h, p.gh, p.gw, H == W, st)) return r;
  // the head launch's arrival counters (per-image mean in the same launch) start from zero; its finisher waves leave them at zero again, but the
  // workspace may have been carved differently by the previous call
  if (mean_out && mode != 2) HIPCHK(hipMemsetAsync(p.mean_cnt, 0, (size_t)B * sizeof(unsigned), st));
  // ---- one-pass input stage: per-image descriptors (filter tables of the image's resize geometry, crop corner) -> workspace ----
  int u8_span = 0;
  if (u8) {
    if (h->lnfold || !h->Wpatch_frag || !cs_patch_fused_supported(H, W, P, C))
      return fail(CS_ERR_UNSUPPORTED, "uint8 input needs the one-launch patch embedding (14-pixel patches, hidden a multiple of 384, LayerNorm fold off)");
    const int n_desc = p.I;  // B + B * N_enc
    cs_model::U8Slot& sl = h->u8_slot[h->u8_next];
    h->u8_next = (h->u8_next + 1) % cs_model::U8_SLOTS;
    if (sl.used) HIPCHK(hipEventSynchronize(sl.ev));  // the copy of four forwards ago: long done
    if (sl.cap < (size_t)n_desc) {
      if (sl.host) HIPCHK(hipHostFree(sl.host));
      sl.host = nullptr; sl.cap = 0;
      HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&sl.host), (size_t)n_desc * sizeof(CsU8Desc), hipHostMallocDefault));
      sl.cap = (size_t)n_desc;
    }
    if (!sl.ev) HIPCHK(hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming));
    // (the table cache holds 64 geometries and is dropped as a whole when a 65th arrives: if that happens while this call gathers its tables, the
    //  pointers gathered before the drop are gone -- gather again; a second drop means the call itself names more than 64 geometries)
    for (int attempt = 0;; ++attempt) {
      unsigned gen0 = 0, gen = 0;
      bool moved = false;
      u8_span = 0;
      for (int i = 0; i < n_desc; ++i) {
        const cs_u8_image& im = i < B ? u8->query[i] : u8->refs[i - B];
        CsU8Desc d{};
        if (im.h <= 0 || im.w <= 0 || im.rs_h <= 0 || im.rs_w <= 0 || im.row_bytes < 3 * im.w || im.crop_y < 0 || im.crop_x < 0 ||
            im.crop_y + H > im.rs_h || im.crop_x + W > im.rs_w)
          return fail(CS_ERR_BAD_ARG, "uint8 input %d: bad sizes (the %d x %d window must lie inside the resized image %d x %d)", i, H, W, im.rs_h, im.rs_w);
        int span = 0;
        HIPCHK(cs_preprocess_tables(im.h, im.w, im.rs_h, im.rs_w, im.crop_y, p.gh, P, &d.t, &span, &gen));
        if (i == 0) gen0 = gen;
        moved = moved || gen != gen0;
        d.data = im.data; d.row_bytes = im.row_bytes; d.crop_y = im.crop_y; d.crop_x = im.crop_x;
        u8_span = std::max(u8_span, span);
        sl.host[i] = d;
      }
      if (!moved) break;
      if (attempt) return fail(CS_ERR_UNSUPPORTED, "uint8 input: more than 64 distinct image geometries in one call");
    }
    if (cs_patch_u8_runs(W, u8_span) <= 0)
      return fail(CS_ERR_UNSUPPORTED, "uint8 input: a patch row reaches %d source rows, more than the one-pass form holds; use cs_op_preprocess_u8 + cs_forward", u8_span);
    HIPCHK(hipMemcpyAsync(p.u8desc, sl.host, (size_t)n_desc * sizeof(CsU8Desc), hipMemcpyHostToDevice, st));
    HIPCHK(hipEventRecord(sl.ev, st));
    sl.used = true;
  }
  const int enc_dh = C / c.enc_heads, dec_dh = C / c.dec_heads;
  const int F = ffn_hidden(c);
  const int KV = 2 * C * c.dec_layers;
  const int bf = c.operand_dtype;  // 16-bit operand type of every activation buffer and packed weight: 0 IEEE half, 1 bfloat16

  // ---- lanes: fork from the caller's stream, join back before returning (everything stays stream-ordered on `st`) ----
  const int NL = p.lanes;
  // The decoder runs as ONE group on the caller's stream: its kernels are small, and splitting the batch over streams only
  // makes them smaller (cfg-2, tools/dec_lanes.py: 1 group 8.77 ms, 2 groups 8.80, 3 groups 8.79, 4 groups 9.31).
  const int ND = 1;
  hipStream_t lst[CS_MAX_LANES] = {st, st, st, st};
  if (NL >= 2) {
    for (int l = 0; l < NL; ++l) {
      if (!h->lane_st[l]) {
        HIPCHK(hipStreamCreateWithFlags(&h->lane_st[l], hipStreamNonBlocking));
        // a lane that shares a hardware queue with the previous lane would run after it, not beside it: probe, and take another
        // stream until the two overlap (the rejected streams are released afterwards so that the runtime does not hand the same
        // queue back at once); one-time set-up cost of ~0.3 ms per probe, with a wait for the probe kernels
        if (l > 0) {
          std::vector<hipStream_t> rejected;
          for (int attempt = 0; attempt < 8; ++attempt) {
            bool ok = false;
            if (int r = streams_overlap(h->lane_st[l - 1], h->lane_st[l], &ok)) return r;
            if (g_debug_stream_log) fprintf(stderr, "[crossscore_hip] lane %d stream candidate %d: %s\n", l, attempt, ok ? "overlaps" : "serialises");
            if (ok) break;
            rejected.push_back(h->lane_st[l]);
            h->lane_st[l] = nullptr;
            HIPCHK(hipStreamCreateWithFlags(&h->lane_st[l], hipStreamNonBlocking));
          }
          for (hipStream_t r : rejected) hipStreamDestroy(r);
        }
      }
      if (!h->ev_join[l]) HIPCHK(hipEventCreateWithFlags(&h->ev_join[l], hipEventDisableTiming));
      if (!h->ev_stag[l]) HIPCHK(hipEventCreateWithFlags(&h->ev_stag[l], hipEventDisableTiming));
      lst[l] = h->lane_st[l];
    }
    if (!h->ev_fork) HIPCHK(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
    for (hipStream_t o : h->lane_st_old) hipStreamDestroy(o);  // the replacements exist now
    h->lane_st_old.clear();
  }
  auto fork = [&](int n) -> int {
    if (NL == 1) return 0;
    HIPCHK(hipEventRecord(h->ev_fork, st));
    for (int l = 0; l < n; ++l) HIPCHK(hipStreamWaitEvent(lst[l], h->ev_fork, 0));
    return 0;
  };
  auto join = [&](int n) -> int {
    if (NL == 1) return 0;
    for (int l = 0; l < n; ++l) {
      HIPCHK(hipEventRecord(h->ev_join[l], lst[l]));
      HIPCHK(hipStreamWaitEvent(st, h->ev_join[l], 0));
    }
    return 0;
  };

  // ================= encoder (Dinov2Model.forward, HF:451-477): image chunks, alternating lanes =================
  // `stage` selects what is enqueued: -1 = patch embedding, 0..L-1 = one encoder layer, L = final LayerNorm / split.  The host
  // enqueues the lanes' chunks stage by stage in turn, so both lanes have work from the first microsecond of the step (a whole
  // chunk is ~90 launches = 0.3 ms of enqueue time, during which the other lane would idle).
  auto enc_chunk = [&](Launcher& L, int slot, int i0, int ic, int stage) {
    hipStream_t s = L.st;
    float* x = p.x[slot]; h16_t* u = p.u[slot]; h16_t* r1 = p.r1[slot];
    const int Mc = ic * p.T;
    // patches are mean-centred per channel before the fp16 rounding; the patch GEMM adds mean * sum(W) back in fp32
    float* pmean = p.pmean[slot];
    if (stage == -1) {
    const bool fold = h->lnfold;
    h16_t* ob = p.ob[slot]; float* stats = p.stats[slot];
    L.begin(32, 0);
    L.misc(cs_cls_rows_launch(x, ic, p.T, C, h->cls, h->pos_tab, fold ? u : nullptr, fold ? stats : nullptr, h->ln_sp, bf, s), "cls");
    L.end();
    if (u8) {
      // the same launch fed from the decoded uint8 images (checked above: the one-launch form is available)
      L.begin(41, 2.0 * ic * p.Np * C * 3.0 * P * P, 3.0 * ic * H * W + 4.0 * ic * p.Np * C + 4.0 * p.Np * C);
      L.misc(cs_patch_fused_u8_launch(p.u8desc, B, N_enc, i0, ic, H, W, C, u8_span, u8->mean3, u8->std3, h->Wpatch_frag, h->bpatch, h->pos_tab, h->wsum,
                                      x, bf, s), "patch_u8");
      L.end();
    } else if (!fold && h->Wpatch_frag && cs_patch_fused_supported(H, W, P, C)) {
      // one launch: strip -> centred fp16 tile in LDS -> MFMA -> token rows (patch.hip).  Algorithmic bytes: the images once, the rows once
      L.begin(41, 2.0 * ic * p.Np * C * 3.0 * P * P, 12.0 * ic * H * W + 4.0 * ic * p.Np * C + 4.0 * p.Np * C);
      L.misc(cs_patch_fused_launch(query, refs, N_enc, i0, ic, H, W, C, h->Wpatch_frag, h->bpatch, h->pos_tab, h->wsum, x, bf, s), "patch");
      L.end();
    } else {
      L.begin(32, 0); L.misc(cs_im2col_launch(query, refs, N_enc, i0, r1, ic, H, W, P, h->Kp, pmean, bf, s), "im2col"); L.end();
      CsGemmParams g = gp(r1, h->Kp, h->Wpatch, h->Kp, ic * p.Np, C, h->Kp, h->bpatch, x, C);
      g.pos = h->pos_tab; g.Np = p.Np; g.pmean = pmean; g.wsum = h->wsum;
      if (fold) { g.out_f16 = u; g.stats_out = stats; g.stats_sp = h->ln_sp; }  // fp16 rows + LayerNorm partial sums for layer 0
      L.gemm(g, CS_EPI_PATCH_F32, 3.0 * P * P);
    }
    (void)ob;
    // tap: Dinov2Embeddings output (CLS row + patch rows + position rows), HF:97-116
    if (h->capture && !L.rc) L.rc = tap_copy(h, "embeddings", x, (size_t)i0 * p.T * C * 4, (size_t)Mc * C * 4, (size_t)p.I * p.T * C * 4, 0, {p.I, p.T, C}, s);
    }
    const bool fold = h->lnfold;
    h16_t* ob = p.ob[slot]; float* stats = p.stats[slot];
    // tap: the residual stream behind encoder layer l (Dinov2Layer output, HF:361-380)
    auto tap_layer = [&](int l) {
      if (h->capture && !L.rc)
        L.rc = tap_copy(h, "enc_layer_" + std::to_string(l), x, (size_t)i0 * p.T * C * 4, (size_t)Mc * C * 4, (size_t)p.I * p.T * C * 4, 0, {p.I, p.T, C}, s);
    };
    // The LayerNorm-folded epilogues (statistics layouts: consumer ln_sp == 1, producer stats_sp == N / 64) exist only in the 256-tile kernel:
    // a chunk takes that branch only if the kernel accepts ALL THREE of its folded shapes -- the same answer for every layer of the chunk, since a
    // layer's producer feeds the next layer's consumer.  Whatever makes it decline (cs_debug_gemm256_enable(0), cs_debug_gemm256_kmin above C, a
    // chunk below 256 rows or beyond the kernel's 32-bit byte offsets) sends the chunk to the plain path below, which is correct for a fold256
    // handle (LayerNorm launches with ones / zeros, gamma in the packed weights, beta in the c vectors).
    bool fold256_chunk = false;
    if (h->fold256 && Mc >= 256) {
      const EncLayer& E0 = h->enc[0];
      CsGemmParams cq = gp(u, C, E0.Wqkv, C, Mc, h->qkv_n, C, E0.c_qkv, r1, h->qkv_n);
      cq.col_s = E0.s_qkv; cq.ln_part = p.lnstat[slot]; cq.ln_sp = 1;
      CsGemmParams c1 = gp(u, C, E0.W1, C, Mc, F, C, E0.c_1, r1, F);
      c1.col_s = E0.s_1; c1.ln_part = p.lnstat[slot]; c1.ln_sp = 1;
      CsGemmParams po = gp(p.ob[slot], C, E0.Wo, C, Mc, C, C, E0.bo, x, C);
      po.resid = x; po.ldr = C; po.out_f16 = u; po.stats_out = p.stats[slot]; po.stats_sp = C / 64;
      CsGemmParams p2 = gp(r1, F, E0.W2, F, Mc, C, F, E0.b2, x, C);
      p2.resid = x; p2.ldr = C; p2.out_f16 = u; p2.stats_out = p.stats[slot]; p2.stats_sp = C / 64;
      fold256_chunk = cs_gemm256_supported(&cq, CS_EPI_LN_F16) && cs_gemm256_supported(&c1, CS_EPI_LN_GELU_F16) &&
                      cs_gemm256_supported(&po, CS_EPI_RESID_F32_LN) && cs_gemm256_supported(&p2, CS_EPI_RESID_F32_LN);
    }
    for (int l = 0; l < c.enc_layers; ++l) {
      if (l != stage) continue;
      const EncLayer& E = h->enc[l];
      const bool last = l == c.enc_layers - 1;
      CsAttnParams a{};
      a.bf16 = bf;
      const int NQ = h->qkv_n;  // row stride of the packed QKV rows (3C, or padded to whole 256-column GEMM tiles)
      a.Q = r1; a.K = r1 + C; a.V = r1 + 2 * C;
      a.ldq = a.ldk = a.ldv = NQ; a.ldo = C;
      a.q_bs = a.k_bs = a.v_bs = (long long)p.T * NQ; a.o_bs = (long long)p.T * C;
      a.Lq = a.Lk = p.T; a.heads = c.enc_heads; a.scale_log2e = 1.0f;  // folded into the Q rows of Wqkv (cs_finalize)
      a.lse = nullptr;
      if (h->panel) {
        // u = fp16 normalised rows (norm1 without gamma/beta: folded into Wqkv / c_qkv), written by the previous layer's panel
        // kernel; layer 0 gets it from the LayerNorm kernel
        if (l == 0) { L.begin(32, 0); L.misc(cs_layernorm_launch(x, Mc, C, h->ones, h->zeros, 1e-6f, nullptr, u, bf, s), "ln1"); L.end(); }
        L.gemm(gp(u, C, E.Wqkv, C, Mc, NQ, C, E.c_qkv, r1, NQ), CS_EPI_BIAS_F16);
        a.O = u;
        L.attn(a, enc_dh, ic);
        CsPanelParams q{};
        q.x = x; q.attn_o = u; q.img = E.panel_img; q.bo = E.bo; q.b1 = E.c_1; q.b2 = E.b2; q.u_out = last ? nullptr : u;
        q.M = Mc; q.eps = 1e-6f;
        L.panel(q);
        tap_layer(l);
        continue;
      }
      if (fold256_chunk) {
        // Wide backbones (r5): LayerNorm folded into the 256-tile GEMM's epilogues.  The residual epilogues (out-projection, fc2) also write
        // u = 16-bit(x) and per-row partial sums, a row-statistics kernel (one thread per row) turns them into (mean, rstd), and the consuming
        // projection (QKV, fc1) applies rstd * (acc - mean * s[n]) + c[n].  No LayerNorm pass over the fp32 stream except layer 0's norm1
        // (its rows come from the patch embedding).  Chunks below 256 rows (tiny images) take the plain path below.
        const int sp = C / 64, Mpad = (Mc + 255) / 256 * 256;
        float* part = p.stats[slot];
        float* stat = p.lnstat[slot];
        auto row_stats = [&]() { L.begin(32, 0); L.misc(cs_ln_finalize_launch(part, Mc, Mpad, sp, C, 1e-6f, stat, s), "ln_stats"); L.end(); };
        auto consumer = [&](const h16_t* W, const float* cvec, const float* svec, h16_t* out, int n, int epi) {
          CsGemmParams g = gp(u, C, W, C, Mc, n, C, cvec, out, n);
          g.col_s = svec; g.ln_part = stat; g.ln_sp = 1; g.ln_eps = 1e-6f;
          L.gemm(g, epi);
        };
        if (l == 0) {
          L.begin(32, 0); L.misc(cs_layernorm_launch(x, Mc, C, h->ones, h->zeros, 1e-6f, nullptr, u, bf, s), "ln1"); L.end();
          L.gemm(gp(u, C, E.Wqkv, C, Mc, NQ, C, E.c_qkv, r1, NQ), CS_EPI_BIAS_F16);  // (gamma is in the packed weights, beta in c)
        } else {
          consumer(E.Wqkv, E.c_qkv, E.s_qkv, r1, NQ, CS_EPI_LN_F16);
        }
        a.O = ob;
        L.attn(a, enc_dh, ic);
        {
          CsGemmParams g = gp(ob, C, E.Wo, C, Mc, C, C, E.bo, x, C);
          g.resid = x; g.ldr = C; g.out_f16 = u; g.stats_out = part; g.stats_sp = sp;
          L.gemm(g, CS_EPI_RESID_F32_LN);
        }
        row_stats();
        consumer(E.W1, E.c_1, E.s_1, r1, F, CS_EPI_LN_GELU_F16);
        {
          CsGemmParams g = gp(r1, F, E.W2, F, Mc, C, F, E.b2, x, C);
          g.resid = x; g.ldr = C;
          if (!last) { g.out_f16 = u; g.stats_out = part; g.stats_sp = sp; }
          L.gemm(g, last ? CS_EPI_RESID_F32 : CS_EPI_RESID_F32_LN);  // the final LayerNorm reads the fp32 stream
        }
        if (!last) row_stats();
        tap_layer(l);
        continue;
      }
      if (fold) {
        // u holds fp16(x) and `stats` the per-row partial sums, both written by the epilogue that produced x: LayerNorm is
        // applied inside the consuming projection's epilogue (CS_EPI_LN_*), there is no separate LN pass over x
        {
          CsGemmParams g = gp(u, C, E.Wqkv, C, Mc, NQ, C, E.c_qkv, r1, NQ);
          g.col_s = E.s_qkv; g.ln_part = stats; g.ln_sp = h->ln_sp; g.ln_eps = 1e-6f;
          L.gemm(g, CS_EPI_LN_F16);
        }
        a.O = ob;
        L.attn(a, enc_dh, ic);
        {
          CsGemmParams g = gp(ob, C, E.Wo, C, Mc, C, C, E.bo, x, C);
          g.resid = x; g.ldr = C; g.out_f16 = u; g.stats_out = stats; g.stats_sp = h->ln_sp;
          L.gemm(g, CS_EPI_RESID_F32_LN);
        }
        {
          CsGemmParams g = gp(u, C, E.W1, C, Mc, F, C, E.c_1, r1, F);
          g.col_s = E.s_1; g.ln_part = stats; g.ln_sp = h->ln_sp; g.ln_eps = 1e-6f;
          L.gemm(g, CS_EPI_LN_GELU_F16);
        }
        {
          CsGemmParams g = gp(r1, F, E.W2, F, Mc, C, F, E.b2, x, C);
          g.resid = x; g.ldr = C;
          if (!last) { g.out_f16 = u; g.stats_out = stats; g.stats_sp = h->ln_sp; }
          L.gemm(g, last ? CS_EPI_RESID_F32 : CS_EPI_RESID_F32_LN);  // the final LayerNorm reads the fp32 stream
        }
        tap_layer(l);
        continue;
      }
      // (a fold256 handle's chunk of fewer than 256 rows lands here with gamma folded into its packed weights and beta into the c vectors:
      //  LayerNorm without gamma / beta then, as in the panel path)
      const bool pf = h->fold256;
      L.begin(32, 0); L.misc(cs_layernorm_launch(x, Mc, C, pf ? h->ones : E.ln1g, pf ? h->zeros : E.ln1b, 1e-6f, nullptr, u, bf, s), "ln1"); L.end();
      L.gemm(gp(u, C, E.Wqkv, C, Mc, NQ, C, pf ? E.c_qkv : E.bqkv, r1, NQ), CS_EPI_BIAS_F16);
      a.O = u;
      L.attn(a, enc_dh, ic);
      {
        CsGemmParams g = gp(u, C, E.Wo, C, Mc, C, C, E.bo, x, C);
        g.resid = x; g.ldr = C;
        L.gemm(g, CS_EPI_RESID_F32);
      }
      L.begin(32, 0); L.misc(cs_layernorm_launch(x, Mc, C, pf ? h->ones : E.ln2g, pf ? h->zeros : E.ln2b, 1e-6f, nullptr, u, bf, s), "ln2"); L.end();
      if (c.swiglu) {
        // Dinov2SwiGLUFFN (HF:300-316): [x1 | x2] = LN2(x) Win^T + b (2F columns), hidden = silu(x1) * x2 in place over the x1 half, x += hidden Wout'^T + b'
        L.gemm(gp(u, C, E.W1, C, Mc, 2 * F, C, E.b1, r1, 2 * F), CS_EPI_BIAS_F16);
        L.begin(32, 0, 6.0 * Mc * F); L.misc(cs_silu_mul_launch(r1, Mc, F, 2 * F, bf, s), "silu_mul"); L.end();
        CsGemmParams g = gp(r1, 2 * F, E.W2, F, Mc, C, F, E.b2, x, C);
        g.resid = x; g.ldr = C;
        L.gemm(g, CS_EPI_RESID_F32);
        tap_layer(l);
        continue;
      }
      L.gemm(gp(u, C, E.W1, C, Mc, F, C, pf ? E.c_1 : E.b1, r1, F), CS_EPI_BIAS_GELU_F16);
      {
        CsGemmParams g = gp(r1, F, E.W2, F, Mc, C, F, E.b2, x, C);
        g.resid = x; g.ldr = C;
        L.gemm(g, CS_EPI_RESID_F32);
      }
      tap_layer(l);
    }
    if (stage != c.enc_layers) return;
    L.begin(32, 0);
    L.misc(cs_final_ln_split_launch(x, ic, i0, p.Np, C, mode == 2 ? -1 : N_enc, h->lnfg, h->lnfb, 1e-6f, h->pe_tab, p.xq, p.q_bf,
                                    mode == 2 ? tokens_out : p.mem_bf, bf, s), "final_ln");
    L.end();
  };

  // ================= decoder (transformer.py:213-268, post-norm layers :157-173) + head, batch items [b0, b0+nb) =================
  auto dec_group = [&](Launcher& L, int b0, int nb) {
    hipStream_t s = L.st;
    const int M = nb * p.Np, Mk = nb * N * p.Np;
    const size_t ro = (size_t)b0 * p.Np;          // first query row of the group
    const size_t ko = (size_t)b0 * N * p.Np;      // first memory row of the group
    float* xq = p.xq + ro * C; float* y = p.y + ro * C; h16_t* q_bf = p.q_bf + ro * C;
    const h16_t* mem = (mode == 1 ? ref_tokens : p.mem_bf) + ko * C; h16_t* kv = p.kv + ko * KV;
    h16_t* dqkv = p.dqkv + ro * 3 * C; h16_t* dq = p.dq + ro * C; h16_t* dob = p.dob + ro * C; h16_t* dhid = p.dhid + ro * C;
    float* lse = p.lse + (size_t)b0 * c.dec_heads * p.Np;
    // K/V projection of the memory (both layers at once).  Nothing before the first cross-attention depends on it, so with lanes
    // it runs on lane stream 1 next to layer 0's self-attention branch (the decoder phase has one small kernel in flight otherwise).
    // each sub-block closes with LN(x + Linear(.)): one launch where the row-complete kernel is built (C = 384), else GEMM + LayerNorm
    const bool fused_ln = cs_rowln_supported(C) != 0 && !g_rowln_off;
    // ... and where it is, the sub-block's NEXT linear rides in the same launch when it is C wide (second stage of rowln.hip): the
    // cross-attention's Q projection behind norm1, linear1 + ReLU behind norm2, the head's first linear + LeakyReLU behind the last
    // norm3 -- 16 launches per decoder + head instead of 21, at the same kernel time (34.5 vs 33.3 us per pair at 10 952 rows).  The next
    // layer's packed QKV projection (3 C wide) stays a GEMM of its own: in this kernel's 64-row shape it costs 62.5 us against 45.7.
    // The flags say which projection is already there.
    const bool fuse_next = fused_ln && !g_rowln_no_next;
    using NL_t = NextLinear;
    auto next_of = [&](const h16_t* W, const float* b, h16_t* out, int n, int act) { NL_t x; x.W = W; x.b = b; x.out = out; x.n = n; x.act = act; return x; };
    bool have_q = false, have_hid = false, have_head0 = false;
    const bool kv_side = NL >= 2 && c.do_self_attn && !h->prof;
    if (kv_side) {
      if (!h->ev_kv0) { if (hipEventCreateWithFlags(&h->ev_kv0, hipEventDisableTiming) != hipSuccess) L.rc = CS_ERR_HIP; }
      if (!h->ev_kv1) { if (hipEventCreateWithFlags(&h->ev_kv1, hipEventDisableTiming) != hipSuccess) L.rc = CS_ERR_HIP; }
      if (!L.rc && (hipEventRecord(h->ev_kv0, s) != hipSuccess || hipStreamWaitEvent(lst[1], h->ev_kv0, 0) != hipSuccess)) L.rc = CS_ERR_HIP;
      Launcher LK{h, lst[1]};
      LK.gemm(gp(mem, C, h->Wkv_all, C, Mk, KV, C, h->bkv_all, kv, KV), CS_EPI_BIAS_F16);
      if (LK.rc) L.rc = LK.rc;
      if (!L.rc && hipEventRecord(h->ev_kv1, lst[1]) != hipSuccess) L.rc = CS_ERR_HIP;
    } else {
      L.gemm(gp(mem, C, h->Wkv_all, C, Mk, KV, C, h->bkv_all, kv, KV), CS_EPI_BIAS_F16);
    }
    for (int l = 0; l < c.dec_layers; ++l) {
      const DecLayer& D = h->dec[l];
      const bool last_l = l == c.dec_layers - 1;
      if (c.do_self_attn) {
        L.gemm(gp(q_bf, C, D.sa_Win, C, M, 3 * C, C, D.sa_bin, dqkv, 3 * C), CS_EPI_BIAS_F16);
        CsAttnParams a{};
        a.bf16 = bf;
        a.Q = dqkv; a.K = dqkv + C; a.V = dqkv + 2 * C; a.O = dob;
        a.ldq = a.ldk = a.ldv = 3 * C; a.ldo = C;
        a.q_bs = a.k_bs = a.v_bs = (long long)p.Np * 3 * C; a.o_bs = (long long)p.Np * C;
        a.Lq = a.Lk = p.Np; a.heads = c.dec_heads; a.scale_log2e = 1.0f;  // folded into the Q projection (cs_finalize)
        L.attn(a, dec_dh, nb);
        if (fused_ln) {
          if (fuse_next) {
            L.rowln(dob, D.sa_Wo, D.sa_bo, c.do_short_cut ? xq : nullptr, D.n1g, D.n1b, 1e-5f, xq, nullptr, M, next_of(D.ca_Wq, D.ca_bq, dq, (int)C, 0));
            have_q = true;
          } else {
            L.rowln(dob, D.sa_Wo, D.sa_bo, c.do_short_cut ? xq : nullptr, D.n1g, D.n1b, 1e-5f, xq, q_bf, M);
          }
        } else {
          CsGemmParams g = gp(dob, C, D.sa_Wo, C, M, C, C, D.sa_bo, y, C);
          g.resid = c.do_short_cut ? xq : nullptr; g.ldr = C;
          L.gemm(g, CS_EPI_RESID_F32);
          L.begin(32, 0); L.misc(cs_layernorm_launch(y, M, C, D.n1g, D.n1b, 1e-5f, xq, q_bf, bf, s), "norm1"); L.end();
        }
      }
      if (kv_side && l == 0 && !L.rc && hipStreamWaitEvent(s, h->ev_kv1, 0) != hipSuccess) L.rc = CS_ERR_HIP;
      if (!have_q) L.gemm(gp(q_bf, C, D.ca_Wq, C, M, C, C, D.ca_bq, dq, C), CS_EPI_BIAS_F16);
      have_q = false;
      CsAttnParams a{};
      a.bf16 = bf;
      a.Q = dq; a.K = kv + (size_t)l * 2 * C; a.V = kv + (size_t)l * 2 * C + C; a.O = dob;
      a.ldq = C; a.ldk = a.ldv = KV; a.ldo = C;
      a.q_bs = (long long)p.Np * C; a.k_bs = a.v_bs = (long long)N * p.Np * KV; a.o_bs = (long long)p.Np * C;
      a.Lq = p.Np; a.Lk = N * p.Np; a.heads = c.dec_heads; a.scale_log2e = 1.0f;  // folded into the Q projection (cs_finalize)
      const bool want_w = attn_out && l == c.dec_layers - 1;  // only the last layer's weights are returned (transformer.py:266-268)
      a.lse = want_w ? lse : nullptr;
      L.attn(a, dec_dh, nb);
      if (want_w && !L.rc) {
        L.begin(32, 0);
        L.misc(cs_attn_weights_launch(&a, dec_dh, nb, head_id, attn_out + (size_t)b0 * p.Np * N * p.Np, s), "attn_weights");
        L.end();
      }
      if (fused_ln) {
        if (fuse_next) {
          L.rowln(dob, D.ca_Wo, D.ca_bo, c.do_short_cut ? xq : nullptr, D.n2g, D.n2b, 1e-5f, xq, nullptr, M, next_of(D.l1W, D.l1b, dhid, (int)C, 1));
          have_hid = true;
        } else {
          L.rowln(dob, D.ca_Wo, D.ca_bo, c.do_short_cut ? xq : nullptr, D.n2g, D.n2b, 1e-5f, xq, q_bf, M);
        }
      } else {
        CsGemmParams g = gp(dob, C, D.ca_Wo, C, M, C, C, D.ca_bo, y, C);
        g.resid = c.do_short_cut ? xq : nullptr; g.ldr = C;
        L.gemm(g, CS_EPI_RESID_F32);
        L.begin(32, 0); L.misc(cs_layernorm_launch(y, M, C, D.n2g, D.n2b, 1e-5f, xq, q_bf, bf, s), "norm2"); L.end();
      }
      if (!have_hid) L.gemm(gp(q_bf, C, D.l1W, C, M, C, C, D.l1b, dhid, C), CS_EPI_BIAS_RELU_F16);
      have_hid = false;
      if (fused_ln) {
        if (fuse_next && last_l) {
          // (the head's hidden rows replace linear1's in dhid: a workgroup writes exactly the 64 rows it staged into LDS at its start)
          L.rowln(dhid, D.l2W, D.l2b, xq, D.n3g, D.n3b, 1e-5f, xq, nullptr, M, next_of(h->Wh0, h->bh0, dhid, (int)C, 2));
          have_head0 = true;
        } else if (fuse_next && !c.do_self_attn) {
          const DecLayer& Dn = h->dec[l + 1];
          L.rowln(dhid, D.l2W, D.l2b, xq, D.n3g, D.n3b, 1e-5f, xq, nullptr, M, next_of(Dn.ca_Wq, Dn.ca_bq, dq, (int)C, 0));
          have_q = true;
        } else {
          L.rowln(dhid, D.l2W, D.l2b, xq, D.n3g, D.n3b, 1e-5f, xq, q_bf, M);
        }
      } else {
        CsGemmParams g = gp(dhid, C, D.l2W, C, M, C, C, D.l2b, y, C);
        g.resid = xq; g.ldr = C;
        L.gemm(g, CS_EPI_RESID_F32);
        L.begin(32, 0); L.misc(cs_layernorm_launch(y, M, C, D.n3g, D.n3b, 1e-5f, xq, q_bf, bf, s), "norm3"); L.end();
      }
      // tap: decoder layer l's output (transformer.py:157-173)
      if (h->capture && !L.rc)
        L.rc = tap_copy(h, "dec" + std::to_string(l) + "_out", xq, ro * C * 4, (size_t)M * C * 4, (size_t)B * p.Np * C * 4, 0, {B, p.Np, C}, s);
    }
    // head + RegressionLayer + jigsaw (cross_reference.py:45-50,82-87)
    if (!have_head0) L.gemm(gp(q_bf, C, h->Wh0, C, M, C, C, h->bh0, dhid, C), CS_EPI_BIAS_LEAKY_F16);
    {
      CsGemmParams g = gp(dhid, C, h->Wh2, C, M, P * P, C, h->bh2, score_out + (size_t)b0 * p.gh * P * p.gw * P, 4);
      g.Np = p.Np; g.gw = p.gw; g.P = P; g.act = c.act; g.powp = c.pow_p;
      if (mean_out) { g.mean_part = p.mean_part + ro * 4 * (size_t)cs_gemm_column_tiles((int)(P * P)); g.mean_cnt = p.mean_cnt + b0; g.mean_out = mean_out + b0; }
      L.gemm(g, CS_EPI_HEAD_SCORE);
    }
    if (h->capture && !L.rc) {
      // tap: the head's second linear before the activation (cross_reference.py:45-50).  The score epilogue applies the activation in
      // registers, so the pre-activation is produced by one more launch of the same GEMM with a plain fp32 store (capture mode only).
      void* pre = nullptr;
      L.rc = tap_buffer(h, "head_pre_activation", (size_t)B * p.Np * P * P * 4, 0, {B, p.Np, (int64_t)P * P}, &pre);
      if (!L.rc) L.gemm(gp(dhid, C, h->Wh2, C, M, P * P, C, h->bh2, static_cast<float*>(pre) + ro * P * P, P * P), CS_EPI_RESID_F32);
    }
  };

  Launcher LL[CS_MAX_LANES] = {Launcher{h, lst[0]}, Launcher{h, lst[1]}, Launcher{h, lst[2]}, Launcher{h, lst[3]}};
  auto lanes_rc = [&]() { for (int l = 0; l < CS_MAX_LANES; ++l) if (LL[l].rc) return LL[l].rc; return 0; };
  // encoder lanes share the GPU: their GEMMs oversubscribe the CUs so that blocks are short and slots change hands often
  // (cfg-2, same box: 9.03 -> 8.84 ms with 3..16 blocks per CU; alone on the GPU two per CU is best: 9.27 vs 9.38..9.73 ms)
  if (NL >= 2) for (int l = 0; l < CS_MAX_LANES; ++l) LL[l].bpc = 4;
  const int per_item = 1 + N_enc;
  // Decoding each chunk's items on its lane right after encoding them (no global join) was measured SLOWER (833 vs 875
  // query-images/s on cfg-2): it doubles the number of small decoder launches and the host enqueue rate becomes the
  // limit.  The decoder therefore runs after a join, as one group on the caller's stream.
  (void)per_item;
  if (int r = fork(NL)) return r;
  {
    // chunk sizes: the short remainder (if any) goes FIRST so that it overlaps the long chunks instead of trailing them;
    // chunk k runs on lane k % NL, and the chunks of one round (one per lane) are enqueued stage by stage in turn
    std::vector<std::pair<int, int>> chunks;  // (first image, images)
    int i0 = 0;
    const int rem = p.I % p.Ic;
    if (rem) { chunks.push_back({0, rem}); i0 = rem; }
    for (; i0 < p.I; i0 += p.Ic) chunks.push_back({i0, p.Ic});
    // The lanes run the same kernel sequence: started together they stay in lockstep (panel beside panel, attention beside
    // attention) and overlap nothing useful -- which is what happens whenever their streams sit on separate hardware queues.  Lane l
    // therefore starts its first chunk when lane l-1 has finished its patch embedding (about half a layer's time): from then on one
    // lane's QKV + attention runs beside the other's panel kernel.
    for (size_t base = 0; base < chunks.size(); base += NL)
      for (int stage = -1; stage <= c.enc_layers; ++stage)
        for (int l = 0; l < NL && base + l < chunks.size(); ++l) {
          const bool stagger = base == 0 && stage == -1 && NL >= 2;
          if (stagger && l > 0 && hipStreamWaitEvent(lst[l], h->ev_stag[l - 1], 0) != hipSuccess) return fail(CS_ERR_HIP, "lane stagger wait failed");
          enc_chunk(LL[l], l, chunks[base + l].first, chunks[base + l].second, stage);
          if (stagger && hipEventRecord(h->ev_stag[l], lst[l]) != hipSuccess) return fail(CS_ERR_HIP, "lane stagger record failed");
        }
  }
  if (int r = join(NL)) return r;
  if (int r = lanes_rc()) return r;
  if (h->capture) {
    // taps: the decoder's inputs = final LayerNorm of the patch tokens + multi-view PE (core.py:141-153,93-98): query rows fp32, reference rows 16 bit
    const int dt16 = bf ? 2 : 1;
    if (mode == 2) {
      if (int r = tap_copy(h, "featmap_ref", tokens_out, 0, (size_t)B * p.Np * C * 2, (size_t)B * p.Np * C * 2, dt16, {B, p.Np, C}, st)) return r;
    } else {
      if (int r = tap_copy(h, "featmap_query", p.xq, 0, (size_t)B * p.Np * C * 4, (size_t)B * p.Np * C * 4, 0, {B, p.Np, C}, st)) return r;
      const h16_t* mem = mode == 1 ? ref_tokens : p.mem_bf;
      if (int r = tap_copy(h, "featmap_ref", mem, 0, (size_t)B * N * p.Np * C * 2, (size_t)B * N * p.Np * C * 2, dt16, {B, (int64_t)N * p.Np, C}, st)) return r;
    }
  }
  if (mode == 2) return 0;
  {  // every image's tokens are in place (join above) before the decoder starts
    (void)ND;
    Launcher LD{h, st};
    dec_group(LD, 0, B);
    if (LD.rc) return LD.rc;
  }
  Launcher L{h, st};
  if (!c.skip_finite_check) L.misc(cs_score_check_launch(score_out, (size_t)B * p.gh * P * p.gw * P, h->nonfinite, st), "score_check");
  return L.rc;
}

static int forward_impl(cs_handle h, int mode, const float* query, const float* refs, const h16_t* ref_tokens, h16_t* tokens_out,
                        int B, int N, int H, int W, float* score_out, float* attn_out, int head_id, float* mean_out,
                        cs_stream stream, const U8In* u8 = nullptr) {
  // The workspace is shared by every call on this handle: a call on a different stream than the previous one first waits for that
  // one to finish (calls on one stream are ordered anyway).
  if (!h) return fail(CS_ERR_BAD_ARG, "null handle");
  hipStream_t st = (hipStream_t)stream;
  if (h->ev_done && h->last_stream != st) HIPCHK(hipStreamWaitEvent(st, h->ev_done, 0));
  h->census.clear();
  const auto t0 = std::chrono::steady_clock::now();
  if (u8) cs_preprocess_tables_hold(1);  // (the filter tables its descriptors point at stay put until everything is queued: preprocess.hip)
  const int rc = forward_body(h, mode, query, refs, ref_tokens, tokens_out, B, N, H, W, score_out, attn_out, head_id, mean_out, stream, u8);
  if (u8) cs_preprocess_tables_hold(0);
  h->host_enqueue_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  if (!h->ev_done) HIPCHK(hipEventCreateWithFlags(&h->ev_done, hipEventDisableTiming));
  HIPCHK(hipEventRecord(h->ev_done, st));
  h->last_stream = st;
  return rc;
}

int cs_forward(cs_handle h, const float* query, const float* refs, int B, int N, int H, int W, float* score_out, float* attn_out,
               int head_id, float* mean_out, cs_stream stream) {
  return forward_impl(h, 0, query, refs, nullptr, nullptr, B, N, H, W, score_out, attn_out, head_id, mean_out, stream);
}

int cs_encode_references(cs_handle h, const float* imgs, int R, int H, int W, uint16_t* tokens_out, cs_stream stream) {
  return forward_impl(h, 2, imgs, nullptr, nullptr, tokens_out, R, 0, H, W, nullptr, nullptr, 0, nullptr, stream);
}

int cs_forward_cached(cs_handle h, const float* query, const uint16_t* ref_tokens, int B, int N, int H, int W, float* score_out,
                      float* attn_out, int head_id, float* mean_out, cs_stream stream) {
  return forward_impl(h, 1, query, nullptr, ref_tokens, nullptr, B, N, H, W, score_out, attn_out, head_id, mean_out, stream);
}

// The three forwards fed from decoded uint8 images (SURVEY.md 8f-4 as worded: uint8 in, tokens out; include/crossscore_hip.h)
int cs_forward_u8(cs_handle h, const cs_u8_image* query, const cs_u8_image* refs, int B, int N, int H, int W, const float* mean3, const float* std3,
                  float* score_out, float* attn_out, int head_id, float* mean_out, cs_stream stream) {
  const U8In u{query, refs, mean3, std3};
  return forward_impl(h, 0, nullptr, nullptr, nullptr, nullptr, B, N, H, W, score_out, attn_out, head_id, mean_out, stream, &u);
}

int cs_encode_references_u8(cs_handle h, const cs_u8_image* imgs, int R, int H, int W, const float* mean3, const float* std3, uint16_t* tokens_out,
                            cs_stream stream) {
  const U8In u{imgs, nullptr, mean3, std3};
  return forward_impl(h, 2, nullptr, nullptr, nullptr, tokens_out, R, 0, H, W, nullptr, nullptr, 0, nullptr, stream, &u);
}

int cs_forward_cached_u8(cs_handle h, const cs_u8_image* query, const uint16_t* ref_tokens, int B, int N, int H, int W, const float* mean3,
                         const float* std3, float* score_out, float* attn_out, int head_id, float* mean_out, cs_stream stream) {
  const U8In u{query, nullptr, mean3, std3};
  return forward_impl(h, 1, nullptr, nullptr, ref_tokens, nullptr, B, N, H, W, score_out, attn_out, head_id, mean_out, stream, &u);
}

int cs_u8_input_supported(cs_handle h, const cs_u8_image* im, int H, int W) {
  if (!h || !h->finalized || !im) return 0;
  const cs_config& c = h->cfg;
  if (h->lnfold || !h->Wpatch_frag || !cs_patch_fused_supported(H, W, c.patch, c.hidden)) return 0;
  if (im->h <= 0 || im->w <= 0 || im->rs_h <= 0 || im->rs_w <= 0 || im->crop_y < 0 || im->crop_x < 0 || im->crop_y + H > im->rs_h || im->crop_x + W > im->rs_w) return 0;
  CsU8Tables t{};
  int span = 0;
  if (cs_preprocess_tables(im->h, im->w, im->rs_h, im->rs_w, im->crop_y, H / c.patch, c.patch, &t, &span, nullptr) != hipSuccess) return 0;
  return cs_patch_u8_runs(W, span) > 0 ? 1 : 0;
}

// What the last forward-class call on this handle launched, and what it cost the host: `launches` = kernel launches (memcpy taps of capture
// mode not counted), `host_ms` = wall time of the call on the calling thread (everything is enqueued, nothing waited for), `names` (optional,
// `names_bytes` long) = "kernel=count kernel=count ..." by kernel: gemm256 / gemm128 (cs_gemm256_kernel / cs_gemm_kernel), attn<dh>, panel,
// rowln, patch, im2col, ln1 / ln2 / ln (layernorm_kernel), cls, final_ln, score_check, ...
int cs_forward_stats(cs_handle h, int* launches, double* host_ms, char* names, size_t names_bytes) {
  if (!h) return fail(CS_ERR_BAD_ARG, "null handle");
  int n = 0;
  std::string txt;
  for (const auto& kv : h->census) { n += kv.second; txt += (txt.empty() ? "" : " ") + kv.first + "=" + std::to_string(kv.second); }
  if (launches) *launches = n;
  if (host_ms) *host_ms = h->host_enqueue_ms;
  if (names && names_bytes) { std::strncpy(names, txt.c_str(), names_bytes - 1); names[names_bytes - 1] = 0; }
  return 0;
}

// Debug taps for the stage-level parity tests: with capture on, every forward also copies its intermediate tensors (stream-ordered
// device-to-device copies into library-owned buffers): "embeddings", "enc_layer_<l>" (fp32 [I][T][C], images in the reference's
// batch-major (query, refs) order), "featmap_query" (fp32 [B][Np][C]), "featmap_ref" (16 bit [B][N*Np][C]), "dec<l>_out" (fp32
// [B][Np][C]), "head_pre_activation" (fp32 [B][Np][P*P]).  Not for timed runs: the first captured forward of a shape allocates.
int cs_debug_capture(cs_handle h, int on) {
  if (!h) return fail(CS_ERR_BAD_ARG, "null handle");
  h->capture = on != 0;
  return 0;
}

// Copies tap `name` of the last captured forward to `dst` (device memory, dst_bytes >= the tap's size) on `stream`; reports its element
// type (CS_DTYPE_*), rank and shape.  dst == NULL: only reports.  CS_ERR_STATE when no forward has captured that tap.
int cs_debug_read(cs_handle h, const char* name, void* dst, size_t dst_bytes, int* dtype, int* ndim, int64_t* shape4, cs_stream stream) {
  if (!h || !name) return fail(CS_ERR_BAD_ARG, "null argument");
  auto it = h->taps.find(name);
  if (it == h->taps.end() || !it->second.d) return fail(CS_ERR_STATE, "no tap named '%s' was captured (cs_debug_capture before the forward?)", name);
  const cs_model::Tap& t = it->second;
  if (dtype) *dtype = t.dtype;
  if (ndim) *ndim = t.ndim;
  if (shape4) for (int k = 0; k < 4; ++k) shape4[k] = t.shape[k];
  if (!dst) return 0;
  if (dst_bytes < t.bytes) return fail(CS_ERR_BAD_ARG, "tap '%s' holds %zu bytes, destination %zu", name, t.bytes, dst_bytes);
  hipStream_t st = (hipStream_t)stream;
  if (h->ev_done && h->last_stream != st) HIPCHK(hipStreamWaitEvent(st, h->ev_done, 0));
  HIPCHK(hipMemcpyAsync(dst, t.d, t.bytes, hipMemcpyDeviceToDevice, st));
  return 0;
}

int cs_profile_enable(cs_handle h, int on) {
  if (!h) return fail(CS_ERR_BAD_ARG, "null handle");
  hipDeviceSynchronize();
  for (auto& r : h->recs) { hipEventDestroy(r.a); hipEventDestroy(r.b); }
  h->recs.clear();
  h->prof = on != 0;
  return 0;
}

// Lanes of the following forwards: 0 = as configured (cs_config.lanes), n >= 1 = at most n.  Same launches either way (score maps are
// bit-identical); for CrossScoreNet.calibrate_lanes, which times the two-lane forward against the one-lane one on the same handle.
int cs_set_lanes(cs_handle h, int lanes) {
  if (!h || lanes < 0 || lanes > CS_MAX_LANES) return fail(CS_ERR_BAD_ARG, "cs_set_lanes: bad arguments");
  h->lanes_now = lanes;
  return 0;
}

// Gives the handle's lane streams back; the next forward draws new ones (and probes them, cs_forward).  A set-up call: waits for the
// lanes' work.  The remedy when a two-lane forward turns out not to overlap although its streams passed the probe.
int cs_redraw_lane_streams(cs_handle h) {
  if (!h) return fail(CS_ERR_BAD_ARG, "null handle");
  for (int l = 0; l < CS_MAX_LANES; ++l)
    if (h->lane_st[l]) {
      HIPCHK(hipStreamSynchronize(h->lane_st[l]));
      // kept alive until the next forward has created the replacements: a stream destroyed now would hand its hardware queue
      // straight back to the very next hipStreamCreate
      h->lane_st_old.push_back(h->lane_st[l]);
      h->lane_st[l] = nullptr;
    }
  return 0;
}

int cs_profile_read_bytes(cs_handle h, int family, double* bytes) {
  if (!h || !bytes) return fail(CS_ERR_BAD_ARG, "null argument");
  double b = 0;
  for (auto& r : h->recs)
    if (r.family == family) b += r.bytes;
  *bytes = b;
  return 0;
}

int cs_profile_read(cs_handle h, int family, double* total_ms, int* launches, double* flops) {
  if (!h) return fail(CS_ERR_BAD_ARG, "null handle");
  HIPCHK(hipDeviceSynchronize());
  double ms = 0, fl = 0; int n = 0;
  for (auto& r : h->recs) {
    if (r.family != family) continue;
    float t = 0;
    HIPCHK(hipEventElapsedTime(&t, r.a, r.b));
    ms += t; fl += r.flops; ++n;
  }
  if (total_ms) *total_ms = ms;
  if (launches) *launches = n;
  if (flops) *flops = fl;
  return 0;
}

// Non-finite values the score maps of this handle's forwards held since the last call (and resets the count).  Waits for the handle's
// last forward (an explicit check, not part of the hot loop): an fp16 operand that overflowed upstream reaches the output as NaN.
int cs_nonfinite_count(cs_handle h, long long* count) {
  if (!h || !count) return fail(CS_ERR_BAD_ARG, "null argument");
  *count = 0;
  if (!h->nonfinite) return 0;
  if (h->ev_done) HIPCHK(hipEventSynchronize(h->ev_done));
  unsigned v = 0;
  HIPCHK(hipMemcpy(&v, h->nonfinite, sizeof v, hipMemcpyDeviceToHost));
  if (v) HIPCHK(hipMemset(h->nonfinite, 0, sizeof v));
  *count = (long long)v;
  return 0;
}

// ---------------------------------------------------------------------------------------------------------
// single-op entry points
// ---------------------------------------------------------------------------------------------------------
static int g_op_bf16 = 0;  // operand type of the cs_op_* entry points below (a handle carries its own: cs_config.operand_dtype)
void cs_debug_stream_probe_log(int on) { g_debug_stream_log = on; }
void cs_debug_rowln_enable(int on) { g_rowln_off = on == 0 ? 1 : 0; g_rowln_no_next = on == 2 ? 1 : 0; }

// out_f32 / out_f16 (M, C) = LayerNorm(resid + A (M, C) W (C, C)^T + bias): the decoder's sub-block closing as the forward runs it (C = 384)
int cs_op_linear_layernorm(const uint16_t* A, const uint16_t* W, const float* bias, const float* resid, const float* gamma, const float* beta,
                           float eps, float* out_f32, uint16_t* out_f16, int M, int C, cs_stream stream) {
  CsRowLnParams q{};
  q.A = A; q.lda = C; q.W = W; q.ldw = C; q.bias = bias; q.resid = resid; q.ldr = C; q.gamma = gamma; q.beta = beta; q.eps = eps;
  q.out_f32 = out_f32; q.out_f16 = out_f16; q.M = M;
  if (const char* e = cs_rowln_check(&q, C)) return fail(CS_ERR_BAD_ARG, "%s", e);
  HIPCHK(cs_rowln_launch(&q, C, g_op_bf16, (hipStream_t)stream));
  return 0;
}

// the same with the sub-block's next linear behind it: out2 (M, n2) = act2(LN rows (rounded to the operand type) x W2 (n2, C)^T + bias2)
int cs_op_linear_layernorm_linear(const uint16_t* A, const uint16_t* W, const float* bias, const float* resid, const float* gamma,
                                  const float* beta, float eps, float* out_f32, uint16_t* out_f16, const uint16_t* W2, const float* bias2,
                                  int n2, int act2, uint16_t* out2, int M, int C, cs_stream stream) {
  CsRowLnParams q{};
  q.A = A; q.lda = C; q.W = W; q.ldw = C; q.bias = bias; q.resid = resid; q.ldr = C; q.gamma = gamma; q.beta = beta; q.eps = eps;
  q.out_f32 = out_f32; q.out_f16 = out_f16; q.M = M;
  q.W2 = W2; q.ldw2 = C; q.bias2 = bias2; q.out2 = out2; q.ld2 = n2; q.n2 = n2; q.act2 = act2;
  if (n2 <= 0) return fail(CS_ERR_BAD_ARG, "linear + LayerNorm + linear: n2 must be positive");
  if (const char* e = cs_rowln_check(&q, C)) return fail(CS_ERR_BAD_ARG, "%s", e);
  HIPCHK(cs_rowln_launch(&q, C, g_op_bf16, (hipStream_t)stream));
  return 0;
}

int cs_debug_set_op_operand_dtype(int dtype) {
  if (dtype != 0 && dtype != 1) return fail(CS_ERR_BAD_ARG, "operand dtype must be 0 (fp16) or 1 (bf16)");
  g_op_bf16 = dtype;
  return 0;
}

int cs_op_gemm(const uint16_t* A, int lda, const uint16_t* W, int ldw, int M, int N, int K, const float* bias,
               const float* resid, int ldr, void* out, int ldc, int epi, const float* pos, int Np, int gw, int P, int act,
               float powp, uint16_t* out_f16, float* stats_out, int stats_sp, const float* ln_part, int ln_sp, const float* col_s,
               float ln_eps, cs_stream stream) {
  CsGemmParams g = gp(A, lda, W, ldw, M, N, K, bias, out, ldc);
  g.out_f16 = out_f16; g.stats_out = stats_out; g.stats_sp = stats_sp; g.ln_part = ln_part; g.ln_sp = ln_sp; g.col_s = col_s;
  g.ln_eps = ln_eps;
  g.resid = resid; g.ldr = ldr; g.pos = pos; g.Np = Np; g.gw = gw; g.P = P; g.act = act; g.powp = powp;
  g.bf16 = g_op_bf16;
  if (epi < 0 || epi > CS_EPI_RESID_F32_LN) return fail(CS_ERR_BAD_ARG, "gemm: unknown epilogue %d", epi);
  if (const char* e = cs_gemm_check(&g, epi)) return fail(CS_ERR_BAD_ARG, "%s", e);
  HIPCHK(cs_gemm_launch(&g, epi, (hipStream_t)stream));
  return 0;
}

int cs_op_head_score(const uint16_t* A, int lda, const uint16_t* W, int ldw, int M, int K, const float* bias, float* score, int Np, int gw, int P,
                     int act, float powp, float* mean_part, unsigned* mean_cnt, float* mean_out, cs_stream stream) {
  CsGemmParams g = gp(A, lda, W, ldw, M, P * P, K, bias, score, 4);
  g.Np = Np; g.gw = gw; g.P = P; g.act = act; g.powp = powp;
  g.mean_part = mean_part; g.mean_cnt = mean_cnt; g.mean_out = mean_out;
  g.bf16 = g_op_bf16;
  if (const char* e = cs_gemm_check(&g, CS_EPI_HEAD_SCORE)) return fail(CS_ERR_BAD_ARG, "%s", e);
  HIPCHK(cs_gemm_launch(&g, CS_EPI_HEAD_SCORE, (hipStream_t)stream));
  return 0;
}

int cs_op_attention(const uint16_t* Q, const uint16_t* K, const uint16_t* V, uint16_t* O, int ldq, int ldk, int ldv, int ldo,
                    long long q_bs, long long k_bs, long long v_bs, long long o_bs, int batch, int heads, int Lq, int Lk, int dh,
                    float q_scale, float* lse, cs_stream stream) {
  CsAttnParams a{};
  a.Q = Q; a.K = K; a.V = V; a.O = O; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo;
  a.q_bs = q_bs; a.k_bs = k_bs; a.v_bs = v_bs; a.o_bs = o_bs; a.Lq = Lq; a.Lk = Lk; a.heads = heads;
  if (!(q_scale >= 0.f)) return fail(CS_ERR_BAD_ARG, "attention: q_scale must be >= 0 (0 = log2(e)/sqrt(dh))");
  a.scale_log2e = q_scale == 0.f ? LOG2E / std::sqrt((float)dh) : q_scale; a.lse = lse;
  a.bf16 = g_op_bf16;
  if (const char* e = cs_attn_check(&a, dh, batch)) return fail(CS_ERR_BAD_ARG, "%s", e);
  HIPCHK(cs_attn_launch(&a, dh, batch, (hipStream_t)stream));
  return 0;
}

int cs_op_attention_weights(const uint16_t* Q, const uint16_t* K, int ldq, int ldk, long long q_bs, long long k_bs, int batch,
                            int heads, int Lq, int Lk, int dh, float q_scale, const float* lse, int head, float* out, cs_stream stream) {
  if (!Q || !K || !lse || !out || !supported_dh(dh) || !(q_scale >= 0.f) || head < 0 || head >= heads || Lq <= 0 || Lk <= 0 || Lq > 65535 || batch <= 0 || batch > 65535)
    return fail(CS_ERR_BAD_ARG, "attention_weights: bad arguments");
  CsAttnParams a{};
  a.Q = Q; a.K = K; a.ldq = ldq; a.ldk = ldk; a.q_bs = q_bs; a.k_bs = k_bs; a.Lq = Lq; a.Lk = Lk; a.heads = heads;
  a.scale_log2e = q_scale == 0.f ? LOG2E / std::sqrt((float)dh) : q_scale; a.lse = const_cast<float*>(lse);
  a.bf16 = g_op_bf16;
  HIPCHK(cs_attn_weights_launch(&a, dh, batch, head, out, (hipStream_t)stream));
  return 0;
}

int cs_op_layernorm(const float* x, int M, int C, const float* gamma, const float* beta, float eps, float* out_f32,
                    uint16_t* out_f16, cs_stream stream) {
  if (!x || !gamma || !beta || M <= 0 || C <= 0 || C % 4 || C > 2048) return fail(CS_ERR_BAD_ARG, "layernorm: C must be a multiple of 4 and <= 2048");
  HIPCHK(cs_layernorm_launch(x, M, C, gamma, beta, eps, out_f32, out_f16, g_op_bf16, (hipStream_t)stream));
  return 0;
}

int cs_op_ln_finalize(const float* part, int M, int rows_padded, int sp, int C, float eps, float* stat, cs_stream stream) {
  if (!part || !stat || M <= 0 || rows_padded < M || sp <= 0 || C <= 0) return fail(CS_ERR_BAD_ARG, "ln_finalize: bad arguments");
  HIPCHK(cs_ln_finalize_launch(part, M, rows_padded, sp, C, eps, stat, (hipStream_t)stream));
  return 0;
}

int cs_op_im2col(const float* x, uint16_t* out, int I, int H, int W, int P, int Kp, cs_stream stream) {
  if (!x || !out || I <= 0 || P <= 0 || H < P || W < P || Kp % 8 || Kp < 3 * P * P) return fail(CS_ERR_BAD_ARG, "im2col: bad arguments");
  HIPCHK(cs_im2col_launch(x, nullptr, 0, 0, out, I, H, W, P, Kp, nullptr, g_op_bf16, (hipStream_t)stream));
  return 0;
}

// Patch embedding in one launch (patch.hip), as the forward runs it for 14-pixel patches and C = 384 n.  Same arguments and result as
// cs_op_patch_embed(centred = 1); CS_ERR_BAD_ARG for shapes the one-launch form does not take.
int cs_op_patch_embed_fused(const float* x, const float* w, const float* bias, const float* pos, int I, int H, int W, int P, int C,
                            float* out, cs_stream stream) {
  if (!x || !w || !bias || !pos || !out || I <= 0 || !cs_patch_fused_supported(H, W, P, C)) return fail(CS_ERR_BAD_ARG, "patch_embed_fused: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  h16_t* wf = nullptr; float* wsum = nullptr;
  HIPCHK(hipMalloc(&wf, cs_patch_pack_elems(C) * sizeof(h16_t)));
  HIPCHK(hipMalloc(&wsum, (size_t)3 * C * sizeof(float)));
  int rc = 0;
  auto chk = [&](hipError_t e, const char* what) { if (e != hipSuccess && !rc) rc = fail(CS_ERR_HIP, "%s: %s", what, hipGetErrorString(e)); };
  chk(cs_patch_pack_launch(w, C, wf, g_op_bf16, st), "pack");
  chk(cs_patch_wsum_launch(w, C, P, wsum, st), "wsum");
  if (!rc) chk(cs_patch_fused_launch(x, nullptr, 0, 0, I, H, W, C, wf, bias, pos, wsum, out, g_op_bf16, st), "patch");
  chk(hipStreamSynchronize(st), "sync");
  hipFree(wf); hipFree(wsum);
  return rc;
}

// The same launch fed from ONE decoded uint8 image geometry (test entry point of the one-pass input stage): imgs = I device images of identical
// size (I, in_h, row_bytes) -> out (I * (1 + Np), C) as cs_op_patch_embed_fused on cs_op_preprocess_u8's output of each image.
int cs_op_patch_embed_fused_u8(const uint8_t* imgs, int I, int in_h, int in_w, int row_bytes, int rs_h, int rs_w, int crop_y, int crop_x, int H, int W,
                               const float* mean3, const float* std3, const float* w, const float* bias, const float* pos, int P, int C, float* out,
                               cs_stream stream) {
  if (!imgs || !w || !bias || !pos || !out || !mean3 || !std3 || I <= 0 || !cs_patch_fused_supported(H, W, P, C) || crop_y < 0 || crop_x < 0 ||
      crop_y + H > rs_h || crop_x + W > rs_w || row_bytes < 3 * in_w)
    return fail(CS_ERR_BAD_ARG, "patch_embed_fused_u8: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  CsU8Tables t{};
  int span = 0;
  HIPCHK(cs_preprocess_tables(in_h, in_w, rs_h, rs_w, crop_y, H / P, P, &t, &span, nullptr));
  if (cs_patch_u8_runs(W, span) <= 0) return fail(CS_ERR_UNSUPPORTED, "patch_embed_fused_u8: %d source rows per patch row do not fit", span);
  std::vector<CsU8Desc> hd(I);
  for (int i = 0; i < I; ++i) { hd[i].data = imgs + (size_t)i * in_h * row_bytes; hd[i].t = t; hd[i].row_bytes = row_bytes; hd[i].crop_y = crop_y; hd[i].crop_x = crop_x; }
  h16_t* wf = nullptr; float* wsum = nullptr; CsU8Desc* dd = nullptr;
  HIPCHK(hipMalloc(&wf, cs_patch_pack_elems(C) * sizeof(h16_t)));
  HIPCHK(hipMalloc(&wsum, (size_t)3 * C * sizeof(float)));
  HIPCHK(hipMalloc(&dd, (size_t)I * sizeof(CsU8Desc)));
  int rc = 0;
  auto chk = [&](hipError_t e, const char* what) { if (e != hipSuccess && !rc) rc = fail(CS_ERR_HIP, "%s: %s", what, hipGetErrorString(e)); };
  chk(hipMemcpy(dd, hd.data(), (size_t)I * sizeof(CsU8Desc), hipMemcpyHostToDevice), "descriptors");
  chk(cs_patch_pack_launch(w, C, wf, g_op_bf16, st), "pack");
  chk(cs_patch_wsum_launch(w, C, P, wsum, st), "wsum");
  if (!rc) chk(cs_patch_fused_u8_launch(dd, I, 0, 0, I, H, W, C, span, mean3, std3, wf, bias, pos, wsum, out, g_op_bf16, st), "patch_u8");
  chk(hipStreamSynchronize(st), "sync");
  hipFree(wf); hipFree(wsum); hipFree(dd);
  return rc;
}

// Patch embedding as the forward ran it before patch.hip (im2col -> MFMA GEMM with the PATCH epilogue), for op-level tests of the mean-centred form:
// centred != 0: every patch's per-channel mean is removed before the fp16 rounding (im2col_rows_kernel) and added back in fp32 as
// mean_ch * sum_taps W[n][ch] by the epilogue (patch_wsum_kernel).  x (I,3,H,W), w (C,3,P,P), bias (C), pos ((1 + Np), C) -> out (I * (1 + Np), C)
// fp32 with the patch rows written (CLS rows untouched).  Allocates its temporaries: a test entry point, not a hot path.
int cs_op_patch_embed(const float* x, const float* w, const float* bias, const float* pos, int I, int H, int W, int P, int C, int centred,
                      float* out, cs_stream stream) {
  if (!x || !w || !bias || !pos || !out || I <= 0 || P != 14 || H < P || W < P || C <= 0 || C % 64) return fail(CS_ERR_BAD_ARG, "patch_embed: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const int gh = H / P, gw = W / P, Np = gh * gw, Kp = ((3 * P * P + 63) / 64) * 64;
  h16_t *A = nullptr, *Wp = nullptr; float *pmean = nullptr, *wsum = nullptr;
  HIPCHK(hipMalloc(&A, (size_t)I * Np * Kp * sizeof(h16_t)));
  HIPCHK(hipMalloc(&Wp, (size_t)C * Kp * sizeof(h16_t)));
  HIPCHK(hipMalloc(&pmean, (size_t)I * Np * 4 * sizeof(float)));
  HIPCHK(hipMalloc(&wsum, (size_t)3 * C * sizeof(float)));
  int rc = 0;
  auto chk = [&](hipError_t e, const char* what) { if (e != hipSuccess && !rc) rc = fail(CS_ERR_HIP, "%s: %s", what, hipGetErrorString(e)); };
  chk(cs_pack_f16_launch(w, C, 3 * P * P, Wp, Kp, nullptr, nullptr, g_op_bf16, st), "pack");
  chk(cs_patch_wsum_launch(w, C, P, wsum, st), "wsum");
  chk(cs_im2col_launch(x, nullptr, 0, 0, A, I, H, W, P, Kp, centred ? pmean : nullptr, g_op_bf16, st), "im2col");
  if (!rc) {
    CsGemmParams g = gp(A, Kp, Wp, Kp, I * Np, C, Kp, bias, out, C);
    g.pos = pos; g.Np = Np; g.bf16 = g_op_bf16;
    if (centred) { g.pmean = pmean; g.wsum = wsum; }
    if (const char* e = cs_gemm_check(&g, CS_EPI_PATCH_F32)) rc = fail(CS_ERR_BAD_ARG, "%s", e);
    else chk(cs_gemm_launch(&g, CS_EPI_PATCH_F32, st), "gemm");
  }
  chk(hipStreamSynchronize(st), "sync");
  hipFree(A); hipFree(Wp); hipFree(pmean); hipFree(wsum);
  return rc;
}

int cs_op_preprocess_u8(const uint8_t* img, int in_h, int in_w, int in_row_bytes, int rs_h, int rs_w, int crop_y, int crop_x, int out_h,
                        int out_w, const float* mean3, const float* std3, float* out, float* scratch, cs_stream stream) {
  if (!img || !out || !mean3 || !std3 || in_h <= 0 || in_w <= 0 || in_row_bytes < 3 * in_w || rs_h <= 0 || rs_w <= 0 || out_h <= 0 ||
      out_w <= 0 || crop_y < 0 || crop_x < 0 || crop_y + out_h > rs_h || crop_x + out_w > rs_w)
    return fail(CS_ERR_BAD_ARG, "preprocess_u8: bad sizes (the crop window must lie inside the resized image)");
  if ((rs_h != in_h || rs_w != in_w) && !scratch) return fail(CS_ERR_BAD_ARG, "preprocess_u8: a resize needs in_h*rs_w*3 floats of scratch");
  for (int c = 0; c < 3; ++c)
    if (!(std3[c] > 0.f)) return fail(CS_ERR_BAD_ARG, "preprocess_u8: std must be positive");
  HIPCHK(cs_preprocess_launch(img, in_h, in_w, in_row_bytes, rs_h, rs_w, crop_y, crop_x, out_h, out_w, mean3, std3, out, scratch,
                              (hipStream_t)stream));
  return 0;
}

int cs_op_score_to_gray16(const float* score, long long n, int signed_range, uint16_t* out, cs_stream stream) {
  if (!score || !out || n <= 0 || (signed_range != 0 && signed_range != 1)) return fail(CS_ERR_BAD_ARG, "score_to_gray16: bad arguments");
  HIPCHK(cs_score_gray16_launch(score, (size_t)n, signed_range, out, (hipStream_t)stream));
  return 0;
}

int cs_op_score_to_rgb(const float* score, long long n, float vmin, float vmax, const uint8_t* lut256x3, uint8_t* out, cs_stream stream) {
  if (!score || !out || !lut256x3 || n <= 0 || !(vmax > vmin)) return fail(CS_ERR_BAD_ARG, "score_to_rgb: bad arguments");
  HIPCHK(cs_score_rgb_launch(score, (size_t)n, vmin, vmax, lut256x3, out, (hipStream_t)stream));
  return 0;
}

int cs_op_pos_bicubic(const float* pos, int G, int C, int gh, int gw, float* out, cs_stream stream) {
  if (!pos || !out || G <= 0 || C <= 0 || gh <= 0 || gw <= 0) return fail(CS_ERR_BAD_ARG, "pos_bicubic: bad arguments");
  HIPCHK(cs_pos_bicubic_launch(pos, G, C, gh, gw, 0.0f, out, (hipStream_t)stream));
  return 0;
}

int cs_op_pos_bicubic_ex(const float* pos, int G, int C, int gh, int gw, int legacy, float* out, cs_stream stream) {
  if (!pos || !out || G <= 0 || C <= 0 || gh <= 0 || gw <= 0) return fail(CS_ERR_BAD_ARG, "pos_bicubic: bad arguments");
  HIPCHK(cs_pos_bicubic_launch(pos, G, C, gh, gw, legacy ? 0.1f : 0.0f, out, (hipStream_t)stream));
  return 0;
}

int cs_op_pe_bilinear(const float* pe, int ph, int pw, int C, int gh, int gw, float* out, cs_stream stream) {
  if (!pe || !out || ph <= 0 || pw <= 0 || C <= 0 || gh <= 0 || gw <= 0) return fail(CS_ERR_BAD_ARG, "pe_bilinear: bad arguments");
  HIPCHK(cs_pe_bilinear_launch(pe, ph, pw, C, gh, gw, out, (hipStream_t)stream));
  return 0;
}

int cs_op_pe_interp(const float* pe, int ph, int pw, int C, int gh, int gw, int mode, float* out, cs_stream stream) {
  if (!pe || !out || ph <= 0 || pw <= 0 || C <= 0 || gh <= 0 || gw <= 0) return fail(CS_ERR_BAD_ARG, "cs_op_pe_interp: bad arguments");
  if (mode != 0 && mode != 1) return fail(CS_ERR_BAD_ARG, "cs_op_pe_interp: mode must be 0 (bilinear) or 1 (bicubic)");
  HIPCHK(cs_pe_interp_launch(pe, ph, pw, C, gh, gw, mode, out, (hipStream_t)stream));
  return 0;
}

int cs_op_streams_overlap(cs_stream a, cs_stream b, int* overlap) {
  if (!overlap || a == b) return fail(CS_ERR_BAD_ARG, "streams_overlap: two different streams and a result pointer are needed");
  bool yes = false;
  if (int r = streams_overlap((hipStream_t)a, (hipStream_t)b, &yes)) return r;
  *overlap = yes ? 1 : 0;
  return 0;
}

int cs_op_pack_f16(const float* w, int rows, int K, uint16_t* out, int ldo, const float* row_scale, const float* col_scale,
                    cs_stream stream) {
  if (!w || !out || rows <= 0 || K <= 0 || ldo < K) return fail(CS_ERR_BAD_ARG, "pack_f16: bad arguments");
  HIPCHK(cs_pack_f16_launch(w, rows, K, out, ldo, row_scale, col_scale, g_op_bf16, (hipStream_t)stream));
  return 0;
}

int cs_op_panel_pack(const float* wo, const float* ls1, const float* w1, const float* g2, const float* w2, const float* ls2,
                     uint16_t* img, cs_stream stream) {
  if (!w1 || !w2 || !img) return fail(CS_ERR_BAD_ARG, "panel_pack: null argument");
  if (g_panel_impl) HIPCHK(cs_panel4_pack_launch(wo, ls1, w1, g2, w2, ls2, img, g_op_bf16, (hipStream_t)stream));
  else HIPCHK(cs_panel_pack_launch(wo, ls1, w1, g2, w2, ls2, img, g_op_bf16, (hipStream_t)stream));
  return 0;
}

size_t cs_panel_image_bytes(int with_outproj) { return g_panel_impl ? cs_panel4_image_bytes(with_outproj) : cs_panel8_image_bytes(with_outproj); }
void cs_debug_panel_impl(int impl) { g_panel_impl = impl ? 1 : 0; }

int cs_op_encoder_panel(float* x, const uint16_t* attn_o, const uint16_t* img, const float* bo, const float* b1, const float* b2,
                        uint16_t* u_out, int M, float eps, cs_stream stream) {
  CsPanelParams q{};
  q.x = x; q.attn_o = attn_o; q.img = img; q.bo = bo; q.b1 = b1; q.b2 = b2; q.u_out = u_out; q.M = M; q.eps = eps;
  q.bf16 = g_op_bf16;
  if (const char* e = cs_panel_check(&q)) return fail(CS_ERR_BAD_ARG, "%s", e);
  if (g_panel_impl) HIPCHK(cs_panel4_launch(&q, (hipStream_t)stream));
  else HIPCHK(cs_panel_launch(&q, (hipStream_t)stream));
  return 0;
}

int cs_op_ln_fold_consts(const uint16_t* w_packed, int ldp, const float* w, const float* beta, const float* bias, int N, int K,
                         float* s_out, float* c_out, cs_stream stream) {
  if (!w || !beta || !c_out || (w_packed && (!s_out || ldp < K)) || N <= 0 || K <= 0) return fail(CS_ERR_BAD_ARG, "ln_fold_consts: bad arguments");
  HIPCHK(cs_ln_fold_consts_launch(w_packed, ldp, w, beta, bias, N, K, s_out, c_out, g_op_bf16, (hipStream_t)stream));
  return 0;
}

}  // extern "C"
