/* C ABI of libcrossscore_hip.so -- the MI355X (gfx950) CrossScore inference hot path.
 *
 * The reference (ActiveVisionLab/CrossScore) is pure Python and has no native interface; the boundary this
 * library sits behind is the model call
 *     CrossScoreNet.forward(query_img, ref_cross_imgs, need_attn_weights, need_attn_weights_head_id, norm_img)
 *         -> {"score_map_ref_cross", "attn_weights_map_ref_cross"}            (task/core.py:58-117)
 * made from CrossScoreLightningModule._core_step (task/core.py:265-272).  Each entry point below names the
 * reference code it replaces.  Plain pointers and sizes only: no torch / Python types cross this line.
 *
 * Ownership: the caller (PyTorch) owns every input / output / weight-source buffer; the library owns only its
 * packed fp16 weights and its workspace.  Threading: a handle is not thread-safe and serves one forward at a time (its
 * workspace is shared by consecutive calls); to keep several batches in flight create one handle per batch in flight (the
 * Python side does: crossscore_amd/pipeline.py) -- handles are independent, also across GPUs of one process (launcher state
 * is kept per device).  All work is enqueued on
 * the caller's hipStream_t (a forward that arrives on another stream than the previous one waits for it with an event: the
 * workspace is shared).  No entry point waits for the device or a stream except cs_finalize, cs_profile_*, cs_destroy,
 * cs_op_streams_overlap and the one-time creation of the encoder lanes' streams in a handle's first forward (which probes that
 * the lanes' streams overlap, ~1 ms):
 * the first forward of a new shape allocates (hipMalloc: position tables of a new patch grid, a larger workspace -- the old
 * one is retired behind an event and freed later), fills what it allocated on the caller's stream, and never overwrites or
 * frees memory that queued work may still read; calls of a shape seen before allocate nothing.  (Only a 17th distinct patch
 * grid, or a 65th distinct resize geometry in cs_preprocess_u8, drops the table cache behind a device synchronisation.)
 * Weights are fp32 at this boundary (cs_set_weight; cs_set_weight_typed also takes fp16 / bf16 sources and widens them): the
 * reference checkpoint stores fp32, and the library packs its own 16-bit copies in cs_finalize.  Every function returning int returns 0 on success; on failure
 * cs_last_error() describes it (CS_ERR_* below) and nothing was launched on the bad-argument paths.
 */
#ifndef CROSSSCORE_HIP_H
#define CROSSSCORE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cs_model* cs_handle;
typedef void* cs_stream; /* hipStream_t */

enum {
  CS_OK = 0,
  CS_ERR_BAD_ARG = 1,     /* shape / config the reference would also reject (ValueError / assert) */
  CS_ERR_UNSUPPORTED = 2, /* legal for the reference, not built here (e.g. head dim outside {16,48,64,96,128,192}) */
  CS_ERR_STATE = 3,       /* wrong call order (forward before finalize, missing weight) */
  CS_ERR_HIP = 4          /* HIP runtime error; message carries hipGetErrorString */
};

/* Architecture + the model.* config keys the forward consumes (config/model/model.yaml:1-32). */
typedef struct cs_config {
  int hidden;        /* C: Dinov2Config.hidden_size (384 small / 768 base), task/core.py:39 */
  int enc_layers;    /* Dinov2Config.num_hidden_layers */
  int enc_heads;     /* Dinov2Config.num_attention_heads (head dim hidden / enc_heads in {16,48,64,96,128}; DINOv2: 64) */
  int mlp_ratio;     /* 4 */
  int patch;         /* model.patch_size = 14 */
  int pos_grid;      /* sqrt(#position_embeddings - 1) = 37 */
  int pe_h, pe_w;    /* model.pos_enc.multi_view.{h,w} = 40,40 (bilinear, align_corners=True) */
  int dec_layers;    /* 2, model/cross_reference.py:38 */
  int dec_heads;     /* 8, model/cross_reference.py:31 */
  int do_self_attn;  /* model.decoder_do_self_attn */
  int do_short_cut;  /* model.decoder_do_short_cut */
  int act;           /* 0 sigmoid (metric.min == 0), 1 tanh (metric.min == -1), model/regression_layer.py:31-38 */
  float pow_p;       /* exponent after the activation; 1 = identity, model/regression_layer.py:40-62 */
  int enc_chunk_images; /* images per encoder pass (0 = library default) */
  int ln_fold;       /* encoder LayerNorms (HF modeling_dinov2.py:361-380) folded into the producing / consuming GEMM epilogues:
                      * 0 (default): by backbone width -- hidden 384: inside the token-panel kernel; hidden 768 / 1024: in the 256-tile GEMM's
                      *    epilogues for chunks of >= 256 rows (residual epilogues write 16-bit(x) + row partial sums, a row-statistics kernel,
                      *    QKV / fc1 apply rstd * (acc - mean * s) + c), separate LayerNorm kernels otherwise;
                      * 1: the 128-row kernel's folded epilogues everywhere (fp16 operands only; slower: A/B and tests);
                      * 2: separate LayerNorm kernels for the wide backbones */
  int lanes;         /* internal streams that run independent image chunks / batch groups concurrently: 0 = default (2), 1 = serial, up to 4 */
  int pos_interp_legacy; /* encoder position-embedding resize (grids other than 37 x 37, or H != W): 0 = F.interpolate(size=(h, w)), the installed
                       * transformers (>= 4.4x; what the goldens were generated with); 1 = scale_factor ((h + 0.1) / 37, (w + 0.1) / 37) as
                       * in the reference's pinned transformers 4.33.3 (environment.yaml:340): source coordinates shrink by h / (h + 0.1) */
  int enc_fused;      /* encoder layer structure: 0 = default: with hidden == 384 (ViT-S) each layer is QKV GEMM + attention + ONE
                       * token-panel kernel (out-projection, residual, norm2, fc1, GELU, fc2, residual and the next layer's norm1; the 4C
                       * hidden activations stay in registers), else the unfused kernels; 1 = always the unfused kernels */
  int operand_dtype;  /* 16-bit operand type of the MFMA kernels (activations between kernels, packed weights): 0 = IEEE half (default: 11 significant
                       * bits, finite to 65504; score-map MAE 1e-4 vs the fp32 reference), 1 = bfloat16 (8 bits, fp32's range; MAE 8e-4): the
                       * choice for a checkpoint whose activations leave the half range (trainer.precision = bf16-mixed in the predict driver).
                       * Accumulation, softmax statistics, LayerNorm, the residual stream and the outputs are fp32 either way. */
  int pe_interp_mode; /* model.pos_enc.multi_view.interpolate_mode, applied when the patch grid differs from (pe_h, pe_w) (model/positional_encoding.py:61-69,
                       * always with align_corners=True): 0 = bilinear (the reference default), 1 = bicubic.  torch rejects every other mode of a 4-D
                       * tensor with align_corners=True, so these two are all the reference can run. */
  int skip_finite_check; /* 0 = default: every forward ends with a pass over the score map that counts non-finite values into a device
                       * counter (cs_nonfinite_count; ~3 us); 1 = skip it */
  int swiglu;         /* encoder MLP: 0 = Dinov2MLP (fc1 -> GELU -> fc2, HF modeling_dinov2.py:286-297); 1 = Dinov2SwiGLUFFN (HF:300-316,
                       * Dinov2Config.use_swiglu_ffn: facebook/dinov2-giant): weights_in (2F x C) -> silu(x1) * x2 -> weights_out (C x F) with
                       * F = (int(hidden * mlp_ratio * 2 / 3) + 7) / 8 * 8, which must be a multiple of 64; the layer then runs as separate
                       * LayerNorm / GEMM launches with one elementwise launch for the gate (no token-panel kernel, no LayerNorm fold) */
} cs_config;

/* Replaces CrossScoreNet.__init__ (task/core.py:27-56). NULL on failure. */
cs_handle cs_create(const cs_config* cfg);
void cs_destroy(cs_handle h);
const char* cs_last_error(void);

/* Replaces load_state_dict for one tensor: `name` is the checkpoint key without the "model." prefix (ckpt layout:
 * SURVEY.md 8b), `data` fp32, row-major, host or device memory (is_device), `shape[ndim]` as in the state dict. */
int cs_set_weight(cs_handle h, const char* name, const float* data, int is_device, int ndim, const int64_t* shape);
/* The same for a source tensor of another storage type (SURVEY.md 8b's `dtype` argument): a half- or bfloat16-precision checkpoint is
 * widened to fp32 (exactly) on the way in. */
enum { CS_DTYPE_F32 = 0, CS_DTYPE_F16 = 1, CS_DTYPE_BF16 = 2 };
int cs_set_weight_typed(cs_handle h, const char* name, const void* data, int is_device, int dtype, int ndim, const int64_t* shape);
/* Number of tensors cs_finalize expects and the i-th expected name (for loaders / strict checking). */
int cs_num_weights(cs_handle h);
const char* cs_weight_name(cs_handle h, int i);
/* Packs weights to fp16 / fused layouts (QKV, both decoder layers' KV). Fails if a tensor is missing. */
int cs_finalize(cs_handle h);

/* Replaces CrossScoreNet.forward (task/core.py:58-117) with norm_img=False.
 *   query:  (B,3,H,W) fp32 device, contiguous;   refs: (B,N,3,H,W) fp32 device, contiguous
 *   score_out: (B, P*(H/P), P*(W/P)) fp32 device
 *   attn_out:  NULL, or (B, h, w, N, h, w) fp32 device = probabilities of head `head_id` of the LAST decoder
 *              layer's cross-attention (model/cross_reference.py:91-93)
 *   mean_out:  NULL, or (B) fp32 device = per-image mean of the score map (utils/io/score_summariser.py:180-192), formed inside the head's
 *              launch (no pass over the map); an item's mean has the same bits alone and at any position of any batch */
int cs_forward(cs_handle h, const float* query, const float* refs, int B, int N, int H, int W, float* score_out,
               float* attn_out, int head_id, float* mean_out, cs_stream stream);
/* Reference-feature cache.  In predict the N references of every query are drawn from one finite
 * reference_dir (dataloading/dataset/simple_reference.py:55-58, utils/neighbour/sampler.py:27-34), and a reference's
 * decoder input -- final LayerNorm of its DINOv2 tokens + multi-view PE, task/core.py:141-153,93-98 -- does not depend on the
 * query or on its view slot.  cs_encode_references encodes R images once into fp16 tokens (R, h*w, C); cs_forward_cached
 * scores B queries against gathered tokens (B, N, h*w, C).  Results are bit-identical to cs_forward on the same images; the
 * encoder work per query drops from 1+N images to 1 (a separate mode: it changes the algorithmic FLOPs). */
int cs_encode_references(cs_handle h, const float* imgs, int R, int H, int W, uint16_t* tokens_out, cs_stream stream);
int cs_forward_cached(cs_handle h, const float* query, const uint16_t* ref_tokens, int B, int N, int H, int W,
                      float* score_out, float* attn_out, int head_id, float* mean_out, cs_stream stream);
/* The same three forwards fed from DECODED uint8 images (SURVEY.md 8f-4 as worded: uint8 in, tokens out).  Replaces, together with the patch
 * embedding, the reference's CPU input transforms -- np.float32(img) / 255 (utils/io/images.py:14-29), T.Resize(short side, BILINEAR, antialias)
 * (task/predict.py:87-93, nvs_dataset.py:218-225), the deterministic / integer-patch crop (dataloading/transformation/crop.py:8-25,
 * nvs_dataset.py:227-241), T.Normalize (task/predict.py:68-74) -- inside the patch-embedding launch: no fp32 image tensor is written or read.  The
 * strip of pixels a patch row needs is formed in LDS by the operations of cs_op_preprocess_u8 in their order, so the token rows, and with them every
 * later value, are bit-identical to cs_op_preprocess_u8 + the fp32 entry point (tested).  query / refs / imgs are HOST arrays of descriptors (B; B * N,
 * item-major; R); an image's `data` is DEVICE memory and must stay valid until the stream has passed the call; data == NULL is the all-zero placeholder
 * image (nvs_dataset.py:459-470: zeros before T.Normalize).  Images of one call may differ in size and geometry, not in the H x W window.  mean3 /
 * std3 are HOST pointers.  CS_ERR_UNSUPPORTED when the handle does not run the one-launch patch embedding (patch 14, hidden a multiple of 384) or a
 * patch row reaches more source rows than the launch holds in LDS (1 170: a down-scale beyond ~70 x): cs_u8_input_supported asks beforehand. */
typedef struct cs_u8_image {
  const uint8_t* data;   /* device, HWC RGB, rows row_bytes apart; NULL: all-zero image */
  int h, w, row_bytes;   /* decoded size */
  int rs_h, rs_w;        /* size after T.Resize (== h, w: no resize) */
  int crop_y, crop_x;    /* top-left corner of the H x W window inside the resized image */
} cs_u8_image;
int cs_forward_u8(cs_handle h, const cs_u8_image* query, const cs_u8_image* refs, int B, int N, int H, int W, const float* mean3, const float* std3,
                  float* score_out, float* attn_out, int head_id, float* mean_out, cs_stream stream);
int cs_encode_references_u8(cs_handle h, const cs_u8_image* imgs, int R, int H, int W, const float* mean3, const float* std3, uint16_t* tokens_out,
                            cs_stream stream);
int cs_forward_cached_u8(cs_handle h, const cs_u8_image* query, const uint16_t* ref_tokens, int B, int N, int H, int W, const float* mean3,
                         const float* std3, float* score_out, float* attn_out, int head_id, float* mean_out, cs_stream stream);
/* 1 when the three calls above take this image geometry on this handle, 0 when not (then: cs_op_preprocess_u8 + the fp32 entry points) */
int cs_u8_input_supported(cs_handle h, const cs_u8_image* img, int H, int W);
/* Overflow report.  With fp16 operands an activation beyond 65504 becomes inf and reaches the score map as NaN (nothing clamps in
 * the hot kernels); every forward counts the non-finite values of its score map on the device.  This call waits for the handle's last
 * forward, returns the count since the previous call in *count and resets it: > 0 means "switch to operand_dtype = 1 (bf16)". */
int cs_nonfinite_count(cs_handle h, long long* count);
/* Bytes of library-owned workspace a forward of this shape needs (grown lazily, never shrunk). */
size_t cs_workspace_bytes(cs_handle h, int B, int N, int H, int W);

/* Lanes (internal streams that run independent image chunks of ONE forward side by side): cs_set_lanes limits the following forwards
 * to at most `lanes` (0 = as configured by cs_config.lanes; same launches, bit-identical results); cs_redraw_lane_streams gives the
 * handle's lane streams back so that the next forward draws and probes new ones (a set-up call: it waits for the lanes' work).  Together
 * they let the host check that a two-lane forward really beats the one-lane one and repair it if not (CrossScoreNet.calibrate_lanes):
 * two streams that passed the overlap probe can still end up serialised when other queues were created in between. */
int cs_set_lanes(cs_handle h, int lanes);
int cs_redraw_lane_streams(cs_handle h);
/* debug switch, process-wide (see cs_debug_* below): 1 = one stderr line per lane-stream candidate of the overlap probe (which streams a handle drew
 * and whether they dispatch side by side); default 0.  Replaces the CS_DEBUG_STREAMS environment read of rounds 3-5: the product path reads no
 * environment variable. */
void cs_debug_stream_probe_log(int on);

/* Launch census of the last cs_forward / cs_forward_cached / cs_encode_references on this handle: kernel launches, host wall time of the
 * call (all work is enqueued, nothing is waited for: this is what a rank's CPU thread pays per forward -- bench.py reports it per rank, the
 * reference pays the same kind of cost in Lightning's predict loop, task/predict.py:119-135) and, optionally, the launches by kernel as
 * "name=count ..." text (gemm256, gemm128, attn<dh>, panel, rowln, patch, im2col, ln*, cls, final_ln, ...; the parity tests use it to prove
 * which kernels a shape was routed to). */
int cs_forward_stats(cs_handle h, int* launches, double* host_ms, char* names, size_t names_bytes);

/* Stage-level taps for the parity tests (tests/test_hip_stages.py): with capture on, every forward of this handle also copies its
 * intermediate tensors into library-owned buffers (stream-ordered device-to-device copies; the first captured forward of a shape
 * allocates -- not for timed runs).  Names follow the reference's module outputs (tests/golden/make_golden.py hooks the same points):
 *   "embeddings"            fp32 (I, T, C)      Dinov2Embeddings output, HF modeling_dinov2.py:97-116; I = B * (1 + N) images in the
 *                                               reference's batch-major (query, refs...) order of task/core.py:134-138, T = 1 + h*w
 *   "enc_layer_<l>"         fp32 (I, T, C)      residual stream behind Dinov2Layer l, HF:361-380
 *   "featmap_query"         fp32 (B, h*w, C)    final LayerNorm of the query's patch tokens + multi-view PE, core.py:141-153,93-98
 *   "featmap_ref"           16 bit (B, N*h*w, C)  the same for the reference views (the decoder's memory), in the handle's operand type
 *   "dec<l>_out"            fp32 (B, h*w, C)    decoder layer l output, transformer.py:157-173
 *   "head_pre_activation"   fp32 (B, h*w, P*P)  head[2] output before the RegressionLayer, cross_reference.py:45-50
 * cs_debug_read copies a tap to dst (device memory of at least the tap's size; NULL = only report) on `stream` and reports its element
 * type (CS_DTYPE_*), rank and shape (4 entries).  CS_ERR_STATE when the last forwards captured no such tap. */
int cs_debug_capture(cs_handle h, int on);
int cs_debug_read(cs_handle h, const char* name, void* dst, size_t dst_bytes, int* dtype, int* ndim, int64_t* shape4, cs_stream stream);

/* Per-kernel-family timing with HIP events on the launch stream (for bench.py's roofline object).
 * Families = kernel symbols: 0..9 cs_gemm_kernel<epilogue> (either GEMM kernel), 16 + dh/16 cs_attn_kernel<dh>, 40 cs_panel_kernel,
 * 41 cs_patch_fused_kernel, 42 cs_rowln_kernel, 32 everything else (LayerNorm, im2col, CLS rows, tables).  `flops` = algorithmic FLOPs (2*M*N*K, 4*B*H*Lq*Lk*dh).  Two events per launch. */
int cs_profile_enable(cs_handle h, int on);
int cs_profile_read(cs_handle h, int family, double* total_ms, int* launches, double* flops);
/* algorithmic HBM bytes (operands and results once each) of the recorded launches of one family */
int cs_profile_read_bytes(cs_handle h, int family, double* bytes);

/* ---- single-op entry points (used by the parity tests; same kernels the forward launches) ------------- */
/* cs_debug_*: PROCESS-WIDE switches and taps for tests and measurement tools.  They are not part of the drop-in surface: a product caller never
 * needs them, and the "handles are independent" promise above holds only while nobody flips them under a running forward.
 * 16-bit operand type of the cs_op_* entry points below (a handle has its own cs_config.operand_dtype): 0 fp16 (default), 1 bf16 */
int cs_debug_set_op_operand_dtype(int dtype);
/* out = epilogue(bias + A[M,K] @ W[N,K]^T): see CsEpilogue in csrc/cs_common.h for `epi`. fp16 = raw uint16. */
int cs_op_gemm(const uint16_t* A, int lda, const uint16_t* W, int ldw, int M, int N, int K, const float* bias,
               const float* resid, int ldr, void* out, int ldc, int epi, const float* pos, int Np,
               int gw, int P, int act, float powp,
               /* LayerNorm fold (CsEpilogue 7-9): producer outputs, then consumer inputs; NULL / 0 when unused */
               uint16_t* out_f16, float* stats_out, int stats_sp, const float* ln_part, int ln_sp, const float* col_s,
               float ln_eps, cs_stream stream);
/* The head's last linear with its epilogue (model/regression_layer.py:26-62 activation, utils/misc/image.py:8-21 jigsaw) and, when the three
 * mean_* pointers are given, the per-image mean of the score map from the SAME launch (utils/io/score_summariser.py:180-192
 * score_map.mean(dim=[-1, -2])): score (M / Np, gh P, gw P) fp32; mean_part (M, 4 * ceil(P*P / 128)) fp32 scratch; mean_cnt (M / Np) uint32,
 * ZERO on entry and zero again on completion; mean_out (M / Np) fp32.  An image's mean has the same bits at any position of any batch. */
int cs_op_head_score(const uint16_t* A, int lda, const uint16_t* W, int ldw, int M, int K, const float* bias, float* score, int Np, int gw, int P,
                     int act, float powp, float* mean_part, unsigned* mean_cnt, float* mean_out, cs_stream stream);
/* softmax(QK^T/sqrt(dh))V for `batch` x `heads`; strides in elements; lse may be NULL. */
/* O = softmax(Q K^T / sqrt(dh)) V computed in the base-2 domain: p = 2^(q_scale * q.k - m).  The forward folds log2(e)/sqrt(dh) into its
 * Q projections and passes q_scale = 1 (Q arrives pre-multiplied); q_scale = 0 means log2(e)/sqrt(dh) applied here, to raw Q, at the
 * price of one more fp16 rounding of Q.  lse (optional, [batch][heads][Lq]) is in base-2 units of the scaled logits. */
int cs_op_attention(const uint16_t* Q, const uint16_t* K, const uint16_t* V, uint16_t* O, int ldq, int ldk, int ldv,
                    int ldo, long long q_bs, long long k_bs, long long v_bs, long long o_bs, int batch, int heads,
                    int Lq, int Lk, int dh, float q_scale, float* lse, cs_stream stream);
int cs_op_attention_weights(const uint16_t* Q, const uint16_t* K, int ldq, int ldk, long long q_bs, long long k_bs,
                            int batch, int heads, int Lq, int Lk, int dh, float q_scale, const float* lse, int head,
                            float* out, cs_stream stream);
int cs_op_layernorm(const float* x, int M, int C, const float* gamma, const float* beta, float eps, float* out_f32,
                    uint16_t* out_f16, cs_stream stream);
/* Row statistics of the LayerNorm folded into the 256-tile GEMM (cs_config.ln_fold = 0 on hidden 768 / 1024): part (M, sp, 2) partial (sum,
 * sum of squares) of every row's 64-column slices as the CS_EPI_RESID_F32_LN epilogue of that kernel writes them (sp = C / 64) ->
 * stat (rows_padded, 2) = (mean, 1 / sqrt(var + eps)), rows [M, rows_padded) zero (the consuming epilogue fetches whole 256-row tiles). */
int cs_op_ln_finalize(const float* part, int M, int rows_padded, int sp, int C, float eps, float* stat, cs_stream stream);
int cs_op_im2col(const float* x, uint16_t* out, int I, int H, int W, int P, int Kp, cs_stream stream);
/* The patch embedding exactly as the forward runs it (HF modeling_dinov2.py:141-149: conv patchify = im2col + GEMM, + position rows), with
 * (centred != 0) or without the mean-centred operand form: x (I,3,H,W), w (C,3,P,P), bias (C), pos (1 + Np, C) -> out (I * (1 + Np), C) fp32,
 * patch rows only.  Test entry point (allocates and synchronises). */
int cs_op_patch_embed(const float* x, const float* w, const float* bias, const float* pos, int I, int H, int W, int P, int C, int centred,
                      float* out, cs_stream stream);
/* The same patch embedding in ONE launch (csrc/patch.hip: image strip -> mean-centred 16-bit tile in LDS -> MFMA -> token rows; no im2col
 * matrix in memory), the form cs_forward uses for 14-pixel patches when C is a multiple of 384 and the LayerNorm-folded epilogues are off.
 * Arguments and result as cs_op_patch_embed(centred = 1); CS_ERR_BAD_ARG for other shapes.  Replaces HF modeling_dinov2.py:141-149. */
int cs_op_patch_embed_fused(const float* x, const float* w, const float* bias, const float* pos, int I, int H, int W, int P, int C,
                            float* out, cs_stream stream);
/* The same launch fed from I decoded uint8 images of one size and geometry (test entry point of the one-pass input stage, cs_forward_u8):
 * out = cs_op_patch_embed_fused applied to cs_op_preprocess_u8's output of every image, bit for bit. */
int cs_op_patch_embed_fused_u8(const uint8_t* imgs, int I, int in_h, int in_w, int row_bytes, int rs_h, int rs_w, int crop_y, int crop_x, int H, int W,
                               const float* mean3, const float* std3, const float* w, const float* bias, const float* pos, int P, int C, float* out,
                               cs_stream stream);
/* debug switch, process-wide (see cs_debug_* above): 0 = cs_forward goes back to im2col + GEMM for the patch embedding; default 1 */
void cs_debug_patch_fused_enable(int on);
/* Input stage: device uint8 HWC image (3 channels, rows in_row_bytes apart) -> fp32 CHW [3][out_h][out_w], the
 * tensor cs_forward consumes.  Same operations, in the same order, as the reference's CPU transforms: x/255 (utils/io/images.py:14-29),
 * antialiased bilinear resize to (rs_h, rs_w) (T.Resize, task/predict.py:87-93; skipped when equal to the input size), crop window
 * (crop_y, crop_x, out_h, out_w) of the resized image (dataloading/transformation/crop.py:8-25, nvs_dataset.py:227-241), then
 * (v - mean) / std (T.Normalize, task/predict.py:68-74).  mean3 / std3 are HOST pointers; scratch (device, in_h*rs_w*3 floats) is
 * needed only when resizing.  Filter tables of the last size pair are cached in the library (not thread-safe, like the handle). */
int cs_op_preprocess_u8(const uint8_t* img, int in_h, int in_w, int in_row_bytes, int rs_h, int rs_w, int crop_y, int crop_x, int out_h,
                        int out_w, const float* mean3, const float* std3, float* out, float* scratch, cs_stream stream);
/* Output stage: score map -> the integer images the reference's writers store (PNG compression stays on the host).
 * gray16: metric_map_write, utils/io/images.py:49-63 (signed_range 1: (m+1)*32767 for the SSIM intrinsic range, 0: m*65535), truncated.
 * rgb: gray2rgb, utils/misc/image.py:37-52 (Normalize(vmin,vmax), 256-entry colormap, u8 truncation); lut256x3 = the colormap's byte
 * table on the device (matplotlib "turbo" in the reference). */
int cs_op_score_to_gray16(const float* score, long long n, int signed_range, uint16_t* out, cs_stream stream);
int cs_op_score_to_rgb(const float* score, long long n, float vmin, float vmax, const uint8_t* lut256x3, uint8_t* out, cs_stream stream);
int cs_op_pos_bicubic(const float* pos, int G, int C, int gh, int gw, float* out, cs_stream stream);
/* the same with the resize convention as an argument: legacy = 0 F.interpolate(size=(gh, gw)) (cs_op_pos_bicubic), 1 the reference's pinned
 * transformers 4.33.3 form scale_factor=((gh + 0.1) / G, (gw + 0.1) / G) (cs_config.pos_interp_legacy; golden tests/golden/g6_pos_legacy.npz) */
int cs_op_pos_bicubic_ex(const float* pos, int G, int C, int gh, int gw, int legacy, float* out, cs_stream stream);
int cs_op_pe_bilinear(const float* pe, int ph, int pw, int C, int gh, int gw, float* out, cs_stream stream);
/* the same resize with the mode as an argument: 0 bilinear (cs_op_pe_bilinear), 1 bicubic; align_corners=True in both (cs_config.pe_interp_mode) */
int cs_op_pe_interp(const float* pe, int ph, int pw, int C, int gh, int gw, int mode, float* out, cs_stream stream);
/* fp32 [rows][K] -> fp16 [rows][ldo] (zero padded); row_scale (rows) / col_scale (K) may be NULL: LayerScale folded into the
 * rows of a projection, LayerNorm gamma into its columns */
/* 1 in *overlap when kernels queued on the two streams run side by side, 0 when the runtime serialises them (streams that share a
 * hardware queue, or hardware queues that share a dispatch pipe: a large grid on `a` then holds back a kernel on `b`).  Runs a two-round
 * grid of idle workgroups on `a` beside one idle wave on `b` and waits for them (~0.4 ms): a set-up helper, not for hot loops. */
int cs_op_streams_overlap(cs_stream a, cs_stream b, int* overlap);
int cs_op_pack_f16(const float* w, int rows, int K, uint16_t* out, int ldo, const float* row_scale, const float* col_scale,
                    cs_stream stream);
/* LayerNorm fold constants of a projection: s[n] = sum_k packed W'[n][k], c[n] = bias[n] + sum_k beta[k] W[n][k] */
int cs_op_ln_fold_consts(const uint16_t* w_packed, int ldp, const float* w, const float* beta, const float* bias, int N, int K,
                         float* s_out, float* c_out, cs_stream stream);
/* Encoder token-panel kernel (csrc/panel.hip; hidden 384, MLP 1536 only).  cs_op_panel_pack builds the weight stream the kernel
 * consumes from fp32 matrices: wo (C,C) [NULL: no out-projection units] with per-row scale ls1 (layer_scale1), w1 (4C,C) with
 * per-column scale g2 (norm2 gamma), w2 (C,4C) with per-row scale ls2 (layer_scale2); scales may be NULL.  img must hold
 * cs_panel_image_bytes(wo != NULL) bytes.  cs_op_encoder_panel then computes, in place on x (M,C) fp32,
 *   x += attn_o Wo'^T + bo   (skipped when attn_o is NULL);   x += GELU(norm(x) W1'^T + b1) W2'^T + b2;
 *   u_out = fp16(norm(x))    (skipped when NULL), norm = LayerNorm without gamma/beta (HF modeling_dinov2.py:361-380). */
/* debug switch, process-wide (see cs_debug_* above): which of the two token-panel kernels handles created from now on (their weight images are
 * packed in cs_finalize) and the cs_op_panel_* entry points use: 0 = csrc/panel.hip (8 waves in role-split pairs, rounds 2-5), 1 = csrc/panel4.hip
 * (4 waves, one per SIMD, column-split residual products; round 6).  The images differ: pack and launch under the same setting. */
void cs_debug_panel_impl(int impl);
int cs_panel_supported(int hidden, int mlp_ratio);
size_t cs_panel_image_bytes(int with_outproj);
int cs_op_panel_pack(const float* wo, const float* ls1, const float* w1, const float* g2, const float* w2, const float* ls2,
                     uint16_t* img, cs_stream stream);
int cs_op_encoder_panel(float* x, const uint16_t* attn_o, const uint16_t* img, const float* bo, const float* b1, const float* b2,
                        uint16_t* u_out, int M, float eps, cs_stream stream);
/* debug switch, process-wide (see cs_debug_* above).  Kernel selection of cs_op_gemm / the forward's linears: 1 (default) = shapes with K >= 384, N a multiple of
 * 256 and M >= 256 run on the 256 x 256 x 64-tile kernel (csrc/gemm256.hip), everything else on the 128-row kernel (csrc/gemm.hip);
 * 0 = the 128-row kernel for every shape (a wide backbone's LayerNorm-folded chunks, whose epilogues exist only in the 256-tile kernel, then run
 * LayerNorm launches + the 128-row kernel's plain epilogues: cs_forward asks cs_gemm256_supported per chunk).  The two kernels add the same products
 * in the same order: their results are bit-identical (tested). */
void cs_debug_gemm256_enable(int on);
/* debug switch, process-wide: the smallest K the 256-tile kernel takes (default 384; tools/qkv_k384_try.py compares 384 with 512) */
void cs_debug_gemm256_kmin(int k);
/* The decoder's sub-block closing  x = LN(x + Linear(y))  (model/customised_transformer/transformer.py:157-173) in one launch (csrc/rowln.hip;
 * K = N = C = 384: the ViT-S decoder): out_f32 / out_f16 (M, C) = LayerNorm(resid + A W^T + bias; gamma, beta, eps); resid may be NULL
 * (decoder_do_short_cut off) and may alias out_f32.  CS_ERR_BAD_ARG for other widths (the forward then runs GEMM + LayerNorm).
 * cs_debug_rowln_enable(0): process-wide debug switch back to the two-launch form (see cs_debug_* above). */
int cs_op_linear_layernorm(const uint16_t* A, const uint16_t* W, const float* bias, const float* resid, const float* gamma, const float* beta,
                           float eps, float* out_f32, uint16_t* out_f16, int M, int C, cs_stream stream);
/* ... and with the sub-block's NEXT linear in the same launch (the forward's default where C = 384 for the C-wide ones: the cross-attention's
 * Q projection behind norm1, linear1 + ReLU behind norm2, the head's first linear + LeakyReLU behind the last norm3; the 3 C-wide form, the next
 * layer's packed QKV projection, is built and tested but measured slower than a GEMM of its own, so the forward does not use it;
 * transformer.py:157-173,182-210, cross_reference.py:45-50): out2 (M, n2) = act2(LN rows, rounded to the operand type, x W2 (n2, C)^T + bias2),
 * n2 = C or 3 C, act2 0 none / 1 ReLU / 2 LeakyReLU(0.01).  out_f32 / out_f16 may be NULL when only out2 is wanted; out2 may alias A (a
 * workgroup writes the 64 rows it read at its start). */
int cs_op_linear_layernorm_linear(const uint16_t* A, const uint16_t* W, const float* bias, const float* resid, const float* gamma,
                                  const float* beta, float eps, float* out_f32, uint16_t* out_f16, const uint16_t* W2, const float* bias2,
                                  int n2, int act2, uint16_t* out2, int M, int C, cs_stream stream);
/* 1 (default): both fusions; 2: linear + LayerNorm in one launch, the next linear as a GEMM of its own; 0: GEMM + LayerNorm + GEMM */
void cs_debug_rowln_enable(int on);
/* number of column tiles the GEMM launcher uses for N output columns (LayerNorm partial-sum slots per row = 4 x this) */
int cs_gemm_column_tiles(int N);

#ifdef __cplusplus
}
#endif
#endif
