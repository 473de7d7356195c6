/* dinodet.h -- C ABI of the MI355X-native dino_detector forward path (libdinodet.so).
 *
 * The reference (mudit1729/dinov2-od) is pure Python and has no FFI of its own; this ABI is the
 * boundary SURVEY.md section 8b defines for the path
 *     DINOv2ObjectDetector.forward   dino_detector/models/detector.py:58-69
 *       DINOv2Backbone.forward       dino_detector/models/dinov2_backbone.py:58-67
 *       DETRDecoder.forward          dino_detector/models/detr_decoder.py:47-83
 * and is what the reference-side binding (a ctypes stub) calls.
 *
 * Conventions
 *  - plain pointers and sizes only; every tensor pointer is a DEVICE pointer (HIP), row-major, fp32
 *    unless stated; `stream` is a hipStream_t passed as void* (NULL = default stream).
 *  - every call returns 0 on success, a non-zero dod_status otherwise; no C++ exception crosses
 *    the ABI.  dod_last_error() returns a message for the last failing call on that handle
 *    (handle may be NULL for errors of dod_create / stateless ops).
 *  - all work is enqueued on the caller's stream; forward calls never synchronise and never
 *    allocate once dod_prepare() has been called for that (H, W).
 *  - a handle is re-entrant across handles but not thread-safe on one handle (the reference is
 *    one Python thread per process, one process per GPU: train.py:1501-1506).
 *  - ownership: I/O buffers, the workspace and the weight tensors passed to dod_set_weight are
 *    caller-owned; weights are only read during dod_finalize_weights(), which builds the handle's
 *    private packed copy (LoRA merged, QKV concatenated, bf16 casts).
 */
#ifndef DINODET_H
#define DINODET_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dod_handle dod_handle;

enum dod_dtype { DOD_F32 = 0, DOD_BF16 = 1 };

enum dod_status {
  DOD_OK = 0,
  DOD_ERR_INVALID = 1,      /* bad argument / unsupported shape  (Python raises ValueError)   */
  DOD_ERR_MISSING = 2,      /* a required state-dict key was never set (KeyError)             */
  DOD_ERR_STATE = 3,        /* call order: not finalized, workspace too small (RuntimeError)  */
  DOD_ERR_HIP = 4           /* HIP runtime error (RuntimeError)                               */
};

enum dod_precision {
  DOD_PREC_FP32 = 0,        /* exact-fp32 MFMA/VALU everywhere: the mode gated at 1e-3 parity  */
  DOD_PREC_BF16 = 1,        /* bf16 MFMA operands in the backbone + value projection, fp32
                               accumulate / residual stream / LayerNorm / softmax / decoder     */
  DOD_PREC_BF16X3 = 3,      /* backbone linears as bf16x3 split products on the bf16 MFMA kernels (A = Ah + Al, W = Wh + Wl,
                               Ah Wh + Ah Wl + Al Wh: ~1e-5 relative, fp32-class accuracy at a third of the bf16 rate);
                               everything else as FP32.  A parity-gated mode that runs on the bf16 matrix cores   */
  DOD_PREC_FP8 = 2,         /* as BF16, with the QKV / MLP-in (/ SwiGLU MLP-out) linears on OCP e4m3 MFMA operands:
                               per-token activation scales from the producing LayerNorm / SwiGLU kernel, per-output-
                               feature weight scales (BASELINE configs[4]: ViT-g/14 fp8)          */
  DOD_PREC_FP16X2 = 4       /* as BF16X3, with the four linears of every backbone block as fp16 main product + both cross terms
                               on ONE block-scaled e4m3 MFMA (x = fp16(x) + remainder; 2.0 bf16-MFMA-equivalents per product
                               instead of 3; ~2e-5 relative per linear).  Parity-gated like BF16X3; operands limited to the
                               fp16 range (|x| <= 65504, clamped beyond)                          */
};

/* Shapes.  Backbone fields mirror HF Dinov2Config as used by dinov2_backbone.py:11-27; decoder
 * fields mirror the DETRDecoder constructor (detr_decoder.py:8-9) / config.py:21-35. */
typedef struct dod_config {
  int32_t hidden;           /* D: 384 / 768 / 1024 / 1536 */
  int32_t layers;
  int32_t heads;            /* head_dim must be 64 in bf16 mode */
  int32_t swiglu;           /* 1 for giant (Dinov2SwiGLUFFN) */
  int32_t patch;            /* 14 */
  int32_t pos_grid;         /* 37: pretrained position grid side */
  int32_t ffn_hidden;       /* 4*D, or the SwiGLU hidden size (4096 for giant) */
  float   ln_eps;           /* 1e-6 */
  int32_t lora_r;           /* informational; LoRA presence is detected per linear from the keys */
  float   lora_alpha;
  int32_t target_dim;       /* 0 = no projection (dinov2_backbone.py:33-37) */
  int32_t num_queries;
  int32_t dec_hidden;       /* Dd */
  int32_t dec_heads;
  int32_t dec_layers;
  int32_t num_classes;
  int32_t dim_feedforward;
  int32_t n_points;
  int32_t use_deformable;
  float   dec_ln_eps;       /* 1e-5 */
  int32_t precision;        /* dod_precision */
} dod_config;

int dod_create(const dod_config* cfg, dod_handle** out);
void dod_destroy(dod_handle* h);
const char* dod_last_error(const dod_handle* h);

/* Register one device tensor (dtype DOD_F32, or DOD_BF16 for half-width checkpoints: widened to fp32 while packing) under its reference state-dict key
 * (e.g. "backbone.dino.encoder.layer.3.attention.attention.query.weight",
 *  "...layer.11.mlp.fc1.lora_A.weight", "decoder.decoder.layers.0.cross_attn.value_proj.bias").
 * A leading "module." (DDP, train.py:700-709) is ignored.  Unknown keys are accepted and unused. */
int dod_set_weight(dod_handle* h, const char* key, const void* dev_ptr, const int64_t* shape, int ndim, int dtype /* dod_dtype */);

/* Build the packed weights (merges W + alpha*B*A, concatenates q/k/v, casts).  Synchronises `stream`
 * once at the end; after it returns the caller's weight tensors are no longer referenced.
 * May be called again after new dod_set_weight calls (e.g. after load_state_dict / an optimizer step). */
int dod_finalize_weights(dod_handle* h, void* stream);

/* Precompute (and cache in the handle) the position table for an H x W input -- bicubic resize of the
 * pretrained grid when (H/patch, W/patch) != (pos_grid, pos_grid), modeling_dinov2.py:57-95.
 * Forward calls do this lazily; call it first when the forward is to be captured in a hipGraph. */
int dod_prepare(dod_handle* h, int H, int W, void* stream);

/* Bytes of scratch a forward of batch B at H x W needs. 0 on error. */
size_t dod_workspace_bytes(const dod_handle* h, int B, int H, int W);
int dod_num_tokens(const dod_handle* h, int H, int W);     /* N = (H/p)*(W/p) + 1 */

/* DINOv2ObjectDetector.forward: pixels [B,3,H,W] -> detections [B, Q, C+4] packed
 * (columns 0..C-1 = pred_logits, C..C+3 = pred_boxes cx,cy,w,h after sigmoid). */
int dod_forward(dod_handle* h, const float* pixels, int B, int H, int W, float* detections,
                void* workspace, size_t workspace_bytes, void* stream);

/* The same forward from the device input pipeline's bytes: pixels_hwc = uint8 [B, H, W, 3] (dod_preprocess_u8: the resampled
 * image before ToTensor); the patch-embedding kernel applies ToTensor's x / 255 in its load stage, so the fp32 CHW batch of
 * train.py:584-587 never exists in HBM.  Same detections, bit for bit, as dod_forward on dod_preprocess's output.
 * bf16 / bf16x3 / fp8 precision only (the fused patch embedding). */
int dod_forward_u8(dod_handle* h, const uint8_t* pixels_hwc, int B, int H, int W, float* detections,
                   void* workspace, size_t workspace_bytes, void* stream);

/* DINOv2Backbone.forward: pixels -> features [B, N, out_dim] fp32 (CLS token at index 0). */
int dod_backbone_forward(dod_handle* h, const float* pixels, int B, int H, int W, float* features,
                         void* workspace, size_t workspace_bytes, void* stream);
/* The frozen prefix of the backbone: embeddings (modeling_dinov2.py:97-116) and the first `nblocks` encoder blocks
 * (:361-380) -> the fp32 residual stream x_out [B, N, hidden].  Training only touches the last two blocks (LoRA,
 * dinov2_backbone.py:47-51), so the blocks before them need no autograd: the train()-mode composite runs them here. */
int dod_backbone_prefix(dod_handle* h, const float* pixels, int B, int H, int W, int nblocks, float* x_out,
                        void* workspace, size_t workspace_bytes, void* stream);

/* DETRDecoder.forward: memory [B, N, Dd] fp32 -> detections [B, Q, C+4] packed. */
int dod_decoder_forward(dod_handle* h, const float* memory, int B, int N, float* detections,
                        void* workspace, size_t workspace_bytes, void* stream);
size_t dod_decoder_workspace_bytes(const dod_handle* h, int B, int N);

/* Debug taps: if set, the next forwards copy that stage (fp32) into `dst` (caller-sized).
 * stage: 0 = embeddings [B,N,D]; 1+i = output of encoder block i [B,N,D]; 1000 = decoder memory
 * [B,N,Dd] (final LayerNorm / projection); 2000 = value projection [B,N,Dd]; 3000+j = decoder layer j
 * output [B,Q,Dd].  dst = NULL clears the tap. */
int dod_set_tap(dod_handle* h, int stage, float* dst);

/* Per-kernel-class timing with HIP events recorded on the caller's stream around each launch of the
 * forward (used by bench.py's roofline leg; adds an event pair per launch, so it is off in the timed
 * region).  dod_profile(h, 1) clears and enables, forwards accumulate, dod_profile_read() waits for the
 * events and returns summed milliseconds, algorithmic FLOPs (2*M*N*K resp. 4*B*N*N*D) and launches.
 * cls: 0 bf16 MFMA GEMM, 1 bf16 flash attention, 2 fp32 MFMA GEMM, 3 fp32 attention, 4 backbone LayerNorm, 6 fp8 MFMA GEMM. */
int dod_profile(dod_handle* h, int enable);
int dod_profile_read(dod_handle* h, int cls, double* ms, double* flops, int* launches);

/* ---- stateless operator entry points (the same kernels the forward uses; for parity tests) ------ */
enum dod_act { DOD_ACT_NONE = 0, DOD_ACT_RELU = 1, DOD_ACT_GELU = 2, DOD_ACT_SIGMOID = 3,
               DOD_ACT_SWIGLU_PAIRS = 4 };   /* dod_op_linear only: the N columns are interleaved SwiGLU pairs (2i: x1_i, 2i+1: x2_i); the
                                                 output has N / 2 columns silu(x1_i) * x2_i at row pitch ldc (bf16 in / out) */

/* out[M,N] = act(A[M,K] W[N,K]^T + bias) * scale + resid ; A, W of dtype `in_dtype`; bias/scale/resid fp32 or NULL */
int dod_op_linear(int in_dtype, const void* A, int lda, const void* W, int ldw, int M, int N, int K,
                  const float* bias, const float* scale, const float* resid, int ldr,
                  void* out, int out_dtype, int ldc, int act, void* stream);
/* The fp32 product in the forms the native training step uses (train.py:1079-1109):
 *   C[z] (+)= alpha * A[z] W[z]^T,  z = zb * hb + zh over `batch` strided views (X[z] = X + zb * x_sb + zh * x_sh, in floats);
 * a_kmajor / w_kmajor: that operand is stored [K, rows] (the transposed products of a backward, no transposed copies);
 * ksplit > 1 splits K over the grid and accumulates atomically (C must already hold the addend).
 * Operands whose pointer, pitch and batch strides are multiples of 4 floats are read as float4: such a buffer must hold whole
 * pitches (rows * ld floats), as any [rows, ld] allocation does. */
int dod_op_gemm_f32x(const float* A, int lda, int a_kmajor, long long a_sb, long long a_sh,
                     const float* W, int ldw, int w_kmajor, long long w_sb, long long w_sh,
                     float* C, int ldc, long long c_sb, long long c_sh,
                     int M, int N, int K, int batch, int hb, float alpha, int accumulate, int ksplit, void* stream);
/* fp8 (OCP e4m3) operands, one byte per element: out = act((Aq Wq^T) * a_scale[m] * w_scale[n] + bias) * scale + resid.
 * K % 64 == 0, lda / ldw in bytes, % 16 == 0.  The ViT-g fp8 configuration's linears (BASELINE configs[4]). */
int dod_op_linear_fp8(const void* A, int lda, const float* a_scale, const void* W, int ldw, const float* w_scale,
                      int M, int N, int K, const float* bias, const float* scale, const float* resid, int ldr,
                      void* out, int out_dtype, int ldc, int act, void* stream);
/* x [rows, cols] (fp32 or bf16, ld in elements) -> q e4m3 [rows, ldq] and scale[rows] = amax_row / 448 (1 for an all-zero
 * row); q = round-to-nearest-even e4m3 of x / scale */
int dod_op_quant_rows_fp8(const void* x, int in_dtype, int ld, int rows, int cols, void* q, int ldq, float* scale,
                          void* stream);
/* Block-scaled ("MX") activations of the fp8 GEMM: x [rows, cols] (cols % 64 == 0) -> q e4m3 [rows, ldq] and one e8m0 byte per 32 elements
 * (value 2^(byte - 127): the smallest power of two >= amax_block / 448), laid out [rows][2][cols / 64]: block b of a row at byte
 * (b & 1) * (cols / 64) + (b >> 1).  dod_op_linear_fp8_mx: the linear on such an A (K % 256 == 0), W as in dod_op_linear_fp8. */
int dod_op_quant_mx_fp8(const void* x, int in_dtype, int ld, int rows, int cols, void* q, int ldq, void* block_scales, void* stream);
int dod_op_linear_fp8_mx(const void* A, int lda, const void* a_block_scales, const void* W, int ldw, const float* w_scale,
                         int M, int N, int K, const float* bias, const float* scale, const float* resid, int ldr,
                         void* out, int out_dtype, int ldc, int act, void* stream);
/* The ViT-g fp8 MLP-in linear with the SwiGLU gate AND the block-scaled quantisation of the gated row in its epilogue: W [N, K] holds the
 * (x1_i, x2_i) rows interleaved (N = 2F, N % 128 == 0); out_q [M, F] e4m3 (pitch ldq bytes) = silu(x1) * x2 / 2^e per 32 columns, the
 * e8m0 bytes in out_block_scales in the dod_op_quant_mx_fp8 layout -- the A operand of dod_op_linear_fp8_mx. */
int dod_op_linear_fp8_glu_mx(const void* A, int lda, const float* a_scale, const void* W, int ldw, const float* w_scale,
                             int M, int N, int K, const float* bias, void* out_q, int ldq, void* out_block_scales, void* stream);
/* Both operands block-scaled (what the fp8 mode's linears run since round 4): W [N, K] e4m3 with e8m0 bytes in the same [N][2][K / 64] layout
 * (dod_op_quant_mx_fp8 on the weight matrix).  glu_out_block_scales != NULL: the weights_in form of dod_op_linear_fp8_glu_mx (out = e4m3 rows
 * of N / 2 gated columns at pitch ldc bytes, out_dtype DOD_BF16 as a byte buffer). */
int dod_op_linear_fp8_mx2(const void* A, int lda, const void* a_block_scales, const void* W, int ldw, const void* w_block_scales,
                          int M, int N, int K, const float* bias, const float* scale, const float* resid, int ldr,
                          void* out, int out_dtype, int ldc, int act, void* glu_out_block_scales, void* stream);
/* bf16x3 (parity-gated mode) operators.  Pair layout: [rows, 2*cols] bf16 = [hi | lo], hi = bf16(x), lo = bf16(x - hi).
 * dod_op_split_pair: fp32 x [rows, cols] (ld) -> pair layout.
 * dod_op_linear_x3: A2 [M, 2K], W2 [N, 2K] pair layouts -> act(A W^T + bias) * scale + resid as the split product
 *   Ah Wh^T + Ah Wl^T + Al Wh^T; out_layout 0: fp32 [M, ldc], 1: bf16 [M, ldc], 2: pair layout [M, 2N] (ldc = 2N).
 * dod_op_attention_x3: qkv2 [B*N, 6*D] = [hi(q|k|v) | lo(q|k|v)] -> ctx2 [B*N, 2*D] pair layout, head_dim 64. */
int dod_op_split_pair(const float* x, int ld, int rows, int cols, void* out, void* stream);
int dod_op_linear_x3(const void* A2, const void* W2, int M, int N, int K, const float* bias, const float* scale,
                     const float* resid, int ldr, void* out, int out_layout, int ldc, int act, void* stream);
int dod_op_attention_x3(const void* qkv2, void* ctx2, int B, int N, int heads, float scale, void* stream);
/* fp16x2 (parity-gated mode) operators.  H2 operand format of a [rows, cols] matrix (cols % 32 == 0), 4*cols bytes per row:
 *   [ fp16(x) x cols | per 32-column block: 32 x e4m3(h 2^e), 32 x e4m3((x - h) 2^(e+11)) ],  h = fp16(x);
 *   activations: e = 0; weights (wexp != NULL): per row e = floor(log2(448 / max|h|)), wexp[row] = 127 - e (E8M0 byte) and the
 *   two 32-byte halves of a block swapped, so that k-slot by k-slot main meets remainder.
 * dod_op_linear_h2: act(A W^T + bias) * scale + resid with A W^T ~ fp16(A) fp16(W)^T + the two cross terms on block-scaled e4m3
 *   MFMA operands; out_layout 0: fp32, 1: bf16, 2: bf16 pair layout [M, 2N], 3: H2 rows (ldc in 2-byte units, >= 2N). */
int dod_op_split_h2(const float* x, int ld, int rows, int cols, void* out, void* wexp, void* stream);
int dod_op_linear_h2(const void* A, const void* W, const void* wexp, int M, int N, int K, const float* bias, const float* scale,
                     const float* resid, int ldr, void* out, int out_layout, int ldc, int act, void* stream);
/* LayerNorm folded into the GEMMs around it (the fast modes' schedule of modeling_dinov2.py:361-380, x -> LN -> linear): with
 * W' = W diag(gamma), c[n] = sum_k W'[n][k] and b' = b + W beta,   LN(x) W^T + b = rstd (x W'^T - mean c) + b'.
 *   consumer (stats, csum set): A holds the residual rows x themselves in the family's operand format, W holds W', bias holds b'; with part_in
 *     (a producer's group sums [M][ceil(K / 128)][2]; stats[m][0] = the shift it used) the consumer finishes the statistics itself in its
 *     epilogue (eps: the LayerNorm's) and the tiles of output column 0 write them to stats_out [M][2] (a different buffer than stats) -- what the
 *     forward does since round 4: no launch of its own merges the groups;
 *   producer (part set; out_layout 0 with a residual): beside the fp32 rows it writes them in the family's operand format to op_out
 *     ([M, N] bf16 / pair layout [M, 2N] / H2 rows) and (sum, sum of squares) of (row - shift[m][0]) per 128-column group to
 *     part [M][ceil(N / 128)][2]; shift [M][2] (or NULL = 0): a value near the row's mean -- the forward passes the row's previous
 *     statistics, which makes the one-pass sums free of cancellation;
 * dod_op_ln_finalize turns the groups into stats [M][2] = (mean, rstd); on entry stats[m][0] must hold the shift the producer used
 * (the same buffer, or zeros); dod_op_rowstats: x -> operand copy + stats (the first block).
 * family: DOD_PREC_BF16 / DOD_PREC_BF16X3 / DOD_PREC_FP16X2; A / W / wexp / out_layout as in dod_op_linear / _x3 / _h2 (bf16: lda = ldw = K). */
typedef struct dod_ln_fold {
  const void* stats;
  const float* csum;
  void* op_out;
  void* part;
  const void* shift;
  const void* part_in;
  void* stats_out;
  float eps;
} dod_ln_fold;
int dod_op_linear_ln(int family, const void* A, const void* W, const void* wexp, int M, int N, int K, const float* bias, const float* scale,
                     const float* resid, int ldr, void* out, int out_layout, int ldc, int act, const dod_ln_fold* ln, void* stream);
int dod_op_rowstats(const float* x, int rows, int D, float eps, void* op_out, int family, void* stats, void* stream);
int dod_op_ln_finalize(const void* part, int rows, int D, float eps, void* stats, void* stream);
/* out = LayerNorm(x + add) ; add may be NULL */
int dod_op_layernorm(const float* x, const float* add, const float* gamma, const float* beta, float eps,
                     int rows, int D, void* out, int out_dtype, void* stream);
/* qkv [B*N, 3*heads*64] bf16 -> ctx [B*N, heads*64] bf16 */
int dod_op_attention_bf16(const void* qkv, void* ctx, int B, int N, int heads, float scale, void* stream);
int dod_op_attention_f32(const float* q, const float* k, const float* v, float* o, int ldq, int ldk, int ldv,
                         int ldo, int Lq, int Lk, int B, int heads, int dh, float scale, void* stream);
/* proj [B*Q, ldp] = [ref logits(2) | offsets(Hd*P*2) | weight logits(Hd*P)], values [B*N, Hd*dh] -> out [B*Q, Hd*dh] */
int dod_op_deform_sample(const float* proj, int ldp, const float* values, int B, int Q, int N, int Hd, int P,
                         int dh, int h, int w, float* out, void* stream);
/* pos_in [G*G+1, D] -> pos_out [gh*gw+1, D] */
int dod_op_pos_resize(const float* pos_in, int G, int gh, int gw, int D, float* pos_out, void* stream);
/* img [B,3,H,W] -> cols [B*(H/p)*(W/p), Kp] of out_dtype */
int dod_op_im2col(const float* img, int B, int H, int W, int patch, int Kp, void* out, int out_dtype, void* stream);

/* ---- detection post-processing on device ---------------------------------------
 * Replaces the per-image / per-class / per-query loop of evaluate_coco, dino_detector/utils.py:195-233:
 * sigmoid scores (:198), background class 0 skipped (:211-212), score > threshold (0.05 at :215), cxcywh -> xyxy
 * (:86-87) -> COCO [x, y, w, h] (:225), one record per kept (image, class, query) in the reference's order
 * (image-major, then class 1..C-1, then query).  `query` is extra (the reference drops it). */
typedef struct dod_detection {
  int64_t image_id;       /* image_ids[b], or b when image_ids is NULL (target.get('image_id', i), utils.py:203) */
  int32_t category_id;    /* class index c, 1..C-1 */
  int32_t query;          /* query index q */
  float bbox[4];          /* x1, y1, x2 - x1, y2 - y1 (normalised, as the reference emits them) */
  float score;            /* sigmoid(logit) */
  int32_t reserved;
} dod_detection;          /* 40 bytes */
size_t dod_postprocess_workspace_bytes(int B, int Q, int C);
/* det: packed detections [B, Q, C+4] fp32 (the output of dod_forward), image_ids: device int64 [B] or NULL.
 * Writes min(total, max_out) records to `out` (device) and the TOTAL number of kept detections to *count (device
 * int64; > max_out means truncated).  Enqueued on `stream`; no host synchronisation. */
int dod_postprocess(const float* det, int B, int Q, int C, const int64_t* image_ids, float threshold,
                    dod_detection* out, int64_t max_out, int64_t* count, void* workspace, size_t workspace_bytes,
                    void* stream);

/* ---- input pipeline on device -------------------------------------------------------
 * Replaces the per-image host transform of dino_detector/train.py:584-587 -- torchvision Resize((R,R)) + ToTensor() on a
 * PIL image, i.e. Pillow's Image.resize(BILINEAR) (two-pass 8-bit fixed-point resample, antialiased when downscaling)
 * followed by uint8 -> float32 / 255 in CHW order -- for a ragged batch.  Bit-exact against Pillow.
 * src: concatenated uint8 HWC RGB images, image b at src + src_offs[b] with heights[b] x widths[b] pixels; tmp: the
 * horizontal-pass images, image b at tmp + tmp_offs[b] (heights[b] * out_w * 3 bytes); out: fp32 [B, 3, out_h, out_w].
 * All pointers are device pointers; max_h / max_w (host) bound the grid and the filter width (scale <= 15). */
int dod_preprocess(const uint8_t* src, const int64_t* src_offs, const int32_t* heights, const int32_t* widths, int B,
                   int max_h, int max_w, int out_h, int out_w, uint8_t* tmp, const int64_t* tmp_offs, float* out,
                   void* stream);

/* as dod_preprocess, stopping before ToTensor: out_hwc = uint8 [B, out_h, out_w, 3], the input of dod_forward_u8 */
int dod_preprocess_u8(const uint8_t* src, const int64_t* src_offs, const int32_t* heights, const int32_t* widths, int B,
                      int max_h, int max_w, int out_h, int out_w, uint8_t* tmp, const int64_t* tmp_offs, uint8_t* out_hwc,
                      void* stream);

/* ---- Hungarian-matcher cost matrices on device -------------------------------------
 * Replaces the per-image cost computation of HungarianMatcher.forward, dino_detector/matching.py:79-98 (focal class
 * cost :80-86, L1 box cost :89, GIoU cost :92-95 with utils.py:124-164, weighted sum :98) for the whole batch.
 * Targets are concatenated: labels int64 [G], gt_boxes fp32 [G,4] (cx,cy,w,h), gt_offsets int32 [B+1] (device).
 * cost (device fp32, G*Q floats): image b's matrix [Q, n_b] row-major at cost + gt_offsets[b]*Q -- exactly the C_valid
 * the reference hands to scipy (:102-105).  rows_from: -1 = image b's own predictions; k >= 0 = the predictions of
 * image k for EVERY image (the reference slices C[:num_queries] of a matrix built over all B*Q rows, i.e. k = 0). */
int dod_match_cost(const float* det, int B, int Q, int C, const int64_t* labels, const float* gt_boxes,
                   const int32_t* gt_offsets, int G, float w_class, float w_bbox, float w_giou, float alpha, float gamma,
                   int rows_from, float* cost, void* stream);

/* ---- native training step of the decoder + heads (first slice) ---------------------------------------
 * What `loss.backward()` at dino_detector/train.py:1101 needs from DETRDecoder.forward (detr_decoder.py:47-83) over the
 * weight-tied DeformableDecoderLayer (deformable_attention.py:215-268, tied at :284): a train-mode forward (dropout at the
 * reference's five sites, masks from a counter-based hash of `seed`) that tapes its activations, and the backward producing
 * the gradients of every decoder / head parameter and of `memory` (which the caller feeds on into the projection and the
 * LoRA-adapted blocks).  Stateless: the shapes come from `cfg`, the fp32 parameters from the caller's own tensors.
 * All pointers are device pointers.  `grads` has the layout of the parameters; its tensors are float ACCUMULATORS (the layers
 * share one set of weights; zero them for a plain gradient).  The same `dropout_p` / `seed` must be given to both calls. */
typedef struct dod_dec_train_params {
  const float *query_embed;                      /* [Q, Dd]            detr_decoder.py:15 */
  const float *class_w, *class_b;                /* [C, Dd], [C]       :40 */
  const float *bb0_w, *bb0_b, *bb2_w, *bb2_b;    /* bbox_embed.mlp.{0,2}  :39-41 */
  const float *in_proj_w, *in_proj_b, *out_proj_w, *out_proj_b;          /* self_attn (nn.MultiheadAttention) */
  const float *norm1_w, *norm1_b, *norm2_w, *norm2_b, *norm3_w, *norm3_b;
  const float *lin1_w, *lin1_b, *lin2_w, *lin2_b;
  const float *refp_w, *refp_b;                  /* reference_points_proj [2, Dd] */
  const float *off_w, *off_b, *aw_w, *aw_b;      /* cross_attn.sampling_offsets / attention_weights */
  const float *vp_w, *vp_b, *op_w, *op_b;        /* cross_attn.value_proj / output_proj */
} dod_dec_train_params;
size_t dod_decoder_train_tape_bytes(const dod_config* cfg, int B, int N);
size_t dod_decoder_train_workspace_bytes(const dod_config* cfg, int B, int N);
/* memory [B, N, Dd] -> detections [B, Q, C+4] (as dod_decoder_forward), activations into `tape` (keep it until the backward) */
int dod_decoder_train_forward(const dod_config* cfg, const dod_dec_train_params* params, const float* memory, int B, int N,
                              float dropout_p, uint64_t seed, float* detections, void* tape, size_t tape_bytes,
                              void* workspace, size_t workspace_bytes, void* stream);
/* d_detections [B, Q, C+4] -> grads (accumulated) and d_memory [B, N, Dd] (overwritten; may be NULL) */
int dod_decoder_train_backward(const dod_config* cfg, const dod_dec_train_params* params, const float* memory, int B, int N,
                               float dropout_p, uint64_t seed, const float* d_detections, const void* tape, size_t tape_bytes,
                               const dod_dec_train_params* grads, float* d_memory, void* workspace, size_t workspace_bytes,
                               void* stream);
const char* dod_decoder_train_last_error(void);

/* The nn.TransformerDecoder branch of DETRDecoder (use_deformable = False; detr_decoder.py:28-35, 62-69) in train() mode:
 * post-norm nn.TransformerDecoderLayer x L with UNTIED weights -- query self-attention, dense cross-attention of the Q queries over
 * all N memory tokens, ReLU FFN, three LayerNorms, dropout at torch's six sites (both attentions' probabilities, dropout1/2/3 and
 * the FFN's inner dropout) -- and the same heads.  Same contract as dod_decoder_train_*: `grads` has the layout of the parameters
 * (float accumulators), `layers` is a HOST array of nlayers entries holding device pointers. */
typedef struct dod_dense_layer_params {
  const float *sa_in_w, *sa_in_b, *sa_out_w, *sa_out_b;      /* self_attn: in_proj [3Dd, Dd], out_proj [Dd, Dd] */
  const float *ca_in_w, *ca_in_b, *ca_out_w, *ca_out_b;      /* multihead_attn: in_proj [3Dd, Dd] = q (queries) | k | v (memory) */
  const float *lin1_w, *lin1_b, *lin2_w, *lin2_b;            /* [F, Dd], [Dd, F] */
  const float *norm1_w, *norm1_b, *norm2_w, *norm2_b, *norm3_w, *norm3_b;
} dod_dense_layer_params;
typedef struct dod_dense_dec_train_params {
  int32_t nlayers, reserved;
  const dod_dense_layer_params* layers;
  const float *query_embed, *class_w, *class_b, *bb0_w, *bb0_b, *bb2_w, *bb2_b;
} dod_dense_dec_train_params;
size_t dod_dense_decoder_train_tape_bytes(const dod_config* cfg, int B, int N);
size_t dod_dense_decoder_train_workspace_bytes(const dod_config* cfg, int B, int N);
int dod_dense_decoder_train_forward(const dod_config* cfg, const dod_dense_dec_train_params* params, const float* memory, int B, int N,
                                    float dropout_p, uint64_t seed, float* detections, void* tape, size_t tape_bytes,
                                    void* workspace, size_t workspace_bytes, void* stream);
int dod_dense_decoder_train_backward(const dod_config* cfg, const dod_dense_dec_train_params* params, const float* memory, int B, int N,
                                     float dropout_p, uint64_t seed, const float* d_detections, const void* tape, size_t tape_bytes,
                                     const dod_dense_dec_train_params* grads, float* d_memory, void* workspace,
                                     size_t workspace_bytes, void* stream);

/* The rest of the trainable subset: the LoRA-adapted encoder blocks (the last two, dinov2_backbone.py:45-51), the final
 * LayerNorm and the projection (dinov2_backbone.py:33-37, 64-65) in train() mode.  x_in [B, N, D] is the residual stream in
 * front of the first adapted block (dod_backbone_prefix: everything before it is frozen and needs no autograd); the forward
 * writes the decoder memory [B, N, Dd] and tapes its activations, the backward turns d(memory) into the gradients of every
 * lora_A / lora_B (dino_detector/utils.py:46-70) and of the projection.  Frozen tensors (w, b, LayerNorm, LayerScale) are read
 * only; in `grads` only A, Bm, proj_w, proj_b are written (float accumulators, same layout).  GELU MLP (ViT-S/B/L) or SwiGLU
 * (ViT-g, cfg->swiglu: the fc1 / fc2 slots then hold mlp.weights_in [2F, D] / mlp.weights_out [D, F]); hidden <= 2048;
 * errors through dod_decoder_train_last_error(). */
typedef struct dod_lora_linear { const float *w, *b, *A, *Bm; } dod_lora_linear;        /* [out,in], [out], [r,in], [out,r] */
typedef struct dod_bb_block_params {
  const float *ln1_w, *ln1_b, *ln2_w, *ln2_b, *ls1, *ls2;                              /* frozen */
  dod_lora_linear q, k, v, o, fc1, fc2;                                                /* attention.attention.{query,key,value}, attention.output.dense, mlp.fc1, mlp.fc2 */
} dod_bb_block_params;
typedef struct dod_bb_tail_params {
  int32_t nblocks, reserved;
  const dod_bb_block_params* blocks;             /* HOST array of nblocks entries (device pointers inside) */
  const float *lnf_w, *lnf_b;                    /* final LayerNorm (frozen) */
  const float *proj_w, *proj_b;                  /* [Dd, D], [Dd]; NULL when the backbone has no projection */
} dod_bb_tail_params;
size_t dod_backbone_tail_tape_bytes(const dod_config* cfg, int B, int N, int nblocks);
size_t dod_backbone_tail_workspace_bytes(const dod_config* cfg, int B, int N, int nblocks);
int dod_backbone_tail_train_forward(const dod_config* cfg, const dod_bb_tail_params* params, const float* x_in, int B, int N,
                                    float* memory_out, void* tape, size_t tape_bytes, void* workspace, size_t workspace_bytes,
                                    void* stream);
int dod_backbone_tail_train_backward(const dod_config* cfg, const dod_bb_tail_params* params, int B, int N, const float* d_memory,
                                     const void* tape, size_t tape_bytes, const dod_bb_tail_params* grads, void* workspace,
                                     size_t workspace_bytes, void* stream);

/* Scratch of the GEMMs' wave-quantisation tail split (K-split partial slabs; gemm_pp.hip) and of the fp32 GEMM's K split (gemm_f32.hip, fixed size).  dod_finalize_weights reserves 64 MiB on the
 * current device; operator-level callers (tests, tools) reserve it themselves.  Never allocated inside a forward / stream capture. */
int dod_reserve_gemm_scratch(size_t bytes);
/* Test hooks.  Process-wide integer options a parity test sets for its own cases and hands back with -1 (= the shipped behaviour); they
 * are NOT environment variables -- nothing outside the calling process can change which kernel runs:
 *   "tailsplit"        GEMM wave-quantisation tail split: 0 off, 1 the shipped heuristic, 2 every qualifying shape
 *   "dec_fused_split"  0 = a split3 launch per decoder query-side linear (the round-2 schedule; bit-identical to the shipped one)
 *   "mha_chunk_images" training-step attention: images per pass (forces ragged passes on small shapes)
 *   "no_fused_patch"   1 = the explicit im2col + GEMM patch embedding (read by dod_finalize_weights)
 *   "ln_fold"          0 = LayerNorm kernels instead of the folded form (read by dod_finalize_weights)
 *   "deterministic"    1 = ordered reductions instead of fp32 atomics in the training step's weight gradients (also DINODET_DETERMINISTIC=1)
 *   "f32_ksplit"       fp32 GEMM K split across workgroups (gemm_f32.hip): 0 = never, 1 = also in dod_op_linear (shipped: the decoder's small linears in
 *                      every mode but the strict fp32 one)
 * dod_test_counter("tail_splits"): GEMM calls that took the tail-split path so far; "rem_cuts": GEMM calls whose short last round ran as a
 * launch of its own (gemm_bf16.hip); "f32_ksplits": fp32 GEMM launches that split K across workgroups (gemm_f32.hip); -1 for an unknown name.
 * The in-kernel time stamps, the register-only MFMA probes and every tile / schedule override of the tuning rounds exist only in
 * -DDINODET_TUNING builds (include/dinodet_tuning.h); the release library exports none of them. */
int dod_test_set_option(const char* name, int value);
long dod_test_counter(const char* name);

const char* dod_version(void);
/* ABI revision of this header: bumped whenever an exported signature or struct layout changes (round 2's dod_set_weight gained its
 * dtype argument at revision 2; revision 4 = this file: the dod_debug_* entry points left the release library, dod_test_* replaced the three
 * the tests use, the folded-LayerNorm operators arrived).  A C caller compiled against DOD_ABI_VERSION checks it once at load. */
#define DOD_ABI_VERSION 4
int dod_abi_version(void);
/* Devices visible to the HIP runtime libdinodet.so is bound to (<= 0: none / error).  The host uses it to
 * verify the library shares PyTorch's HIP runtime (pointers and streams cross this ABI). */
int dod_device_count(void);

#ifdef __cplusplus
}
#endif
#endif
