// Shared device helpers + launcher prototypes for the dino_detector forward path (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef unsigned short bf16_t;   // storage type of bf16 in memory

#define DOD_WAVE 64

// Tuning switches (tile overrides, A/B schedules, in-kernel time stamps, the register-only MFMA probes) exist only in -DDINODET_TUNING builds
// (`python -m dinov2_od_amd._build --tuning` -> lib/libdinodet_tuning.so, which tools/ load through DINODET_LIB).  The release library reads
// the few OPERATIONAL variables INTEGRATION.md lists and nothing else: no environment variable can change which kernel runs or what it computes.
// Test hooks (extern "C" dod_test_set_option(name, value), include/dinodet.h): process-wide integer options a parity test sets for its own cases
// and hands back (-1 = the shipped behaviour).  Not environment variables: nothing outside the calling process can flip them.
enum { DOD_OPT_TAILSPLIT = 0,         // GEMM wave-quantisation tail split: 0 off, 1 shipped heuristic, 2 every qualifying shape
       DOD_OPT_DEC_FUSED_SPLIT,       // decoder split operands written by their producers: 0 = a split3 launch per linear (round-2 schedule)
       DOD_OPT_MHA_CHUNK_IMAGES,      // training-step attention: images per pass (forces several ragged passes on small shapes)
       DOD_OPT_NO_FUSED_PATCH,        // 1 = pack for the explicit im2col + GEMM patch embedding instead of the fused kernel (read at dod_finalize_weights)
       DOD_OPT_LN_FOLD,               // 0 = LayerNorm kernels instead of the folded form (read at dod_finalize_weights)
       DOD_OPT_DETERMINISTIC,         // 1 = ordered reductions instead of fp32 atomics in the training step's weight gradients
       DOD_OPT_F32_KSPLIT,            // fp32 GEMM K split across workgroups (gemm_f32.hip): 0 = never, 1 = also for the operator dod_op_linear
       DOD_OPT_COUNT };
int dod_option(int which);            // dod_api.hip; -1 when unset
long gemm_tail_split_count();         // gemm_pp.hip: GEMM calls that took the tail-split path so far
long gemm_rem_cut_count();            // gemm_bf16.hip: GEMM calls whose short last round ran as a launch of its own

#ifdef DINODET_TUNING
#define DOD_TUNE_ENV(name) getenv(name)
#else
#define DOD_TUNE_ENV(name) (static_cast<const char*>(nullptr))
#endif

__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 b = (__bf16)f;   // v_cvt_pk_bf16_f32: RNE, NaN preserved
  return __builtin_bit_cast(bf16_t, b);
}
__device__ __forceinline__ float bf2f(bf16_t h) {
  return __builtin_bit_cast(float, ((uint32_t)h) << 16);
}
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  f32x2 v = {lo, hi};
  bf16x2 h = __builtin_convertvector(v, bf16x2);
  return __builtin_bit_cast(uint32_t, h);
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// four floats -> four OCP e4m3 bytes (v_cvt_pk_fp8_f32, round to nearest even), little-endian
__device__ __forceinline__ unsigned pack4_fp8(float a, float b, float c, float d) {
  int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
  return (unsigned)w;
}
// ---- block-scaled ("MX") e4m3 activations of the fp8 GEMM (gemm_fp8.hip gemm_fp8mx_256x128_kernel): one e8m0 byte per 32 elements along K,
// value 2^(byte - 127), the smallest power of two with amax_block / 448 <= 2^e (no element saturates); a zero block gets byte 1.
// Scale bytes [rows][2][K / 64]: block bi of a row (K-tile bi >> 1, half bi & 1) at byte (bi & 1) * (K / 64) + (bi >> 1) of its K / 32 bytes.
__device__ __forceinline__ unsigned mx_ebyte(float amax) {
  const unsigned b = __float_as_uint(amax * (1.0f / 448.0f));
  unsigned eb = ((b >> 23) & 0xffu) + ((b & 0x7fffffu) ? 1u : 0u);
  eb = eb < 1u ? 1u : (eb > 253u ? 253u : eb);
  return eb;
}
__device__ __forceinline__ float mx_inv_scale(unsigned eb) { return __uint_as_float((254u - eb) << 23); }      // 2^-(eb - 127)
__device__ __forceinline__ size_t mx_scale_off(int K, int bi) { return (size_t)(bi & 1) * (K >> 6) + (bi >> 1); }
// ---- "H2" operand format (fp16 main product + e4m3 compensation terms; gemm_pp.hip gemm_h2_256x256_kernel, the fp16x2 precision mode)
// x = h + l with h = fp16(x) (11 significant bits) and l = x - h (|l| <= 2^-11 |x|).  x.w ~ hx hw + [hx lw + lx hw]: the main product
// on the fp16 MFMA, the two cross terms -- which only need ~4 significant bits to sit below 2^-16 of the result -- as ONE
// v_mfma_scale_f32_32x32x64_f8f6f4 whose 64 k-slots per lane pair e4m3(hx) with e4m3(lw 2^(e+11)) and e4m3(lx 2^11) with
// e4m3(hw 2^e): every product carries the same factor 2^(e+11), which the weight row's E8M0 block scale 2^-(e+11) removes before
// the fp32 accumulation: 2.0 bf16-MFMA-equivalents per product instead of the 3 of the bf16 split form.
//   activation row of K elements (K % 32 == 0), 4K bytes: [ fp16 x K | per 16-k group: 16 x e4m3(h), 16 x e4m3(l 2^11) ]
//   weight row, 3K bytes + one exponent byte:             [ fp16 x K | K x e4m3(l 2^(e+11)) ],  e = floor(log2(448 / max|h|)) per row;
//     e4m3(h 2^e) is derived from the fp16 fragments in the kernel (v_cvt_scalef32_pk_fp8_f16: 25 % fewer weight bytes through
//     the per-CU global->LDS path, which -- not the MFMAs -- paces this kernel).  The activations keep their e4m3(h) bytes: that
//     conversion turns |h| > 464 into NaN (no saturating form exists), so it needs the producer's clamp.
// Values are clamped to the fp16 / e4m3 finite ranges (|x| <= 65504; beyond it the mode degrades, it does not overflow).
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
__device__ __forceinline__ float clampf_(float v, float lim) { return __builtin_fminf(__builtin_fmaxf(v, -lim), lim); }
// four values -> fp16 (2 dwords), e4m3 of the fp16 value times 2^ehi and of the remainder times 2^(ehi + 11) (one dword each)
__device__ __forceinline__ void h2_quad(float4 v, float shi, uint2& f16, unsigned& hi8, unsigned& lo8) {
  v.x = clampf_(v.x, 65504.f); v.y = clampf_(v.y, 65504.f); v.z = clampf_(v.z, 65504.f); v.w = clampf_(v.w, 65504.f);
  const f16x2 p0 = {(_Float16)v.x, (_Float16)v.y}, p1 = {(_Float16)v.z, (_Float16)v.w};
  f16.x = __builtin_bit_cast(unsigned, p0); f16.y = __builtin_bit_cast(unsigned, p1);
  const float hx = (float)p0.x, hy = (float)p0.y, hz = (float)p1.x, hw = (float)p1.y;
  const float slo = shi * 2048.0f;
  hi8 = pack4_fp8(clampf_(hx * shi, 448.f), clampf_(hy * shi, 448.f), clampf_(hz * shi, 448.f), clampf_(hw * shi, 448.f));
  lo8 = pack4_fp8(clampf_((v.x - hx) * slo, 448.f), clampf_((v.y - hy) * slo, 448.f), clampf_((v.z - hz) * slo, 448.f), clampf_((v.w - hw) * slo, 448.f));
}
// byte offset of the e4m3(h) bytes of column c (c % 4 == 0) inside an H2 activation row of K elements; the remainder bytes follow 16 bytes on
__device__ __forceinline__ size_t h2_off8(int K, int c) { return (size_t)2 * K + (size_t)(c >> 4) * 32 + (c & 15); }

// exact-erf GELU (HF ACT2FN["gelu"], modeling_dinov2.py:288-296)
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// GELU for the bf16 path: erf by Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, far below the bf16
// rounding of the result) with one v_rcp and one v_exp instead of libm erff's ~40-instruction body,
// which made the fc1 epilogue cost ~25 % of that GEMM.  The fp32 (strict) kernels keep erff.
// Two values per packed instruction (v_pk_fma_f32 / v_pk_mul_f32), rearranged so that no compare / select
// and no separate log2(e) scaling is left:  z' = |x| sqrt(log2(e) / 2),  t = 1 / (1 + k z'),  gelu(x) = (x + |x|) / 2 - |x| t p(t) / 2 * 2^(-z'^2)
// -- x * erf(x / sqrt 2) is even in x.  38 VALU instructions per four values instead of 52 (the fc1 epilogue runs alone on its CU).
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_t gelu_fast2(f32x2_t x) {
  f32x2_t ax; ax.x = __builtin_fabsf(x.x); ax.y = __builtin_fabsf(x.y);
  const f32x2_t z = ax * 0.8493218002880191f;
  const f32x2_t d = __builtin_elementwise_fma(z, (f32x2_t)(0.2727374808792225f), (f32x2_t)(1.0f));
  f32x2_t t; t.x = __builtin_amdgcn_rcpf(d.x); t.y = __builtin_amdgcn_rcpf(d.y);
  f32x2_t p = __builtin_elementwise_fma(t, (f32x2_t)(0.5f * 1.061405429f), (f32x2_t)(0.5f * -1.453152027f));
  p = __builtin_elementwise_fma(p, t, (f32x2_t)(0.5f * 1.421413741f));
  p = __builtin_elementwise_fma(p, t, (f32x2_t)(0.5f * -0.284496736f));
  p = __builtin_elementwise_fma(p, t, (f32x2_t)(0.5f * 0.254829592f));
  const f32x2_t nz2 = -z * z;
  f32x2_t e; e.x = __builtin_amdgcn_exp2f(nz2.x); e.y = __builtin_amdgcn_exp2f(nz2.y);
  const f32x2_t q = ax * (t * p);
  const f32x2_t r = __builtin_elementwise_fma(ax, (f32x2_t)(0.5f), x * 0.5f);
  return __builtin_elementwise_fma(-q, e, r);
}
__device__ __forceinline__ void gelu_fast4(float4& v) {
  f32x2_t a = {v.x, v.y}, b = {v.z, v.w};
  a = gelu_fast2(a); b = gelu_fast2(b);
  v.x = a.x; v.y = a.y; v.z = b.x; v.w = b.y;
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// ----------------------------------------------------------------------------- epilogue description
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2, ACT_SIGMOID = 3 };

// C[m][n] = act( (A W^T)[m][n] + bias[n] ) * scale[n] + resid[m'][n]   (each optional)
// Output row m' = m unless rows_per_img > 0, in which case (patch-embed, K1/K2 fusion):
//   b = m / rows_per_img, p = m % rows_per_img, m' = b * out_rows_per_img + 1 + p and pos[(1+p)][n] is added.
struct GemmEpi {
  const float* bias;      // [N] or null
  const float* scale;     // [N] LayerScale lambda or null
  const float* resid;     // fp32 [M', ldr] or null (may alias out_f32)
  int ldr;
  float* out_f32;         // one of out_f32 / out_bf16 non-null
  bf16_t* out_bf16;
  int ldc;
  int act;
  const float* pos;       // [out_rows_per_img, N] (patch-embed only)
  int rows_per_img;
  int out_rows_per_img;
  int out_split;          // with out_bf16: > 0 bf16x3 activation layout [hi | hi | lo], each out_split columns wide (ldc = 3*out_split);
                          // < 0 pair layout [hi | lo], each -out_split wide (ldc = -2*out_split)
  const float* a_scale;   // fp8 GEMM only: per-row (token) dequant scale of A, [M]; null otherwise
  const float* w_scale;   // fp8 GEMM only: per-output-feature dequant scale of W, [N]
  unsigned char* out_bs;       // fp8 GEMM with glu only: the gated rows leave as e4m3 (out_bf16 = byte rows, pitch ldc bytes) with these e8m0 block scales [M][2][F / 64]
  const unsigned char* a_bs;   // fp8 GEMM only: e8m0 BLOCK scales of A (one byte per 32 elements along K, layout [M][2][K / 64]: gemm_fp8.hip) instead of a_scale
  const unsigned char* w_bs;   // fp8 GEMM with a_bs only: e8m0 block scales of W in the same layout [N][2][K / 64] instead of w_scale
  int out_h2;             // with out_bf16: H2 activation rows (above) of N columns at row pitch ldc (2-byte units, >= 2N)
  const unsigned char* h2_wexp;   // H2 GEMM only: E8M0 byte (127 - e) of every weight row's e4m3 scale 2^e, [N]
  int ksplit;             // ping-pong / H2 kernels only: > 1 = the grid's y index is a K slice of kslice_len k; the slice's fp32 partial
  int kslice_len;         //   product goes to out_f32 + slice * kslice_stride (no other epilogue term may be set)
  long long kslice_stride;
  int glu;                // with out_bf16 (plain layout): the GEMM's columns are INTERLEAVED SwiGLU pairs (2i: x1_i, 2i + 1: x2_i;
                          //   Dinov2SwiGLUFFN, modeling_dinov2.py:310-314) and the epilogue writes silu(x1_i) * x2_i to column i: N / 2 columns at
                          //   row pitch ldc -- the gate costs no pass over the [M, 2F] intermediate
  // ---- LayerNorm folded into the GEMMs around it (round 4; modeling_dinov2.py:361-380: x -> LN -> linear).  With W' = W diag(gamma),
  // c[n] = sum_k W'[n][k] and b' = b + W beta:   LN(x) W^T + b  =  rstd (x W'^T - mean c) + b'  -- the linear reads the residual row x itself
  // (in its operand format) and normalises in its epilogue, so no LayerNorm pass reads the fp32 stream.
  //   producer (the in-place residual epilogue of out-proj / fc2): beside the fp32 row it writes the row in the next GEMM's operand format
  //   and, per 128-column group g, (sum, sum of squares) of the new row minus a per-row shift to ln_part[m * ln_npart + g] (ln_finalize_kernel);
  void* ln_op;            //   operand copy of the new residual rows, or null
  int ln_op_kind;         //   LNOP_BF16 (pitch ln_op_ld bf16) / LNOP_PAIR ([hi | lo], each ln_op_ld / 2 wide) / LNOP_H2 (H2 rows, pitch 2 * ln_op_ld bytes)
  int ln_op_ld;
  float2* ln_part;        //   [M][ln_npart] (sum, sum of squares) of (row - shift) per group
  int ln_npart;           //   = ceil(N / 128)
  const float2* ln_shift; //   [M]: .x = the shift of row m (its previous mean; null = 0) -- ln_finalize_kernel reads the same value
  //   consumer (QKV, fc1 / weights_in): v = (acc - mean[m] c[n]) rstd[m] before the bias
  const float2* ln_stats; //   [M] (mean, rstd), or null.  With ln_part_in: .x = the shift the producer subtracted, and the consumer finishes the statistics
  const float* ln_c;      //   [N]
  const float2* ln_part_in; // [M][ln_npart] the producer's group sums, or null (ln_stats final: the first block, written by rowstats_kernel).  The tile
  float2* ln_stats_out;   //   column n0 == 0 also writes the finished (mean, rstd) here -- ANOTHER buffer than ln_stats (other tiles still read the shift
  float ln_eps;           //   there): the next producer's shift.  No kernel of its own merges the groups (24 launches of 3-5 us per forward).
};
enum { LNOP_BF16 = 1, LNOP_PAIR = 2, LNOP_H2 = 3 };
// The epilogue of rows m0.. of the same GEMM as a launch of its own (row 0 of the result = row m0 of `e`): every per-row pointer moves down m0
// rows.  Only for epilogues whose rows are independent of their index (no rows_per_img / pos map, no K slices, no fp8 scales): gemm_epi_rows_ok.
inline bool gemm_epi_rows_ok(const GemmEpi& e) {
  return e.rows_per_img == 0 && !e.pos && e.ksplit <= 1 && !e.a_scale && !e.a_bs && !e.out_bs;
}
inline GemmEpi gemm_epi_rows(const GemmEpi& e, int m0) {
  GemmEpi r = e;
  const size_t m = (size_t)m0;
  if (r.resid) r.resid += m * r.ldr;
  if (r.out_f32) r.out_f32 += m * r.ldc;
  if (r.out_bf16) r.out_bf16 += m * r.ldc;
  if (r.ln_op) r.ln_op = reinterpret_cast<char*>(r.ln_op) + m * (size_t)r.ln_op_ld * 2;      // every kind: ln_op_ld 2-byte units per row (gemm_epi.h ln_store_op)
  if (r.ln_part) r.ln_part += m * r.ln_npart;
  if (r.ln_shift) r.ln_shift += m;
  if (r.ln_stats) r.ln_stats += m;
  if (r.ln_part_in) r.ln_part_in += m * r.ln_npart;
  if (r.ln_stats_out) r.ln_stats_out += m;
  return r;
}

// ----------------------------------------------------------------------------- launchers (all enqueue on `s`, no sync)
// gemm: C = A[M,K] (row-major, lda) x W[N,K]^T (row-major, ldw)
int launch_gemm_bf16(const bf16_t* A, int lda, const bf16_t* W, int ldw, int M, int N, int K,
                     const GemmEpi& e, hipStream_t s);
int launch_gemm_f32(const float* A, int lda, const float* W, int ldw, int M, int N, int K,
                    const GemmEpi& e, hipStream_t s, bool allow_ksplit = false);      // allow_ksplit: gemm_f32.hip "K split across workgroups" (never in the strict fp32 mode)
int gemm_f32_ksplit_reserve();      // scratch of that split; outside any stream capture
long gemm_f32_ksplit_count();
// The same exact-fp32 MFMA main loop for the training step's products (dec_train.hip): either operand may be given k-major
// ([K, rows]: the transposed products of a backward need no transposed copies), a two-level batch (image, head) walks strided views
// (attention scores / context / their adjoints as batched GEMMs), the K range may be split over grid.z with an atomic accumulate.
//   C[z] (+)= alpha * A[z] W[z]^T (+ bias, activation);   z = zb * hb + zh,  X[z] = X + zb * x_sb + zh * x_sh   (strides in floats)
struct GemmF32X {
  const float* A; const float* W; float* C; const float* bias;
  int lda, ldw, ldc, M, N, K;
  long long a_sb, a_sh, w_sb, w_sh, c_sb, c_sh;
  int batch, hb;            // batch = number of z, hb = inner (head) count; 1, 1 for a plain product
  int a_kmajor, w_kmajor;   // operand stored [K, rows] (rows contiguous) instead of [rows, K]
  float alpha;
  int accumulate;           // C += instead of C =
  int act;                  // ACT_* (not with ksplit > 1)
  int ksplit;               // > 1: K split over grid.z, partial products added atomically (implies accumulate; C must hold the addend)
};
int launch_gemm_f32x(const GemmF32X& g, hipStream_t s);
// fp8 (OCP e4m3) operands, one byte per element, K contiguous; e.a_scale / e.w_scale are the dequant scales
// (e.a_bs instead of e.a_scale: block-scaled activations, K % 256 == 0)
int launch_quant_mx_fp8(const void* x, int in_bf16, int ld, int rows, int cols, unsigned char* q, int ldq, unsigned char* bs, hipStream_t s);
int launch_gemm_fp8(const unsigned char* A, int lda, const unsigned char* W, int ldw, int M, int N, int K,
                    const GemmEpi& e, hipStream_t s);
// rows x cols fp32 or bf16 -> e4m3 with one scale per row: scale[r] = amax_r / 448 (1 if the row is all zero),
// q = rne_e4m3(x / scale[r]).  in_bf16: input element type.  ld in elements.
// bf16x3-mode attention: qkv2 [B*N, 6*D] = [hi(q|k|v) | lo(q|k|v)] -> ctx2 [B*N, 2*D] = [hi | lo]; head_dim 64
// ctx_h2: write the context in the H2 operand format (row pitch 4*D bytes) instead of the pair layout
int launch_attn_x3(const bf16_t* qkv2, bf16_t* ctx3, int B, int N, int heads, float scale, hipStream_t s, int ctx_h2 = 0);
// SwiGLU (silu(a) * b of the bf16 [rows, 2*Fh] input) -> e4m3 [rows, Fh] + per-row scale
int launch_swiglu_fp8(const bf16_t* in, int rows, int Fh, unsigned char* q, float* scale, hipStream_t s);
int launch_quant_rows_fp8(const void* x, int in_bf16, int ld, int rows, int cols, unsigned char* q, int ldq, float* scale,
                          hipStream_t s);

// rows x D LayerNorm, optional pre-add (y = LN(x + add)), fp32 statistics; out bf16 or fp32
// out_fp8 / out_scale (both or neither): e4m3 row + per-row scale (amax / 448) instead of the fp32 / bf16 output;
// out_split3: split-product pair layout [hi | lo] (row pitch 2*D) instead
// h2: out_split3 is written in the H2 operand format (row pitch 4*D bytes) instead
int launch_layernorm(const float* x, const float* add, const float* gamma, const float* beta, float eps,
                     int rows, int D, float* out_f32, bf16_t* out_bf16, hipStream_t s,
                     unsigned char* out_fp8 = nullptr, float* out_scale = nullptr, bf16_t* out_split3 = nullptr, int h2 = 0,
                     bf16_t* out_a3 = nullptr,       // out_a3 (with out_f32 only): also the bf16x3 activation layout [hi | hi | lo], row pitch 3*D
                     unsigned char* out_bs = nullptr);   // with out_fp8 (and no out_scale): block-scaled e4m3 rows, e8m0 bytes [rows][2][D / 64]
// ---- folded LayerNorm (GemmEpi::ln_*): row statistics only.
// rows of x -> operand copy of x ITSELF (kind LNOP_*, as GemmEpi::ln_op) + (mean, rstd) per row: what the first block's QKV needs (its
// producer is the patch embedding, not a residual GEMM)
int launch_rowstats(const float* x, int rows, int D, float eps, void* op, int op_kind, float2* stats, hipStream_t s);
// (sum, sum of squares) of (row - shift) per 128-column group [rows][npart] -> stats[row] = (mean, rstd); IN: stats[row].x = the shift the
// producer used (GemmEpi::ln_shift pointed here, or zeros)
int launch_ln_finalize(const float2* part, int npart, int rows, int D, float eps, float2* stats, hipStream_t s);
// pack time: Wout[n][k] = W[n][k] gamma[k],  bias_out[n] = bias_in[n] + sum_k W[n][k] beta[k]   (Wout may alias W, bias_out bias_in)
int launch_ln_fold(const float* W, int rows, int cols, const float* gamma, const float* beta, const float* bias_in, float* Wout, float* bias_out, hipStream_t s);
// c[n] = sum_k W[n][k]; round_bf16: of the bf16-rounded elements (what the single-pass bf16 MFMA multiplies)
int launch_rowsum(const float* W, int rows, int cols, int round_bf16, float* c, hipStream_t s);

// backbone attention, bf16 MFMA flash kernel, head_dim 64.  qkv [B*N, 3*D] bf16 -> ctx [B*N, D] bf16
// ctx_bs != null (fp8 mode): ctx is a BYTE buffer of block-scaled e4m3 rows [B*N, D], ctx_bs its e8m0 bytes [B*N][2][D / 64]
int launch_attn_bf16(const bf16_t* qkv, bf16_t* ctx, int B, int N, int heads, float scale, hipStream_t s, unsigned char* ctx_bs = nullptr);

// generic fp32 attention (strict backbone, decoder self-attn, dense cross-attn). head_dim <= 128, %4 == 0
struct AttnF32 {
  const float *q, *k, *v; float* o;
  int ldq, ldk, ldv, ldo;     // row strides (floats)
  int Lq, Lk;                 // rows per batch item
  int B, heads, dh;
  float scale;
  bf16_t* o3 = nullptr;       // optional: the output also as a bf16x3 activation operand [hi | hi | lo] of heads*dh columns each (row pitch
                              // 3*heads*dh), what the decoder's next query-side linear reads -- saves its split3 launch
  float* lse = nullptr;       // optional [B, heads, Lq, 2]: (max_j(s c), sum_j 2^(s c - max)) of every score row, c = scale log2(e) -- what
                              // the flash-style adjoint (launch_attn_f32_bwd) recomputes the probabilities from; MFMA kernel only
};
int launch_attn_f32(const AttnF32& a, hipStream_t s);
// Adjoint of launch_attn_f32 without materialised scores (attn_f32m.hip; head_dim 64, exact-fp32 MFMA): q | k | v and their gradients
// as strided [rows, ld] views like the forward's, o / d_o [B*Lq, ldo], lse from the forward, delta [B, heads, Lq] scratch.
// Returns 2 when the shape is not taken: callers decide BEFORE sizing their scratch (dec_train.hip tail_flash_bwd is the one predicate and
// implies every check made here), so behind that predicate a 2 is a programming error, not a fallback.
struct AttnF32Bwd {
  const float *q, *k, *v, *o, *d_o, *lse;
  float *dq, *dk, *dv, *delta;
  int ldq, ldk, ldv, ldo, lddq, lddk, lddv;
  int Lq, Lk, B, heads, dh;
  float scale;
};
int launch_attn_f32_bwd(const AttnF32Bwd& a, hipStream_t s);

// patch im2col: img [B,3,H,W] fp32 -> A [B*gh*gw, Kp] (k = c*p*p + i*p + j, zero padded to Kp)
int launch_im2col(const float* img, int B, int H, int W, int patch, int Kp, float* out_f32, bf16_t* out_bf16, hipStream_t s);
// fused patch embedding with implicit im2col (patch_embed.hip): W [D,3,p,p] -> kernel order (launch_patch_pack), then
// img (fp32 CHW, or uint8 HWC with ToTensor's /255 applied in the load) -> x[b][1 + m][:] = conv + bias + pos[1 + m]
int launch_patch_pack(const float* W, int D, int p, bf16_t* out, int x3, hipStream_t s);
int launch_patch_embed(const void* img, int u8, int B, int H, int W, int p, const bf16_t* Wp, int x3, const float* bias, const float* pos,
                       float* out, int D, hipStream_t s);
// x[b][0][:] = cls + pos[0]
int launch_cls_row(const float* cls, const float* pos, float* x, int B, int N, int D, hipStream_t s);
// bicubic resize of the patch position table (torch upsample_bicubic2d, A=-0.75, align_corners=False)
int launch_pos_resize(const float* pos_in, int G, int gh, int gw, int D, float* pos_out, hipStream_t s);
// dst [2F, cols] = rows of src [2F, cols] interleaved: dst[2i] = src[i], dst[2i + 1] = src[F + i] (SwiGLU pairs adjacent: GemmEpi::glu)
int launch_interleave_halves(const float* src, float* dst, int F, int cols, hipStream_t s);
// silu(x1)*x2 over [rows, 2*Fh] -> [rows, Fh]
int launch_swiglu(const float* in_f32, const bf16_t* in_bf16, int rows, int Fh, float* out_f32, bf16_t* out_bf16, hipStream_t s);
int launch_cast_bf16(const float* in, bf16_t* out, size_t n, hipStream_t s);
// dst[r][c] (ld_dst) = src[r][c] for a [rows, cols] fp32 block, zero-filling cols..cols_pad
int launch_copy2d(const float* src, int ld_src, float* dst, int ld_dst, int rows, int cols, int cols_pad, hipStream_t s);
// W' = W + alpha * B[out,r] A[r,in]
int launch_lora_merge(const float* W, const float* A, const float* Bm, float alpha, int out_f, int in_f, int r, float* dst, hipStream_t s);
// bf16x3 split: fp32 [rows,K] -> bf16 [rows,3K]; mode 0 = [hi|hi|lo] (activations), 1 = [hi|lo|hi] (weights)
int launch_split3(const float* in, int ld_in, bf16_t* out, int rows, int K, int mode, hipStream_t s);
int launch_split2(const float* in, int ld_in, bf16_t* out, int rows, int K, hipStream_t s);   // [hi | lo], pitch 2K
// fp32 [rows, K] -> H2 operand rows.  wexp == null: activation form (4K bytes per row); else the weight form (3K bytes per row,
// wexp[row] = 127 - e)
int launch_split_h2(const float* in, int ld_in, void* out, int rows, int K, unsigned char* wexp, hipStream_t s);
// H2 GEMM (gemm_pp.hip): A [M, K] activation rows (pitch lda BYTES >= 4K), W [N, K] weight rows (pitch ldw BYTES >= 3K), e.h2_wexp set
int launch_gemm_h2(const void* A, int lda, const void* W, int ldw, int M, int N, int K, const GemmEpi& e, hipStream_t s);
// Wave-quantisation tail of the 256x256-tile GEMMs (gemm_pp.hip): when tiles % CUs leaves a short last round (1029 tiles on 256
// CUs: a fifth round for five tiles, +24 %), the rows of that round are cut off the main launch and computed by a K-SPLIT launch
// (every tile of the remainder as S slices on S CUs, fp32 partials to a scratch buffer) plus a reduce + epilogue launch.
// kind: 0 plain bf16 (ppm kernel), 1 split product, 2 H2.  Returns 0 when done, -1 when the shape does not qualify (caller
// launches normally), > 0 on error.  gemm_tail_reserve(bytes): allocate the scratch outside any stream capture.  The scratch is one
// slab per launching stream (four per device, least-recently-used first: a forward may run as two concurrent micro-batches on two
// streams, engine.py).  The slices are summed in a fixed order, so results do not depend on scheduling.
int gemm_tail_split(int kind, const void* A, int lda, const void* W, int ldw, int M, int N, int K, const GemmEpi& e, hipStream_t s);
int gemm_tail_reserve(size_t bytes);
// split-product GEMM on pair-layout operands A2 [M, 2K], W2 [N, 2K] (gemm_x3.hip)
int launch_gemm_bf16_k64(const bf16_t* A, int lda, const bf16_t* W, int ldw, int M, int N, int K, const GemmEpi& e, hipStream_t s);
int gemm_tile_mode();   // tile-order mode word of the 256-row kernels (gemm_x3.hip; gemm_epi.h tile_map)
// 256x256x64 ping-pong kernel (gemm_pp.hip): K % 64 == 0
int launch_gemm_bf16_ppm(const bf16_t* A, int lda, const bf16_t* W, int ldw, int M, int N, int K, const GemmEpi& e, hipStream_t s);
int launch_gemm_x3_pp(const bf16_t* A2, int lda, const bf16_t* W2, int ldw, int M, int N, int K, const GemmEpi& e, hipStream_t s);
int launch_gemm_x3(const bf16_t* A2, int lda, const bf16_t* W2, int ldw, int M, int N, int K, const GemmEpi& e, hipStream_t s);
// tgt[b][q][:] = query_embed[q][:]
int launch_bcast_rows(const float* src, float* dst, int B, int rows, int D, hipStream_t s);

// K12/K13/K15: proj [B*Q, ldp] = [ref logits(2) | offsets(Hd*P*2) | weight logits(Hd*P)], values fp32 [B*N, Dd]
int launch_deform_sample(const float* proj, int ldp, const float* values, int B, int Q, int N, int Hd, int P, int dh,
                         int h, int w, float* out, hipStream_t s, int proj_shared = 0, bf16_t* out3 = nullptr);   // out3: also [hi | hi | lo], pitch 3*Hd*dh
