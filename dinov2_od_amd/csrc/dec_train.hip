// Native training step of the decoder + heads (first slice): train-mode forward with a tape and
// the backward of DETRDecoder.forward (detr_decoder.py:47-83) over the weight-tied DeformableDecoderLayer
// (deformable_attention.py:215-268, :284) -- what `loss.backward()` at train.py:1101 computes for the decoder, the heads and
// d(memory) (which then flows into the projection / LoRA blocks).  fp32 throughout (master weights, exact-fp32 MFMA GEMMs).
//
//   forward  : query tiling -> per layer { MHA self-attention (dropout on the probabilities, nn.MultiheadAttention) -> +dropout1
//              -> LN1 -> sigmoid reference points, offsets, point weights -> bilinear gather -> output_proj -> +dropout2 -> LN2
//              -> linear1 -> ReLU -> dropout3 -> linear2 -> +dropout4 -> LN3 } -> class / box heads.
//   backward : the exact adjoint of each step; weight gradients ACCUMULATE (the layers share one set of weights, and the
//              caller may accumulate over micro-batches); d(values) is a float-atomic scatter-add of the same four corners the
//              forward gathered (floor / clamp carry no gradient; torch.clamp passes it inside [0, 1] inclusive).
// Linear backward runs on the fp32 MFMA GEMM of the forward in its k-major-operand form (gemm_f32.hip, launch_gemm_f32x): dX = dY W
// reads W as the k-major operand, dW += dY^T X reads both operands k-major with the row dimension split over the grid (atomic
// accumulate) -- no transposed copies.  Self-attention (forward and adjoint) is batched GEMMs over (image, head) around one row
// kernel.  Dropout masks come from a counter-based hash of (seed, site, element): the backward regenerates them, nothing but
// activations is taped.
#include <mutex>
#include "dod_common.h"
#include "../../include/dinodet.h"

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

namespace {

inline size_t up4(size_t x) { return (x + 3) & ~(size_t)3; }
inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

// ------------------------------------------------------------------------------------------------ RNG
__device__ __forceinline__ float u01(unsigned long long key, unsigned long long idx) {
  unsigned long long z = key + idx * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (float)(z >> 40) * (1.0f / 16777216.0f);
}
__host__ __device__ inline unsigned long long site_key(unsigned long long seed, int layer, int site) {
  return seed * 0xD1342543DE82EF95ull + (unsigned long long)(layer * 8 + site + 1) * 0x9E3779B97F4A7C15ull;
}

// ------------------------------------------------------------------------------------------------ deterministic mode
// DINODET_DETERMINISTIC=1 (or the test option "deterministic"): every reduction that the fast step spreads over workgroups and merges with
// fp32 atomics -- K-split weight / activation gradient products, bias column sums, LayerNorm and LoRA parameter gradients, the scatter of
// the deformable sampling adjoint and its shared reference-logit columns -- runs in a FIXED order instead: run-to-run bit-identical
// gradients (the fast step agrees with itself to ~1e-6: r3_t8.log), at roughly twice the step time.
static bool det_mode() {
  const int o = dod_option(DOD_OPT_DETERMINISTIC);
  if (o >= 0) return o != 0;
  static const bool env = [] { const char* v = getenv("DINODET_DETERMINISTIC"); return v && v[0] == '1'; }();
  return env;
}
// partial-sum scratch of the ordered reductions (per device, grown on demand; the training step is never stream-captured)
static float* det_scratch(size_t floats) {
  static std::mutex mu;
  static float* buf[16] = {};
  static size_t cap[16] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  std::lock_guard<std::mutex> lk(mu);
  if (cap[dev] < floats) {
    if (buf[dev]) { (void)hipDeviceSynchronize(); (void)hipFree(buf[dev]); buf[dev] = nullptr; cap[dev] = 0; }
    if (hipMalloc((void**)&buf[dev], floats * 4) != hipSuccess) return nullptr;
    cap[dev] = floats;
  }
  return buf[dev];
}
// dst[c] += part[0][c] + part[1][c] + ... in that order (part: [nparts][cols])
__global__ void ordered_add_kernel(const float* __restrict__ part, int nparts, int stride, int cols, float* __restrict__ dst) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= cols) return;
  float acc = 0.f;
  for (int i = 0; i < nparts; ++i) acc += part[(size_t)i * stride + c];
  dst[c] += acc;
}

// ------------------------------------------------------------------------------------------------ small kernels
// dst[c] += sum_r src[r][c]
__global__ void colsum_kernel(const float* __restrict__ src, int ld, int rows, int cols, float* __restrict__ dst) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int w = threadIdx.x >> 6;
  float acc = 0.f;
  if (c < cols)
    for (int r = blockIdx.y * 4 + w; r < rows; r += gridDim.y * 4) acc += src[(size_t)r * ld + c];
  __shared__ float red[4][64];
  red[w][threadIdx.x & 63] = acc;
  __syncthreads();
  if (w == 0 && c < cols) atomicAdd(dst + c, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}
int colsum_add(const float* src, int ld, int rows, int cols, float* dst, hipStream_t s) {
  int gy = (rows + 63) / 64; gy = gy < 1 ? 1 : (gy > 128 ? 128 : gy);
  if (det_mode()) gy = 1;      // one workgroup per 64 columns walks every row: a single adder per column
  hipLaunchKernelGGL(colsum_kernel, dim3((cols + 63) / 64, gy), dim3(256), 0, s, src, ld, rows, cols, dst);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}

// out = a + keep(b) / (1 - p)   (p == 0: plain add); also used with a == nullptr (out = dropped b)
__global__ void dropout_add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, size_t n,
                                   float p, unsigned long long key) {
  const float inv = 1.0f / (1.0f - p);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float v = b[i];
    if (p > 0.f) v = u01(key, i) >= p ? v * inv : 0.f;
    out[i] = a ? a[i] + v : v;
  }
}
int dropout_add(const float* a, const float* b, float* out, size_t n, float p, unsigned long long key, hipStream_t s) {
  const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  hipLaunchKernelGGL(dropout_add_kernel, dim3(blocks), dim3(256), 0, s, a, b, out, n, p, key);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}
// g = dy * (y > 0 ? 1 : 0) [* dropout mask / (1 - p)]
__global__ void relu_drop_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ g, size_t n, float p,
                                     unsigned long long key) {
  const float inv = 1.0f / (1.0f - p);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float v = dy[i];
    if (p > 0.f) v = u01(key, i) >= p ? v * inv : 0.f;
    g[i] = y[i] > 0.f ? v : 0.f;
  }
}
__global__ void add_inplace_kernel(float* __restrict__ a, const float* __restrict__ b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] += b[i];
}
int add_inplace(float* a, const float* b, size_t n, hipStream_t s) {
  const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  hipLaunchKernelGGL(add_inplace_kernel, dim3(blocks), dim3(256), 0, s, a, b, n);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}
// dz[r][0..3] = dbox[r][0..3] * s (1 - s), s = the forward's sigmoid output
__global__ void sigmoid_bwd4_kernel(const float* __restrict__ dbox, int ldd, const float* __restrict__ box, int ldb, float* __restrict__ dz, int rows) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * 4) return;
  const int r = i >> 2, c = i & 3;
  const float sg = box[(size_t)r * ldb + c];
  dz[i] = dbox[(size_t)r * ldd + c] * sg * (1.0f - sg);
}
// exact-erf GELU backward (modeling_dinov2.py:288-296): g = dy * (Phi(x) + x phi(x))
__global__ void gelu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ pre, float* __restrict__ g, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float x = pre[i];
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * expf(-0.5f * x * x);
    g[i] = dy[i] * (cdf + x * pdf);
  }
}
__global__ void gelu_fwd_kernel(const float* __restrict__ pre, float* __restrict__ h, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) h[i] = gelu_erf(pre[i]);
}
// SwiGLU (Dinov2SwiGLUFFN, modeling_dinov2.py:300-314): pre [rows, 2F] = [x1 | x2] -> h = silu(x1) * x2
__global__ void swiglu_fwd_kernel(const float* __restrict__ pre, float* __restrict__ h, size_t rows, int F) {
  const size_t n = rows * (size_t)F;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = i / F; const int c = (int)(i - r * F);
    const float x1 = pre[r * 2 * F + c], x2 = pre[r * 2 * F + F + c];
    h[i] = x1 / (1.0f + expf(-x1)) * x2;
  }
}
// its adjoint: d(x1) = dh x2 s (1 + x1 (1 - s)), d(x2) = dh x1 s, s = sigmoid(x1); written as [d(x1) | d(x2)] rows of 2F
__global__ void swiglu_bwd_kernel(const float* __restrict__ dh, const float* __restrict__ pre, float* __restrict__ dpre, size_t rows, int F) {
  const size_t n = rows * (size_t)F;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = i / F; const int c = (int)(i - r * F);
    const float x1 = pre[r * 2 * F + c], x2 = pre[r * 2 * F + F + c], g = dh[i];
    const float sg = 1.0f / (1.0f + expf(-x1));
    dpre[r * 2 * F + c] = g * x2 * sg * (1.0f + x1 * (1.0f - sg));
    dpre[r * 2 * F + F + c] = g * x1 * sg;
  }
}
// out[i] = a[i] * v[i % D]   (LayerScale on the gradient)
__global__ void colscale_kernel(const float* __restrict__ a, const float* __restrict__ v, float* __restrict__ out, size_t n, int D) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = a[i] * v[i % D];
}
// dq[q][:] += sum_b d[b][q][:]
__global__ void batch_sum_kernel(const float* __restrict__ d, float* __restrict__ dq, int B, size_t per) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < per; i += (size_t)gridDim.x * blockDim.x) {
    float acc = 0.f;
    for (int b = 0; b < B; ++b) acc += d[(size_t)b * per + i];
    dq[i] += acc;
  }
}

// ------------------------------------------------------------------------------------------------ LayerNorm backward
// x: the pre-norm input (t = residual + branch), dy: gradient of the LayerNorm output.  dx per row; dgamma / dbeta accumulated
// per wave over its rows, then one float atomic per column and wave.
template <int MAXC>       // 64-column chunks per lane: 16 (D <= 1024) or 32 (D <= 2048: ViT-g's 1536)
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ dy,
                                                     float eps, int rows, int D, float* __restrict__ dx, float* __restrict__ dgamma,
                                                     float* __restrict__ dbeta, float* __restrict__ part) {
  const int lane = threadIdx.x & 63, wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
  float dg[MAXC], db[MAXC];
#pragma unroll
  for (int c = 0; c < MAXC; ++c) { dg[c] = 0.f; db[c] = 0.f; }
  for (int r = wave; r < rows; r += nwaves) {
    const float* xr = x + (size_t)r * D;
    const float* dyr = dy + (size_t)r * D;
    float xv[MAXC], gv[MAXC];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) { const int k = c * 64 + lane; xv[c] = k < D ? xr[k] : 0.f; s += xv[c]; }
    const float mu = wave_sum(s) / (float)D;
    float v = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) { const int k = c * 64 + lane; const float d = k < D ? xv[c] - mu : 0.f; v += d * d; }
    const float rstd = rsqrtf(wave_sum(v) / (float)D + eps);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      const int k = c * 64 + lane;
      if (k < D) {
        const float xh = (xv[c] - mu) * rstd, d = dyr[k];
        gv[c] = d * gamma[k];
        xv[c] = xh;
        s1 += gv[c]; s2 += gv[c] * xh;
        dg[c] += d * xh; db[c] += d;
      } else { gv[c] = 0.f; xv[c] = 0.f; }
    }
    s1 = wave_sum(s1) / (float)D; s2 = wave_sum(s2) / (float)D;
    float* dxr = dx + (size_t)r * D;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) { const int k = c * 64 + lane; if (k < D) dxr[k] = rstd * (gv[c] - s1 - xv[c] * s2); }
  }
  // the workgroup's four waves reduce their parameter-gradient partials in LDS: one atomic per column per workgroup
  __shared__ float sg[4][MAXC * 64], sb[4][MAXC * 64];
  const int w = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c < MAXC; ++c) { sg[w][c * 64 + lane] = dg[c]; sb[w][c * 64 + lane] = db[c]; }
  __syncthreads();
  for (int k = threadIdx.x; k < D; k += 256) {
    const float g_ = (sg[0][k] + sg[1][k]) + (sg[2][k] + sg[3][k]), b_ = (sb[0][k] + sb[1][k]) + (sb[2][k] + sb[3][k]);
    if (part) { part[((size_t)blockIdx.x * 2) * D + k] = g_; part[((size_t)blockIdx.x * 2 + 1) * D + k] = b_; }      // deterministic mode: merged in block order
    else { atomicAdd(dgamma + k, g_); atomicAdd(dbeta + k, b_); }
  }
}
int ln_bwd(const float* x, const float* gamma, const float* dy, float eps, int rows, int D, float* dx, float* dgamma, float* dbeta, hipStream_t s) {
  if (D > 2048) return 2;
  int blocks = (rows + 3) / 4; blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);      // one row per wave up to 8 192 rows
  float* part = nullptr;
  if (det_mode()) {
    blocks = blocks > 256 ? 256 : blocks;
    part = det_scratch((size_t)blocks * 2 * D);
    if (!part) return 3;
  }
  if (D <= 1024) hipLaunchKernelGGL(ln_bwd_kernel<16>, dim3(blocks), dim3(256), 0, s, x, gamma, dy, eps, rows, D, dx, dgamma, dbeta, part);
  else hipLaunchKernelGGL(ln_bwd_kernel<32>, dim3(blocks), dim3(256), 0, s, x, gamma, dy, eps, rows, D, dx, dgamma, dbeta, part);
  if (part) {      // part rows alternate (dgamma, dbeta) per block: two strided ordered sums
    hipLaunchKernelGGL(ordered_add_kernel, dim3((D + 255) / 256), dim3(256), 0, s, part, blocks, 2 * D, D, dgamma);
    hipLaunchKernelGGL(ordered_add_kernel, dim3((D + 255) / 256), dim3(256), 0, s, part + D, blocks, 2 * D, D, dbeta);
  }
  return hipGetLastError() == hipSuccess ? 0 : 3;
}

// ------------------------------------------------------------------------------------------------ multi-head self-attention (Q x Q)
// nn.MultiheadAttention (deformable_attention.py:195, 233): softmax((q k^T) / sqrt(dh)), dropout on the probabilities, times v.
// qkv [B*Q, 3*Dd] = [q | k | v], head h at columns h*dh.  Scores, probabilities and their adjoints are [B*Hd, Q, Qp] fp32 scratch
// (Qp = Q rounded up to 4); every product is one batched fp32-MFMA GEMM over (image, head) on strided views of qkv / dO / dqkv, a chunk of images per pass:
//   forward : S = scale q k^T  ->  row kernel: Pd = dropout(softmax(S))  ->  O = Pd v
//   backward: S = scale q k^T, dP = dO v^T  ->  row kernel: Pd, dS = P (keep dP - sum_j keep dP P)
//             ->  dq = scale dS k,  dk = scale dS^T q,  dv = Pd^T dO
// (Round 2 first ran one wave per query row on the VALU -- 2.2 ms per ViT-B block backward at 16 x 257 tokens, every wave
// re-reading K and V from L2 -- and an LDS-resident workgroup per (image, head), which was 2x slower still: 192 workgroups of four
// waves left each SIMD one latency-bound wave.)
#define MHA_MAXQ 1408      // decoder queries, and the 1370 tokens of a 518x518 image in the backbone-tail backward
#define MHA_RT (MHA_MAXQ / 64)

// one wave per score row; item = (b*Hd + h)*Q + i is also the dropout counter base (mask element = item*Q + j)
__global__ __launch_bounds__(256) void mha_softmax_fwd_kernel(float* __restrict__ S, int Q, int Qp, long nrows, long item_base, float p,
                                                              unsigned long long key) {
  const int lane = threadIdx.x & 63;
  const long local = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (local >= nrows) return;
  const long item = item_base + local;            // rows of an image chunk; the dropout counter runs over the whole batch
  float* row = S + (size_t)local * Qp;
  float v[MHA_RT];
  float mx = -INFINITY;
#pragma unroll
  for (int t = 0; t < MHA_RT; ++t) { const int j = lane + 64 * t; v[t] = j < Q ? row[j] : -INFINITY; mx = fmaxf(mx, v[t]); }
  mx = wave_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int t = 0; t < MHA_RT; ++t) { const int j = lane + 64 * t; v[t] = j < Q ? expf(v[t] - mx) : 0.f; sum += v[t]; }
  sum = wave_sum(sum);
  const float inv = 1.0f / sum, invk = 1.0f / (1.0f - p);
#pragma unroll
  for (int t = 0; t < MHA_RT; ++t) {
    const int j = lane + 64 * t;
    if (j < Qp) {
      float o = v[t] * inv;
      if (p > 0.f && j < Q) o = u01(key, (unsigned long long)item * Q + j) >= p ? o * invk : 0.f;
      row[j] = j < Q ? o : 0.f;
    }
  }
}
// SP: scores in, dropped probabilities out;  DD: d(loss)/d(dropped probabilities) in, d(loss)/d(scores) out
__global__ __launch_bounds__(256) void mha_softmax_bwd_kernel(float* __restrict__ SP, float* __restrict__ DD, int Q, int Qp, long nrows, long item_base,
                                                              float p, unsigned long long key) {
  const int lane = threadIdx.x & 63;
  const long local = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (local >= nrows) return;
  const long item = item_base + local;
  float* srow = SP + (size_t)local * Qp;
  float* drow = DD + (size_t)local * Qp;
  float v[MHA_RT], g[MHA_RT];
  float mx = -INFINITY;
#pragma unroll
  for (int t = 0; t < MHA_RT; ++t) { const int j = lane + 64 * t; v[t] = j < Q ? srow[j] : -INFINITY; g[t] = j < Q ? drow[j] : 0.f; mx = fmaxf(mx, v[t]); }
  mx = wave_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int t = 0; t < MHA_RT; ++t) { const int j = lane + 64 * t; v[t] = j < Q ? expf(v[t] - mx) : 0.f; sum += v[t]; }
  sum = wave_sum(sum);
  const float inv = 1.0f / sum, invk = 1.0f / (1.0f - p);
  float dot = 0.f;
#pragma unroll
  for (int t = 0; t < MHA_RT; ++t) {
    const int j = lane + 64 * t;
    v[t] *= inv;
    float keepf = 1.0f;
    if (p > 0.f && j < Q) keepf = u01(key, (unsigned long long)item * Q + j) >= p ? invk : 0.f;
    g[t] *= keepf;                         // d(loss) / d(P_ij) through the dropout
    dot += g[t] * v[t];
    if (j < Qp) srow[j] = j < Q ? v[t] * keepf : 0.f;
  }
  dot = wave_sum(dot);
#pragma unroll
  for (int t = 0; t < MHA_RT; ++t) {
    const int j = lane + 64 * t;
    if (j < Qp) drow[j] = j < Q ? v[t] * (g[t] - dot) : 0.f;
  }
}

// batched product over z = (image, head); operands are strided views: per-image stride, per-head stride
GemmF32X mha_gemm(const float* A, int lda, long long a_sb, long long a_sh, bool a_km, const float* W, int ldw, long long w_sb, long long w_sh, bool w_km,
                  float* C, int ldc, long long c_sb, long long c_sh, int M, int N, int K, int B, int Hd, float alpha) {
  GemmF32X g; memset(&g, 0, sizeof g);
  g.A = A; g.lda = lda; g.a_sb = a_sb; g.a_sh = a_sh; g.a_kmajor = a_km;
  g.W = W; g.ldw = ldw; g.w_sb = w_sb; g.w_sh = w_sh; g.w_kmajor = w_km;
  g.C = C; g.ldc = ldc; g.c_sb = c_sb; g.c_sh = c_sh;
  g.M = M; g.N = N; g.K = K; g.batch = B * Hd; g.hb = Hd; g.alpha = alpha; g.ksplit = 1;
  return g;
}
// Images per pass: the score / adjoint scratch ([images*Hd, Q, Qp] fp32, two of them in the backward) is capped at 1 GB per buffer
// (DINODET_MHA_CHUNK_MB) instead of growing with the batch (1 370 tokens x 12 heads: 90 MB per image per buffer).  Passes small
// enough to keep the scores in the 256 MB Infinity Cache between launches were measured and do not pay: ViT-B 518x518, batch 8,
// one image per pass 23.0 ms per step, two 22.0, the whole batch in one pass 21.7.
inline int mha_chunk_images(int B, int Hd, int Lq, int Lk) {
  const int forced = dod_option(DOD_OPT_MHA_CHUNK_IMAGES);      // tests: force several (ragged) passes on small shapes
  if (forced > 0) return forced > B ? B : forced;
  const char* e = getenv("DINODET_MHA_CHUNK_MB");
  const size_t mb = e && atoi(e) > 0 ? (size_t)atoi(e) : (size_t)1024;
  const size_t per = (size_t)Hd * Lq * up4((size_t)Lk) * 4;
  size_t c = (mb << 20) / (per ? per : 1);
  if (c < 1) c = 1;
  return c > (size_t)B ? B : (int)c;
}
inline int mha_chunk_images(int B, int Hd, int Q) { return mha_chunk_images(B, Hd, Q, Q); }
inline size_t mha_scratch_floats(int B, int Hd, int Lq, int Lk) { return (size_t)mha_chunk_images(B, Hd, Lq, Lk) * Hd * Lq * up4((size_t)Lk); }
inline size_t mha_scratch_floats(int B, int Hd, int Q) { return mha_scratch_floats(B, Hd, Q, Q); }

// General (rectangular) form: queries q [B*Lq, ldq], keys / values k, v [B*Lk, ldkv] (head h at columns h*dh of each), Lk <= MHA_MAXQ.
// The decoder's self-attention passes q | k | v of one packed buffer (Lq = Lk = Q); the dense cross-attention of the
// nn.TransformerDecoder branch (detr_decoder.py:28-35) passes Lq = Q queries against the Lk = N memory tokens.
// S: mha_scratch_floats(B, Hd, Lq, Lk) of scratch
static int launch_mha_fwd_rect(const float* q, int ldq, const float* k, const float* v, int ldkv, float* out, int ldo, float* S, int B, int Lq, int Lk,
                               int Hd, int dh, float scale, float p, unsigned long long key, hipStream_t s) {
  if (Lk > MHA_MAXQ) return 2;
  const int Lkp = (int)up4((size_t)Lk);
  const long long qs = (long long)Lq * ldq, ks = (long long)Lk * ldkv, ss = (long long)Lq * Lkp;
  const int cb = mha_chunk_images(B, Hd, Lq, Lk);
  for (int b0 = 0; b0 < B; b0 += cb) {
    const int nb = B - b0 < cb ? B - b0 : cb;
    const float* q0 = q + (size_t)b0 * qs;
    const float* k0 = k + (size_t)b0 * ks;
    const float* v0 = v + (size_t)b0 * ks;
    int rc = launch_gemm_f32x(mha_gemm(q0, ldq, qs, dh, false, k0, ldkv, ks, dh, false, S, Lkp, ss * Hd, ss, Lq, Lk, dh, nb, Hd, scale), s);
    if (rc) return rc;
    const long nrows = (long)nb * Hd * Lq;
    hipLaunchKernelGGL(mha_softmax_fwd_kernel, dim3((unsigned)((nrows + 3) / 4)), dim3(256), 0, s, S, Lk, Lkp, nrows, (long)b0 * Hd * Lq, p, key);
    if (hipGetLastError() != hipSuccess) return 3;
    rc = launch_gemm_f32x(mha_gemm(S, Lkp, ss * Hd, ss, false, v0, ldkv, ks, dh, true, out + (size_t)b0 * Lq * ldo, ldo, (long long)Lq * ldo, dh, Lq, dh, Lk,
                                   nb, Hd, 1.0f), s);
    if (rc) return rc;
  }
  return 0;
}
// dS, Pd: mha_scratch_floats(B, Hd, Lq, Lk) of scratch each; dq [B*Lq, lddq], dk, dv [B*Lk, lddkv] are WRITTEN (not accumulated)
static int launch_mha_bwd_rect(const float* q, int ldq, const float* k, const float* v, int ldkv, const float* dO, int ldo, float* dq, int lddq, float* dk,
                               float* dv, int lddkv, float* dS, float* Pd, int B, int Lq, int Lk, int Hd, int dh, float scale, float p,
                               unsigned long long key, hipStream_t s) {
  if (Lk > MHA_MAXQ) return 2;
  const int Lkp = (int)up4((size_t)Lk);
  const long long qs = (long long)Lq * ldq, ks = (long long)Lk * ldkv, os = (long long)Lq * ldo, ss = (long long)Lq * Lkp;
  const long long dqs = (long long)Lq * lddq, dks = (long long)Lk * lddkv;
  const int cb = mha_chunk_images(B, Hd, Lq, Lk);
  for (int b0 = 0; b0 < B; b0 += cb) {
    const int nb = B - b0 < cb ? B - b0 : cb;
    const float* q0 = q + (size_t)b0 * qs;
    const float* k0 = k + (size_t)b0 * ks;
    const float* v0 = v + (size_t)b0 * ks;
    const float* o0 = dO + (size_t)b0 * os;
    int rc = launch_gemm_f32x(mha_gemm(q0, ldq, qs, dh, false, k0, ldkv, ks, dh, false, Pd, Lkp, ss * Hd, ss, Lq, Lk, dh, nb, Hd, scale), s);
    if (rc) return rc;
    rc = launch_gemm_f32x(mha_gemm(o0, ldo, os, dh, false, v0, ldkv, ks, dh, false, dS, Lkp, ss * Hd, ss, Lq, Lk, dh, nb, Hd, 1.0f), s);
    if (rc) return rc;
    const long nrows = (long)nb * Hd * Lq;
    hipLaunchKernelGGL(mha_softmax_bwd_kernel, dim3((unsigned)((nrows + 3) / 4)), dim3(256), 0, s, Pd, dS, Lk, Lkp, nrows, (long)b0 * Hd * Lq, p, key);
    if (hipGetLastError() != hipSuccess) return 3;
    // dq = scale dS k;  dk = scale dS^T q;  dv = Pd^T dO      (k, q, dO enter as the k-major operand: [token, dh] views)
    rc = launch_gemm_f32x(mha_gemm(dS, Lkp, ss * Hd, ss, false, k0, ldkv, ks, dh, true, dq + (size_t)b0 * dqs, lddq, dqs, dh, Lq, dh, Lk, nb, Hd, scale), s);
    if (rc) return rc;
    rc = launch_gemm_f32x(mha_gemm(dS, Lkp, ss * Hd, ss, true, q0, ldq, qs, dh, true, dk + (size_t)b0 * dks, lddkv, dks, dh, Lk, dh, Lq, nb, Hd, scale), s);
    if (rc) return rc;
    rc = launch_gemm_f32x(mha_gemm(Pd, Lkp, ss * Hd, ss, true, o0, ldo, os, dh, true, dv + (size_t)b0 * dks, lddkv, dks, dh, Lk, dh, Lq, nb, Hd, 1.0f), s);
    if (rc) return rc;
  }
  return 0;
}
// the packed self-attention forms: qkv [B*Q, ld] = [q | k | v]
static int launch_mha_fwd_train(const float* qkv, int ld, float* out, int ldo, float* S, int B, int Q, int Hd, int Dd, int dh, float scale, float p,
                                unsigned long long key, hipStream_t s) {
  return launch_mha_fwd_rect(qkv, ld, qkv + Dd, qkv + 2 * Dd, ld, out, ldo, S, B, Q, Q, Hd, dh, scale, p, key, s);
}
static int launch_mha_bwd(const float* qkv, int ld, const float* dO, int ldo, float* dqkv, float* dS, float* Pd, int B, int Q, int Hd, int Dd,
                          int dh, float scale, float p, unsigned long long key, hipStream_t s) {
  return launch_mha_bwd_rect(qkv, ld, qkv + Dd, qkv + 2 * Dd, ld, dO, ldo, dqkv, ld, dqkv + Dd, dqkv + 2 * Dd, ld, dS, Pd, B, Q, Q, Hd, dh, scale, p, key, s);
}

// ------------------------------------------------------------------------------------------------ deformable gather backward
// Adjoint of deform_sample_kernel (deform.hip; deformable_attention.py:101-174).  One wave per (b, q, head), lanes along dh.
// dproj must be zero on entry (the two reference-logit columns are shared by all heads: float atomics); dvalues accumulates.
// DET (deterministic mode): no scatter and no atomics here -- the value gradient comes from deform_bwd_values_det_kernel (a gather in a
// fixed order), the heads' contributions to the two shared reference-logit columns go to dref_part [B*Q*Hd][2] and are summed head by head.
template <bool DET>
__global__ __launch_bounds__(256) void deform_bwd_kernel(const float* __restrict__ proj, int ldp, const float* __restrict__ values,
                                                         const float* __restrict__ dout, int B, int Q, int N, int Hd, int P, int dh, int h,
                                                         int w, float* __restrict__ dproj, float* __restrict__ dvalues, float* __restrict__ dref_part) {
  const int lane = threadIdx.x & 63;
  const long item = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (item >= (long)B * Q * Hd) return;
  const int hd = (int)(item % Hd);
  const long bq = item / Hd;
  const int b = (int)(bq / Q);
  const float* pr = proj + (size_t)bq * ldp;
  float* dpr = dproj + (size_t)bq * ldp;
  const float refx = sigmoidf_(pr[0]), refy = sigmoidf_(pr[1]);
  const float* off = pr + 2 + hd * P * 2;
  const float* awl = pr + 2 + Hd * P * 2 + hd * P;
  float mx = -INFINITY;
  for (int p = 0; p < P; ++p) mx = fmaxf(mx, awl[p]);
  float aw[8], den = 0.f;
#pragma unroll
  for (int p = 0; p < 8; ++p) { aw[p] = p < P ? expf(awl[p] - mx) : 0.f; den += aw[p]; }
  const int Dd = Hd * dh;
  const float* vb = values + (size_t)b * N * Dd + hd * dh;
  float* dvb = dvalues + (size_t)b * N * Dd + hd * dh;
  const float* g = dout + (size_t)bq * Dd + hd * dh;
  const bool d0ok = lane < dh, d1ok = lane + 64 < dh;
  const float g0 = d0ok ? g[lane] : 0.f, g1 = d1ok ? g[lane + 64] : 0.f;
  float da[8];
  float drefx = 0.f, drefy = 0.f;
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    da[p] = 0.f;
    if (p < P) {
      const float sx = refx + off[2 * p], sy = refy + off[2 * p + 1];
      float lx = fminf(fmaxf(sx, 0.f), 1.f), ly = fminf(fmaxf(sy, 0.f), 1.f);
      lx = lx * (float)(w - 1);
      ly = ly * (float)(h - 1);
      int x0 = (int)floorf(lx), y0 = (int)floorf(ly);
      int x1 = x0 + 1, y1 = y0 + 1;
      x0 = min(max(x0, 0), w - 1); x1 = min(max(x1, 0), w - 1);
      y0 = min(max(y0, 0), h - 1); y1 = min(max(y1, 0), h - 1);
      const float wx1 = lx - (float)x0, wx0 = 1.0f - wx1;
      const float wy1 = ly - (float)y0, wy0 = 1.0f - wy1;
      const float a = aw[p] / den;
      const size_t i00 = (size_t)(y0 * w + x0) * Dd, i01 = (size_t)(y1 * w + x0) * Dd, i10 = (size_t)(y0 * w + x1) * Dd, i11 = (size_t)(y1 * w + x1) * Dd;
      // <g, V_c> per corner
      float p00 = 0.f, p01 = 0.f, p10 = 0.f, p11 = 0.f;
      if (d0ok) { p00 += g0 * vb[i00 + lane]; p01 += g0 * vb[i01 + lane]; p10 += g0 * vb[i10 + lane]; p11 += g0 * vb[i11 + lane]; }
      if (d1ok) { p00 += g1 * vb[i00 + lane + 64]; p01 += g1 * vb[i01 + lane + 64]; p10 += g1 * vb[i10 + lane + 64]; p11 += g1 * vb[i11 + lane + 64]; }
      p00 = wave_sum(p00); p01 = wave_sum(p01); p10 = wave_sum(p10); p11 = wave_sum(p11);
      const float w00 = wx0 * wy0, w01 = wx0 * wy1, w10 = wx1 * wy0, w11 = wx1 * wy1;
      da[p] = p00 * w00 + p01 * w01 + p10 * w10 + p11 * w11;      // d out / d a_p
      // corner scatter: dV_c += a w_c g
      if (!DET && d0ok) {
        atomicAdd(dvb + i00 + lane, a * w00 * g0); atomicAdd(dvb + i01 + lane, a * w01 * g0);
        atomicAdd(dvb + i10 + lane, a * w10 * g0); atomicAdd(dvb + i11 + lane, a * w11 * g0);
      }
      if (!DET && d1ok) {
        atomicAdd(dvb + i00 + lane + 64, a * w00 * g1); atomicAdd(dvb + i01 + lane + 64, a * w01 * g1);
        atomicAdd(dvb + i10 + lane + 64, a * w10 * g1); atomicAdd(dvb + i11 + lane + 64, a * w11 * g1);
      }
      // bilinear weights -> pixel coordinates -> normalised location (clamp passes the gradient inside [0, 1] inclusive)
      const float dwx0 = a * (p00 * wy0 + p01 * wy1), dwx1 = a * (p10 * wy0 + p11 * wy1);
      const float dwy0 = a * (p00 * wx0 + p10 * wx1), dwy1 = a * (p01 * wx0 + p11 * wx1);
      float dsx = (dwx1 - dwx0) * (float)(w - 1), dsy = (dwy1 - dwy0) * (float)(h - 1);
      if (!(sx >= 0.f && sx <= 1.f)) dsx = 0.f;
      if (!(sy >= 0.f && sy <= 1.f)) dsy = 0.f;
      if (lane == 0) { dpr[2 + hd * P * 2 + 2 * p] = dsx; dpr[2 + hd * P * 2 + 2 * p + 1] = dsy; }
      drefx += dsx; drefy += dsy;
    }
  }
  // point-weight softmax backward
  float dotp = 0.f;
#pragma unroll
  for (int p = 0; p < 8; ++p) if (p < P) dotp += (aw[p] / den) * da[p];
  if (lane == 0) {
#pragma unroll
    for (int p = 0; p < 8; ++p) if (p < P) dpr[2 + Hd * P * 2 + hd * P + p] = (aw[p] / den) * (da[p] - dotp);
    if (DET) {
      dref_part[item * 2] = drefx * refx * (1.0f - refx);
      dref_part[item * 2 + 1] = drefy * refy * (1.0f - refy);
    } else {
      atomicAdd(dpr + 0, drefx * refx * (1.0f - refx));        // sigmoid of the reference logits
      atomicAdd(dpr + 1, drefy * refy * (1.0f - refy));
    }
  }
}
// deterministic mode: dproj[bq][0..1] = sum over heads, in head order, of dref_part
__global__ void deform_dref_det_kernel(const float* __restrict__ dref_part, int BQ, int Hd, int ldp, float* __restrict__ dproj) {
  const int bq = blockIdx.x * 256 + threadIdx.x;
  if (bq >= BQ) return;
  float ax = 0.f, ay = 0.f;
  for (int hd = 0; hd < Hd; ++hd) { ax += dref_part[((size_t)bq * Hd + hd) * 2]; ay += dref_part[((size_t)bq * Hd + hd) * 2 + 1]; }
  dproj[(size_t)bq * ldp] += ax;
  dproj[(size_t)bq * ldp + 1] += ay;
}
// deterministic mode: the adjoint of the bilinear gather as a GATHER -- one wave per (image, token, head) walks the image's Q x P samples
// in order (lanes along the samples: each recomputes its sample's corners as deform_bwd_kernel does and keeps its weight on THIS token),
// then adds the matching samples' a w g rows in ascending (q, p) order.  One writer per dvalues row: no atomics, a fixed order.
__global__ __launch_bounds__(256) void deform_bwd_values_det_kernel(const float* __restrict__ proj, int ldp, const float* __restrict__ dout, int B, int Q, int N,
                                                                    int Hd, int P, int dh, int h, int w, float* __restrict__ dvalues) {
  const int lane = threadIdx.x & 63;
  const long item = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (item >= (long)B * N * Hd) return;
  const int hd = (int)(item % Hd);
  const long bn = item / Hd;
  const int b = (int)(bn / N), n = (int)(bn % N);
  const int Dd = Hd * dh, QP = Q * P;
  float acc0 = 0.f, acc1 = 0.f;
  for (int s0 = 0; s0 < QP; s0 += 64) {
    const int sidx = s0 + lane;
    float wt = 0.f;
    if (sidx < QP) {
      const int q = sidx / P, p = sidx - q * P;
      const float* pr = proj + ((size_t)b * Q + q) * ldp;
      const float refx = sigmoidf_(pr[0]), refy = sigmoidf_(pr[1]);
      const float* awl = pr + 2 + Hd * P * 2 + hd * P;
      float mx = -INFINITY;
      for (int pp = 0; pp < P; ++pp) mx = fmaxf(mx, awl[pp]);
      float den = 0.f;                                            // the same 8-slot sum as deform_bwd_kernel
#pragma unroll
      for (int pp = 0; pp < 8; ++pp) den += pp < P ? expf(awl[pp] - mx) : 0.f;
      const float a = expf(awl[p] - mx) / den;
      const float* off = pr + 2 + hd * P * 2;
      const float sx = refx + off[2 * p], sy = refy + off[2 * p + 1];
      float lx = fminf(fmaxf(sx, 0.f), 1.f), ly = fminf(fmaxf(sy, 0.f), 1.f);
      lx = lx * (float)(w - 1);
      ly = ly * (float)(h - 1);
      int x0 = (int)floorf(lx), y0 = (int)floorf(ly);
      int x1 = x0 + 1, y1 = y0 + 1;
      x0 = min(max(x0, 0), w - 1); x1 = min(max(x1, 0), w - 1);
      y0 = min(max(y0, 0), h - 1); y1 = min(max(y1, 0), h - 1);
      const float wx1 = lx - (float)x0, wx0 = 1.0f - wx1;
      const float wy1 = ly - (float)y0, wy0 = 1.0f - wy1;
      // corners in the scatter kernel's order 00, 01, 10, 11 (clamped corners may coincide: their weights add, as their atomics did)
      if (y0 * w + x0 == n) wt += a * (wx0 * wy0);
      if (y1 * w + x0 == n) wt += a * (wx0 * wy1);
      if (y0 * w + x1 == n) wt += a * (wx1 * wy0);
      if (y1 * w + x1 == n) wt += a * (wx1 * wy1);
    }
    unsigned long long hit = __ballot(wt != 0.f);
    while (hit) {
      const int l = __ffsll((long long)hit) - 1;
      hit &= hit - 1;
      const float wl = __shfl(wt, l, 64);
      const int q = (s0 + l) / P;
      const float* g = dout + ((size_t)b * Q + q) * Dd + hd * dh;
      if (lane < dh) acc0 = fmaf(wl, g[lane], acc0);
      if (lane + 64 < dh) acc1 = fmaf(wl, g[lane + 64], acc1);
    }
  }
  float* dv = dvalues + ((size_t)b * N + n) * Dd + hd * dh;
  if (lane < dh) dv[lane] += acc0;
  if (lane + 64 < dh) dv[lane + 64] += acc1;
}

// ------------------------------------------------------------------------------------------------ orchestration
thread_local std::string g_terr;   // forward and backward of a step may run on different threads (autograd engine)
int tfail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
  g_terr = buf;
  return code;
}
#define TK(x) do { int r_ = (x); if (r_) return tfail(r_ == 3 ? DOD_ERR_HIP : DOD_ERR_INVALID, "decoder train: %s failed (%d)", #x, r_); } while (0)
#define TH(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return tfail(DOD_ERR_HIP, "decoder train: %s: %s", #x, hipGetErrorString(e_)); } while (0)

struct Dims { int B, N, Q, Dd, Hd, P, F, C, L, dh, ncat, ncp, BQ, M, fh, fw; };
bool make_dims(const dod_config* c, int B, int N, Dims* d) {
  if (!c || B <= 0 || N <= 0 || !c->use_deformable) return false;
  d->B = B; d->N = N; d->Q = c->num_queries; d->Dd = c->dec_hidden; d->Hd = c->dec_heads; d->P = c->n_points; d->F = c->dim_feedforward;
  d->C = c->num_classes; d->L = c->dec_layers; d->dh = d->Dd / d->Hd; d->ncat = 2 + 3 * d->Hd * d->P; d->ncp = (int)up4(d->ncat);
  d->BQ = B * d->Q; d->M = B * N;
  if (d->Dd % d->Hd || d->dh > 128 || d->dh % 4 || d->Dd % 4 || d->F % 4 || (d->Dd / 2) % 4 || d->Dd > 1024 || d->Q > MHA_MAXQ || d->P > 8 || d->P < 1) return false;
  int s = 1; while ((s + 1) * (s + 1) <= N) ++s;          // (h, w) of deformable_attention.py:241-256
  d->fh = s; d->fw = s;
  if (s * s != N) for (int i = s; i > 0; --i) if (N % i == 0) { d->fh = i; d->fw = N / i; break; }
  return true;
}

// tape layout (floats)
struct Tape {
  float *values, *hs, *hb, *boxes;
  struct Layer { float *tgt_in, *qkv, *att, *t1, *tgt1, *proj, *samp, *t2, *tgt2, *hid, *t3; } l[64];
};
size_t carve_tape(const Dims& d, void* base, Tape* t) {
  size_t off = 0;
  auto take = [&](size_t n) { float* p = base ? (float*)((char*)base + off) : nullptr; off += al256(n * 4); return p; };
  const size_t BQ = d.BQ, Dd = d.Dd;
  Tape tt;
  tt.values = take((size_t)d.M * Dd); tt.hs = take(BQ * Dd); tt.hb = take(BQ * (Dd / 2)); tt.boxes = take(BQ * 4);
  for (int j = 0; j < d.L; ++j) {
    auto& L = tt.l[j];
    L.tgt_in = take(BQ * Dd); L.qkv = take(BQ * 3 * Dd); L.att = take(BQ * Dd); L.t1 = take(BQ * Dd); L.tgt1 = take(BQ * Dd);
    L.proj = take(BQ * d.ncp); L.samp = take(BQ * Dd); L.t2 = take(BQ * Dd); L.tgt2 = take(BQ * Dd); L.hid = take(BQ * (size_t)d.F); L.t3 = take(BQ * Dd);
  }
  if (t) *t = tt;
  return off;
}

struct Scratch {
  float *y, *cat_w, *cat_b;                       // forward: branch output, fused [ref | offsets | weights] linear
  // backward
  float *dtgt, *dt, *dbr, *dbig, *dproj, *dcat_w, *dcat_b, *dqkv, *dS, *Pd, *dvalues, *dhb, *dz;
};
size_t carve_scratch(const Dims& d, void* base, Scratch* sc) {
  size_t off = 0;
  auto take = [&](size_t n) { float* p = base ? (float*)((char*)base + off) : nullptr; off += al256(n * 4); return p; };
  const size_t BQ = d.BQ, Dd = d.Dd, F = d.F, M = d.M;
  const size_t maxcols = (size_t)(3 * Dd > F ? 3 * Dd : F);           // widest activation of the query side
  Scratch s;
  s.y = take(BQ * maxcols); s.cat_w = take((size_t)d.ncp * Dd); s.cat_b = take(d.ncp);
  s.dtgt = take(BQ * Dd); s.dt = take(BQ * Dd); s.dbr = take(BQ * Dd); s.dbig = take(BQ * maxcols); s.dproj = take(BQ * d.ncp);
  s.dcat_w = take((size_t)d.ncp * Dd); s.dcat_b = take(d.ncp); s.dqkv = take(BQ * 3 * Dd);
  s.dS = take(mha_scratch_floats(d.B, d.Hd, d.Q)); s.Pd = take(mha_scratch_floats(d.B, d.Hd, d.Q));
  s.dvalues = take(M * Dd); s.dhb = take(BQ * (Dd / 2)); s.dz = take(BQ * 4);
  if (sc) *sc = s;
  return off;
}

GemmEpi gepi(const float* bias, float* out, int ldc, int act = ACT_NONE, const float* resid = nullptr, int ldr = 0) {
  GemmEpi e; memset(&e, 0, sizeof e);
  e.bias = bias; e.out_f32 = out; e.ldc = ldc; e.act = act; e.resid = resid; e.ldr = ldr;
  return e;
}
GemmF32X xgemm(const float* A, int lda, bool a_km, const float* W, int ldw, bool w_km, float* C, int ldc, int M, int N, int K, float alpha, bool accumulate) {
  GemmF32X g; memset(&g, 0, sizeof g);
  g.A = A; g.lda = lda; g.a_kmajor = a_km; g.W = W; g.ldw = ldw; g.w_kmajor = w_km; g.C = C; g.ldc = ldc;
  g.M = M; g.N = N; g.K = K; g.batch = 1; g.hb = 1; g.alpha = alpha; g.accumulate = accumulate; g.ksplit = 1;
  return g;
}
// K slices for a product whose 64x64 tiles leave most of the chip idle (the decoder's 1 600-row linears: 300 tiles, a lone
// workgroup's 16-k tile takes ~1 us): target ~768 workgroups of at least 8 k-tiles each; 1 = do not split
int ksplit_for(int rows, int cols, int K) {
  if (det_mode()) return 1;      // one workgroup owns an output tile: no atomic merge of K slices
  static const int target = [] { const char* e = DOD_TUNE_ENV("DINODET_F32_KSPLIT_WGS"); return e && atoi(e) > 0 ? atoi(e) : 768; }();
  const int tiles = ((rows + 63) / 64) * ((cols + 63) / 64), nkt = (K + 15) / 16;
  if (tiles >= target) return 1;
  int ks = (target + tiles - 1) / tiles;
  const int cap = nkt / 8 > 1 ? nkt / 8 : 1;
  return ks > cap ? cap : ks;
}
// Y[M,N] = act(X[M,K] W[N,K]^T + b).  Never K-split: the forward stays a bit-reproducible function of (inputs, seed).
int lin_fwd(const float* X, int ldx, const float* W, const float* b, int M, int N, int K, float* Y, int ldy, int act, hipStream_t s) {
  return launch_gemm_f32(X, ldx, W, K, M, N, K, gepi(b, Y, ldy, act), s);
}
// dX[M,K] (+)= dY[M,N] W[N,K]: W [N, K] is the k-major operand of the product over n
int lin_bwd_x(const float* dY, int ldy, const float* W, int M, int N, int K, float* dX, bool accumulate, hipStream_t s) {
  GemmF32X g = xgemm(dY, ldy, false, W, K, true, dX, K, M, K, N, 1.0f, accumulate);
  g.ksplit = ksplit_for(M, K, N);
  if (g.ksplit > 1 && !accumulate) {
    if (hipMemsetAsync(dX, 0, (size_t)M * K * 4, s) != hipSuccess) return 3;
    g.accumulate = 1;
  }
  return launch_gemm_f32x(g, s);
}
// C[R,Cc] += alpha * Y[M,R]^T X[M,Cc]: both operands k-major over the M rows.  A small output (weight gradients: a few dozen to
// a few hundred tiles against a reduction over thousands of rows) splits the rows over grid.z and accumulates atomically.
int gemm_tn_acc(const float* Y, int ldy, const float* X, int ldx, int M, int R, int Cc, float* C, int ldc, float alpha, hipStream_t s) {
  GemmF32X g = xgemm(Y, ldy, true, X, ldx, true, C, ldc, R, Cc, M, alpha, true);
  const int ks = ksplit_for(R, Cc, M);
  g.ksplit = ks;
  return launch_gemm_f32x(g, s);
}
// dW[N,K] += dY[M,N]^T X[M,K];  db[N] += colsum(dY)
int lin_bwd_w(const float* dY, int ldy, const float* X, int ldx, int M, int N, int K, float* dW, float* db, hipStream_t s) {
  int r = gemm_tn_acc(dY, ldy, X, ldx, M, N, K, dW, K, 1.0f, s);
  if (r) return r;
  return db ? colsum_add(dY, ldy, M, N, db, s) : 0;
}

int build_cat(const Dims& d, const dod_dec_train_params* p, const Scratch& sc, hipStream_t s) {
  const int Dd = d.Dd, HP = d.Hd * d.P;
  TH(hipMemsetAsync(sc.cat_w, 0, (size_t)d.ncp * Dd * 4, s));
  TH(hipMemsetAsync(sc.cat_b, 0, (size_t)d.ncp * 4, s));
  TH(hipMemcpyAsync(sc.cat_w, p->refp_w, (size_t)2 * Dd * 4, hipMemcpyDeviceToDevice, s));
  TH(hipMemcpyAsync(sc.cat_w + (size_t)2 * Dd, p->off_w, (size_t)HP * 2 * Dd * 4, hipMemcpyDeviceToDevice, s));
  TH(hipMemcpyAsync(sc.cat_w + (size_t)(2 + HP * 2) * Dd, p->aw_w, (size_t)HP * Dd * 4, hipMemcpyDeviceToDevice, s));
  TH(hipMemcpyAsync(sc.cat_b, p->refp_b, 2 * 4, hipMemcpyDeviceToDevice, s));
  TH(hipMemcpyAsync(sc.cat_b + 2, p->off_b, (size_t)HP * 2 * 4, hipMemcpyDeviceToDevice, s));
  TH(hipMemcpyAsync(sc.cat_b + 2 + HP * 2, p->aw_b, (size_t)HP * 4, hipMemcpyDeviceToDevice, s));
  return 0;
}

}  // namespace

extern "C" {

const char* dod_decoder_train_last_error(void) { return g_terr.c_str(); }

size_t dod_decoder_train_tape_bytes(const dod_config* cfg, int B, int N) {
  Dims d; if (!make_dims(cfg, B, N, &d)) return 0;
  return carve_tape(d, nullptr, nullptr) + 256;
}
size_t dod_decoder_train_workspace_bytes(const dod_config* cfg, int B, int N) {
  Dims d; if (!make_dims(cfg, B, N, &d)) return 0;
  return carve_scratch(d, nullptr, nullptr) + 256;
}

int dod_decoder_train_forward(const dod_config* cfg, const dod_dec_train_params* p, const float* memory, int B, int N, float dropout_p,
                              uint64_t seed, float* det, void* tape, size_t tape_bytes, void* ws, size_t ws_bytes, void* stream) {
  Dims d;
  if (!make_dims(cfg, B, N, &d)) return tfail(DOD_ERR_INVALID, "decoder train: unsupported configuration (deformable branch, head_dim <= 128, Dd <= 1024, Q <= %d)", MHA_MAXQ);
  if (!p || !memory || !det || !tape || !ws) return tfail(DOD_ERR_INVALID, "decoder train: null buffer");
  if (dropout_p < 0.f || dropout_p >= 1.f) return tfail(DOD_ERR_INVALID, "decoder train: dropout %g outside [0, 1)", dropout_p);
  if (tape_bytes < dod_decoder_train_tape_bytes(cfg, B, N) || ws_bytes < dod_decoder_train_workspace_bytes(cfg, B, N))
    return tfail(DOD_ERR_STATE, "decoder train: tape / workspace too small");
  hipStream_t s = (hipStream_t)stream;
  Tape t; Scratch sc;
  carve_tape(d, (void*)(((uintptr_t)tape + 255) & ~(uintptr_t)255), &t);
  carve_scratch(d, (void*)(((uintptr_t)ws + 255) & ~(uintptr_t)255), &sc);
  const int BQ = d.BQ, Dd = d.Dd, Q = d.Q;
  const float scale = 1.0f / sqrtf((float)d.dh);
  int rc = build_cat(d, p, sc, s); if (rc) return rc;
  TK(lin_fwd(memory, Dd, p->vp_w, p->vp_b, d.M, Dd, Dd, t.values, Dd, ACT_NONE, s));                     // tied layers: once
  TK(launch_bcast_rows(p->query_embed, t.l[0].tgt_in, B, Q, Dd, s));
  for (int j = 0; j < d.L; ++j) {
    auto& L = t.l[j];
    TK(lin_fwd(L.tgt_in, Dd, p->in_proj_w, p->in_proj_b, BQ, 3 * Dd, Dd, L.qkv, 3 * Dd, ACT_NONE, s));
    TK(launch_mha_fwd_train(L.qkv, 3 * Dd, L.att, Dd, sc.Pd, B, Q, d.Hd, Dd, d.dh, scale, dropout_p, site_key(seed, j, 0), s));
    TK(lin_fwd(L.att, Dd, p->out_proj_w, p->out_proj_b, BQ, Dd, Dd, sc.y, Dd, ACT_NONE, s));
    TK(dropout_add(L.tgt_in, sc.y, L.t1, (size_t)BQ * Dd, dropout_p, site_key(seed, j, 1), s));
    TK(launch_layernorm(L.t1, nullptr, p->norm1_w, p->norm1_b, cfg->dec_ln_eps, BQ, Dd, L.tgt1, nullptr, s));
    TH(hipMemsetAsync(L.proj, 0, (size_t)BQ * d.ncp * 4, s));
    TK(launch_gemm_f32(L.tgt1, Dd, sc.cat_w, Dd, BQ, d.ncat, Dd, gepi(sc.cat_b, L.proj, d.ncp), s));
    TK(launch_deform_sample(L.proj, d.ncp, t.values, B, Q, N, d.Hd, d.P, d.dh, d.fh, d.fw, L.samp, s, 0));
    TK(lin_fwd(L.samp, Dd, p->op_w, p->op_b, BQ, Dd, Dd, sc.y, Dd, ACT_NONE, s));
    TK(dropout_add(L.tgt1, sc.y, L.t2, (size_t)BQ * Dd, dropout_p, site_key(seed, j, 2), s));
    TK(launch_layernorm(L.t2, nullptr, p->norm2_w, p->norm2_b, cfg->dec_ln_eps, BQ, Dd, L.tgt2, nullptr, s));
    TK(lin_fwd(L.tgt2, Dd, p->lin1_w, p->lin1_b, BQ, d.F, Dd, L.hid, d.F, ACT_RELU, s));                   // taped: post-ReLU, pre-dropout
    const float* hin = L.hid;
    if (dropout_p > 0.f) { TK(dropout_add(nullptr, L.hid, sc.y, (size_t)BQ * d.F, dropout_p, site_key(seed, j, 3), s)); hin = sc.y; }
    float* y2 = sc.dbig;                                                                                  // free during the forward
    TK(lin_fwd(hin, d.F, p->lin2_w, p->lin2_b, BQ, Dd, d.F, y2, Dd, ACT_NONE, s));
    TK(dropout_add(L.tgt2, y2, L.t3, (size_t)BQ * Dd, dropout_p, site_key(seed, j, 4), s));
    float* nxt = j + 1 < d.L ? t.l[j + 1].tgt_in : t.hs;
    TK(launch_layernorm(L.t3, nullptr, p->norm3_w, p->norm3_b, cfg->dec_ln_eps, BQ, Dd, nxt, nullptr, s));
  }
  const int C = d.C;
  TK(launch_gemm_f32(t.hs, Dd, p->class_w, Dd, BQ, C, Dd, gepi(p->class_b, det, C + 4), s));
  TK(lin_fwd(t.hs, Dd, p->bb0_w, p->bb0_b, BQ, Dd / 2, Dd, t.hb, Dd / 2, ACT_RELU, s));
  TK(launch_gemm_f32(t.hb, Dd / 2, p->bb2_w, Dd / 2, BQ, 4, Dd / 2, gepi(p->bb2_b, det + C, C + 4, ACT_SIGMOID), s));
  TK(launch_copy2d(det + C, C + 4, t.boxes, 4, BQ, 4, 4, s));
  return DOD_OK;
}

int dod_decoder_train_backward(const dod_config* cfg, const dod_dec_train_params* p, const float* memory, int B, int N, float dropout_p,
                               uint64_t seed, const float* d_det, const void* tape, size_t tape_bytes, const dod_dec_train_params* grads,
                               float* d_memory, void* ws, size_t ws_bytes, void* stream) {
  Dims d;
  if (!make_dims(cfg, B, N, &d)) return tfail(DOD_ERR_INVALID, "decoder train: unsupported configuration");
  if (!p || !memory || !d_det || !tape || !grads || !ws) return tfail(DOD_ERR_INVALID, "decoder train: null buffer");
  if (tape_bytes < dod_decoder_train_tape_bytes(cfg, B, N) || ws_bytes < dod_decoder_train_workspace_bytes(cfg, B, N))
    return tfail(DOD_ERR_STATE, "decoder train: tape / workspace too small");
  hipStream_t s = (hipStream_t)stream;
  Tape t; Scratch sc;
  carve_tape(d, (void*)(((uintptr_t)tape + 255) & ~(uintptr_t)255), &t);
  carve_scratch(d, (void*)(((uintptr_t)ws + 255) & ~(uintptr_t)255), &sc);
  // gradients are written through the const-qualified struct's pointers (same layout as the parameters, float accumulators)
  auto G = [](const float* q) { return const_cast<float*>(q); };
  const int BQ = d.BQ, Dd = d.Dd, Q = d.Q, C = d.C, F = d.F, HP = d.Hd * d.P;
  const float scale = 1.0f / sqrtf((float)d.dh);
  const size_t nBD = (size_t)BQ * Dd;
  int rc = build_cat(d, p, sc, s); if (rc) return rc;
  TH(hipMemsetAsync(sc.dcat_w, 0, (size_t)d.ncp * Dd * 4, s));
  TH(hipMemsetAsync(sc.dcat_b, 0, (size_t)d.ncp * 4, s));
  TH(hipMemsetAsync(sc.dvalues, 0, (size_t)d.M * Dd * 4, s));
  // ---- heads (detr_decoder.py:80-81; utils.py:14-30)
  hipLaunchKernelGGL(sigmoid_bwd4_kernel, dim3((BQ * 4 + 255) / 256), dim3(256), 0, s, d_det + C, C + 4, t.boxes, 4, sc.dz, BQ);
  TH(hipGetLastError());
  TK(lin_bwd_w(sc.dz, 4, t.hb, Dd / 2, BQ, 4, Dd / 2, G(grads->bb2_w), G(grads->bb2_b), s));
  TK(lin_bwd_x(sc.dz, 4, p->bb2_w, BQ, 4, Dd / 2, sc.dhb, false, s));
  {
    const size_t n = (size_t)BQ * (Dd / 2);
    hipLaunchKernelGGL(relu_drop_bwd_kernel, dim3((unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048)), dim3(256), 0, s, sc.dhb, t.hb, sc.dhb, n, 0.f, 0ull);
    TH(hipGetLastError());
  }
  TK(lin_bwd_w(sc.dhb, Dd / 2, t.hs, Dd, BQ, Dd / 2, Dd, G(grads->bb0_w), G(grads->bb0_b), s));
  TK(lin_bwd_x(sc.dhb, Dd / 2, p->bb0_w, BQ, Dd / 2, Dd, sc.dtgt, false, s));
  TK(lin_bwd_w(d_det, C + 4, t.hs, Dd, BQ, C, Dd, G(grads->class_w), G(grads->class_b), s));
  TK(lin_bwd_x(d_det, C + 4, p->class_w, BQ, C, Dd, sc.dtgt, true, s));
  // ---- layers, last to first (weights tied: every layer adds into the same gradient tensors)
  for (int j = d.L - 1; j >= 0; --j) {
    const auto& L = t.l[j];
    // LN3 <- tgt2 + dropout4(linear2(dropout3(relu(linear1(tgt2)))))
    TK(ln_bwd(L.t3, p->norm3_w, sc.dtgt, cfg->dec_ln_eps, BQ, Dd, sc.dt, G(grads->norm3_w), G(grads->norm3_b), s));
    TK(dropout_add(nullptr, sc.dt, sc.dbr, nBD, dropout_p, site_key(seed, j, 4), s));                       // d(linear2 output)
    const float* hin = L.hid;
    if (dropout_p > 0.f) { TK(dropout_add(nullptr, L.hid, sc.y, (size_t)BQ * F, dropout_p, site_key(seed, j, 3), s)); hin = sc.y; }
    TK(lin_bwd_w(sc.dbr, Dd, hin, F, BQ, Dd, F, G(grads->lin2_w), G(grads->lin2_b), s));
    TK(lin_bwd_x(sc.dbr, Dd, p->lin2_w, BQ, Dd, F, sc.dbig, false, s));
    {
      const size_t n = (size_t)BQ * F;
      hipLaunchKernelGGL(relu_drop_bwd_kernel, dim3((unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048)), dim3(256), 0, s, sc.dbig, L.hid, sc.dbig, n,
                         dropout_p, site_key(seed, j, 3));
      TH(hipGetLastError());
    }
    TK(lin_bwd_w(sc.dbig, F, L.tgt2, Dd, BQ, F, Dd, G(grads->lin1_w), G(grads->lin1_b), s));
    TK(lin_bwd_x(sc.dbig, F, p->lin1_w, BQ, F, Dd, sc.dt, true, s));                                   // dt = d(tgt2): residual + FFN input
    // LN2 <- tgt1 + dropout2(output_proj(samp))
    TK(ln_bwd(L.t2, p->norm2_w, sc.dt, cfg->dec_ln_eps, BQ, Dd, sc.dtgt, G(grads->norm2_w), G(grads->norm2_b), s));   // dtgt = d(t2)
    TK(dropout_add(nullptr, sc.dtgt, sc.dbr, nBD, dropout_p, site_key(seed, j, 2), s));
    TK(lin_bwd_w(sc.dbr, Dd, L.samp, Dd, BQ, Dd, Dd, G(grads->op_w), G(grads->op_b), s));
    TK(lin_bwd_x(sc.dbr, Dd, p->op_w, BQ, Dd, Dd, sc.dt, false, s));                                   // dt = d(samp)
    TH(hipMemsetAsync(sc.dproj, 0, (size_t)BQ * d.ncp * 4, s));
    if (det_mode()) {
      float* dref_part = det_scratch((size_t)BQ * d.Hd * 2);
      if (!dref_part) return tfail(DOD_ERR_HIP, "deterministic mode: scratch allocation failed");
      hipLaunchKernelGGL(deform_bwd_kernel<true>, dim3((unsigned)(((long)BQ * d.Hd + 3) / 4)), dim3(256), 0, s, L.proj, d.ncp, t.values, sc.dt, B, Q, N, d.Hd,
                         d.P, d.dh, d.fh, d.fw, sc.dproj, sc.dvalues, dref_part);
      hipLaunchKernelGGL(deform_dref_det_kernel, dim3((BQ + 255) / 256), dim3(256), 0, s, dref_part, BQ, d.Hd, d.ncp, sc.dproj);
      hipLaunchKernelGGL(deform_bwd_values_det_kernel, dim3((unsigned)(((long)B * N * d.Hd + 3) / 4)), dim3(256), 0, s, L.proj, d.ncp, sc.dt, B, Q, N, d.Hd,
                         d.P, d.dh, d.fh, d.fw, sc.dvalues);
    } else {
      hipLaunchKernelGGL(deform_bwd_kernel<false>, dim3((unsigned)(((long)BQ * d.Hd + 3) / 4)), dim3(256), 0, s, L.proj, d.ncp, t.values, sc.dt, B, Q, N, d.Hd,
                         d.P, d.dh, d.fh, d.fw, sc.dproj, sc.dvalues, nullptr);
    }
    TH(hipGetLastError());
    TK(lin_bwd_w(sc.dproj, d.ncp, L.tgt1, Dd, BQ, d.ncat, Dd, sc.dcat_w, sc.dcat_b, s));
    TK(lin_bwd_x(sc.dproj, d.ncp, sc.cat_w, BQ, d.ncat, Dd, sc.dtgt, true, s));                        // dtgt = d(tgt1)
    // LN1 <- tgt_in + dropout1(out_proj(att))
    TK(ln_bwd(L.t1, p->norm1_w, sc.dtgt, cfg->dec_ln_eps, BQ, Dd, sc.dt, G(grads->norm1_w), G(grads->norm1_b), s));   // dt = d(t1)
    TK(dropout_add(nullptr, sc.dt, sc.dbr, nBD, dropout_p, site_key(seed, j, 1), s));
    TK(lin_bwd_w(sc.dbr, Dd, L.att, Dd, BQ, Dd, Dd, G(grads->out_proj_w), G(grads->out_proj_b), s));
    TK(lin_bwd_x(sc.dbr, Dd, p->out_proj_w, BQ, Dd, Dd, sc.dtgt, false, s));                           // dtgt = d(att)
    TK(launch_mha_bwd(L.qkv, 3 * Dd, sc.dtgt, Dd, sc.dqkv, sc.dS, sc.Pd, B, Q, d.Hd, Dd, d.dh, scale, dropout_p, site_key(seed, j, 0), s));
    TK(lin_bwd_w(sc.dqkv, 3 * Dd, L.tgt_in, Dd, BQ, 3 * Dd, Dd, G(grads->in_proj_w), G(grads->in_proj_b), s));
    TK(lin_bwd_x(sc.dqkv, 3 * Dd, p->in_proj_w, BQ, 3 * Dd, Dd, sc.dt, true, s));                      // dt = d(tgt_in): next (earlier) layer's d(output)
    TH(hipMemcpyAsync(sc.dtgt, sc.dt, nBD * 4, hipMemcpyDeviceToDevice, s));
  }
  // query embedding: tgt_0[b] = query_embed for every image (detr_decoder.py:59)
  hipLaunchKernelGGL(batch_sum_kernel, dim3((unsigned)(((size_t)Q * Dd + 255) / 256)), dim3(256), 0, s, sc.dtgt, G(grads->query_embed), B, (size_t)Q * Dd);
  TH(hipGetLastError());
  // fused small linear -> its three parameters
  {
    float* dw = sc.dcat_w; float* dbv = sc.dcat_b;
    TK(add_inplace(G(grads->refp_w), dw, (size_t)2 * Dd, s));
    TK(add_inplace(G(grads->off_w), dw + (size_t)2 * Dd, (size_t)HP * 2 * Dd, s));
    TK(add_inplace(G(grads->aw_w), dw + (size_t)(2 + HP * 2) * Dd, (size_t)HP * Dd, s));
    TK(add_inplace(G(grads->refp_b), dbv, 2, s));
    TK(add_inplace(G(grads->off_b), dbv + 2, (size_t)HP * 2, s));
    TK(add_inplace(G(grads->aw_b), dbv + 2 + HP * 2, (size_t)HP, s));
  }
  // value projection (computed once for the tied layers: d(values) is the sum over layers)
  TK(lin_bwd_w(sc.dvalues, Dd, memory, Dd, d.M, Dd, Dd, G(grads->vp_w), G(grads->vp_b), s));
  if (d_memory) TK(lin_bwd_x(sc.dvalues, Dd, p->vp_w, d.M, Dd, Dd, d_memory, false, s));
  return DOD_OK;
}

}  // extern "C"

// =============================================================================================================================
// Dense decoder: the nn.TransformerDecoder branch (detr_decoder.py:28-35, 62-69; torch's TransformerDecoderLayer, post-norm, ReLU):
//   x  = LN1(x + drop1(MHA(x, x, x)))              self-attention over the Q queries (probabilities dropped inside the MHA)
//   x  = LN2(x + drop2(MHA(x, memory, memory)))    dense cross-attention: Q queries x N memory tokens per head
//   x  = LN3(x + drop3(lin2(drop(relu(lin1(x))))))
// layers untied; heads as in the deformable branch.  Dropout sites per layer: 0 self-attn probabilities, 1 dropout1, 2 dropout2,
// 3 the FFN's inner dropout, 4 dropout3, 5 cross-attn probabilities.
namespace {

struct DDims { int B, N, Q, Dd, Hd, F, C, L, dh, BQ, M; };
bool make_ddims(const dod_config* c, int B, int N, DDims* d) {
  if (!c || B <= 0 || N <= 0 || c->use_deformable) return false;
  d->B = B; d->N = N; d->Q = c->num_queries; d->Dd = c->dec_hidden; d->Hd = c->dec_heads; d->F = c->dim_feedforward; d->C = c->num_classes;
  d->L = c->dec_layers; d->dh = d->Dd / d->Hd; d->BQ = B * d->Q; d->M = B * N;
  if (d->Dd % d->Hd || d->dh > 128 || d->dh % 4 || d->Dd % 4 || d->F % 4 || (d->Dd / 2) % 4 || d->Dd > 1024 || d->Q > MHA_MAXQ || N > MHA_MAXQ || d->L < 1 || d->L > 16)
    return false;
  return true;
}
struct DTape {
  float *hs, *hb, *boxes;
  struct Layer { float *x_in, *qkv, *att, *t1, *x1, *cq, *ckv, *catt, *t2, *x2, *hid, *t3; } l[16];
};
size_t carve_dtape(const DDims& d, void* base, DTape* t) {
  size_t off = 0;
  auto take = [&](size_t n) { float* p = base ? (float*)((char*)base + off) : nullptr; off += al256(n * 4); return p; };
  const size_t BQ = d.BQ, Dd = d.Dd;
  DTape tt;
  tt.hs = take(BQ * Dd); tt.hb = take(BQ * (Dd / 2)); tt.boxes = take(BQ * 4);
  for (int j = 0; j < d.L; ++j) {
    auto& L = tt.l[j];
    L.x_in = take(BQ * Dd); L.qkv = take(BQ * 3 * Dd); L.att = take(BQ * Dd); L.t1 = take(BQ * Dd); L.x1 = take(BQ * Dd);
    L.cq = take(BQ * Dd); L.ckv = take((size_t)d.M * 2 * Dd); L.catt = take(BQ * Dd); L.t2 = take(BQ * Dd); L.x2 = take(BQ * Dd);
    L.hid = take(BQ * (size_t)d.F); L.t3 = take(BQ * Dd);
  }
  if (t) *t = tt;
  return off;
}
struct DScratch { float *y, *dx, *dt, *dbr, *dbig, *dqkv, *dcq, *dckv, *dS, *Pd, *dhb, *dz, *dmem; };
size_t carve_dscratch(const DDims& d, void* base, DScratch* sc) {
  size_t off = 0;
  auto take = [&](size_t n) { float* p = base ? (float*)((char*)base + off) : nullptr; off += al256(n * 4); return p; };
  const size_t BQ = d.BQ, Dd = d.Dd, F = d.F;
  const size_t maxcols = (size_t)(3 * Dd > F ? 3 * Dd : F);
  const size_t sq = mha_scratch_floats(d.B, d.Hd, d.Q, d.Q), sr = mha_scratch_floats(d.B, d.Hd, d.Q, d.N);
  DScratch s;
  s.y = take(BQ * maxcols); s.dx = take(BQ * Dd); s.dt = take(BQ * Dd); s.dbr = take(BQ * Dd); s.dbig = take(BQ * maxcols);
  s.dqkv = take(BQ * 3 * Dd); s.dcq = take(BQ * Dd); s.dckv = take((size_t)d.M * 2 * Dd);
  s.dS = take(sq > sr ? sq : sr); s.Pd = take(sq > sr ? sq : sr); s.dhb = take(BQ * (Dd / 2)); s.dz = take(BQ * 4);
  s.dmem = take((size_t)d.M * Dd);
  if (sc) *sc = s;
  return off;
}

}  // namespace

extern "C" {

size_t dod_dense_decoder_train_tape_bytes(const dod_config* cfg, int B, int N) {
  DDims d; if (!make_ddims(cfg, B, N, &d)) return 0;
  return carve_dtape(d, nullptr, nullptr) + 256;
}
size_t dod_dense_decoder_train_workspace_bytes(const dod_config* cfg, int B, int N) {
  DDims d; if (!make_ddims(cfg, B, N, &d)) return 0;
  return carve_dscratch(d, nullptr, nullptr) + 256;
}

int dod_dense_decoder_train_forward(const dod_config* cfg, const dod_dense_dec_train_params* p, const float* memory, int B, int N, float dropout_p,
                                    uint64_t seed, float* det, void* tape, size_t tape_bytes, void* ws, size_t ws_bytes, void* stream) {
  DDims d;
  if (!make_ddims(cfg, B, N, &d)) return tfail(DOD_ERR_INVALID, "dense decoder train: unsupported configuration (nn.TransformerDecoder branch, head_dim <= 128, Dd <= 1024, Q and N <= %d, <= 16 layers)", MHA_MAXQ);
  if (!p || !p->layers || p->nlayers != d.L || !memory || !det || !tape || !ws) return tfail(DOD_ERR_INVALID, "dense decoder train: null buffer / layer count mismatch");
  if (dropout_p < 0.f || dropout_p >= 1.f) return tfail(DOD_ERR_INVALID, "dense decoder train: dropout %g outside [0, 1)", dropout_p);
  if (tape_bytes < dod_dense_decoder_train_tape_bytes(cfg, B, N) || ws_bytes < dod_dense_decoder_train_workspace_bytes(cfg, B, N))
    return tfail(DOD_ERR_STATE, "dense decoder train: tape / workspace too small");
  hipStream_t s = (hipStream_t)stream;
  DTape t; DScratch sc;
  carve_dtape(d, (void*)(((uintptr_t)tape + 255) & ~(uintptr_t)255), &t);
  carve_dscratch(d, (void*)(((uintptr_t)ws + 255) & ~(uintptr_t)255), &sc);
  const int BQ = d.BQ, Dd = d.Dd, Q = d.Q, F = d.F;
  const float scale = 1.0f / sqrtf((float)d.dh);
  const size_t nBD = (size_t)BQ * Dd;
  TK(launch_bcast_rows(p->query_embed, t.l[0].x_in, B, Q, Dd, s));
  for (int j = 0; j < d.L; ++j) {
    const dod_dense_layer_params& W = p->layers[j];
    auto& L = t.l[j];
    // self-attention block
    TK(lin_fwd(L.x_in, Dd, W.sa_in_w, W.sa_in_b, BQ, 3 * Dd, Dd, L.qkv, 3 * Dd, ACT_NONE, s));
    TK(launch_mha_fwd_train(L.qkv, 3 * Dd, L.att, Dd, sc.Pd, B, Q, d.Hd, Dd, d.dh, scale, dropout_p, site_key(seed, j, 0), s));
    TK(lin_fwd(L.att, Dd, W.sa_out_w, W.sa_out_b, BQ, Dd, Dd, sc.y, Dd, ACT_NONE, s));
    TK(dropout_add(L.x_in, sc.y, L.t1, nBD, dropout_p, site_key(seed, j, 1), s));
    TK(launch_layernorm(L.t1, nullptr, W.norm1_w, W.norm1_b, cfg->dec_ln_eps, BQ, Dd, L.x1, nullptr, s));
    // dense cross-attention: q from the queries, k | v from the memory (in_proj rows 0..Dd-1 / Dd..3Dd-1)
    TK(lin_fwd(L.x1, Dd, W.ca_in_w, W.ca_in_b, BQ, Dd, Dd, L.cq, Dd, ACT_NONE, s));
    TK(lin_fwd(memory, Dd, W.ca_in_w + (size_t)Dd * Dd, W.ca_in_b + Dd, d.M, 2 * Dd, Dd, L.ckv, 2 * Dd, ACT_NONE, s));
    TK(launch_mha_fwd_rect(L.cq, Dd, L.ckv, L.ckv + Dd, 2 * Dd, L.catt, Dd, sc.Pd, B, Q, N, d.Hd, d.dh, scale, dropout_p, site_key(seed, j, 5), s));
    TK(lin_fwd(L.catt, Dd, W.ca_out_w, W.ca_out_b, BQ, Dd, Dd, sc.y, Dd, ACT_NONE, s));
    TK(dropout_add(L.x1, sc.y, L.t2, nBD, dropout_p, site_key(seed, j, 2), s));
    TK(launch_layernorm(L.t2, nullptr, W.norm2_w, W.norm2_b, cfg->dec_ln_eps, BQ, Dd, L.x2, nullptr, s));
    // FFN
    TK(lin_fwd(L.x2, Dd, W.lin1_w, W.lin1_b, BQ, F, Dd, L.hid, F, ACT_RELU, s));                         // taped: post-ReLU, pre-dropout
    const float* hin = L.hid;
    if (dropout_p > 0.f) { TK(dropout_add(nullptr, L.hid, sc.y, (size_t)BQ * F, dropout_p, site_key(seed, j, 3), s)); hin = sc.y; }
    TK(lin_fwd(hin, F, W.lin2_w, W.lin2_b, BQ, Dd, F, sc.dbig, Dd, ACT_NONE, s));
    TK(dropout_add(L.x2, sc.dbig, L.t3, nBD, dropout_p, site_key(seed, j, 4), s));
    float* nxt = j + 1 < d.L ? t.l[j + 1].x_in : t.hs;
    TK(launch_layernorm(L.t3, nullptr, W.norm3_w, W.norm3_b, cfg->dec_ln_eps, BQ, Dd, nxt, nullptr, s));
  }
  const int C = d.C;
  TK(launch_gemm_f32(t.hs, Dd, p->class_w, Dd, BQ, C, Dd, gepi(p->class_b, det, C + 4), s));
  TK(lin_fwd(t.hs, Dd, p->bb0_w, p->bb0_b, BQ, Dd / 2, Dd, t.hb, Dd / 2, ACT_RELU, s));
  TK(launch_gemm_f32(t.hb, Dd / 2, p->bb2_w, Dd / 2, BQ, 4, Dd / 2, gepi(p->bb2_b, det + C, C + 4, ACT_SIGMOID), s));
  TK(launch_copy2d(det + C, C + 4, t.boxes, 4, BQ, 4, 4, s));
  return DOD_OK;
}

int dod_dense_decoder_train_backward(const dod_config* cfg, const dod_dense_dec_train_params* p, const float* memory, int B, int N, float dropout_p,
                                     uint64_t seed, const float* d_det, const void* tape, size_t tape_bytes, const dod_dense_dec_train_params* grads,
                                     float* d_memory, void* ws, size_t ws_bytes, void* stream) {
  DDims d;
  if (!make_ddims(cfg, B, N, &d)) return tfail(DOD_ERR_INVALID, "dense decoder train: unsupported configuration");
  if (!p || !p->layers || p->nlayers != d.L || !grads || !grads->layers || grads->nlayers != d.L || !memory || !d_det || !tape || !ws)
    return tfail(DOD_ERR_INVALID, "dense decoder train: null buffer / layer count mismatch");
  if (tape_bytes < dod_dense_decoder_train_tape_bytes(cfg, B, N) || ws_bytes < dod_dense_decoder_train_workspace_bytes(cfg, B, N))
    return tfail(DOD_ERR_STATE, "dense decoder train: tape / workspace too small");
  hipStream_t s = (hipStream_t)stream;
  DTape t; DScratch sc;
  carve_dtape(d, (void*)(((uintptr_t)tape + 255) & ~(uintptr_t)255), &t);
  carve_dscratch(d, (void*)(((uintptr_t)ws + 255) & ~(uintptr_t)255), &sc);
  auto G = [](const float* q) { return const_cast<float*>(q); };
  const int BQ = d.BQ, Dd = d.Dd, Q = d.Q, C = d.C, F = d.F;
  const float scale = 1.0f / sqrtf((float)d.dh);
  const size_t nBD = (size_t)BQ * Dd;
  float* dmem = d_memory ? d_memory : sc.dmem;                 // d(memory): the sum over the layers' k | v projections
  TH(hipMemsetAsync(dmem, 0, (size_t)d.M * Dd * 4, s));
  // ---- heads (detr_decoder.py:80-81; utils.py:14-30)
  hipLaunchKernelGGL(sigmoid_bwd4_kernel, dim3((BQ * 4 + 255) / 256), dim3(256), 0, s, d_det + C, C + 4, t.boxes, 4, sc.dz, BQ);
  TH(hipGetLastError());
  TK(lin_bwd_w(sc.dz, 4, t.hb, Dd / 2, BQ, 4, Dd / 2, G(grads->bb2_w), G(grads->bb2_b), s));
  TK(lin_bwd_x(sc.dz, 4, p->bb2_w, BQ, 4, Dd / 2, sc.dhb, false, s));
  {
    const size_t n = (size_t)BQ * (Dd / 2);
    hipLaunchKernelGGL(relu_drop_bwd_kernel, dim3((unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048)), dim3(256), 0, s, sc.dhb, t.hb, sc.dhb, n, 0.f, 0ull);
    TH(hipGetLastError());
  }
  TK(lin_bwd_w(sc.dhb, Dd / 2, t.hs, Dd, BQ, Dd / 2, Dd, G(grads->bb0_w), G(grads->bb0_b), s));
  TK(lin_bwd_x(sc.dhb, Dd / 2, p->bb0_w, BQ, Dd / 2, Dd, sc.dx, false, s));
  TK(lin_bwd_w(d_det, C + 4, t.hs, Dd, BQ, C, Dd, G(grads->class_w), G(grads->class_b), s));
  TK(lin_bwd_x(d_det, C + 4, p->class_w, BQ, C, Dd, sc.dx, true, s));                                   // dx = d(layer output)
  for (int j = d.L - 1; j >= 0; --j) {
    const dod_dense_layer_params& W = p->layers[j];
    const dod_dense_layer_params& Gw = grads->layers[j];
    const auto& L = t.l[j];
    // LN3 <- x2 + drop3(lin2(drop(relu(lin1(x2)))))
    TK(ln_bwd(L.t3, W.norm3_w, sc.dx, cfg->dec_ln_eps, BQ, Dd, sc.dt, G(Gw.norm3_w), G(Gw.norm3_b), s));           // dt = d(t3)
    TK(dropout_add(nullptr, sc.dt, sc.dbr, nBD, dropout_p, site_key(seed, j, 4), s));                              // d(lin2 output)
    const float* hin = L.hid;
    if (dropout_p > 0.f) { TK(dropout_add(nullptr, L.hid, sc.y, (size_t)BQ * F, dropout_p, site_key(seed, j, 3), s)); hin = sc.y; }
    TK(lin_bwd_w(sc.dbr, Dd, hin, F, BQ, Dd, F, G(Gw.lin2_w), G(Gw.lin2_b), s));
    TK(lin_bwd_x(sc.dbr, Dd, W.lin2_w, BQ, Dd, F, sc.dbig, false, s));
    {
      const size_t n = (size_t)BQ * F;
      hipLaunchKernelGGL(relu_drop_bwd_kernel, dim3((unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048)), dim3(256), 0, s, sc.dbig, L.hid, sc.dbig, n,
                         dropout_p, site_key(seed, j, 3));
      TH(hipGetLastError());
    }
    TK(lin_bwd_w(sc.dbig, F, L.x2, Dd, BQ, F, Dd, G(Gw.lin1_w), G(Gw.lin1_b), s));
    TK(lin_bwd_x(sc.dbig, F, W.lin1_w, BQ, F, Dd, sc.dt, true, s));                                               // dt = d(x2): residual + FFN input
    // LN2 <- x1 + drop2(ca_out(catt))
    TK(ln_bwd(L.t2, W.norm2_w, sc.dt, cfg->dec_ln_eps, BQ, Dd, sc.dx, G(Gw.norm2_w), G(Gw.norm2_b), s));           // dx = d(t2)
    TK(dropout_add(nullptr, sc.dx, sc.dbr, nBD, dropout_p, site_key(seed, j, 2), s));
    TK(lin_bwd_w(sc.dbr, Dd, L.catt, Dd, BQ, Dd, Dd, G(Gw.ca_out_w), G(Gw.ca_out_b), s));
    TK(lin_bwd_x(sc.dbr, Dd, W.ca_out_w, BQ, Dd, Dd, sc.dt, false, s));                                            // dt = d(catt)
    TK(launch_mha_bwd_rect(L.cq, Dd, L.ckv, L.ckv + Dd, 2 * Dd, sc.dt, Dd, sc.dcq, Dd, sc.dckv, sc.dckv + Dd, 2 * Dd, sc.dS, sc.Pd, B, Q, N, d.Hd, d.dh,
                           scale, dropout_p, site_key(seed, j, 5), s));
    // in_proj of the cross-attention: rows 0..Dd-1 see the queries, rows Dd..3Dd-1 the memory
    TK(lin_bwd_w(sc.dcq, Dd, L.x1, Dd, BQ, Dd, Dd, G(Gw.ca_in_w), G(Gw.ca_in_b), s));
    TK(lin_bwd_w(sc.dckv, 2 * Dd, memory, Dd, d.M, 2 * Dd, Dd, G(Gw.ca_in_w) + (size_t)Dd * Dd, G(Gw.ca_in_b) + Dd, s));
    TK(lin_bwd_x(sc.dckv, 2 * Dd, W.ca_in_w + (size_t)Dd * Dd, d.M, 2 * Dd, Dd, dmem, true, s));                   // d(memory) += d(k | v) W_kv
    TK(lin_bwd_x(sc.dcq, Dd, W.ca_in_w, BQ, Dd, Dd, sc.dx, true, s));                                              // dx = d(x1): residual + query input
    // LN1 <- x_in + drop1(sa_out(att))
    TK(ln_bwd(L.t1, W.norm1_w, sc.dx, cfg->dec_ln_eps, BQ, Dd, sc.dt, G(Gw.norm1_w), G(Gw.norm1_b), s));           // dt = d(t1)
    TK(dropout_add(nullptr, sc.dt, sc.dbr, nBD, dropout_p, site_key(seed, j, 1), s));
    TK(lin_bwd_w(sc.dbr, Dd, L.att, Dd, BQ, Dd, Dd, G(Gw.sa_out_w), G(Gw.sa_out_b), s));
    TK(lin_bwd_x(sc.dbr, Dd, W.sa_out_w, BQ, Dd, Dd, sc.dx, false, s));                                            // dx = d(att)
    TK(launch_mha_bwd(L.qkv, 3 * Dd, sc.dx, Dd, sc.dqkv, sc.dS, sc.Pd, B, Q, d.Hd, Dd, d.dh, scale, dropout_p, site_key(seed, j, 0), s));
    TK(lin_bwd_w(sc.dqkv, 3 * Dd, L.x_in, Dd, BQ, 3 * Dd, Dd, G(Gw.sa_in_w), G(Gw.sa_in_b), s));
    TK(lin_bwd_x(sc.dqkv, 3 * Dd, W.sa_in_w, BQ, 3 * Dd, Dd, sc.dt, true, s));                                     // dt = d(x_in): the layer below's d(output)
    TH(hipMemcpyAsync(sc.dx, sc.dt, nBD * 4, hipMemcpyDeviceToDevice, s));
  }
  // query embedding: x_0[b] = query_embed for every image (detr_decoder.py:59)
  hipLaunchKernelGGL(batch_sum_kernel, dim3((unsigned)(((size_t)Q * Dd + 255) / 256)), dim3(256), 0, s, sc.dx, G(grads->query_embed), B, (size_t)Q * Dd);
  TH(hipGetLastError());
  return DOD_OK;
}

}  // extern "C"

// =============================================================================================================================
// Backbone tail: the LoRA-adapted encoder blocks (dinov2_backbone.py:45-51: the last two), the final LayerNorm and the projection
// (dinov2_backbone.py:33-37, 64-65) in train() mode -- the rest of what `loss.backward()` (train.py:1101) reaches: gradients of every
// lora_A / lora_B (utils.py:46-70) and of the projection.  The DINOv2 weights, LayerNorms, LayerScales and biases are frozen
// (dinov2_backbone.py:40-41), and so is everything in front of the first adapted block (it runs in the inference kernels,
// dod_backbone_prefix): the backward stops at the tail's input.  GELU MLP (ViT-S/B/L) or SwiGLU (ViT-g: modeling_dinov2.py:300-314; the
// fc1 / fc2 slots of dod_bb_block_params then hold mlp.weights_in [2F, D] / mlp.weights_out [D, F]).
//   forward : x -> LN1 -> q|k|v (W' = W + alpha B A, merged in fp32 as the eval path does) -> softmax(q k^T / sqrt(dh)) v -> dense
//             -> x + ls1 * . -> LN2 -> fc1 -> GELU(erf) -> fc2 -> + ls2 * .  ->  final LN -> projection
//   backward: dX = dY W' on the fp32 MFMA GEMM; per LoRA linear  dB += alpha dY^T (X A^T),  dA += alpha (dY B)^T X  (rank-r GEMMs);
//             attention backward = the row / column passes of the decoder's self-attention with Q := N tokens.
namespace {

struct TDims { int B, N, M, D, H, dh, F, F1, Dd, nb, r, swiglu; float alpha, eps; };   // F1: width of the first MLP linear (2F for SwiGLU)
bool make_tdims(const dod_config* c, int B, int N, int nblocks, TDims* d) {
  if (!c || B <= 0 || N <= 0 || nblocks <= 0 || nblocks > 8) return false;
  d->B = B; d->N = N; d->M = B * N; d->D = c->hidden; d->H = c->heads; d->dh = d->D / d->H; d->F = c->ffn_hidden;
  d->swiglu = c->swiglu ? 1 : 0; d->F1 = c->swiglu ? 2 * c->ffn_hidden : c->ffn_hidden;
  d->Dd = c->target_dim ? c->target_dim : c->hidden; d->nb = nblocks; d->r = c->lora_r; d->alpha = c->lora_alpha; d->eps = c->ln_eps;
  if (d->D % d->H || d->dh > 128 || d->dh % 4 || d->D % 4 || d->F % 4 || d->D > 2048 || N > MHA_MAXQ || d->r < 1 || d->r > 64) return false;   // D: ln_bwd / launch_layernorm
  return true;
}
struct TTape {
  float* xout; float* f;
  struct Blk { float *x, *y1, *qkv, *ctx, *x1, *y2, *pre, *h, *Wqkv, *Wo, *W1, *W2, *bqkv, *lse; } b[8];
};
size_t carve_ttape(const TDims& d, void* base, TTape* t) {
  size_t off = 0;
  auto take = [&](size_t n) { float* p = base ? (float*)((char*)base + off) : nullptr; off += al256(n * 4); return p; };
  const size_t M = d.M, D = d.D, F = d.F;
  TTape tt;
  tt.xout = take(M * D); tt.f = take(M * D);
  for (int i = 0; i < d.nb; ++i) {
    auto& b = tt.b[i];
    b.x = take(M * D); b.y1 = take(M * D); b.qkv = take(M * 3 * D); b.ctx = take(M * D); b.x1 = take(M * D); b.y2 = take(M * D);
    b.pre = take(M * (size_t)d.F1); b.h = take(M * F);
    b.Wqkv = take(3 * D * D); b.Wo = take(D * D); b.W1 = take((size_t)d.F1 * D); b.W2 = take(D * F); b.bqkv = take(3 * D);
    b.lse = take(2 * M * (size_t)d.H);             // (max, sum) of every score row [B, H, N, 2]: the flash-style attention adjoint
  }
  if (t) *t = tt;
  return off;
}
struct TScratch { float *dx, *da, *db, *dbig, *dqkv, *dS, *Pd, *T, *U, *dump, *dh, *delta; };
// head_dim 64 (every DINOv2 variant): the attention adjoint may recompute its scores tile by tile (attn_f32m.hip launch_attn_f32_bwd) -- no
// [B*H, N, N] score / adjoint scratch -- instead of the batched-GEMM form (which other head sizes always take).
// Taken from 1 024 tokens per image up (518x518 inputs: 19.8 vs 21.9 ms per ViT-B batch-8 step, and no 2 x 720 MB of scratch); below
// that the batched form is as fast (224x224: 10.0 vs 9.9 ms) and sits closer to a float64 evaluation -- the flash form takes
// delta = <dO, O> from the forward's rounded output instead of sum_j P dP over the probabilities it multiplies (1.7e-5 vs 5.3e-5 from
// float64 on the worst LoRA gradient at 1 370 tokens, the PyTorch composite 2.3e-5).  DINODET_ATTN_BWD_FLASH = 0 / 1 forces either.
// This is the SINGLE predicate: the scratch carve (no dS / Pd when it holds), the forward's log-sum-exp tape and the backward all ask it, and it
// contains everything launch_attn_f32_bwd itself checks (head_dim 64; q / k / v / o pitches 3D and D multiples of 4: D = heads * 64) -- so that
// launcher's "not taken" return (2) cannot occur behind it; there is no second scratch layout to fall back to.
inline bool tail_flash_bwd(const TDims& d) {
  static const char* e = getenv("DINODET_ATTN_BWD_FLASH");
  if (d.dh != 64 || d.D != d.H * 64 || d.D % 4 != 0 || d.B <= 0 || d.N <= 0 || (e && e[0] == '0')) return false;
  return (e && e[0] == '1') || d.N >= 1024;
}
size_t carve_tscratch(const TDims& d, void* base, TScratch* sc) {
  size_t off = 0;
  auto take = [&](size_t n) { float* p = base ? (float*)((char*)base + off) : nullptr; off += al256(n * 4); return p; };
  const size_t M = d.M, D = d.D, F = d.F1, big = F > 3 * D ? F : 3 * D;
  TScratch s;
  s.dx = take(M * D); s.da = take(M * D); s.db = take(M * D); s.dbig = take(M * big); s.dqkv = take(M * 3 * D);
  const bool flash = tail_flash_bwd(d);
  s.dS = take(flash ? 0 : mha_scratch_floats(d.B, d.H, d.N)); s.Pd = take(flash ? 0 : mha_scratch_floats(d.B, d.H, d.N));
  s.delta = take(M * (size_t)d.H);
  s.T = take(M * up4(d.r)); s.U = take(M * up4(d.r)); s.dump = take(2 * big);
  s.dh = d.swiglu ? take(M * (size_t)d.F) : nullptr;        // SwiGLU: d(h) [M, F] beside d(pre) [M, 2F] (the tape stays read-only)
  if (sc) *sc = s;
  return off;
}
// Rank-r products of a LoRA pair, r <= 8 (the reference trains r = 2: dinov2_backbone.py:47-51): both are bound by reading the
// [M, features] activation once, which a 64x64-tile GEMM with r useful columns cannot do (12 workgroups walking 4 112 rows: 115 us).
//   down: T[m, c] = alpha * sum_k X[m, k] * A(c, k)            one wave per row, lanes along k;   A(c, k) = A[c * sa_c + k * sa_k]
//   up  : G(o, c) += sum_m Y[m, o] * T[m, c]                   one thread per column o, 64 rows per workgroup, atomic accumulate
#define LORA_RMAX 8
#define LORA_UP_ROWS 64
__global__ __launch_bounds__(256) void lora_down_kernel(const float* __restrict__ X, int ldx, const float* __restrict__ A, int sa_c, int sa_k, int M, int K,
                                                        int r, float alpha, float* __restrict__ T, int ldt) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  const float* x = X + (size_t)m * ldx;
  float acc[LORA_RMAX];
#pragma unroll
  for (int c = 0; c < LORA_RMAX; ++c) acc[c] = 0.f;
#pragma unroll 4
  for (int k = lane; k < K; k += 64) {
    const float xv = x[k];
    const float* ak = A + (size_t)k * sa_k;
#pragma unroll
    for (int c = 0; c < LORA_RMAX; ++c)
      if (c < r) acc[c] = fmaf(xv, ak[(size_t)c * sa_c], acc[c]);
  }
#pragma unroll
  for (int c = 0; c < LORA_RMAX; ++c) {
    if (c < r) {
      const float v = wave_sum(acc[c]);
      if (lane == 0) T[(size_t)m * ldt + c] = alpha * v;
    }
  }
}
__global__ __launch_bounds__(256) void lora_up_kernel(const float* __restrict__ Y, int ldy, const float* __restrict__ T, int ldt, int M, int O, int r,
                                                      float* __restrict__ G, int sg_o, int sg_c) {
  __shared__ float sT[LORA_UP_ROWS][LORA_RMAX];
  const int o = blockIdx.x * 256 + threadIdx.x;
  float acc[LORA_RMAX];
#pragma unroll
  for (int c = 0; c < LORA_RMAX; ++c) acc[c] = 0.f;
  // a workgroup walks the row chunks blockIdx.y, blockIdx.y + gridDim.y, ...: gridDim.y = 1 (deterministic mode) makes it the only adder
  for (int m0 = blockIdx.y * LORA_UP_ROWS; m0 < M; m0 += gridDim.y * LORA_UP_ROWS) {
    const int nm = M - m0 < LORA_UP_ROWS ? M - m0 : LORA_UP_ROWS;
    __syncthreads();
    for (int i = threadIdx.x; i < LORA_UP_ROWS * LORA_RMAX; i += 256) {
      const int mm = i / LORA_RMAX, c = i % LORA_RMAX;
      sT[mm][c] = (mm < nm && c < r) ? T[(size_t)(m0 + mm) * ldt + c] : 0.f;
    }
    __syncthreads();
    if (o < O) {
      const float* y = Y + (size_t)m0 * ldy + o;
      for (int mm = 0; mm < nm; ++mm) {
        const float yv = y[(size_t)mm * ldy];
#pragma unroll
        for (int c = 0; c < LORA_RMAX; ++c) acc[c] = fmaf(yv, sT[mm][c], acc[c]);
      }
    }
  }
  if (o >= O) return;
#pragma unroll
  for (int c = 0; c < LORA_RMAX; ++c)
    if (c < r) unsafeAtomicAdd(G + (size_t)o * sg_o + (size_t)c * sg_c, acc[c]);
}
int lora_down(const float* X, int ldx, const float* A, int sa_c, int sa_k, int M, int K, int r, float alpha, float* T, int ldt, hipStream_t s) {
  hipLaunchKernelGGL(lora_down_kernel, dim3((M + 3) / 4), dim3(256), 0, s, X, ldx, A, sa_c, sa_k, M, K, r, alpha, T, ldt);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}
int lora_up(const float* Y, int ldy, const float* T, int ldt, int M, int O, int r, float* G, int sg_o, int sg_c, hipStream_t s) {
  hipLaunchKernelGGL(lora_up_kernel, dim3((O + 255) / 256, det_mode() ? 1 : (M + LORA_UP_ROWS - 1) / LORA_UP_ROWS), dim3(256), 0, s, Y, ldy, T, ldt, M, O, r, G, sg_o, sg_c);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}

// gradients of one LoRA pair for out = X W'^T: dB [out, r] += alpha dY^T (X A^T), dA [r, in] += alpha (dY B)^T X
int lora_grads(const TDims& d, const float* X, int in_f, const float* dY, int ldy, int out_f, const float* A, const float* Bm, float* dA, float* dB,
               const TScratch& t, hipStream_t s) {
  if (!dA || !dB) return 0;
  const int r = d.r, rp = (int)up4(r), M = d.M;
  int rc;
  if (r <= LORA_RMAX) {
    rc = lora_down(X, in_f, A, in_f, 1, M, in_f, r, d.alpha, t.T, rp, s); if (rc) return rc;            // T = alpha X A^T   [M, r]
    rc = lora_up(dY, ldy, t.T, rp, M, out_f, r, dB, r, 1, s); if (rc) return rc;                        // dB += dY^T T
    rc = lora_down(dY, ldy, Bm, 1, r, M, out_f, r, d.alpha, t.U, rp, s); if (rc) return rc;             // U = alpha dY B    [M, r]
    return lora_up(X, in_f, t.U, rp, M, in_f, r, dA, 1, in_f, s);                                       // dA += U^T X
  }
  rc = launch_gemm_f32x(xgemm(X, in_f, false, A, in_f, false, t.T, rp, M, r, in_f, d.alpha, false), s); if (rc) return rc;
  rc = gemm_tn_acc(dY, ldy, t.T, rp, M, out_f, r, dB, r, 1.0f, s); if (rc) return rc;
  rc = launch_gemm_f32x(xgemm(dY, ldy, false, Bm, r, true, t.U, rp, M, r, out_f, d.alpha, false), s); if (rc) return rc;
  return gemm_tn_acc(t.U, rp, X, in_f, M, r, in_f, dA, in_f, 1.0f, s);
}

}  // namespace

extern "C" {

size_t dod_backbone_tail_tape_bytes(const dod_config* cfg, int B, int N, int nblocks) {
  TDims d; if (!make_tdims(cfg, B, N, nblocks, &d)) return 0;
  return carve_ttape(d, nullptr, nullptr) + 256;
}
size_t dod_backbone_tail_workspace_bytes(const dod_config* cfg, int B, int N, int nblocks) {
  TDims d; if (!make_tdims(cfg, B, N, nblocks, &d)) return 0;
  return carve_tscratch(d, nullptr, nullptr) + 256;
}

int dod_backbone_tail_train_forward(const dod_config* cfg, const dod_bb_tail_params* p, const float* x_in, int B, int N, float* mem_out,
                                    void* tape, size_t tape_bytes, void* ws, size_t ws_bytes, void* stream) {
  if (!p || !p->blocks) return tfail(DOD_ERR_INVALID, "backbone tail: null parameters");
  TDims d;
  if (!make_tdims(cfg, B, N, p->nblocks, &d)) return tfail(DOD_ERR_INVALID, "backbone tail: unsupported configuration (head_dim <= 128, N <= %d, 1 <= lora_r <= 64, at most 8 blocks)", MHA_MAXQ);
  if (!x_in || !mem_out || !tape || !ws) return tfail(DOD_ERR_INVALID, "backbone tail: null buffer");
  if (tape_bytes < dod_backbone_tail_tape_bytes(cfg, B, N, p->nblocks) || ws_bytes < dod_backbone_tail_workspace_bytes(cfg, B, N, p->nblocks))
    return tfail(DOD_ERR_STATE, "backbone tail: tape / workspace too small");
  if (cfg->target_dim && (!p->proj_w || !p->proj_b)) return tfail(DOD_ERR_MISSING, "backbone tail: projection weights missing");
  hipStream_t s = (hipStream_t)stream;
  TTape t; TScratch sc;
  carve_ttape(d, (void*)(((uintptr_t)tape + 255) & ~(uintptr_t)255), &t);
  carve_tscratch(d, (void*)(((uintptr_t)ws + 255) & ~(uintptr_t)255), &sc);
  const int M = d.M, D = d.D, F = d.F, F1 = d.F1;
  const float scale = 1.0f / sqrtf((float)d.dh);
  TH(hipMemcpyAsync(t.b[0].x, x_in, (size_t)M * D * 4, hipMemcpyDeviceToDevice, s));
  for (int i = 0; i < d.nb; ++i) {
    const dod_bb_block_params& bp = p->blocks[i];
    auto& tb = t.b[i];
    // merged weights W' = W + alpha B A (utils.py:68-70), q | k | v concatenated
    const dod_lora_linear* qkv3[3] = {&bp.q, &bp.k, &bp.v};
    for (int c = 0; c < 3; ++c) {
      TK(launch_lora_merge(qkv3[c]->w, qkv3[c]->A, qkv3[c]->Bm, d.alpha, D, D, d.r, tb.Wqkv + (size_t)c * D * D, s));
      TH(hipMemcpyAsync(tb.bqkv + (size_t)c * D, qkv3[c]->b, (size_t)D * 4, hipMemcpyDeviceToDevice, s));
    }
    TK(launch_lora_merge(bp.o.w, bp.o.A, bp.o.Bm, d.alpha, D, D, d.r, tb.Wo, s));
    TK(launch_lora_merge(bp.fc1.w, bp.fc1.A, bp.fc1.Bm, d.alpha, F1, D, d.r, tb.W1, s));
    TK(launch_lora_merge(bp.fc2.w, bp.fc2.A, bp.fc2.Bm, d.alpha, D, F, d.r, tb.W2, s));
    TK(launch_layernorm(tb.x, nullptr, bp.ln1_w, bp.ln1_b, d.eps, M, D, tb.y1, nullptr, s));
    TK(lin_fwd(tb.y1, D, tb.Wqkv, tb.bqkv, M, 3 * D, D, tb.qkv, 3 * D, ACT_NONE, s));
    {
      AttnF32 a; a.q = tb.qkv; a.k = tb.qkv + D; a.v = tb.qkv + 2 * D; a.o = tb.ctx; a.ldq = a.ldk = a.ldv = 3 * D; a.ldo = D;
      a.Lq = a.Lk = N; a.B = B; a.heads = d.H; a.dh = d.dh; a.scale = scale;
      if (d.dh == 64) a.lse = tb.lse;            // fp32-MFMA flash kernel: the adjoint's log-sum-exp comes for free
      TK(launch_attn_f32(a, s));
    }
    {   // x1 = x + ls1 * (ctx Wo'^T + bo)
      GemmEpi e = gepi(bp.o.b, tb.x1, D, ACT_NONE, tb.x, D); e.scale = bp.ls1;
      TK(launch_gemm_f32(tb.ctx, D, tb.Wo, D, M, D, D, e, s));
    }
    TK(launch_layernorm(tb.x1, nullptr, bp.ln2_w, bp.ln2_b, d.eps, M, D, tb.y2, nullptr, s));
    TK(lin_fwd(tb.y2, D, tb.W1, bp.fc1.b, M, F1, D, tb.pre, F1, ACT_NONE, s));      // taped: the backward needs the pre-activation
    {
      const size_t n = (size_t)M * F;
      const dim3 grid((unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096));
      if (d.swiglu) hipLaunchKernelGGL(swiglu_fwd_kernel, grid, dim3(256), 0, s, tb.pre, tb.h, (size_t)M, F);   // fc1 / fc2 = weights_in / weights_out
      else hipLaunchKernelGGL(gelu_fwd_kernel, grid, dim3(256), 0, s, tb.pre, tb.h, n);
      TH(hipGetLastError());
    }
    float* xnext = i + 1 < d.nb ? t.b[i + 1].x : t.xout;
    {
      GemmEpi e = gepi(bp.fc2.b, xnext, D, ACT_NONE, tb.x1, D); e.scale = bp.ls2;
      TK(launch_gemm_f32(tb.h, F, tb.W2, F, M, D, F, e, s));
    }
  }
  if (cfg->target_dim) {
    TK(launch_layernorm(t.xout, nullptr, p->lnf_w, p->lnf_b, d.eps, M, D, t.f, nullptr, s));
    TK(lin_fwd(t.f, D, p->proj_w, p->proj_b, M, d.Dd, D, mem_out, d.Dd, ACT_NONE, s));
  } else {
    TK(launch_layernorm(t.xout, nullptr, p->lnf_w, p->lnf_b, d.eps, M, D, mem_out, nullptr, s));
  }
  return DOD_OK;
}

int dod_backbone_tail_train_backward(const dod_config* cfg, const dod_bb_tail_params* p, int B, int N, const float* d_mem, const void* tape,
                                     size_t tape_bytes, const dod_bb_tail_params* grads, void* ws, size_t ws_bytes, void* stream) {
  if (!p || !p->blocks || !grads || !grads->blocks || grads->nblocks != p->nblocks) return tfail(DOD_ERR_INVALID, "backbone tail: null / mismatched parameters");
  TDims d;
  if (!make_tdims(cfg, B, N, p->nblocks, &d)) return tfail(DOD_ERR_INVALID, "backbone tail: unsupported configuration");
  if (!d_mem || !tape || !ws) return tfail(DOD_ERR_INVALID, "backbone tail: null buffer");
  if (tape_bytes < dod_backbone_tail_tape_bytes(cfg, B, N, p->nblocks) || ws_bytes < dod_backbone_tail_workspace_bytes(cfg, B, N, p->nblocks))
    return tfail(DOD_ERR_STATE, "backbone tail: tape / workspace too small");
  hipStream_t s = (hipStream_t)stream;
  TTape t; TScratch sc;
  carve_ttape(d, (void*)(((uintptr_t)tape + 255) & ~(uintptr_t)255), &t);
  carve_tscratch(d, (void*)(((uintptr_t)ws + 255) & ~(uintptr_t)255), &sc);
  auto G = [](const float* q) { return const_cast<float*>(q); };
  const int M = d.M, D = d.D, F = d.F, F1 = d.F1;
  const size_t nMD = (size_t)M * D;
  const float scale = 1.0f / sqrtf((float)d.dh);
  auto blocks_for = [](size_t n) { return dim3((unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096)); };
  // ---- projection + final LayerNorm (frozen affine: its parameter gradients go to a dump)
  if (cfg->target_dim) {
    TK(lin_bwd_w(d_mem, d.Dd, t.f, D, M, d.Dd, D, G(grads->proj_w), G(grads->proj_b), s));
    TK(lin_bwd_x(d_mem, d.Dd, p->proj_w, M, d.Dd, D, sc.da, false, s));
    TK(ln_bwd(t.xout, p->lnf_w, sc.da, d.eps, M, D, sc.dx, sc.dump, sc.dump + D, s));
  } else {
    TK(ln_bwd(t.xout, p->lnf_w, d_mem, d.eps, M, D, sc.dx, sc.dump, sc.dump + D, s));
  }
  // sc.dx = d(block output)
  for (int i = d.nb - 1; i >= 0; --i) {
    const dod_bb_block_params& bp = p->blocks[i];
    const dod_bb_block_params& gp = grads->blocks[i];
    const auto& tb = t.b[i];
    // x2 = x1 + ls2 * (h W2'^T + b2)
    hipLaunchKernelGGL(colscale_kernel, blocks_for(nMD), dim3(256), 0, s, sc.dx, bp.ls2, sc.da, nMD, D);                  // da = d(fc2 out)
    TH(hipGetLastError());
    TK(lora_grads(d, tb.h, F, sc.da, D, D, bp.fc2.A, bp.fc2.Bm, G(gp.fc2.A), G(gp.fc2.Bm), sc, s));
    if (d.swiglu) {     // d(h) [M, F], then d(pre) = [d(x1) | d(x2)] in dbig
      TK(lin_bwd_x(sc.da, D, tb.W2, M, D, F, sc.dh, false, s));
      hipLaunchKernelGGL(swiglu_bwd_kernel, blocks_for((size_t)M * F), dim3(256), 0, s, sc.dh, tb.pre, sc.dbig, (size_t)M, F);
    } else {
      TK(lin_bwd_x(sc.da, D, tb.W2, M, D, F, sc.dbig, false, s));                                                     // d(h)
      hipLaunchKernelGGL(gelu_bwd_kernel, blocks_for((size_t)M * F), dim3(256), 0, s, sc.dbig, tb.pre, sc.dbig, (size_t)M * F);
    }
    TH(hipGetLastError());
    TK(lora_grads(d, tb.y2, D, sc.dbig, F1, F1, bp.fc1.A, bp.fc1.Bm, G(gp.fc1.A), G(gp.fc1.Bm), sc, s));
    TK(lin_bwd_x(sc.dbig, F1, tb.W1, M, F1, D, sc.da, false, s));                                                     // d(y2)
    TK(ln_bwd(tb.x1, bp.ln2_w, sc.da, d.eps, M, D, sc.db, sc.dump, sc.dump + D, s));
    TK(add_inplace(sc.dx, sc.db, nMD, s));                                                                                // dx = d(x1)
    // x1 = x + ls1 * (ctx Wo'^T + bo)
    hipLaunchKernelGGL(colscale_kernel, blocks_for(nMD), dim3(256), 0, s, sc.dx, bp.ls1, sc.da, nMD, D);
    TH(hipGetLastError());
    TK(lora_grads(d, tb.ctx, D, sc.da, D, D, bp.o.A, bp.o.Bm, G(gp.o.A), G(gp.o.Bm), sc, s));
    TK(lin_bwd_x(sc.da, D, tb.Wo, M, D, D, sc.db, false, s));                                                         // db = d(ctx)
    if (tail_flash_bwd(d)) {
      AttnF32Bwd g;
      g.q = tb.qkv; g.k = tb.qkv + D; g.v = tb.qkv + 2 * D; g.o = tb.ctx; g.d_o = sc.db; g.lse = tb.lse;
      g.dq = sc.dqkv; g.dk = sc.dqkv + D; g.dv = sc.dqkv + 2 * D; g.delta = sc.delta;
      g.ldq = g.ldk = g.ldv = g.lddq = g.lddk = g.lddv = 3 * D; g.ldo = D;
      g.Lq = g.Lk = N; g.B = B; g.heads = d.H; g.dh = d.dh; g.scale = scale;
      TK(launch_attn_f32_bwd(g, s));
    } else {
      TK(launch_mha_bwd(tb.qkv, 3 * D, sc.db, D, sc.dqkv, sc.dS, sc.Pd, B, N, d.H, D, d.dh, scale, 0.f, 0ull, s));
    }
    const dod_lora_linear* qkv3[3] = {&bp.q, &bp.k, &bp.v};
    const dod_lora_linear* gqkv3[3] = {&gp.q, &gp.k, &gp.v};
    for (int c = 0; c < 3; ++c)
      TK(lora_grads(d, tb.y1, D, sc.dqkv + (size_t)c * D, 3 * D, D, qkv3[c]->A, qkv3[c]->Bm, G(gqkv3[c]->A), G(gqkv3[c]->Bm), sc, s));
    if (i > 0) {        // the tail's input is the frozen prefix's output: nothing below block 0 needs a gradient
      TK(lin_bwd_x(sc.dqkv, 3 * D, tb.Wqkv, M, 3 * D, D, sc.da, false, s));                                           // d(y1)
      TK(ln_bwd(tb.x, bp.ln1_w, sc.da, d.eps, M, D, sc.db, sc.dump, sc.dump + D, s));
      TK(add_inplace(sc.dx, sc.db, nMD, s));                                                                              // dx = d(x): the block below's output
    }
  }
  return DOD_OK;
}

}  // extern "C"
