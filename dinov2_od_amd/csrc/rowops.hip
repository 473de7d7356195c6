// Row-wise / elementwise kernels of the forward path: LayerNorm (K3, K17/K18 post-norms), patch im2col
// (K1 front end), CLS row (K2), bicubic position-table resize (K2), SwiGLU gate (K7g) and the
// one-time weight packing helpers (LoRA merge K8, casts, concatenation).  All HBM-bound:
// wave-per-row with 16-byte accesses, fp32 statistics, shuffle reductions.
#include "dod_common.h"

// ----------------------------------------------------------------------------- LayerNorm
// nn.LayerNorm over the last dim (modeling_dinov2.py:348,353,441 eps 1e-6; deformable_attention.py:197-209
// eps 1e-5).  One wave per row, D % 4 == 0, D <= 2048.  Two-pass statistics in registers
// (mean, then centred variance) like ATen's RowwiseMoments result to fp32 rounding.
#define LN_MAXC 8
// Output forms (compile-time: the round-2 kernel took them as runtime pointers / an int and carried every writer's registers -- 100
// VGPRs, four waves per SIMD -- which cost 27 % of its bandwidth).
enum { LN_F32 = 1, LN_BF16 = 2, LN_PAIR = 4, LN_H2 = 8, LN_FP8 = 16, LN_S3 = 32, LN_FP8MX = 64 };
struct LnOut { float* f32; bf16_t* bf16; unsigned char* fp8; float* scale; bf16_t* split; bf16_t* a3; unsigned char* bs; };
// NCH float4 chunks per lane (compile-time trip count: D = 768 holds a row in 3 x 4 registers instead of LN_MAXC x 4); EXACT: D == 256 NCH
template <int NCH, bool EXACT, int OUT>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ add,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float eps, int rows, int D, LnOut o) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nc = D >> 2;
  const float4* xr = reinterpret_cast<const float4*>(x + (size_t)row * D);
  const float4* ar = add ? reinterpret_cast<const float4*>(add + (size_t)row * D) : nullptr;
  float4 v[NCH];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + 64 * i;
    if (EXACT || c < nc) v[i] = xr[c];
  }
  if (ar) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + 64 * i;
      if (EXACT || c < nc) { const float4 a = ar[c]; v[i].x += a.x; v[i].y += a.y; v[i].z += a.z; v[i].w += a.w; }
    }
  }
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + 64 * i;
    if (EXACT || c < nc) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + 64 * i;
    if (EXACT || c < nc) {
      const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
      q += (a * a + b * b) + (cc * cc + d * d);
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + 64 * i;
    if (EXACT || c < nc) {
      const float4 g = reinterpret_cast<const float4*>(gamma)[c];
      const float4 b = reinterpret_cast<const float4*>(beta)[c];
      v[i].x = (v[i].x - mean) * rstd * g.x + b.x;
      v[i].y = (v[i].y - mean) * rstd * g.y + b.y;
      v[i].z = (v[i].z - mean) * rstd * g.z + b.z;
      v[i].w = (v[i].w - mean) * rstd * g.w + b.w;
    }
  }
  if (OUT & LN_FP8MX) {   // block-scaled e4m3 operand (dod_common.h mx_ebyte): one e8m0 byte per 32 columns = 8 consecutive lanes' float4s
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + 64 * i;
      const bool ok = EXACT || c < nc;
      float amax = ok ? fmaxf(fmaxf(fabsf(v[i].x), fabsf(v[i].y)), fmaxf(fabsf(v[i].z), fabsf(v[i].w))) : 0.f;
      amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
      amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
      amax = fmaxf(amax, __shfl_xor(amax, 4, 64));
      const unsigned eb = mx_ebyte(amax);
      const float inv = mx_inv_scale(eb);
      if (ok) {
        reinterpret_cast<unsigned*>(o.fp8 + (size_t)row * D)[c] = pack4_fp8(v[i].x * inv, v[i].y * inv, v[i].z * inv, v[i].w * inv);
        if ((lane & 7) == 0) o.bs[(size_t)row * (D >> 5) + mx_scale_off(D, c >> 3)] = (unsigned char)eb;
      }
    }
    return;
  }
  if (OUT & LN_FP8) {   // fp8 operand for the next GEMM: normalised row kept in registers, one scale per row (amax / 448)
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + 64 * i;
      if (EXACT || c < nc) amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v[i].x), fabsf(v[i].y)), fmaxf(fabsf(v[i].z), fabsf(v[i].w))));
    }
    amax = wave_max(amax);
    const float sc = amax > 0.f ? amax / 448.0f : 1.0f;
    const float inv = 1.0f / sc;
    if (lane == 0) o.scale[row] = sc;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + 64 * i;
      if (EXACT || c < nc) reinterpret_cast<unsigned*>(o.fp8 + (size_t)row * D)[c] = pack4_fp8(v[i].x * inv, v[i].y * inv, v[i].z * inv, v[i].w * inv);
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + 64 * i;
    if (EXACT || c < nc) {
      const float4 w = v[i];
      if (OUT & LN_F32) reinterpret_cast<float4*>(o.f32 + (size_t)row * D)[c] = w;
      if (OUT & LN_H2) {   // H2 operand row (fp16 | e4m3 main, e4m3 remainder per 32-k block; dod_common.h)
        uint2 f16; unsigned hi8, lo8;
        h2_quad(w, 1.0f, f16, hi8, lo8);
        char* ob = reinterpret_cast<char*>(o.split) + (size_t)row * 4 * D;
        reinterpret_cast<uint2*>(ob)[c] = f16;
        char* p8 = ob + h2_off8(D, 4 * c);
        *reinterpret_cast<unsigned*>(p8) = hi8;
        *reinterpret_cast<unsigned*>(p8 + 16) = lo8;
      }
      if (OUT & LN_PAIR) {   // split-product operand in the pair layout [hi | lo] (row pitch 2*D)
        uint2 hi, lo;
        hi.x = pack2bf(w.x, w.y);
        hi.y = pack2bf(w.z, w.w);
        lo.x = pack2bf(w.x - __uint_as_float(hi.x << 16), w.y - __uint_as_float(hi.x & 0xffff0000u));
        lo.y = pack2bf(w.z - __uint_as_float(hi.y << 16), w.w - __uint_as_float(hi.y & 0xffff0000u));
        bf16_t* ob = o.split + (size_t)row * 2 * D;
        reinterpret_cast<uint2*>(ob)[c] = hi;
        reinterpret_cast<uint2*>(ob + D)[c] = lo;
      }
      if (OUT & LN_S3) {     // bf16x3 activation operand [hi | hi | lo] (row pitch 3*D) of the decoder's query-side linears: saves their split3 launch
        uint2 hi, lo;
        hi.x = pack2bf(w.x, w.y);
        hi.y = pack2bf(w.z, w.w);
        lo.x = pack2bf(w.x - __uint_as_float(hi.x << 16), w.y - __uint_as_float(hi.x & 0xffff0000u));
        lo.y = pack2bf(w.z - __uint_as_float(hi.y << 16), w.w - __uint_as_float(hi.y & 0xffff0000u));
        bf16_t* ob = o.a3 + (size_t)row * 3 * D;
        reinterpret_cast<uint2*>(ob)[c] = hi;
        reinterpret_cast<uint2*>(ob + D)[c] = hi;
        reinterpret_cast<uint2*>(ob + 2 * D)[c] = lo;
      }
      if (OUT & LN_BF16) {
        uint2 p;
        p.x = pack2bf(w.x, w.y);
        p.y = pack2bf(w.z, w.w);
        reinterpret_cast<uint2*>(o.bf16 + (size_t)row * D)[c] = p;
      }
    }
  }
}

template <int OUT>
static void ln_dispatch(const float* x, const float* add, const float* gamma, const float* beta, float eps, int rows, int D,
                        const LnOut& o, hipStream_t s) {
  const dim3 grid((rows + 3) / 4), block(256);
  const int nc = D >> 2, nch = (nc + 63) / 64;
  const bool exact = nc == 64 * nch;
#define LN_GO(N, E) hipLaunchKernelGGL((layernorm_kernel<N, E, OUT>), grid, block, 0, s, x, add, gamma, beta, eps, rows, D, o)
  if (exact) {
    switch (nch) {
      case 1: LN_GO(1, true); return;
      case 2: LN_GO(2, true); return;
      case 3: LN_GO(3, true); return;     // D = 768 (ViT-B, the decoder)
      case 4: LN_GO(4, true); return;     // D = 1024 (ViT-L)
      case 6: LN_GO(6, true); return;     // D = 1536 (ViT-g)
      default: break;
    }
  }
  if (nch <= 1) LN_GO(1, false);
  else if (nch <= 2) LN_GO(2, false);     // D = 384 (ViT-S)
  else if (nch <= 4) LN_GO(4, false);
  else LN_GO(LN_MAXC, false);
#undef LN_GO
}

int launch_layernorm(const float* x, const float* add, const float* gamma, const float* beta, float eps,
                     int rows, int D, float* out_f32, bf16_t* out_bf16, hipStream_t s, unsigned char* out_fp8,
                     float* out_scale, bf16_t* out_split3, int h2, bf16_t* out_a3, unsigned char* out_bs) {
  if (rows <= 0) return 1;
  if (D % 4 != 0 || D > 256 * LN_MAXC) return 2;
  if (h2 && (!out_split3 || D % 32 != 0)) return 2;
  if (out_bs) {        // block-scaled e4m3 rows: out_fp8 + one e8m0 byte per 32 columns (layout [rows][2][D / 64])
    if (!out_fp8 || out_scale || out_f32 || out_bf16 || out_split3 || out_a3 || D % 64 != 0) return 2;
    LnOut o{nullptr, nullptr, out_fp8, nullptr, nullptr, nullptr, out_bs};
    ln_dispatch<LN_FP8MX>(x, add, gamma, beta, eps, rows, D, o, s);
    return hipGetLastError() == hipSuccess ? 0 : 3;
  }
  if ((out_fp8 == nullptr) != (out_scale == nullptr)) return 2;
  LnOut o{out_f32, out_bf16, out_fp8, out_scale, out_split3, out_a3, nullptr};
  if (out_a3) {
    if (!out_f32 || out_bf16 || out_fp8 || out_split3) return 2;
    ln_dispatch<LN_F32 | LN_S3>(x, add, gamma, beta, eps, rows, D, o, s);
  } else if (out_fp8) ln_dispatch<LN_FP8>(x, add, gamma, beta, eps, rows, D, o, s);
  else if (out_split3 && (out_f32 || out_bf16)) return 2;
  else if (out_split3) { if (h2) ln_dispatch<LN_H2>(x, add, gamma, beta, eps, rows, D, o, s); else ln_dispatch<LN_PAIR>(x, add, gamma, beta, eps, rows, D, o, s); }
  else if (out_f32 && out_bf16) ln_dispatch<LN_F32 | LN_BF16>(x, add, gamma, beta, eps, rows, D, o, s);
  else if (out_bf16) ln_dispatch<LN_BF16>(x, add, gamma, beta, eps, rows, D, o, s);
  else if (out_f32) ln_dispatch<LN_F32>(x, add, gamma, beta, eps, rows, D, o, s);
  else return 2;
  return hipGetLastError() == hipSuccess ? 0 : 3;
}

// ----------------------------------------------------------------------------- folded LayerNorm (GemmEpi::ln_*, dod_common.h)
// The backbone's pre-norm LayerNorms (modeling_dinov2.py:348, 353) are folded into the GEMMs around them: the residual GEMM that produces a
// row also writes it in the next GEMM's operand format and its (sum, centred square sum) per 128-column group; the next GEMM normalises
// in its epilogue.  What is left as kernels of their own: the first block's input (rowstats_kernel: its producer is the patch embedding),
// the merge of the group statistics (ln_finalize_kernel) and the one-time weight fold.
template <int NCH, bool EXACT, int KIND>
__global__ __launch_bounds__(256) void rowstats_kernel(const float* __restrict__ x, int rows, int D, float eps, void* __restrict__ op, float2* __restrict__ stats) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nc = D >> 2;
  const float4* xr = reinterpret_cast<const float4*>(x + (size_t)row * D);
  float4 v[NCH];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + 64 * i;
    if (EXACT || c < nc) { v[i] = xr[c]; s += (v[i].x + v[i].y) + (v[i].z + v[i].w); }
  }
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + 64 * i;
    if (EXACT || c < nc) {
      const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
      q += (a * a + b * b) + (cc * cc + d * d);
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
  if (lane == 0) stats[row] = make_float2(mean, rstd);
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + 64 * i;
    if (!(EXACT || c < nc)) continue;
    const float4 w = v[i];
    if (KIND == LNOP_H2) {
      uint2 f16; unsigned hi8, lo8;
      h2_quad(w, 1.0f, f16, hi8, lo8);
      char* ob = reinterpret_cast<char*>(op) + (size_t)row * 4 * D;
      reinterpret_cast<uint2*>(ob)[c] = f16;
      char* p8 = ob + h2_off8(D, 4 * c);
      *reinterpret_cast<unsigned*>(p8) = hi8;
      *reinterpret_cast<unsigned*>(p8 + 16) = lo8;
    } else {
      uint2 hi;
      hi.x = pack2bf(w.x, w.y);
      hi.y = pack2bf(w.z, w.w);
      bf16_t* ob = reinterpret_cast<bf16_t*>(op) + (size_t)row * (KIND == LNOP_PAIR ? 2 : 1) * D;
      reinterpret_cast<uint2*>(ob)[c] = hi;
      if (KIND == LNOP_PAIR) {
        uint2 lo;
        lo.x = pack2bf(w.x - __uint_as_float(hi.x << 16), w.y - __uint_as_float(hi.x & 0xffff0000u));
        lo.y = pack2bf(w.z - __uint_as_float(hi.y << 16), w.w - __uint_as_float(hi.y & 0xffff0000u));
        reinterpret_cast<uint2*>(ob + D)[c] = lo;
      }
    }
  }
}

template <int KIND>
static void rowstats_dispatch(const float* x, int rows, int D, float eps, void* op, float2* stats, hipStream_t s) {
  const dim3 grid((rows + 3) / 4), block(256);
  const int nc = D >> 2, nch = (nc + 63) / 64;
  const bool exact = nc == 64 * nch;
#define RS_GO(N, E) hipLaunchKernelGGL((rowstats_kernel<N, E, KIND>), grid, block, 0, s, x, rows, D, eps, op, stats)
  if (exact && nch == 3) RS_GO(3, true);
  else if (exact && nch == 4) RS_GO(4, true);
  else if (exact && nch == 6) RS_GO(6, true);
  else if (nch <= 2) RS_GO(2, false);
  else if (nch <= 4) RS_GO(4, false);
  else RS_GO(LN_MAXC, false);
#undef RS_GO
}

int launch_rowstats(const float* x, int rows, int D, float eps, void* op, int op_kind, float2* stats, hipStream_t s) {
  if (rows <= 0) return 1;
  if (D % 4 != 0 || D > 256 * LN_MAXC || !op || !stats) return 2;
  if (op_kind == LNOP_H2) { if (D % 32) return 2; rowstats_dispatch<LNOP_H2>(x, rows, D, eps, op, stats, s); }
  else if (op_kind == LNOP_PAIR) rowstats_dispatch<LNOP_PAIR>(x, rows, D, eps, op, stats, s);
  else if (op_kind == LNOP_BF16) rowstats_dispatch<LNOP_BF16>(x, rows, D, eps, op, stats, s);
  else return 2;
  return hipGetLastError() == hipSuccess ? 0 : 3;
}

// group sums of (x - k) and (x - k)^2, k = stats[row].x on entry (the shift the producer subtracted: the row's previous mean) ->
// mean = k + S / D,  var = Q / D - (S / D)^2  (S / D is the CHANGE of the mean: a fraction of the spread, so nothing cancels)
__global__ __launch_bounds__(256) void ln_finalize_kernel(const float2* __restrict__ part, int npart, int rows, int D, float eps, float2* __restrict__ stats) {
  const int row = blockIdx.x * 256 + threadIdx.x;
  if (row >= rows) return;
  const float2* p = part + (size_t)row * npart;
  float S = 0.f, Q = 0.f;
  for (int g = 0; g < npart; ++g) { S += p[g].x; Q += p[g].y; }
  const float k = stats[row].x, dm = S / (float)D;
  const float var = fmaxf(Q / (float)D - dm * dm, 0.f);
  stats[row] = make_float2(k + dm, 1.0f / sqrtf(var + eps));
}

int launch_ln_finalize(const float2* part, int npart, int rows, int D, float eps, float2* stats, hipStream_t s) {
  if (rows <= 0) return 1;
  if (npart != (D + 127) / 128 || !part || !stats) return 2;
  hipLaunchKernelGGL(ln_finalize_kernel, dim3((rows + 255) / 256), dim3(256), 0, s, part, npart, rows, D, eps, stats);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}

// pack time, a wave per output feature n:  Wout[n][k] = W[n][k] gamma[k],  bias_out[n] = bias_in[n] + sum_k W[n][k] beta[k]
__global__ __launch_bounds__(256) void ln_fold_kernel(const float* __restrict__ W, int rows, int cols, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      const float* __restrict__ bias_in, float* __restrict__ Wout, float* __restrict__ bias_out) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= rows) return;
  float acc = 0.f;
  for (int k = lane; k < cols; k += 64) {
    const float w = W[(size_t)n * cols + k];
    acc = fmaf(w, beta[k], acc);
    Wout[(size_t)n * cols + k] = w * gamma[k];
  }
  acc = wave_sum(acc);
  if (lane == 0) bias_out[n] = bias_in[n] + acc;
}

int launch_ln_fold(const float* W, int rows, int cols, const float* gamma, const float* beta, const float* bias_in, float* Wout, float* bias_out, hipStream_t s) {
  if (rows <= 0 || cols <= 0) return 1;
  hipLaunchKernelGGL(ln_fold_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, W, rows, cols, gamma, beta, bias_in, Wout, bias_out);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}

__global__ __launch_bounds__(256) void rowsum_kernel(const float* __restrict__ W, int rows, int cols, int round_bf16, float* __restrict__ c) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= rows) return;
  float acc = 0.f;
  for (int k = lane; k < cols; k += 64) {
    const float w = W[(size_t)n * cols + k];
    acc += round_bf16 ? bf2f(f2bf(w)) : w;
  }
  acc = wave_sum(acc);
  if (lane == 0) c[n] = acc;
}

int launch_rowsum(const float* W, int rows, int cols, int round_bf16, float* c, hipStream_t s) {
  if (rows <= 0 || cols <= 0) return 1;
  hipLaunchKernelGGL(rowsum_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, W, rows, cols, round_bf16, c);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}

// ----------------------------------------------------------------------------- im2col for Conv2d(3->D, k=p, s=p)
// A[m][k]: m = b*gh*gw + py*gw + px, k = c*p*p + i*p + j  (weight.reshape(D, 3*p*p) order,
// modeling_dinov2.py:139,148).  Thread per output element; consecutive k within an (c,i) run read
// consecutive pixels.  Columns k >= 3*p*p are zero (K padding for the bf16 kernel's BK).
__global__ void im2col_kernel(const float* __restrict__ img, int B, int H, int W, int p, int gh, int gw, int Kp,
                              float* __restrict__ out_f32, bf16_t* __restrict__ out_bf16) {
  const size_t total = (size_t)B * gh * gw * Kp;
  const int K = 3 * p * p;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int k = (int)(idx % Kp);
    const size_t m = idx / Kp;
    float v = 0.f;
    if (k < K) {
      const int c = k / (p * p), r = k - c * p * p, i = r / p, j = r - i * p;
      const int px = (int)(m % gw);
      const size_t t = m / gw;
      const int py = (int)(t % gh);
      const int b = (int)(t / gh);
      v = img[(((size_t)b * 3 + c) * H + (py * p + i)) * W + (px * p + j)];
    }
    if (out_f32) out_f32[idx] = v; else out_bf16[idx] = f2bf(v);
  }
}

int launch_im2col(const float* img, int B, int H, int W, int patch, int Kp, float* out_f32, bf16_t* out_bf16, hipStream_t s) {
  const int gh = H / patch, gw = W / patch;
  if (B <= 0 || gh <= 0 || gw <= 0 || Kp < 3 * patch * patch) return 2;
  const size_t total = (size_t)B * gh * gw * Kp;
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(im2col_kernel, dim3(blocks), dim3(256), 0, s, img, B, H, W, patch, gh, gw, Kp, out_f32, out_bf16);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}

// ----------------------------------------------------------------------------- CLS row: x[b][0][:] = cls + pos[0]
__global__ void cls_row_kernel(const float* __restrict__ cls, const float* __restrict__ pos, float* __restrict__ x,
                               int B, int N, int D) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * D) return;
  const int b = i / D, d = i - b * D;
  x[(size_t)b * N * D + d] = cls[d] + pos[d];
}
int launch_cls_row(const float* cls, const float* pos, float* x, int B, int N, int D, hipStream_t s) {
  hipLaunchKernelGGL(cls_row_kernel, dim3((B * D + 255) / 256), dim3(256), 0, s, cls, pos, x, B, N, D);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}

// ----------------------------------------------------------------------------- bicubic position-table resize
// Dinov2Embeddings.interpolate_pos_encoding (modeling_dinov2.py:57-95): the G x G patch position
// table is resized to gh x gw with F.interpolate(mode="bicubic", align_corners=False) in fp32; row 0
// (class position) is copied.  Restates ATen upsample_bicubic2d: scale = in/out, src = scale*(dst+0.5)-0.5
// (not clamped), A = -0.75, taps at floor(src)-1..+2 clamped to the border, x-interpolation first.
__device__ __forceinline__ float cubic1(float x, float A) { return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }
__device__ __forceinline__ float cubic2(float x, float A) { return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }
__global__ void pos_resize_kernel(const float* __restrict__ pin, int G, int gh, int gw, int D, float* __restrict__ pout) {
  const size_t total = (size_t)(gh * gw + 1) * D;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int d = (int)(idx % D);
  const int tok = (int)(idx / D);
  if (tok == 0) { pout[idx] = pin[d]; return; }
  const int oy = (tok - 1) / gw, ox = (tok - 1) - oy * gw;
  const float A = -0.75f;
  const float sy = (float)G / (float)gh, sx = (float)G / (float)gw;
  const float ry = sy * ((float)oy + 0.5f) - 0.5f, rx = sx * ((float)ox + 0.5f) - 0.5f;
  const float fy = floorf(ry), fx = floorf(rx);
  const int iy = (int)fy, ix = (int)fx;
  const float ty = ry - fy, tx = rx - fx;
  const float cy[4] = {cubic2(ty + 1.f, A), cubic1(ty, A), cubic1(1.f - ty, A), cubic2(2.f - ty, A)};
  const float cx[4] = {cubic2(tx + 1.f, A), cubic1(tx, A), cubic1(1.f - tx, A), cubic2(2.f - tx, A)};
  float acc = 0.f;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    int yy = iy - 1 + a; yy = yy < 0 ? 0 : (yy > G - 1 ? G - 1 : yy);
    float rowv = 0.f;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      int xx = ix - 1 + b; xx = xx < 0 ? 0 : (xx > G - 1 ? G - 1 : xx);
      rowv += pin[(size_t)(1 + yy * G + xx) * D + d] * cx[b];
    }
    acc += rowv * cy[a];
  }
  pout[idx] = acc;
}
int launch_pos_resize(const float* pos_in, int G, int gh, int gw, int D, float* pos_out, hipStream_t s) {
  const size_t total = (size_t)(gh * gw + 1) * D;
  hipLaunchKernelGGL(pos_resize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, pos_in, G, gh, gw, D, pos_out);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}

// ----------------------------------------------------------------------------- SwiGLU gate (giant): silu(x1) * x2
// Dinov2SwiGLUFFN.forward, modeling_dinov2.py:310-314: x1, x2 = chunk(2, dim=-1)
__global__ void swiglu_kernel(const float* __restrict__ in_f32, const bf16_t* __restrict__ in_bf16, int rows, int Fh,
                              float* __restrict__ out_f32, bf16_t* __restrict__ out_bf16) {
  const size_t total = (size_t)rows * Fh;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const size_t r = idx / Fh;
    const int c = (int)(idx - r * Fh);
    float a, b;
    if (in_f32) { a = in_f32[r * 2 * Fh + c]; b = in_f32[r * 2 * Fh + Fh + c]; }
    else { a = bf2f(in_bf16[r * 2 * Fh + c]); b = bf2f(in_bf16[r * 2 * Fh + Fh + c]); }
    const float v = a / (1.0f + expf(-a)) * b;
    if (out_f32) out_f32[idx] = v; else out_bf16[idx] = f2bf(v);
  }
}
int launch_swiglu(const float* in_f32, const bf16_t* in_bf16, int rows, int Fh, float* out_f32, bf16_t* out_bf16, hipStream_t s) {
  const size_t total = (size_t)rows * Fh;
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(swiglu_kernel, dim3(blocks), dim3(256), 0, s, in_f32, in_bf16, rows, Fh, out_f32, out_bf16);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}

// ----------------------------------------------------------------------------- packing helpers (one-time, at finalize)
__global__ void cast_bf16_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = f2bf(in[i]);
}
int launch_cast_bf16(const float* in, bf16_t* out, size_t n, hipStream_t s) {
  const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  hipLaunchKernelGGL(cast_bf16_kernel, dim3(blocks), dim3(256), 0, s, in, out, n);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}

__global__ void copy2d_kernel(const float* __restrict__ src, int ld_src, float* __restrict__ dst, int ld_dst, int rows,
                              int cols, int cols_pad) {
  const size_t total = (size_t)rows * cols_pad;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = i / cols_pad;
    const int c = (int)(i - r * cols_pad);
    dst[r * ld_dst + c] = c < cols ? src[r * ld_src + c] : 0.f;
  }
}
int launch_copy2d(const float* src, int ld_src, float* dst, int ld_dst, int rows, int cols, int cols_pad, hipStream_t s) {
  const size_t total = (size_t)rows * cols_pad;
  if (total == 0) return 0;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(copy2d_kernel, dim3(blocks), dim3(256), 0, s, src, ld_src, dst, ld_dst, rows, cols, cols_pad);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}

// LoraLinear (dino_detector/utils.py:68-70): linear(x) + alpha * lora_B(lora_A(x))  ==  x (W + alpha B A)^T + b.
// fp32 merge, summed over r in index order.
__global__ void lora_merge_kernel(const float* __restrict__ W, const float* __restrict__ A, const float* __restrict__ Bm,
                                  float alpha, int out_f, int in_f, int r, float* __restrict__ dst) {
  const size_t total = (size_t)out_f * in_f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t o = i / in_f;
    const int c = (int)(i - o * in_f);
    float acc = 0.f;
    for (int t = 0; t < r; ++t) acc += Bm[o * r + t] * A[(size_t)t * in_f + c];
    dst[i] = W[i] + alpha * acc;
  }
}
int launch_lora_merge(const float* W, const float* A, const float* Bm, float alpha, int out_f, int in_f, int r, float* dst, hipStream_t s) {
  const size_t total = (size_t)out_f * in_f;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(lora_merge_kernel, dim3(blocks), dim3(256), 0, s, W, A, Bm, alpha, out_f, in_f, r, dst);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}

// tgt = query_embed.weight.unsqueeze(0).repeat(B,1,1)   detr_decoder.py:59
__global__ void bcast_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, size_t n) {
  const size_t total = (size_t)B * n;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i % n];
}
int launch_bcast_rows(const float* src, float* dst, int B, int rows, int D, hipStream_t s) {
  const size_t n = (size_t)rows * D, total = n * B;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(bcast_rows_kernel, dim3(blocks), dim3(256), 0, s, src, dst, B, n);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}

// ----------------------------------------------------------------------------- bf16x3 operand split
// x = hi + lo with hi = bf16(x), lo = bf16(x - hi): A W^T ~= Ah Wh^T + Ah Wl^T + Al Wh^T (the lo*lo term, 2^-18
// relative, is dropped).  Written as ONE bf16 GEMM with K' = 3K:  A' = [Ah | Ah | Al],  W' = [Wh | Wl | Wh].
// Used for the decoder's query-side linears in bf16 mode: ~fp32 accuracy (1e-5) at bf16 MFMA rate / 3 instead of the
// fp32 MFMA rate (1/16).   mode 0: activation layout [hi | hi | lo];  mode 1: weight layout [hi | lo | hi].
__global__ void split3_kernel(const float* __restrict__ in, int ld_in, bf16_t* __restrict__ out, int rows, int K, int mode) {
  const size_t total = (size_t)rows * K;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = i / K;
    const int c = (int)(i - r * K);
    const float x = in[r * ld_in + c];
    const bf16_t hi = f2bf(x);
    const bf16_t lo = f2bf(x - bf2f(hi));
    bf16_t* o = out + r * 3 * (size_t)K + c;
    o[0] = hi;
    o[K] = mode ? lo : hi;
    o[2 * (size_t)K] = mode ? hi : lo;
  }
}
// pair layout [hi | lo] (row pitch 2K): operands of the split-product kernels (gemm_x3.hip, attn_x3.hip)
__global__ void split2_kernel(const float* __restrict__ in, int ld_in, bf16_t* __restrict__ out, int rows, int K) {
  const size_t total = (size_t)rows * K;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = i / K;
    const int c = (int)(i - r * K);
    const float x = in[r * ld_in + c];
    const bf16_t hi = f2bf(x);
    bf16_t* o = out + r * 2 * (size_t)K + c;
    o[0] = hi;
    o[K] = f2bf(x - bf2f(hi));
  }
}
int launch_split2(const float* in, int ld_in, bf16_t* out, int rows, int K, hipStream_t s) {
  const size_t total = (size_t)rows * K;
  if (total == 0) return 0;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(split2_kernel, dim3(blocks), dim3(256), 0, s, in, ld_in, out, rows, K);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}
// fp32 [rows, K] -> H2 operand rows (dod_common.h).  One wave per row.  Weight form (wexp != null): the row's e4m3 scale 2^e is the
// power of two that brings max|fp16(w)| into (224, 448]; only the remainder bytes are stored (3K-byte rows).
__global__ __launch_bounds__(256) void split_h2_kernel(const float* __restrict__ in, int ld_in, char* __restrict__ out, int rows, int K,
                                                       unsigned char* __restrict__ wexp) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = in + (size_t)row * ld_in;
  float shi = 1.0f;
  if (wexp) {
    float amax = 0.f;
    for (int c = lane; c < K; c += 64) amax = fmaxf(amax, fabsf((float)(_Float16)clampf_(xr[c], 65504.f)));
    amax = wave_max(amax);
    int e = 0;
    if (amax > 0.f) {
      int ex; const float m = frexpf(amax, &ex);           // amax = m 2^ex, m in [0.5, 1): amax 2^e <= 448 = 0.875 * 2^9
      e = (m <= 0.875f ? 9 : 8) - ex;
      e = e < -100 ? -100 : (e > 100 ? 100 : e);
    }
    shi = ldexpf(1.0f, e);
    if (lane == 0) wexp[row] = (unsigned char)(127 - e);
  }
  char* ob = out + (size_t)row * (wexp ? 3 : 4) * K;
  for (int c = lane * 4; c < K; c += 256) {
    const float4 v = make_float4(xr[c], xr[c + 1], xr[c + 2], xr[c + 3]);
    uint2 f16; unsigned hi8, lo8;
    h2_quad(v, shi, f16, hi8, lo8);
    *reinterpret_cast<uint2*>(ob + 2 * c) = f16;
    if (wexp) *reinterpret_cast<unsigned*>(ob + 2 * K + c) = lo8;
    else {
      char* p8 = ob + h2_off8(K, c);
      *reinterpret_cast<unsigned*>(p8) = hi8;
      *reinterpret_cast<unsigned*>(p8 + 16) = lo8;
    }
  }
}
int launch_split_h2(const float* in, int ld_in, void* out, int rows, int K, unsigned char* wexp, hipStream_t s) {
  if (rows <= 0) return 1;
  if (K % 32 != 0) return 2;
  hipLaunchKernelGGL(split_h2_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, in, ld_in, (char*)out, rows, K, wexp);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}

int launch_split3(const float* in, int ld_in, bf16_t* out, int rows, int K, int mode, hipStream_t s) {
  const size_t total = (size_t)rows * K;
  if (total == 0) return 0;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(split3_kernel, dim3(blocks), dim3(256), 0, s, in, ld_in, out, rows, K, mode);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}

// dst[2i] = src[i], dst[2i + 1] = src[F + i] over rows of `cols` floats (weight pack time: SwiGLU pairs adjacent for GemmEpi::glu)
__global__ void interleave_halves_kernel(const float* __restrict__ src, float* __restrict__ dst, int F, int cols) {
  const size_t total = (size_t)2 * F * cols;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const size_t r = idx / cols; const int c = (int)(idx - r * cols);
    const size_t sr = (r & 1) ? (size_t)F + (r >> 1) : (r >> 1);
    dst[idx] = src[sr * cols + c];
  }
}
int launch_interleave_halves(const float* src, float* dst, int F, int cols, hipStream_t s) {
  if (F <= 0 || cols <= 0) return 1;
  const size_t total = (size_t)2 * F * cols;
  hipLaunchKernelGGL(interleave_halves_kernel, dim3((unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096)), dim3(256), 0, s, src, dst, F, cols);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}
