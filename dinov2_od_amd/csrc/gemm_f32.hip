// Exact-fp32 MFMA GEMM (v_mfma_f32_32x32x2_f32: bit-for-bit a k-ordered fmaf chain) with the same
// fused epilogues as the bf16 kernel.  Used for (a) the strict-parity mode of the backbone, the mode
// gated at 1e-3 against the reference's fp32 CPU forward, and (b) every decoder/head linear in both
// modes (K11-K13, K17-K19: deformable_attention.py:86-94,181,232-238,264-266; detr_decoder.py:80-81),
// whose sampling-coordinate math must stay fp32 (floor() at deformable_attention.py:114-115).
//   C[M,N] = A[M,K] * W[N,K]^T, arbitrary M, N (guarded), K % 4 == 0.
// Tile 64x64x16, 256 threads = 4 waves (2x2), one 32x32 accumulator per wave, computed transposed
// (D = W_tile * A_tile^T) so a lane owns an output row and register quads run along n.
// LDS holds both operands k-major ([k][row], stride 66 floats) so the one-float-per-lane MFMA operands
// are conflict-free ds_read_b32 (lanes 0-31 consecutive rows at k, lanes 32-63 at k+1).
#include "dod_common.h"
#include <cstdlib>

#define FBM 64
#define FBN 64
#define FBK 16
#define FLD 66   // 4*FLD == 8 (mod 32): the transposing ds_write_b32 pattern below is conflict-free

__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, int lda,
                                                       const float* __restrict__ W, int ldw,
                                                       int M, int N, int K, GemmEpi e) {
  __shared__ float sA[2][FBK * FLD];
  __shared__ float sW[2][FBK * FLD];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int tiles_m = (M + FBM - 1) / FBM;
  const int tn = blockIdx.x / tiles_m, tm = blockIdx.x - tn * tiles_m;
  const int m0 = tm * FBM, n0 = tn * FBN;

  // staging: thread -> (row = tid/4, kq = tid%4): 4 lanes read one row's 64 contiguous bytes
  const int srow = tid >> 2, kq = tid & 3;
  const bool va = (m0 + srow) < M, vw = (n0 + srow) < N;
  const float* gA = A + (size_t)(va ? m0 + srow : 0) * lda + kq * 4;
  const float* gW = W + (size_t)(vw ? n0 + srow : 0) * ldw + kq * 4;
  float4 ra, rw;
  auto gload = [&](int k0) {
    const bool kin = (k0 + kq * 4) < K;   // K % 4 == 0
    ra = (va && kin) ? *reinterpret_cast<const float4*>(gA + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
    rw = (vw && kin) ? *reinterpret_cast<const float4*>(gW + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  auto lwrite = [&](int st) {
    float* a = &sA[st][(kq * 4) * FLD + srow];
    a[0] = ra.x; a[FLD] = ra.y; a[2 * FLD] = ra.z; a[3 * FLD] = ra.w;
    float* w = &sW[st][(kq * 4) * FLD + srow];
    w[0] = rw.x; w[FLD] = rw.y; w[2 * FLD] = rw.z; w[3 * FLD] = rw.w;
  };

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int nk = (K + FBK - 1) / FBK;
  gload(0);
  lwrite(0);
  __syncthreads();
  const int lr = lane & 31, lh = lane >> 5;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) gload((kt + 1) * FBK);
    const float* a = &sA[kt & 1][wm * 32 + lr];
    const float* w = &sW[kt & 1][wn * 32 + lr];
#pragma unroll
    for (int s = 0; s < FBK / 2; ++s) {
      const float av = a[(2 * s + lh) * FLD];
      const float wv = w[(2 * s + lh) * FLD];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv, av, acc, 0, 0, 0);
    }
    if (kt + 1 < nk) lwrite((kt + 1) & 1);
    __syncthreads();
  }

  const int m = m0 + wm * 32 + lr;
  if (m >= M) return;
  size_t orow = (size_t)m;
  const float* posrow = nullptr;
  if (e.rows_per_img > 0) {
    const int b = m / e.rows_per_img, p = m - b * e.rows_per_img;
    orow = (size_t)b * e.out_rows_per_img + 1 + p;
    posrow = e.pos + (size_t)(1 + p) * N;
  }
#pragma unroll
  for (int g = 0; g < 4; ++g) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int n = n0 + wn * 32 + 8 * g + 4 * lh + t;
      if (n >= N) continue;
      float v = acc[4 * g + t];
      if (e.bias) v += e.bias[n];
      if (e.act == ACT_GELU) v = gelu_erf(v);
      else if (e.act == ACT_RELU) v = fmaxf(v, 0.f);
      else if (e.act == ACT_SIGMOID) v = sigmoidf_(v);
      if (e.scale) v *= e.scale[n];
      if (posrow) v += posrow[n];
      if (e.resid) v += e.resid[orow * e.ldr + n];
      if (e.out_f32) e.out_f32[orow * e.ldc + n] = v;
      else e.out_bf16[orow * e.ldc + n] = f2bf(v);
    }
  }
}

int launch_gemm_f32(const float* A, int lda, const float* W, int ldw, int M, int N, int K,
                    const GemmEpi& e, hipStream_t s) {
  if (M <= 0 || N <= 0 || K <= 0) return 1;
  if (K % 4 != 0 || lda % 4 != 0 || ldw % 4 != 0) return 2;
  if (!e.out_f32 && !e.out_bf16) return 2;
  const int tiles = ((M + FBM - 1) / FBM) * ((N + FBN - 1) / FBN);
  hipLaunchKernelGGL(gemm_f32_kernel, dim3(tiles), dim3(256), 0, s, A, lda, W, ldw, M, N, K, e);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}
