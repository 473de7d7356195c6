// bf16 MFMA GEMM with fused epilogues for the DINOv2 blocks (K4, K6, K7, K9, K14 and the patch-embed K1).
//   C[M,N] = A[M,K] * W[N,K]^T   A, W bf16 row-major (K contiguous), fp32 accumulate.
// Reference ops replaced: nn.Linear in Dinov2SelfAttention / Dinov2SelfOutput / Dinov2MLP
// (site-packages/transformers/models/dinov2/modeling_dinov2.py:199-201, 246-252, 281-297),
// LayerScale + residual (:367-370, :377-380), Conv2d patch projection (:139,148).
//
// Tiling (gfx950): 128x128x64 block tile, 256 threads = 4 waves (2x2), each wave a 64x64 sub-tile
// as 2x2 v_mfma_f32_32x32x16_bf16 accumulators (64 acc VGPRs).  The product is computed
// TRANSPOSED (D = W_tile * A_tile^T) so that a lane owns one output row m and its registers
// run along n: 4 consecutive n per register quad -> 8-byte (bf16) / 16-byte (fp32) row-major
// stores and float4 bias/scale/residual reads in the epilogue.
// Staging: global -> VGPR (16 B/lane, coalesced 128-B rows) -> LDS with a 16-B-chunk XOR
// swizzle (chunk ^= (row>>1)&7 on 128-B rows: conflict-free for the ds_read_b128 lane groups),
// double-buffered; the next tile's global loads are issued before the MFMA phase and written to
// the other LDS buffer after it (one barrier per K-tile).
#include "dod_common.h"

#define BM 128
#define BN 128
#define BK 64
#define STAGE_BYTES (2 * BM * BK * 2)   // A + W tile, bf16

__device__ __forceinline__ int swz128(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

__global__ __launch_bounds__(256) void gemm_bf16_kernel(const bf16_t* __restrict__ A, int lda,
                                                        const bf16_t* __restrict__ W, int ldw,
                                                        int M, int N, int K, GemmEpi e) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  // XCD-aware remap (bijective for any grid): blocks b, b+8, ... share an XCD/L2; give each XCD a
  // contiguous run of tiles that walk M for a fixed weight panel.
  const int nwg = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tn = bid / tiles_m, tm = bid - tn * tiles_m;
  const int m0 = tm * BM, n0 = tn * BN;

  // staging assignment: 4 chunks of A and 4 of W per thread per K-tile
  const bf16_t* gA[4];
  const bf16_t* gW[4];
  int ldsoff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int id = tid + 256 * i, row = id >> 3, c = id & 7;
    int ra = m0 + row; ra = ra < M ? ra : M - 1;
    int rw = n0 + row; rw = rw < N ? rw : N - 1;
    gA[i] = A + (size_t)ra * lda + c * 8;
    gW[i] = W + (size_t)rw * ldw + c * 8;
    ldsoff[i] = row * 128 + swz128(row, c) * 16;
  }
  uint4 ra_[4], rw_[4];
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ra_[i] = *reinterpret_cast<const uint4*>(gA[i] + k0);
      rw_[i] = *reinterpret_cast<const uint4*>(gW[i] + k0);
    }
  };
  auto lwrite = [&](int stage) {
    char* sA = smem + stage * STAGE_BYTES;
    char* sW = sA + BM * BK * 2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<uint4*>(sA + ldsoff[i]) = ra_[i];
      *reinterpret_cast<uint4*>(sW + ldsoff[i]) = rw_[i];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = K / BK;
  gload(0);
  lwrite(0);
  __syncthreads();
  const int lr = lane & 31, lh = lane >> 5;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) gload((kt + 1) * BK);
    const char* sA = smem + (kt & 1) * STAGE_BYTES;
    const char* sW = sA + BM * BK * 2;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      bf16x8 af[2], wf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int rowa = wm * 64 + i * 32 + lr;
        af[i] = *reinterpret_cast<const bf16x8*>(sA + rowa * 128 + swz128(rowa, kk * 2 + lh) * 16);
        const int roww = wn * 64 + i * 32 + lr;
        wf[i] = *reinterpret_cast<const bf16x8*>(sW + roww * 128 + swz128(roww, kk * 2 + lh) * 16);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) lwrite((kt + 1) & 1);
    __syncthreads();
  }

  // ---- epilogue: lane owns row m, register quads own 4 consecutive n
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + wm * 64 + i * 32 + lr;
    if (m >= M) continue;
    size_t orow = (size_t)m;
    const float* posrow = nullptr;
    if (e.rows_per_img > 0) {
      const int b = m / e.rows_per_img, p = m - b * e.rows_per_img;
      orow = (size_t)b * e.out_rows_per_img + 1 + p;
      posrow = e.pos + (size_t)(1 + p) * N;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = n0 + wn * 64 + j * 32 + 8 * g + 4 * lh;
        if (n >= N) continue;
        float v[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) v[t] = acc[i][j][4 * g + t];
        if (e.bias) {
          const float4 b4 = *reinterpret_cast<const float4*>(e.bias + n);
          v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
        }
        if (e.act == ACT_GELU) {
#pragma unroll
          for (int t = 0; t < 4; ++t) v[t] = gelu_erf(v[t]);
        } else if (e.act == ACT_RELU) {
#pragma unroll
          for (int t = 0; t < 4; ++t) v[t] = fmaxf(v[t], 0.f);
        }
        if (e.scale) {
          const float4 s4 = *reinterpret_cast<const float4*>(e.scale + n);
          v[0] *= s4.x; v[1] *= s4.y; v[2] *= s4.z; v[3] *= s4.w;
        }
        if (posrow) {
          const float4 p4 = *reinterpret_cast<const float4*>(posrow + n);
          v[0] += p4.x; v[1] += p4.y; v[2] += p4.z; v[3] += p4.w;
        }
        if (e.resid) {
          const float4 r4 = *reinterpret_cast<const float4*>(e.resid + orow * e.ldr + n);
          v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
        }
        if (e.out_f32) {
          *reinterpret_cast<float4*>(e.out_f32 + orow * e.ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
          uint2 o;
          o.x = pack2bf(v[0], v[1]);
          o.y = pack2bf(v[2], v[3]);
          *reinterpret_cast<uint2*>(e.out_bf16 + orow * e.ldc + n) = o;
        }
      }
    }
  }
}

int launch_gemm_bf16(const bf16_t* A, int lda, const bf16_t* W, int ldw, int M, int N, int K,
                     const GemmEpi& e, hipStream_t s) {
  if (M <= 0 || N <= 0 || K <= 0) return 1;
  if (K % BK != 0 || N % 4 != 0 || lda % 8 != 0 || ldw % 8 != 0 || e.ldc % 4 != 0) return 2;
  if (e.resid && e.ldr % 4 != 0) return 2;
  if (!e.out_f32 && !e.out_bf16) return 2;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_kernel),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES);
    attr_set = true;
  }
  const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  hipLaunchKernelGGL(gemm_bf16_kernel, dim3(tiles), dim3(256), 2 * STAGE_BYTES, s, A, lda, W, ldw, M, N, K, e);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}
