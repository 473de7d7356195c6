// Backbone self-attention (K5), flash-style, bf16 MFMA, head_dim 64 (every DINOv2 variant).
// Replaces Dinov2SelfAttention.forward's softmax(Q K^T * dh^-0.5) V
// (site-packages/transformers/models/dinov2/modeling_dinov2.py:153-178, 215-229): no mask, non-causal.
//
// Layout: qkv is the fused-QKV GEMM output [B*N, 3*D] bf16 (q | k | v column blocks, head h at
// columns h*64..h*64+63 of each block); ctx is [B*N, D] bf16.
// One workgroup = 4 waves = 128 query rows of one (batch, head); each wave owns 32 query rows.
// K/V tiles of 64 keys are staged global -> VGPR -> LDS (double-buffered, next tile's loads issued
// before the MFMA phase, written after it; one barrier per tile).
// "Swapped" products keep the query on the MFMA lane so the softmax row state (m, l) is per-lane:
//   S^T = K Q^T   (A = K tile rows from LDS via ds_read_b128, B = Q fragments held in registers)
//   O^T += V^T P^T (A = V^T via ds_read_b64_tr_b16 from the row-major V tile, B = the S^T accumulator
//                  registers themselves, converted to bf16 - no LDS round trip for P)
// N = 1370 is not a multiple of 64: the last tile's keys >= N get -inf before the row max; loads of
// rows >= N are clamped to row N-1 (finite values times p = 0).
#include "dod_common.h"

#define AT_WAVES 4
#define AT_QW 32
#define AT_KV 64

__device__ __forceinline__ int kswz(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }
// V tile: rows r and r+2 of a 4-row tr-read block are 256 B apart (same banks): flip the 64-B half on row bit 1
__device__ __forceinline__ int vswz(int row, int chunk) { return chunk ^ (((row >> 1) & 1) << 2); }

typedef bf16x4 __attribute__((address_space(3))) * lds_bf16x4_ptr;

// One K/V tile (64 keys) for one wave (32 query rows on the lanes).  TAIL masks keys >= N (last tile only,
// so the full tiles carry no compare/select).  Softmax in the exp2 domain with the scale folded into one FMA:
//   p = exp2(s*c - m*c), m = running max of the raw scores (c > 0).
template <bool TAIL>
__device__ __forceinline__ void attn_tile(const char* sK, const char* sV, const bf16x8 (&qf)[4], f32x16 (&o)[2],
                                          float& m_run, float& l_run, float c, int kbase, int N,
                                          int lr, int lh, int g16, int tq, int tp) {
  f32x16 s[2];
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) {
    const int row = kb * 32 + lr;
    const bf16x8 kf0 = *reinterpret_cast<const bf16x8*>(sK + row * 128 + kswz(row, lh) * 16);
    s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf0, qf[0], zero, 0, 0, 0);   // C = inline 0: no zeroing moves
#pragma unroll
    for (int t = 1; t < 4; ++t) {
      const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sK + row * 128 + kswz(row, 2 * t + lh) * 16);
      s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[t], s[kb], 0, 0, 0);
    }
  }
  if (TAIL) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kbase + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (key >= N) s[kb][r] = -INFINITY;
      }
  }
  float mx = s[0][0];
#pragma unroll
  for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[0][r]);
#pragma unroll
  for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[1][r]);
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  const float m_new = fmaxf(m_run, mx * c);            // scaled (exp2-domain) running max
  const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
  m_run = m_new;
  const float nm = -m_new;
  f32x16 p0, p1;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    p0[r] = __builtin_amdgcn_exp2f(fmaf(s[0][r], c, nm));
    p1[r] = __builtin_amdgcn_exp2f(fmaf(s[1][r], c, nm));
  }
  const f32x16 ps = p0 + p1;
  const float lsum = ((ps[0] + ps[1]) + (ps[2] + ps[3])) + ((ps[4] + ps[5]) + (ps[6] + ps[7])) +
                     (((ps[8] + ps[9]) + (ps[10] + ps[11])) + ((ps[12] + ps[13]) + (ps[14] + ps[15])));
  l_run = fmaf(l_run, alpha, lsum);   // per-half partial; the halves are combined once at the end
  o[0] *= alpha;
  o[1] *= alpha;
  // P^T fragments: accumulator registers 8u..8u+7 of key block kb are the B operand of k-step 2kb+u
  bf16x8 pf[4];
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4) {
    const f32x16& pp = (s4 >> 1) ? p1 : p0;
    const int u = (s4 & 1) * 8;
    uint4 pk;
    pk.x = pack2bf(pp[u + 0], pp[u + 1]);
    pk.y = pack2bf(pp[u + 2], pp[u + 3]);
    pk.z = pack2bf(pp[u + 4], pp[u + 5]);
    pk.w = pack2bf(pp[u + 6], pp[u + 7]);
    pf[s4] = __builtin_bit_cast(bf16x8, pk);
  }
#pragma unroll
  for (int db = 0; db < 2; ++db) {
    const int col = db * 32 + 16 * (g16 & 1) + 4 * tp;     // first of this lane's 4 source columns
    const int chunk = col >> 3, inb = (col & 7) * 2;
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const int r1 = 16 * s4 + 4 * lh + tq, r2 = r1 + 8;
      const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(sV + r1 * 128 + vswz(r1, chunk) * 16 + inb));
      const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(sV + r2 * 128 + vswz(r2, chunk) * 16 + inb));
      const bf16x8 vf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[s4], o[db], 0, 0, 0);
    }
  }
}

__global__ __launch_bounds__(256, 2) void attn_bf16_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ ctx,
                                                        int N, int heads, float scale_log2e) {
  __shared__ __attribute__((aligned(16))) char smem[2 * 2 * AT_KV * 128];   // [stage][K|V][64 rows][128 B]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int D = heads * 64, ld = 3 * D;
  const int b = blockIdx.z, h = blockIdx.y;
  const int q0 = blockIdx.x * (AT_WAVES * AT_QW) + wid * AT_QW;
  const bf16_t* base = qkv + (size_t)b * N * ld;

  // Q fragments: B operand of S^T = K Q^T: lane holds Q[q = lr][d = 16 t + 8 lh + 0..7]
  bf16x8 qf[4];
  {
    int qr = q0 + lr; qr = qr < N ? qr : N - 1;
    const bf16_t* qp = base + (size_t)qr * ld + h * 64 + lh * 8;
#pragma unroll
    for (int t = 0; t < 4; ++t) qf[t] = *reinterpret_cast<const bf16x8*>(qp + 16 * t);
  }

  // staging assignment: 2 chunks of K and 2 of V per thread per tile (rows srow and srow + 32)
  const int srow = tid >> 3, sc = tid & 7;
  const bf16_t* kvbase = base + D + h * 64 + sc * 8;
  const int kls0 = srow * 128 + kswz(srow, sc) * 16;          // (row + 32) has the same swizzle term
  const int vls0 = srow * 128 + vswz(srow, sc) * 16;
  uint4 rk0, rk1, rv0, rv1;
#define AT_GLOAD(kt_)                                                              \
  {                                                                                \
    int key0 = (kt_) * AT_KV + srow, key1 = key0 + 32;                             \
    key0 = key0 < N ? key0 : N - 1; key1 = key1 < N ? key1 : N - 1;                \
    const bf16_t* p0 = kvbase + (size_t)key0 * ld;                                 \
    const bf16_t* p1 = kvbase + (size_t)key1 * ld;                                 \
    rk0 = *reinterpret_cast<const uint4*>(p0); rv0 = *reinterpret_cast<const uint4*>(p0 + D); \
    rk1 = *reinterpret_cast<const uint4*>(p1); rv1 = *reinterpret_cast<const uint4*>(p1 + D); \
  }
#define AT_LWRITE(st_)                                                             \
  {                                                                                \
    char* sK_ = smem + (st_) * (2 * AT_KV * 128);                                  \
    char* sV_ = sK_ + AT_KV * 128;                                                 \
    *reinterpret_cast<uint4*>(sK_ + kls0) = rk0; *reinterpret_cast<uint4*>(sK_ + kls0 + 4096) = rk1; \
    *reinterpret_cast<uint4*>(sV_ + vls0) = rv0; *reinterpret_cast<uint4*>(sV_ + vls0 + 4096) = rv1; \
  }

  f32x16 o[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; }
  float m_run = -INFINITY, l_run = 0.f;

  // tr-read lane geometry (ds_read_b64_tr_b16: 16-lane groups, lane 4q+p supplies row q, cols 4p..4p+3)
  const int g16 = lane >> 4, i16 = lane & 15, tq = i16 >> 2, tp = i16 & 3;

  const int nkt = (N + AT_KV - 1) / AT_KV;
  AT_GLOAD(0)
  AT_LWRITE(0)
  __syncthreads();
  const int nfull = N / AT_KV;                 // tiles without a masked key
  for (int kt = 0; kt < nkt; ++kt) {
    if (kt + 1 < nkt) AT_GLOAD(kt + 1)
    const char* sK = smem + (kt & 1) * (2 * AT_KV * 128);
    const char* sV = sK + AT_KV * 128;
    if (kt < nfull) attn_tile<false>(sK, sV, qf, o, m_run, l_run, scale_log2e, kt * AT_KV, N, lr, lh, g16, tq, tp);
    else attn_tile<true>(sK, sV, qf, o, m_run, l_run, scale_log2e, kt * AT_KV, N, lr, lh, g16, tq, tp);
    if (kt + 1 < nkt) AT_LWRITE((kt + 1) & 1)
    __syncthreads();
  }

  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  const int q = q0 + lr;
  if (q < N) {
    bf16_t* op = ctx + ((size_t)b * N + q) * D + h * 64;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        uint2 pk;
        pk.x = pack2bf(o[db][4 * g] * inv, o[db][4 * g + 1] * inv);
        pk.y = pack2bf(o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv);
        *reinterpret_cast<uint2*>(op + db * 32 + 8 * g + 4 * lh) = pk;
      }
  }
}

int launch_attn_bf16(const bf16_t* qkv, bf16_t* ctx, int B, int N, int heads, float scale, hipStream_t s) {
  if (B <= 0 || N <= 0 || heads <= 0) return 1;
  const dim3 grid((N + AT_WAVES * AT_QW - 1) / (AT_WAVES * AT_QW), heads, B);
  hipLaunchKernelGGL(attn_bf16_kernel, grid, dim3(256), 0, s, qkv, ctx, N, heads, scale * 1.44269504088896340736f);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}
