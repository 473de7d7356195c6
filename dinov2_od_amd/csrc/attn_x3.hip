// Backbone self-attention for the bf16x3 (parity-gated) mode: flash-style, head_dim 64, every product as a split
// product on the bf16 MFMA cores.  Replaces the same reference op as attn_bf16.hip (Dinov2SelfAttention.forward,
// modeling_dinov2.py:203-234) at fp32-class accuracy:
//   x = hi + lo, hi = bf16(x), lo = bf16(x - hi);   a.b ~= ah.bh + ah.bl + al.bh   (lo.lo, 2^-18 relative, dropped)
//   S^T = K Q^T  as  Kh Qh^T + Kh Ql^T + Kl Qh^T          (24 MFMAs per 64-key tile and 32-query block)
//   O^T += V^T P^T  as  Vh^T Ph^T + Vh^T Pl^T + Vl^T Ph^T  (24 MFMAs), P split in registers after the fp32 softmax
// Input: the QKV linear's epilogue writes the pair layout [hi(q|k|v) | lo(q|k|v)] (row pitch 6*D bf16);
// output: the context in the same pair layout [hi | lo] (row pitch 2*D), the A operand of the split out-proj GEMM.
// Structure as attn_bf16.hip with one 32-query block per wave: workgroup = 4 waves = 128 query rows of one (image, head);
// K/V tiles of 64 keys (four planes: Kh, Kl, Vh, Vl = 32 KiB) stream by LDS-DMA into TWO slots (64 KiB, two workgroups
// per CU): tile t+1 is in flight while tile t is computed; same swizzles, query-on-lane products, tr-reads for V^T.
#include "dod_common.h"

#define X3_WAVES 4
#define X3_KV 64
#define X3_PLANE (X3_KV * 128)        // one 64-key x 64-dim bf16 plane: 8 KiB
#define X3_SLOT (4 * X3_PLANE)        // Kh | Kl | Vh | Vl

__device__ __forceinline__ int x3_kswz(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }
__device__ __forceinline__ int x3_vswz(int row, int chunk) { return chunk ^ (((row >> 1) & 1) << 2); }

// eight transposed 64-bit reads (rows 16*s4 (+8) of one d-block) + their wait, as ONE asm statement: the builtin form makes
// hipcc drain the DMA ring before every read (attn_bf16.hip)
#define X3_TR8(v0, v1, v2, v3, v4, v5, v6, v7, addr)                                                                   \
  asm volatile("ds_read_b64_tr_b16 %0, %8\n\tds_read_b64_tr_b16 %1, %8 offset:1024\n\t"                                \
               "ds_read_b64_tr_b16 %2, %8 offset:2048\n\tds_read_b64_tr_b16 %3, %8 offset:3072\n\t"                    \
               "ds_read_b64_tr_b16 %4, %8 offset:4096\n\tds_read_b64_tr_b16 %5, %8 offset:5120\n\t"                    \
               "ds_read_b64_tr_b16 %6, %8 offset:6144\n\tds_read_b64_tr_b16 %7, %8 offset:7168\n\ts_waitcnt lgkmcnt(0)" \
               : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7)                \
               : "v"(addr)                                                                                             \
               : "memory")

template <bool TAIL>
__device__ __forceinline__ void x3_tile(const char* st, const bf16x8 (&qh)[4], const bf16x8 (&ql)[4], f32x16 (&o)[2],
                                        float& m_run, float& l_run, float c, int kbase, int N, int lr, int lh, int g16,
                                        int tq, int tp) {
  const char* sKh = st;
  const char* sKl = st + X3_PLANE;
  const char* sVh = st + 2 * X3_PLANE;
  const char* sVl = st + 3 * X3_PLANE;
  f32x16 s[2];
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) {
    const int row = kb * 32 + lr;
    s[kb] = zero;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int off = row * 128 + x3_kswz(row, 2 * t + lh) * 16;
      const bf16x8 kh = *reinterpret_cast<const bf16x8*>(sKh + off);
      const bf16x8 kl = *reinterpret_cast<const bf16x8*>(sKl + off);
      s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl, qh[t], s[kb], 0, 0, 0);    // small terms first
      s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, ql[t], s[kb], 0, 0, 0);
      s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qh[t], s[kb], 0, 0, 0);
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
  {
    const unsigned mb = __float_as_uint(mx);
    const auto sw = __builtin_amdgcn_permlane32_swap(mb, mb, false, false);
    mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
  }
  const float m_new = fmaxf(m_run, mx * c);
  const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
  m_run = m_new;
  const float nm = -m_new;
  // (round 3: the same loop on explicit packed instructions -- v_pk_fma_f32 for the exponent, inline-asm v_pk_add_f32 for the row sum and for
  //  the residual of the bf16 split -- has 35 fewer VALU instructions per tile, 208 instead of 243, and is 4.5 % SLOWER, 15.2 vs 14.5 ms per
  //  step: hipcc leaves packed fp32 operations packed only where no MFMA is in flight and splits them elsewhere, and with the matrix pipe 50 %
  //  busy that choice is the right one here.  attn_bf16.hip, matrix pipe 35 % busy, gains 2.5 % from the packed row sum.)
  float lsum = 0.f;
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s[kb][r] = __builtin_amdgcn_exp2f(fmaf(s[kb][r], c, nm));
      lsum += s[kb][r];
    }
  l_run = fmaf(l_run, alpha, lsum);
#pragma unroll
  for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }      // per element: a vector multiply becomes v_pk_mul_f32, which does not overlap the MFMAs (attn_bf16.hip)
  // P^T fragments, split: accumulator registers 8u..8u+7 of key block kb are the B operand of k-step 2kb+u
  bf16x8 ph[4], pl[4];
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4) {
    const f32x16& pp = s[s4 >> 1];
    const int u = (s4 & 1) * 8;
    uint4 hi, lo;
    hi.x = pack2bf(pp[u + 0], pp[u + 1]);
    hi.y = pack2bf(pp[u + 2], pp[u + 3]);
    hi.z = pack2bf(pp[u + 4], pp[u + 5]);
    hi.w = pack2bf(pp[u + 6], pp[u + 7]);
    lo.x = pack2bf(pp[u + 0] - __uint_as_float(hi.x << 16), pp[u + 1] - __uint_as_float(hi.x & 0xffff0000u));
    lo.y = pack2bf(pp[u + 2] - __uint_as_float(hi.y << 16), pp[u + 3] - __uint_as_float(hi.y & 0xffff0000u));
    lo.z = pack2bf(pp[u + 4] - __uint_as_float(hi.z << 16), pp[u + 5] - __uint_as_float(hi.z & 0xffff0000u));
    lo.w = pack2bf(pp[u + 6] - __uint_as_float(hi.w << 16), pp[u + 7] - __uint_as_float(hi.w & 0xffff0000u));
    ph[s4] = __builtin_bit_cast(bf16x8, hi);
    pl[s4] = __builtin_bit_cast(bf16x8, lo);
  }
  {
    const int rowb = 4 * lh + tq;
    const int col0 = 16 * (g16 & 1) + 4 * tp;
    const int chunk0 = col0 >> 3, inb = (col0 & 7) * 2;
    const unsigned a0 = rowb * 128 + x3_vswz(rowb, chunk0) * 16 + inb;        // d-block 0
    const unsigned a1 = rowb * 128 + x3_vswz(rowb, chunk0 + 4) * 16 + inb;    // d-block 1
    const unsigned vh_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)sVh;
    const unsigned vl_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)sVl;
#pragma unroll
    for (int db = 0; db < 2; ++db) {
      bf16x4 v0, v1, v2, v3, v4, v5, v6, v7;
      X3_TR8(v0, v1, v2, v3, v4, v5, v6, v7, vl_base + (db ? a1 : a0));
      {
        const bf16x8 f0 = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7), f1 = __builtin_shufflevector(v2, v3, 0, 1, 2, 3, 4, 5, 6, 7);
        const bf16x8 f2 = __builtin_shufflevector(v4, v5, 0, 1, 2, 3, 4, 5, 6, 7), f3 = __builtin_shufflevector(v6, v7, 0, 1, 2, 3, 4, 5, 6, 7);
        o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f0, ph[0], o[db], 0, 0, 0);
        o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f1, ph[1], o[db], 0, 0, 0);
        o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f2, ph[2], o[db], 0, 0, 0);
        o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f3, ph[3], o[db], 0, 0, 0);
      }
      X3_TR8(v0, v1, v2, v3, v4, v5, v6, v7, vh_base + (db ? a1 : a0));
      {
        const bf16x8 f0 = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7), f1 = __builtin_shufflevector(v2, v3, 0, 1, 2, 3, 4, 5, 6, 7);
        const bf16x8 f2 = __builtin_shufflevector(v4, v5, 0, 1, 2, 3, 4, 5, 6, 7), f3 = __builtin_shufflevector(v6, v7, 0, 1, 2, 3, 4, 5, 6, 7);
        o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f0, pl[0], o[db], 0, 0, 0);
        o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f1, pl[1], o[db], 0, 0, 0);
        o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f2, pl[2], o[db], 0, 0, 0);
        o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f3, pl[3], o[db], 0, 0, 0);
        o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f0, ph[0], o[db], 0, 0, 0);
        o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f1, ph[1], o[db], 0, 0, 0);
        o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f2, ph[2], o[db], 0, 0, 0);
        o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f3, ph[3], o[db], 0, 0, 0);
      }
    }
  }
}

__global__ __launch_bounds__(256, 2) void attn_x3_kernel(const bf16_t* __restrict__ qkv2, bf16_t* __restrict__ ctx3, int N,
                                                         int heads, int npairs, float scale_log2e, int ctx_h2) {
  __shared__ __attribute__((aligned(16))) char smem[2 * X3_SLOT];   // 64 KiB
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int D = heads * 64, ld = 6 * D, lo_off = 3 * D;
  const int nqb = (N + X3_WAVES * 32 - 1) / (X3_WAVES * 32);
  int b, h, qb;
  {   // XCD-aware order: the q-blocks of one (image, head) run on one XCD (attn_bf16.hip)
    const int L = blockIdx.x, xcd = L & 7, s = L >> 3;
    const int pair = (s / nqb) * 8 + xcd;
    qb = s - (s / nqb) * nqb;
    if (pair >= npairs) return;
    b = pair / heads;
    h = pair - b * heads;
  }
  const int q0 = qb * (X3_WAVES * 32) + wid * 32;
  const bf16_t* base = qkv2 + (size_t)b * N * ld;
  bf16x8 qh[4], ql[4];
  {
    int qr = q0 + lr; qr = qr < N ? qr : N - 1;
    const bf16_t* qp = base + (size_t)qr * ld + h * 64 + lh * 8;
#pragma unroll
    for (int t = 0; t < 4; ++t) { qh[t] = *reinterpret_cast<const bf16x8*>(qp + 16 * t); ql[t] = *reinterpret_cast<const bf16x8*>(qp + lo_off + 16 * t); }
  }
  // retire the Q loads before any LDS-DMA is in flight (see attn_bf16.hip)
#pragma unroll
  for (int t = 0; t < 4; ++t) asm volatile("" : "+v"(qh[t]), "+v"(ql[t]));

  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const int wu = __builtin_amdgcn_readfirstlane(wid);
  const int prow0 = wu * 16 + (lane >> 3), prow1 = prow0 + 8;
  const int kc0 = ((lane & 7) ^ ((prow0 >> 1) & 7)) * 8, kc1 = ((lane & 7) ^ ((prow1 >> 1) & 7)) * 8;
  const int vc0 = ((lane & 7) ^ (((prow0 >> 1) & 1) << 2)) * 8, vc1 = ((lane & 7) ^ (((prow1 >> 1) & 1) << 2)) * 8;
  const bf16_t* kbase = base + D + h * 64;
#define X3_STAGE(slot_, kt_)                                                                                   \
  {                                                                                                            \
    int key0 = (kt_) * X3_KV + prow0, key1 = (kt_) * X3_KV + prow1;                                            \
    key0 = key0 < N ? key0 : N - 1; key1 = key1 < N ? key1 : N - 1;                                            \
    const bf16_t* r0 = kbase + (size_t)key0 * ld;                                                              \
    const bf16_t* r1 = kbase + (size_t)key1 * ld;                                                              \
    char* s_ = smem + (slot_) * X3_SLOT + wu * 2048;                                                           \
    __builtin_amdgcn_global_load_lds((gptr_t)(r0 + kc0), (lptr_t)(s_), 16, 0, 0);                              \
    __builtin_amdgcn_global_load_lds((gptr_t)(r1 + kc1), (lptr_t)(s_ + 1024), 16, 0, 0);                       \
    __builtin_amdgcn_global_load_lds((gptr_t)(r0 + lo_off + kc0), (lptr_t)(s_ + X3_PLANE), 16, 0, 0);          \
    __builtin_amdgcn_global_load_lds((gptr_t)(r1 + lo_off + kc1), (lptr_t)(s_ + X3_PLANE + 1024), 16, 0, 0);   \
    __builtin_amdgcn_global_load_lds((gptr_t)(r0 + D + vc0), (lptr_t)(s_ + 2 * X3_PLANE), 16, 0, 0);           \
    __builtin_amdgcn_global_load_lds((gptr_t)(r1 + D + vc1), (lptr_t)(s_ + 2 * X3_PLANE + 1024), 16, 0, 0);    \
    __builtin_amdgcn_global_load_lds((gptr_t)(r0 + lo_off + D + vc0), (lptr_t)(s_ + 3 * X3_PLANE), 16, 0, 0);  \
    __builtin_amdgcn_global_load_lds((gptr_t)(r1 + lo_off + D + vc1), (lptr_t)(s_ + 3 * X3_PLANE + 1024), 16, 0, 0); \
  }
  f32x16 o[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; }
  float m_run = -INFINITY, l_run = 0.f;
  const bool active = __builtin_amdgcn_readfirstlane(q0) < N;
  const int g16 = lane >> 4, i16 = lane & 15, tq = i16 >> 2, tp = i16 & 3;
  const int nkt = (N + X3_KV - 1) / X3_KV, nfull = N / X3_KV;
  X3_STAGE(0, 0)
  for (int kt = 0; kt < nkt; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // tile kt landed (the only one in flight)
    __builtin_amdgcn_s_barrier();                          // ... for every wave, and every wave is done with tile kt-1
    asm volatile("" ::: "memory");
    if (kt + 1 < nkt) X3_STAGE((kt + 1) & 1, kt + 1)       // streams in under this tile's 48 MFMAs
    const char* st = smem + (kt & 1) * X3_SLOT;
    if (active) {
      if (kt < nfull) x3_tile<false>(st, qh, ql, o, m_run, l_run, scale_log2e, kt * X3_KV, N, lr, lh, g16, tq, tp);
      else x3_tile<true>(st, qh, ql, o, m_run, l_run, scale_log2e, kt * X3_KV, N, lr, lh, g16, tq, tp);
    }
  }
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  const int q = q0 + lr;
  if (active && q < N && ctx_h2) {       // H2 operand rows for the out-proj of the fp16x2 mode (dod_common.h)
    char* row = reinterpret_cast<char*>(ctx3) + ((size_t)b * N + q) * 4 * D;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 v = make_float4(o[db][4 * g] * inv, o[db][4 * g + 1] * inv, o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv);
        uint2 f16; unsigned hi8, lo8;
        h2_quad(v, 1.0f, f16, hi8, lo8);
        const int c = h * 64 + db * 32 + 8 * g + 4 * lh;
        *reinterpret_cast<uint2*>(row + 2 * c) = f16;
        char* p8 = row + h2_off8(D, c);
        *reinterpret_cast<unsigned*>(p8) = hi8;
        *reinterpret_cast<unsigned*>(p8 + 16) = lo8;
      }
  } else if (active && q < N) {
    bf16_t* op = ctx3 + ((size_t)b * N + q) * 2 * D + h * 64;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float v0 = o[db][4 * g] * inv, v1 = o[db][4 * g + 1] * inv, v2 = o[db][4 * g + 2] * inv, v3 = o[db][4 * g + 3] * inv;
        uint2 hi, lo;
        hi.x = pack2bf(v0, v1);
        hi.y = pack2bf(v2, v3);
        lo.x = pack2bf(v0 - __uint_as_float(hi.x << 16), v1 - __uint_as_float(hi.x & 0xffff0000u));
        lo.y = pack2bf(v2 - __uint_as_float(hi.y << 16), v3 - __uint_as_float(hi.y & 0xffff0000u));
        bf16_t* p = op + db * 32 + 8 * g + 4 * lh;
        *reinterpret_cast<uint2*>(p) = hi;
        *reinterpret_cast<uint2*>(p + D) = lo;
      }
  }
}

// qkv2 [B*N, 6*D] bf16 = [hi(q|k|v) | lo(q|k|v)]  ->  ctx3 [B*N, 2*D] bf16 = [hi | lo]
int launch_attn_x3(const bf16_t* qkv2, bf16_t* ctx3, int B, int N, int heads, float scale, hipStream_t s, int ctx_h2) {
  if (B <= 0 || N <= 0 || heads <= 0) return 1;
  const int npairs = B * heads, pairs8 = (npairs + 7) / 8 * 8;
  const int nqb = (N + X3_WAVES * 32 - 1) / (X3_WAVES * 32);
  hipLaunchKernelGGL(attn_x3_kernel, dim3(pairs8 * nqb), dim3(256), 0, s, qkv2, ctx3, N, heads, npairs,
                     scale * 1.44269504088896340736f, ctx_h2);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}
