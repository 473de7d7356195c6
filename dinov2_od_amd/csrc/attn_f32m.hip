// fp32 flash attention on the fp32 MFMA (v_mfma_f32_32x32x2_f32: a k-ordered fmaf chain, exact fp32 products), head_dim DH in
// {32, 64, 96}: the strict-mode backbone attention (K5, modeling_dinov2.py:203-234; DH = 64) and -- round 3 -- the decoder's query
// self-attention K11 and the dense cross-attention K20 over all memory tokens (nn.MultiheadAttention / nn.TransformerDecoder,
// deformable_attention.py:195,233; detr_decoder.py:28-35,62-69; DH = 96 for the default 768 / 8 decoder): every dense QK^T / AV of
// the path is on the matrix cores.  attn_f32.hip's VALU kernel stays for the remaining head sizes (micro test models).
// Workgroup = 4 waves = 128 query rows of one (image, head); K/V tiles of 64 keys in LDS (K transposed to [d][key] so the one-
// float-per-lane MFMA operands are conflict-free ds_read_b32; V row-major [key][d]), query on the MFMA lane as in attn_bf16.hip:
//   S^T[key][q] = sum_d K[key][d] Q[q][d]      32 k-steps of 2 per 32-key block, Q held in registers (one float per k-step)
//   O^T[d][q]  += sum_key V[key][d] P[key][q]  the S^T accumulator registers ARE the B operand: k-step t pairs the keys
//                                              (t&3) + 8(t>>2) (lanes 0..31) and that + 4 (lanes 32..63), i.e. register t of
//                                              each half -- the contraction order over keys is free, so no shuffle is needed
// 128 MFMAs x 64 cycles per tile and wave: MFMA-bound (fp32 MFMA peak 157 TFLOP/s); softmax statistics per lane (= per query).
#include "dod_common.h"

#define FM_KV 64
#define FM_LDK 97     // K^T rows [d][key]: odd pitch -> conflict-free transposing ds_write_b32; the two lane halves (d, d+1) land 33 banks apart

template <int DH>
__global__ __launch_bounds__(256, 2) void attn_f32m_kernel(AttnF32 a) {
  constexpr int FM_LDV = DH + 8;         // V rows [key][d]: the two lane halves read keys 4 apart: 4 * (DH + 8) = 32 (mod 64) banks apart
  constexpr int NDB = DH / 32;           // 32-wide d blocks of the output
  constexpr int DPT = DH / 4;            // d's staged per thread (four threads per key)
  __shared__ float sK[DH * FM_LDK];      // [d][key]
  __shared__ float sV[FM_KV * FM_LDV];   // [key][d]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y;
  const int q0 = blockIdx.x * 128 + wid * 32;
  const float* Q = a.q + (size_t)b * a.Lq * a.ldq + h * DH;
  const float* K = a.k + (size_t)b * a.Lk * a.ldk + h * DH;
  const float* V = a.v + (size_t)b * a.Lk * a.ldv + h * DH;
  // Q operand of k-step s (d = 2s + lh) for query lr: DH / 2 registers
  float qf[DH / 2];
  {
    int qr = q0 + lr; qr = qr < a.Lq ? qr : a.Lq - 1;
    const float* qp = Q + (size_t)qr * a.ldq + lh;
#pragma unroll
    for (int s = 0; s < DH / 2; ++s) qf[s] = qp[2 * s];
  }
  f32x16 o[NDB];
#pragma unroll
  for (int db = 0; db < NDB; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const float c = a.scale * 1.44269504088896340736f;
  const bool active = q0 < a.Lq;
  const int nkt = (a.Lk + FM_KV - 1) / FM_KV;
  for (int kt = 0; kt < nkt; ++kt) {
    __syncthreads();
    // stage the tile: thread -> (key = tid / 4, DH / 4 d's): K transposed, V row-major
    {
      const int key = tid >> 2, dq = (tid & 3) * DPT;
      int kr = kt * FM_KV + key; kr = kr < a.Lk ? kr : a.Lk - 1;
      const float* kp = K + (size_t)kr * a.ldk + dq;
      const float* vp = V + (size_t)kr * a.ldv + dq;
#pragma unroll
      for (int v4 = 0; v4 < DPT / 4; ++v4) {
        const float4 kv = *reinterpret_cast<const float4*>(kp + 4 * v4);
        const float4 vv = *reinterpret_cast<const float4*>(vp + 4 * v4);
        const int d = dq + 4 * v4;
        sK[(d + 0) * FM_LDK + key] = kv.x; sK[(d + 1) * FM_LDK + key] = kv.y;
        sK[(d + 2) * FM_LDK + key] = kv.z; sK[(d + 3) * FM_LDK + key] = kv.w;
        *reinterpret_cast<float4*>(sV + key * FM_LDV + d) = vv;
      }
    }
    __syncthreads();
    if (!active) continue;
    f32x16 s[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
#pragma unroll
      for (int st0 = 0; st0 < DH / 2; st0 += 16) {       // groups of 16 k-steps: a scheduling fence between them keeps the compiler from
#pragma unroll                                            // hoisting all DH / 2 LDS reads of a key block (DH = 96: 48 registers, spills)
        for (int st = st0; st < st0 + 16; ++st) {
          const float kv = sK[(2 * st + lh) * FM_LDK + kb * 32 + lr];     // A: K[key = lr][d = 2 st + lh]
          s[kb] = __builtin_amdgcn_mfma_f32_32x32x2f32(kv, qf[st], s[kb], 0, 0, 0);
        }
        if (DH > 64) __builtin_amdgcn_sched_barrier(0);
      }
    }
    const int kbase = kt * FM_KV;
    if (kbase + FM_KV > a.Lk) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kbase + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (key >= a.Lk) s[kb][r] = -INFINITY;
        }
    }
    float mx = s[0][0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[0][r]);
#pragma unroll
    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[1][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx * c);
    const float alpha = exp2f(m_run - m_new);
    m_run = m_new;
    float lsum = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s[kb][r] = exp2f(fmaf(s[kb][r], c, -m_new));
        lsum += s[kb][r];
      }
    l_run = fmaf(l_run, alpha, lsum);
#pragma unroll
    for (int db = 0; db < NDB; ++db) o[db] *= alpha;
    // O^T += V^T P^T: k-step t of key block kb uses keys kb*32 + (t&3) + 8(t>>2) + 4 lh
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int key = kb * 32 + (t & 3) + 8 * (t >> 2) + 4 * lh;
#pragma unroll
        for (int db = 0; db < NDB; ++db)                  // A: V^T[d = 32 db + lr][key]
          o[db] = __builtin_amdgcn_mfma_f32_32x32x2f32(sV[key * FM_LDV + 32 * db + lr], s[kb][t], o[db], 0, 0, 0);
      }
  }
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  const int q = q0 + lr;
  // (max, sum) of the row in the log2 domain, kept SEPARATE: folded into one log-sum-exp float (magnitude ~20, ulp 2e-6) the adjoint's
  // recomputed probabilities carry a 1e-6 relative error per row -- measured as 3-5x the gradient error of the batched-GEMM adjoint
  if (a.lse && active && q < a.Lq && lh == 0)
    *reinterpret_cast<float2*>(a.lse + (((size_t)b * a.heads + h) * a.Lq + q) * 2) = make_float2(m_run, l_tot);
  if (active && q < a.Lq) {
    float* op = a.o + ((size_t)b * a.Lq + q) * a.ldo + h * DH;
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(op + db * 32 + 8 * g + 4 * lh) =
            make_float4(o[db][4 * g] * inv, o[db][4 * g + 1] * inv, o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv);
    if (a.o3) {          // the same values as the bf16x3 operand [hi | hi | lo] of the next query-side linear
      const int Dm = a.heads * DH;
      bf16_t* o3 = a.o3 + ((size_t)b * a.Lq + q) * 3 * Dm + h * DH;
#pragma unroll
      for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          // opaque to the optimiser: split the ROUNDED products the fp32 output holds (hipcc would fold the multiply into the residual's
          // subtraction as an fma and split a different number than split3 of the stored output does)
          float y0 = o[db][4 * g] * inv, y1 = o[db][4 * g + 1] * inv, y2 = o[db][4 * g + 2] * inv, y3 = o[db][4 * g + 3] * inv;
          asm volatile("" : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3));
          uint2 hi, lo;
          hi.x = pack2bf(y0, y1); hi.y = pack2bf(y2, y3);
          lo.x = pack2bf(y0 - __uint_as_float(hi.x << 16), y1 - __uint_as_float(hi.x & 0xffff0000u));
          lo.y = pack2bf(y2 - __uint_as_float(hi.y << 16), y3 - __uint_as_float(hi.y & 0xffff0000u));
          bf16_t* d = o3 + db * 32 + 8 * g + 4 * lh;
          *reinterpret_cast<uint2*>(d) = hi;
          *reinterpret_cast<uint2*>(d + Dm) = hi;
          *reinterpret_cast<uint2*>(d + 2 * Dm) = lo;
        }
    }
  }
}

// dh in {32, 64, 96}, 16-byte aligned rows; otherwise the caller falls back to the generic VALU kernel
int launch_attn_f32_mfma(const AttnF32& a, hipStream_t s) {
  if ((a.dh != 32 && a.dh != 64 && a.dh != 96) || a.B <= 0 || a.heads <= 0 || a.Lq <= 0 || a.Lk <= 0) return 2;
  if ((a.ldq | a.ldk | a.ldv | a.ldo) % 4) return 2;
  const dim3 grid((a.Lq + 127) / 128, a.heads, a.B);
  if (a.dh == 32) hipLaunchKernelGGL(attn_f32m_kernel<32>, grid, dim3(256), 0, s, a);
  else if (a.dh == 64) hipLaunchKernelGGL(attn_f32m_kernel<64>, grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL(attn_f32m_kernel<96>, grid, dim3(256), 0, s, a);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}


// =============================================================================================================================
// Flash-style adjoint (head_dim 64): the training step's backbone-tail attention backward (dec_train.hip) without the [B*H, N, N]
// score / probability / adjoint buffers of the batched-GEMM form (~6 GB of traffic per ViT-B block at 8 x 1 370 tokens).  Scores are
// recomputed tile by tile from q, k and the forward's log-sum-exp; every product on the exact-fp32 MFMA:
//   P = 2^(s c - m_q) / l_q  (the forward's row max and sum),  dP = dO V^T,  dS = P o (dP - delta_q),  delta_q = <dO_q, O_q>
//   dV = P^T dO,  dK = scale dS^T Q                 (attn_f32m_bwd_kv: a wave owns 32 KEYS on the MFMA lane, loops over query tiles)
//   dQ = scale dS K                                 (attn_f32m_bwd_q : a wave owns 32 QUERIES on the MFMA lane, loops over key tiles)
// Operand layouts follow the forward: the tile being looped over sits TRANSPOSED in LDS ([d][row], pitch 97) -- read with consecutive
// lanes along `row` it is the A operand of the score-like products (S, dP), read with lane = d at a fixed row it is the A operand of
// the products that contract over the tile's rows (dV, dK, dQ), whose B operand is the score-layout accumulator itself (register t of
// a 32-row block <-> row (t & 3) + 8 (t >> 2) + 4 (lane >> 5)): no shuffle, no second copy of the tile.
#define FB_T 64            // rows of the looped-over tile
__global__ __launch_bounds__(256) void attn_f32m_delta_kernel(AttnF32Bwd a) {
  // delta[b][h][q] = sum_d dO[q][h*64 + d] O[q][h*64 + d]: 16 lanes per (row, head), float4 each
  const long item = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
  const int sub = threadIdx.x & 15;
  const long total = (long)a.B * a.Lq * a.heads;
  float acc = 0.f;
  if (item < total) {
    const int h = (int)(item % a.heads);
    const long row = item / a.heads;
    const float4 x = *reinterpret_cast<const float4*>(a.d_o + (size_t)row * a.ldo + h * 64 + 4 * sub);
    const float4 y = *reinterpret_cast<const float4*>(a.o + (size_t)row * a.ldo + h * 64 + 4 * sub);
    acc = (x.x * y.x + x.y * y.y) + (x.z * y.z + x.w * y.w);
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (item < total && sub == 0) {
    const int h = (int)(item % a.heads);
    const long row = item / a.heads;
    const int b = (int)(row / a.Lq), q = (int)(row - (long)b * a.Lq);
    a.delta[((size_t)b * a.heads + h) * a.Lq + q] = acc;
  }
}

// stage a 64-row x 64-col tile of `src` (row pitch ld, rows clamped to [0, nrows), zero beyond) TRANSPOSED into dst[d][row]
__device__ __forceinline__ void fb_stage_t(float* dst, const float* src, int ld, int row0, int nrows, int tid, bool zero_tail) {
  const int r = tid >> 2, dq = (tid & 3) * 16;
  const int row = row0 + r;
  const bool ok = row < nrows;
  const float* p = src + (size_t)(ok ? row : nrows - 1) * ld + dq;
#pragma unroll
  for (int v4 = 0; v4 < 4; ++v4) {
    float4 x = *reinterpret_cast<const float4*>(p + 4 * v4);
    if (zero_tail && !ok) x = make_float4(0.f, 0.f, 0.f, 0.f);
    const int d = dq + 4 * v4;
    dst[(d + 0) * FM_LDK + r] = x.x; dst[(d + 1) * FM_LDK + r] = x.y; dst[(d + 2) * FM_LDK + r] = x.z; dst[(d + 3) * FM_LDK + r] = x.w;
  }
}

__global__ __launch_bounds__(256, 2) void attn_f32m_bwd_kv_kernel(AttnF32Bwd a) {
  __shared__ float sQt[64 * FM_LDK];      // [d][query]
  __shared__ float sOt[64 * FM_LDK];      // dO, [d][query]
  __shared__ float sL[FB_T], sI[FB_T], sD[FB_T];        // row max, 1 / row sum, delta
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y;
  const int k0 = blockIdx.x * 128 + wid * 32;
  const float* Q = a.q + (size_t)b * a.Lq * a.ldq + h * 64;
  const float* K = a.k + (size_t)b * a.Lk * a.ldk + h * 64;
  const float* V = a.v + (size_t)b * a.Lk * a.ldv + h * 64;
  const float* DO = a.d_o + (size_t)b * a.Lq * a.ldo + h * 64;
  const float* LSE = a.lse + ((size_t)b * a.heads + h) * a.Lq * 2;
  const float* DEL = a.delta + ((size_t)b * a.heads + h) * a.Lq;
  float kf[32], vf[32];                   // B operands of k-step st (d = 2 st + lh) for key lr
  {
    int kr = k0 + lr; kr = kr < a.Lk ? kr : a.Lk - 1;
    const float* kp = K + (size_t)kr * a.ldk + lh;
    const float* vp = V + (size_t)kr * a.ldv + lh;
#pragma unroll
    for (int st = 0; st < 32; ++st) { kf[st] = kp[2 * st]; vf[st] = vp[2 * st]; }
  }
  f32x16 dk[2], dv[2];
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[db][r] = 0.f; dv[db][r] = 0.f; }
  const float c = a.scale * 1.44269504088896340736f;
  const bool active = k0 < a.Lk;
  const int nqt = (a.Lq + FB_T - 1) / FB_T;
  for (int qt = 0; qt < nqt; ++qt) {
    __syncthreads();
    fb_stage_t(sQt, Q, a.ldq, qt * FB_T, a.Lq, tid, false);
    fb_stage_t(sOt, DO, a.ldo, qt * FB_T, a.Lq, tid, true);         // rows beyond Lq: dO = 0 (and L = +inf below: P = 0)
    if (tid < FB_T) {
      const int q = qt * FB_T + tid;
      sL[tid] = q < a.Lq ? LSE[2 * q] : INFINITY;         // rows beyond Lq: 2^(-inf) = 0
      sI[tid] = q < a.Lq ? 1.0f / LSE[2 * q + 1] : 0.f;
      sD[tid] = q < a.Lq ? DEL[q] : 0.f;
    }
    __syncthreads();
    if (!active) continue;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int st0 = 0; st0 < 32; st0 += 8) { // S[q][key], dP[q][key]: A = Q^T / dO^T tile rows (lane = query), B = this wave's k / v
#pragma unroll                                 // (groups of 8 k-steps behind a scheduling fence: a full unroll hoists 64 LDS reads and spills)
        for (int st = st0; st < st0 + 8; ++st) {
          s = __builtin_amdgcn_mfma_f32_32x32x2f32(sQt[(2 * st + lh) * FM_LDK + qb * 32 + lr], kf[st], s, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_32x32x2f32(sOt[(2 * st + lh) * FM_LDK + qb * 32 + lr], vf[st], dp, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      // register r <-> query qb*32 + (r & 3) + 8 (r >> 2) + 4 lh
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 L4 = *reinterpret_cast<const float4*>(sL + qb * 32 + 8 * g + 4 * lh);
        const float4 I4 = *reinterpret_cast<const float4*>(sI + qb * 32 + 8 * g + 4 * lh);
        const float4 D4 = *reinterpret_cast<const float4*>(sD + qb * 32 + 8 * g + 4 * lh);
        const float Lv[4] = {L4.x, L4.y, L4.z, L4.w}, Iv[4] = {I4.x, I4.y, I4.z, I4.w}, Dv[4] = {D4.x, D4.y, D4.z, D4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float p = exp2f(fmaf(s[4 * g + e], c, -Lv[e])) * Iv[e];
          s[4 * g + e] = p;                                   // P
          dp[4 * g + e] = p * (dp[4 * g + e] - Dv[e]);          // dS
        }
      }
      // dV^T[d][key] += dO^T[d][q] P[q][key];  dK^T[d][key] += Q^T[d][q] dS[q][key]   (k-step t contracts query (t&3) + 8(t>>2) + 4 lh)
#pragma unroll
      for (int t0 = 0; t0 < 16; t0 += 4) {
#pragma unroll
        for (int t = t0; t < t0 + 4; ++t) {
          const int qq = qb * 32 + (t & 3) + 8 * (t >> 2) + 4 * lh;
#pragma unroll
          for (int db = 0; db < 2; ++db) {
            dv[db] = __builtin_amdgcn_mfma_f32_32x32x2f32(sOt[(32 * db + lr) * FM_LDK + qq], s[t], dv[db], 0, 0, 0);
            dk[db] = __builtin_amdgcn_mfma_f32_32x32x2f32(sQt[(32 * db + lr) * FM_LDK + qq], dp[t], dk[db], 0, 0, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  const int key = k0 + lr;
  if (active && key < a.Lk) {
    float* dkp = a.dk + ((size_t)b * a.Lk + key) * a.lddk + h * 64;
    float* dvp = a.dv + ((size_t)b * a.Lk + key) * a.lddv + h * 64;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        *reinterpret_cast<float4*>(dkp + db * 32 + 8 * g + 4 * lh) =
            make_float4(dk[db][4 * g] * a.scale, dk[db][4 * g + 1] * a.scale, dk[db][4 * g + 2] * a.scale, dk[db][4 * g + 3] * a.scale);
        *reinterpret_cast<float4*>(dvp + db * 32 + 8 * g + 4 * lh) = make_float4(dv[db][4 * g], dv[db][4 * g + 1], dv[db][4 * g + 2], dv[db][4 * g + 3]);
      }
  }
}

__global__ __launch_bounds__(256, 2) void attn_f32m_bwd_q_kernel(AttnF32Bwd a) {
  __shared__ float sKt[64 * FM_LDK];      // [d][key]
  __shared__ float sVt[64 * FM_LDK];      // [d][key]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y;
  const int q0 = blockIdx.x * 128 + wid * 32;
  const float* Q = a.q + (size_t)b * a.Lq * a.ldq + h * 64;
  const float* K = a.k + (size_t)b * a.Lk * a.ldk + h * 64;
  const float* V = a.v + (size_t)b * a.Lk * a.ldv + h * 64;
  const float* DO = a.d_o + (size_t)b * a.Lq * a.ldo + h * 64;
  float qf[32], dof[32];                  // B operands of k-step st (d = 2 st + lh) for query lr
  int qr = q0 + lr; qr = qr < a.Lq ? qr : a.Lq - 1;
  {
    const float* qp = Q + (size_t)qr * a.ldq + lh;
    const float* op = DO + (size_t)qr * a.ldo + lh;
#pragma unroll
    for (int st = 0; st < 32; ++st) { qf[st] = qp[2 * st]; dof[st] = op[2 * st]; }
  }
  const float2 ml_ = *reinterpret_cast<const float2*>(a.lse + (((size_t)b * a.heads + h) * a.Lq + qr) * 2);
  const float Lq_ = ml_.x, Iq_ = 1.0f / ml_.y;
  const float Dq_ = a.delta[((size_t)b * a.heads + h) * a.Lq + qr];
  f32x16 dq[2];
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[db][r] = 0.f;
  const float c = a.scale * 1.44269504088896340736f;
  const bool active = q0 < a.Lq;
  const int nkt = (a.Lk + FB_T - 1) / FB_T;
  for (int kt = 0; kt < nkt; ++kt) {
    __syncthreads();
    fb_stage_t(sKt, K, a.ldk, kt * FB_T, a.Lk, tid, false);
    fb_stage_t(sVt, V, a.ldv, kt * FB_T, a.Lk, tid, false);
    __syncthreads();
    if (!active) continue;
    const int kbase = kt * FB_T;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int st0 = 0; st0 < 32; st0 += 8) { // S^T[key][q], dP^T[key][q]: A = K^T / V^T tile rows (lane = key), B = this wave's q / dO
#pragma unroll
        for (int st = st0; st < st0 + 8; ++st) {
          s = __builtin_amdgcn_mfma_f32_32x32x2f32(sKt[(2 * st + lh) * FM_LDK + kb * 32 + lr], qf[st], s, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_32x32x2f32(sVt[(2 * st + lh) * FM_LDK + kb * 32 + lr], dof[st], dp, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kbase + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const float p = key < a.Lk ? exp2f(fmaf(s[r], c, -Lq_)) * Iq_ : 0.f;
        dp[r] = p * (dp[r] - Dq_);                              // dS^T[key][q]
      }
      // dQ^T[d][q] += K^T[d][key] dS^T[key][q]   (k-step t contracts key kb*32 + (t&3) + 8(t>>2) + 4 lh)
#pragma unroll
      for (int t0 = 0; t0 < 16; t0 += 8) {
#pragma unroll
        for (int t = t0; t < t0 + 8; ++t) {
          const int kk = kb * 32 + (t & 3) + 8 * (t >> 2) + 4 * lh;
#pragma unroll
          for (int db = 0; db < 2; ++db)
            dq[db] = __builtin_amdgcn_mfma_f32_32x32x2f32(sKt[(32 * db + lr) * FM_LDK + kk], dp[t], dq[db], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  const int q = q0 + lr;
  if (active && q < a.Lq) {
    float* dqp = a.dq + ((size_t)b * a.Lq + q) * a.lddq + h * 64;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(dqp + db * 32 + 8 * g + 4 * lh) =
            make_float4(dq[db][4 * g] * a.scale, dq[db][4 * g + 1] * a.scale, dq[db][4 * g + 2] * a.scale, dq[db][4 * g + 3] * a.scale);
  }
}

int launch_attn_f32_bwd(const AttnF32Bwd& a, hipStream_t s) {
  if (a.dh != 64 || a.B <= 0 || a.heads <= 0 || a.Lq <= 0 || a.Lk <= 0) return 2;
  if ((a.ldq | a.ldk | a.ldv | a.ldo | a.lddq | a.lddk | a.lddv) % 4) return 2;
  if (!a.lse || !a.delta || !a.o || !a.d_o) return 2;
  const long items = (long)a.B * a.Lq * a.heads;
  hipLaunchKernelGGL(attn_f32m_delta_kernel, dim3((unsigned)((items + 15) / 16)), dim3(256), 0, s, a);
  hipLaunchKernelGGL(attn_f32m_bwd_kv_kernel, dim3((a.Lk + 127) / 128, a.heads, a.B), dim3(256), 0, s, a);
  hipLaunchKernelGGL(attn_f32m_bwd_q_kernel, dim3((a.Lq + 127) / 128, a.heads, a.B), dim3(256), 0, s, a);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}
