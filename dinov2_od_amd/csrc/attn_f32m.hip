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
  if (active && q < a.Lq) {
    float* op = a.o + ((size_t)b * a.Lq + q) * a.ldo + h * DH;
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(op + db * 32 + 8 * g + 4 * lh) =
            make_float4(o[db][4 * g] * inv, o[db][4 * g + 1] * inv, o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv);
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
