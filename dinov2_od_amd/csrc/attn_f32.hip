// Generic fp32 attention (online softmax, VALU fp32): softmax(Q K^T * scale) V per (batch, head).
// Used where exact-fp32 arithmetic is required or the shape is tiny:
//   - strict-parity mode of the backbone attention (K5, modeling_dinov2.py:153-178)
//   - decoder self-attention over the Q queries (K11: nn.MultiheadAttention, deformable_attention.py:232-235)
//   - dense cross-attention of the nn.TransformerDecoder branch (K20, detr_decoder.py:28-35, 62-69)
// Arbitrary head_dim <= 128 (multiple of 4), arbitrary Lq/Lk, strided row-major operands.
// Workgroup = 4 waves; each wave owns 8 query rows (32 rows per workgroup); keys are processed in
// tiles of 64 staged in LDS (K padded to dh+4 floats per row: conflict-free ds_read_b128 with the
// key on the lane; V unpadded: consecutive lanes read consecutive d).  QK phase: lane = key;
// PV phase: lane = output column d (two columns per lane when dh > 64), probabilities broadcast
// from a per-wave LDS tile.
#include "dod_common.h"
#include <cstdlib>

#define AF_R 8
#define AF_WAVES 4
#define AF_KV 64

__global__ __launch_bounds__(256) void attn_f32_kernel(AttnF32 a) {
  extern __shared__ __attribute__((aligned(16))) float fsm[];
  const int dh = a.dh, ldk_s = dh + 4;
  float* sK = fsm;                               // [64][dh+4]
  float* sV = sK + AF_KV * ldk_s;                // [64][dh]
  float* sQ = sV + AF_KV * dh;                   // [32][dh]
  float* sP = sQ + AF_WAVES * AF_R * dh;         // [4][8][64]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int b = blockIdx.z, h = blockIdx.y;
  const int q0 = blockIdx.x * (AF_WAVES * AF_R);
  const float* Q = a.q + (size_t)b * a.Lq * a.ldq + h * dh;
  const float* K = a.k + (size_t)b * a.Lk * a.ldk + h * dh;
  const float* V = a.v + (size_t)b * a.Lk * a.ldv + h * dh;
  const int nvec = dh >> 2;

  // stage the workgroup's 32 query rows (clamped)
  for (int i = tid; i < AF_WAVES * AF_R * nvec; i += 256) {
    const int r = i / nvec, c = i - r * nvec;
    int qr = q0 + r; qr = qr < a.Lq ? qr : a.Lq - 1;
    reinterpret_cast<float4*>(sQ + r * dh)[c] = *reinterpret_cast<const float4*>(Q + (size_t)qr * a.ldq + 4 * c);
  }
  float m_run[AF_R], l_run[AF_R], o0[AF_R], o1[AF_R];
#pragma unroll
  for (int r = 0; r < AF_R; ++r) { m_run[r] = -INFINITY; l_run[r] = 0.f; o0[r] = 0.f; o1[r] = 0.f; }
  const float* myQ = sQ + wid * AF_R * dh;
  float* myP = sP + wid * AF_R * AF_KV;
  const bool d0ok = lane < dh, d1ok = (lane + 64) < dh;

  const int nkt = (a.Lk + AF_KV - 1) / AF_KV;
  for (int kt = 0; kt < nkt; ++kt) {
    __syncthreads();   // previous tile fully consumed (also orders the sQ fill before first use)
    for (int i = tid; i < AF_KV * nvec; i += 256) {
      const int r = i / nvec, c = i - r * nvec;
      int kr = kt * AF_KV + r; kr = kr < a.Lk ? kr : a.Lk - 1;
      *reinterpret_cast<float4*>(sK + r * ldk_s + 4 * c) = *reinterpret_cast<const float4*>(K + (size_t)kr * a.ldk + 4 * c);
      *reinterpret_cast<float4*>(sV + r * dh + 4 * c) = *reinterpret_cast<const float4*>(V + (size_t)kr * a.ldv + 4 * c);
    }
    __syncthreads();
    // ---- scores: lane = key
    float sc[AF_R];
#pragma unroll
    for (int r = 0; r < AF_R; ++r) sc[r] = 0.f;
    const float* kp = sK + lane * ldk_s;
    for (int c = 0; c < nvec; ++c) {
      const float4 kv = *reinterpret_cast<const float4*>(kp + 4 * c);
#pragma unroll
      for (int r = 0; r < AF_R; ++r) {
        const float4 qv = *reinterpret_cast<const float4*>(myQ + r * dh + 4 * c);   // wave-uniform (broadcast)
        sc[r] = fmaf(qv.x, kv.x, sc[r]);
        sc[r] = fmaf(qv.y, kv.y, sc[r]);
        sc[r] = fmaf(qv.z, kv.z, sc[r]);
        sc[r] = fmaf(qv.w, kv.w, sc[r]);
      }
    }
    const bool kvalid = (kt * AF_KV + lane) < a.Lk;
    float alpha[AF_R];
#pragma unroll
    for (int r = 0; r < AF_R; ++r) {
      const float v = kvalid ? sc[r] * a.scale : -INFINITY;
      const float mx = wave_max(v);
      const float m_new = fmaxf(m_run[r], mx);
      alpha[r] = expf(m_run[r] - m_new);
      const float p = expf(v - m_new);
      l_run[r] = l_run[r] * alpha[r] + wave_sum(p);
      m_run[r] = m_new;
      myP[r * AF_KV + lane] = p;
      o0[r] *= alpha[r];
      o1[r] *= alpha[r];
    }
    // wave-private LDS tile: make the writes visible to the other lanes of this wave
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ---- PV: lane = d
    for (int j = 0; j < AF_KV; j += 4) {
      float v0[4], v1[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        v0[t] = d0ok ? sV[(j + t) * dh + lane] : 0.f;
        v1[t] = d1ok ? sV[(j + t) * dh + lane + 64] : 0.f;
      }
#pragma unroll
      for (int r = 0; r < AF_R; ++r) {
        const float4 p4 = *reinterpret_cast<const float4*>(myP + r * AF_KV + j);   // broadcast
        o0[r] = fmaf(p4.x, v0[0], o0[r]); o0[r] = fmaf(p4.y, v0[1], o0[r]);
        o0[r] = fmaf(p4.z, v0[2], o0[r]); o0[r] = fmaf(p4.w, v0[3], o0[r]);
        o1[r] = fmaf(p4.x, v1[0], o1[r]); o1[r] = fmaf(p4.y, v1[1], o1[r]);
        o1[r] = fmaf(p4.z, v1[2], o1[r]); o1[r] = fmaf(p4.w, v1[3], o1[r]);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < AF_R; ++r) {
    const int q = q0 + wid * AF_R + r;
    if (q < a.Lq) {
      float* op = a.o + ((size_t)b * a.Lq + q) * a.ldo + h * dh;
      const float inv = 1.0f / l_run[r];
      if (d0ok) op[lane] = o0[r] * inv;
      if (d1ok) op[lane + 64] = o1[r] * inv;
      if (a.o3) {        // also as the bf16x3 operand [hi | hi | lo] of the next query-side linear
        const int Dm = a.heads * dh;
        bf16_t* o3 = a.o3 + ((size_t)b * a.Lq + q) * 3 * Dm + h * dh;
        // the empty asm makes y opaque: the fp32 output above is the ROUNDED product, and hipcc (-ffp-contract=fast) would otherwise fold the
        // multiply into the residual's subtraction, fma(o, inv, -hi), and split a different number than split3 of the stored output does
        if (d0ok) { float y = o0[r] * inv; asm volatile("" : "+v"(y)); const bf16_t hi = f2bf(y); o3[lane] = hi; o3[Dm + lane] = hi; o3[2 * Dm + lane] = f2bf(y - bf2f(hi)); }
        if (d1ok) { float y = o1[r] * inv; asm volatile("" : "+v"(y)); const bf16_t hi = f2bf(y); o3[lane + 64] = hi; o3[Dm + lane + 64] = hi; o3[2 * Dm + lane + 64] = f2bf(y - bf2f(hi)); }
      }
    }
  }
}

int launch_attn_f32_mfma(const AttnF32& a, hipStream_t s);   // attn_f32m.hip: head_dim 32 / 64 / 96 on the fp32 MFMA

int launch_attn_f32(const AttnF32& a, hipStream_t s) {
  if (a.B <= 0 || a.heads <= 0 || a.Lq <= 0 || a.Lk <= 0) return 1;
  {
    static const char* fm = DOD_TUNE_ENV("DINODET_ATTN_F32_MFMA");     // "0": always the generic VALU kernel (A/B)
    if ((a.lse || !fm || fm[0] != '0') && (a.dh == 32 || a.dh == 64 || a.dh == 96) && a.ldo % 4 == 0 && a.ldq % 4 == 0 && a.ldk % 4 == 0 && a.ldv % 4 == 0)
      return launch_attn_f32_mfma(a, s);
  }
  if (a.dh % 4 != 0 || a.dh > 128 || a.dh <= 0) return 2;
  if (a.ldq % 4 || a.ldk % 4 || a.ldv % 4) return 2;
  if (a.lse) return 2;                       // the log-sum-exp output exists in the MFMA kernel only
  const size_t lds = sizeof(float) * ((size_t)AF_KV * (a.dh + 4) + (size_t)AF_KV * a.dh +
                                      (size_t)AF_WAVES * AF_R * a.dh + (size_t)AF_WAVES * AF_R * AF_KV);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_f32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    attr_set = true;
  }
  const dim3 grid((a.Lq + AF_WAVES * AF_R - 1) / (AF_WAVES * AF_R), a.heads, a.B);
  hipLaunchKernelGGL(attn_f32_kernel, grid, dim3(256), lds, s, a);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}
