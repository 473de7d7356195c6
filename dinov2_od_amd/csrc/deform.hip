// Deformable bilinear gather (K12 + K13 + K15), fp32.
// Replaces the 4-deep Python loop of DeformableAttention.forward
// (dino_detector/models/deformable_attention.py:101-174) and the point-weight softmax (:92-94),
// reference-point sigmoid (:238).  Arithmetic order follows the reference statement by statement:
//   loc = clamp(ref + off, 0, 1); x = loc_x*(w-1); y = loc_y*(h-1); x0 = floor(x); x1 = x0+1;
//   clamp all four to the grid; wx1 = x - x0; wx0 = 1 - wx1 (same in y);
//   out = sum_p aw[p] * (v00*wx0*wy0 + v01*wx0*wy1 + v10*wx1*wy0 + v11*wx1*wy1)
// where v01 is (y1,x0) and v10 is (y0,x1) as at :150-153.  The `idx < hw` guard (:164) is always
// true after clamping.  Coordinates are fp32 throughout: floor() is discontinuous.
//
// proj row layout (one small fused GEMM): [ref logit x, ref logit y | Hd*P*2 offsets | Hd*P weight logits].
// One wave per (b, q, head); lanes run along the head's dh value columns (coalesced row reads of V).
#include "dod_common.h"

#define DF_MAXP 8

__global__ __launch_bounds__(256) void deform_sample_kernel(const float* __restrict__ proj, int ldp,
                                                            const float* __restrict__ values, int B, int Q, int N,
                                                            int Hd, int P, int dh, int h, int w,
                                                            float* __restrict__ out, int proj_shared, bf16_t* __restrict__ out3) {
  const int lane = threadIdx.x & 63;
  const long item = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (item >= (long)B * Q * Hd) return;
  const int hd = (int)(item % Hd);
  const long bq = item / Hd;
  const int b = (int)(bq / Q);
  // proj_shared: the projections depend on the query only (decoder layer 0: tgt = query_embed for every image)
  const float* pr = proj + (size_t)(proj_shared ? (bq - (long)b * Q) : bq) * ldp;
  const float refx = sigmoidf_(pr[0]), refy = sigmoidf_(pr[1]);
  const float* off = pr + 2 + hd * P * 2;
  const float* awl = pr + 2 + Hd * P * 2 + hd * P;
  // softmax over the P points
  float mx = -INFINITY;
  for (int p = 0; p < P; ++p) mx = fmaxf(mx, awl[p]);
  float aw[DF_MAXP], den = 0.f;
#pragma unroll
  for (int p = 0; p < DF_MAXP; ++p) {
    aw[p] = p < P ? expf(awl[p] - mx) : 0.f;
    den += aw[p];
  }
  const int Dd = Hd * dh;
  const float* vb = values + (size_t)b * N * Dd + hd * dh;
  float acc0 = 0.f, acc1 = 0.f;
  const bool d0ok = lane < dh, d1ok = lane + 64 < dh;
#pragma unroll
  for (int p = 0; p < DF_MAXP; ++p) {
    if (p < P) {
      float lx = fminf(fmaxf(refx + off[2 * p], 0.f), 1.f);
      float ly = fminf(fmaxf(refy + off[2 * p + 1], 0.f), 1.f);
      lx = lx * (float)(w - 1);
      ly = ly * (float)(h - 1);
      int x0 = (int)floorf(lx), y0 = (int)floorf(ly);
      int x1 = x0 + 1, y1 = y0 + 1;
      x0 = min(max(x0, 0), w - 1); x1 = min(max(x1, 0), w - 1);
      y0 = min(max(y0, 0), h - 1); y1 = min(max(y1, 0), h - 1);
      const float wx1 = lx - (float)x0, wx0 = 1.0f - wx1;
      const float wy1 = ly - (float)y0, wy0 = 1.0f - wy1;
      const float w00 = wx0 * wy0, w01 = wx0 * wy1, w10 = wx1 * wy0, w11 = wx1 * wy1;
      const float* r00 = vb + (size_t)(y0 * w + x0) * Dd;
      const float* r01 = vb + (size_t)(y1 * w + x0) * Dd;
      const float* r10 = vb + (size_t)(y0 * w + x1) * Dd;
      const float* r11 = vb + (size_t)(y1 * w + x1) * Dd;
      const float a = aw[p] / den;
      if (d0ok) {
        const float v = ((r00[lane] * w00 + r01[lane] * w01) + r10[lane] * w10) + r11[lane] * w11;
        acc0 += v * a;
      }
      if (d1ok) {
        const float v = ((r00[lane + 64] * w00 + r01[lane + 64] * w01) + r10[lane + 64] * w10) + r11[lane + 64] * w11;
        acc1 += v * a;
      }
    }
  }
  float* op = out + (size_t)bq * Dd + hd * dh;
  if (d0ok) op[lane] = acc0;
  if (d1ok) op[lane + 64] = acc1;
  if (out3) {        // the same values as the bf16x3 operand [hi | hi | lo] of the output projection (K17)
    bf16_t* o3 = out3 + (size_t)bq * 3 * Dd + hd * dh;
    if (d0ok) { const bf16_t hi = f2bf(acc0); o3[lane] = hi; o3[Dd + lane] = hi; o3[2 * Dd + lane] = f2bf(acc0 - bf2f(hi)); }
    if (d1ok) { const bf16_t hi = f2bf(acc1); o3[lane + 64] = hi; o3[Dd + lane + 64] = hi; o3[2 * Dd + lane + 64] = f2bf(acc1 - bf2f(hi)); }
  }
}

int launch_deform_sample(const float* proj, int ldp, const float* values, int B, int Q, int N, int Hd, int P, int dh,
                         int h, int w, float* out, hipStream_t s, int proj_shared, bf16_t* out3) {
  if (P > DF_MAXP || P <= 0 || dh > 128 || dh <= 0) return 2;
  if (h * w != N) return 2;   // the reference raises / re-infers here (deformable_attention.py:76-83)
  const long items = (long)B * Q * Hd;
  hipLaunchKernelGGL(deform_sample_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, s, proj, ldp, values, B, Q, N,
                     Hd, P, dh, h, w, out, proj_shared, out3);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}
