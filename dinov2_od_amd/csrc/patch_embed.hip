// Fused patch embedding (K1 + K2 of SURVEY.md): Conv2d(3 -> D, kernel p, stride p) + flatten + transpose + bias + position add
// (modeling_dinov2.py:139-149, 107-112) as ONE kernel with an IMPLICIT im2col -- the A tile of the GEMM is gathered from the
// image in the load stage (coalesced reads of whole pixel rows), converted to bf16 in registers and written to LDS; no
// [B*Np, 640] column buffer goes through HBM.  The same kernel also takes the input pipeline's uint8 HWC image directly
// (train.py:584-587: Resize + ToTensor; preproc.hip's resampled bytes), applying ToTensor's x / 255 in the load, which removes the
// fp32 CHW round trip as well.
//
//   out[b][1 + m][n] = sum_{c,i,j} W[n][c][i][j] * img[b][c][py*p + i][px*p + j] + bias[n] + pos[1 + m][n],   m = py*gw + px
//
// K is walked in steps of TWO pixel rows of one channel: k' = r*16 + j (r = 0, 1; j < p <= 16, zero padded), 3*p/2 steps (21 at
// p = 14); the weight is packed to that order at load time.  Tile: 128 patches x 128 channels, 4 waves (2 x 2 of 64 x 64,
// v_mfma_f32_16x16x32_bf16, product computed transposed as in the GEMMs), two LDS stages fed from a three-step-deep register ring
// (pixels AND weight slice: the loads of steps s+1 .. s+3 are in flight while step s feeds the MFMAs).
//   X3 = false: bf16 operands (bf16 / fp8 modes);  X3 = true: split products on hi / lo planes (parity-gated bf16x3 mode):
//   pixel = hi + lo in registers, weight in the pair layout [Wh | Wl], Wl Ah + Wh Al + Wh Ah.
#include "dod_common.h"

#define PE_M 128
#define PE_N 128
#define PE_PLANE (128 * 64)        // 128 rows x 32 k' bf16: 8 KiB

__device__ __forceinline__ int pe_swz(int row, int chunk) { return chunk ^ (((row >> 3) & 1) << 1); }

template <bool X3, bool U8, int P>
__global__ __launch_bounds__(256) void patch_embed_kernel(const void* __restrict__ img_, const bf16_t* __restrict__ Wp, int ldw,
                                                          const float* __restrict__ bias, const float* __restrict__ pos,
                                                          float* __restrict__ out, int B, int H, int W, int gh, int gw, int D) {
  constexpr int p = P;                                  // compile-time: every pass below is unconditional (a runtime bound makes hipcc
                                                        // branch around each load and wait for it alone)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NPL = X3 ? 4 : 2;                      // planes per stage: A(h) [, A(l)], W(h) [, W(l)]
  constexpr int STAGE = NPL * PE_PLANE;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wu = __builtin_amdgcn_readfirstlane(wid);
  const int wm = wu >> 1, wn = wu & 1;
  const int Np = gh * gw, Mtot = B * Np;
  const int tiles_m = (Mtot + PE_M - 1) / PE_M;
  // XCD-aware order (round 3): workgroup b runs on XCD b % 8 (round-robin dispatch), and each XCD has its own L2.  All n-tiles of a
  // patch tile are given to ONE XCD, back to back in its dispatch order -- the pixels are then fetched from HBM once and hit that
  // XCD's L2 for the other tiles_n - 1 readers; the packed weight (1 MB at D = 768) stays in every L2 anyway.  With consecutive
  // blocks sharing a patch tile (rounds 1-2) its six n-tiles sat on six XCDs: 1 796 MB of HBM-side traffic per batch of 64
  // against 475 MB algorithmic (profiles/r02_pmc_traffic_bf16.json).
  const int tiles_n = (D + PE_N - 1) / PE_N;
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int tn = idx % tiles_n, tm = (idx / tiles_n) * 8 + xcd;
  if (tm >= tiles_m) return;                            // the grid is padded to whole groups of 8 patch tiles
  const int m0 = tm * PE_M, n0 = tn * PE_N;
  const int nsteps = 3 * (p / 2), kp = nsteps * 32;

  // ---- A gather.  Per step the tile needs 2 pixel rows x 128 patches x p pixels; element e = (r * 128 + pl) * p + j is fetched by
  // thread e % 256 in pass e / 256, so a wave instruction reads 64 CONSECUTIVE pixels of an image row (patches that follow each
  // other in a patch row are contiguous there) -- and is scattered to its [patch][k' = r*16 + j] slot in LDS as one bf16.
  // passes: pixel PAIRS (2 * 128 * p / 2) / 256 = p / 2 exactly, every thread owns one pair (8-byte load: p and W even keep it
  // aligned; a dword per lane made the kernel bound by the number of vector-memory instructions) per pass
  constexpr int NP2 = P / 2;
  unsigned goff[NP2];                                   // element offset of (channel 0, row pair 0), per pass (images < 4 G elements)
  int lofs[NP2];                                        // LDS byte offset inside the A plane, per pass
  size_t cstride, rstride;
  if (U8) { cstride = 1; rstride = (size_t)W * 3; } else { cstride = (size_t)H * W; rstride = (size_t)W; }
#pragma unroll
  for (int i = 0; i < NP2; ++i) {
    const int e = tid + 256 * i;
    const int rr = e / (PE_M * NP2), rem_e = e - rr * (PE_M * NP2), pl = rem_e / NP2, j = 2 * (rem_e - pl * NP2);
    int m = m0 + pl; m = m < Mtot ? m : Mtot - 1;
    const int b = m / Np, rem = m - b * Np, py = rem / gw, px = rem - py * gw;
    if (U8) goff[i] = (unsigned)((((size_t)b * H + (size_t)py * p + rr) * W + (size_t)px * p + j) * 3);
    else goff[i] = (unsigned)((((size_t)b * 3) * H + (size_t)py * p + rr) * W + (size_t)px * p + j);
    const int kq = rr * 16 + j;                         // k' of the pair's first pixel (even: both sit in one 16-byte chunk)
    lofs[i] = pl * 64 + pe_swz(pl, kq >> 3) * 16 + (kq & 7) * 2;
  }
  // Register ring, three steps deep: the loads of steps s+1 .. s+3 are in flight while step s computes -- a step is only 16 (48)
  // MFMAs per wave, far shorter than a memory round trip, so one step of prefetch left the kernel latency-bound (3.5 us per step).
  // The weight slice goes through registers too (plain loads: beside an LDS-DMA in flight hipcc would wait vmcnt(0) for every
  // ordinary load and drain the ring each step).
  // (macros over plain locals: arrays of uint4 handed to lambdas by reference stayed in scratch)
  float pv0[P], pv1[P], pv2[P];
  uint4 wA0, wB0, wC0, wD0, wA1, wB1, wC1, wD1, wA2, wB2, wC2, wD2;       // per ring slot: W(h) halves [, W(l) halves]
  // W slice: thread -> row (tid >> 1), 32-byte half (tid & 1) of the step's 64-byte row
  const bf16_t* gWr;
  int wofs0, wofs1;
  {
    const int rl = tid >> 1, hf = tid & 1;
    int rr = n0 + rl; rr = rr < D ? rr : D - 1;
    gWr = Wp + (size_t)rr * ldw + hf * 16;
    wofs0 = rl * 64 + pe_swz(rl, 2 * hf) * 16;
    wofs1 = rl * 64 + pe_swz(rl, 2 * hf + 1) * 16;
  }
#define PE_GLOAD(PV, WA, WB, WC, WD, s_)                                                                    \
  {                                                                                                         \
    const int c_ = (s_) / (p / 2), si_ = (s_) - c_ * (p / 2);                                               \
    const size_t o_ = c_ * cstride + (size_t)(2 * si_) * rstride;                                           \
    _Pragma("unroll") for (int i = 0; i < NP2; ++i) {                                                       \
      if (U8) {                                                                                             \
        const unsigned char* q_ = (const unsigned char*)img_ + goff[i] + o_;                                \
        PV[2 * i] = (float)q_[0] / 255.0f; PV[2 * i + 1] = (float)q_[3] / 255.0f;   /* ToTensor */           \
      } else {                                                                                              \
        const float2 v_ = *reinterpret_cast<const float2*>((const float*)img_ + goff[i] + o_);              \
        PV[2 * i] = v_.x; PV[2 * i + 1] = v_.y;                                                             \
      }                                                                                                     \
    }                                                                                                       \
    const uint4* wp_ = reinterpret_cast<const uint4*>(gWr + (s_) * 32);                                     \
    WA = wp_[0]; WB = wp_[1];                                                                               \
    if (X3) { const uint4* wq_ = reinterpret_cast<const uint4*>(gWr + kp + (s_) * 32); WC = wq_[0]; WD = wq_[1]; } \
  }
#define PE_SWRITE(PV, WA, WB, WC, WD, stage_)                                                               \
  {                                                                                                         \
    char* sa_ = smem + (stage_) * STAGE;                                                                    \
    _Pragma("unroll") for (int i = 0; i < NP2; ++i) {                                                       \
      const unsigned hi_ = pack2bf(PV[2 * i], PV[2 * i + 1]);                                               \
      *reinterpret_cast<unsigned*>(sa_ + lofs[i]) = hi_;                                                    \
      if (X3) *reinterpret_cast<unsigned*>(sa_ + PE_PLANE + lofs[i]) =                                      \
          pack2bf(PV[2 * i] - __uint_as_float(hi_ << 16), PV[2 * i + 1] - __uint_as_float(hi_ & 0xffff0000u)); \
    }                                                                                                       \
    char* sw_ = sa_ + (X3 ? 2 : 1) * PE_PLANE;                                                              \
    *reinterpret_cast<uint4*>(sw_ + wofs0) = WA;                                                            \
    *reinterpret_cast<uint4*>(sw_ + wofs1) = WB;                                                            \
    if (X3) { *reinterpret_cast<uint4*>(sw_ + PE_PLANE + wofs0) = WC; *reinterpret_cast<uint4*>(sw_ + PE_PLANE + wofs1) = WD; } \
  }

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int l15 = lane & 15, l4 = lane >> 4;
  int offA[4], offW[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { const int row = wm * 64 + i * 16 + l15; offA[i] = row * 64 + pe_swz(row, l4) * 16; }
#pragma unroll
  for (int j = 0; j < 4; ++j) { const int row = wn * 64 + j * 16 + l15; offW[j] = (X3 ? 2 : 1) * PE_PLANE + row * 64 + pe_swz(row, l4) * 16; }

  // the padding columns j = p..15 of the A planes are never written: zero both stages once
  for (int i = tid; i < 2 * STAGE / 16; i += 256) {
    const int st_ = i / (STAGE / 16), o_ = (i - st_ * (STAGE / 16)) * 16;
    if (o_ < (X3 ? 2 : 1) * PE_PLANE) *reinterpret_cast<uint4*>(smem + st_ * STAGE + o_) = make_uint4(0u, 0u, 0u, 0u);
  }
  __syncthreads();
  auto compute = [&](int stage) {
    const char* st = smem + stage * STAGE;
    bf16x8 wh[4], wl[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      wh[j] = *reinterpret_cast<const bf16x8*>(st + offW[j]);
      if (X3) wl[j] = *reinterpret_cast<const bf16x8*>(st + offW[j] + PE_PLANE);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bf16x8 ah = *reinterpret_cast<const bf16x8*>(st + offA[i]);
      bf16x8 al;
      if (X3) al = *reinterpret_cast<const bf16x8*>(st + offA[i] + PE_PLANE);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (X3) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[j], ah, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[j], al, acc[i][j], 0, 0, 0);
        }
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[j], ah, acc[i][j], 0, 0, 0);
      }
    }
  };
  // nsteps = 3 * (p / 2) is a multiple of 3: step s lives in ring register s % 3 and LDS stage s & 1
  PE_GLOAD(pv0, wA0, wB0, wC0, wD0, 0)
  PE_GLOAD(pv1, wA1, wB1, wC1, wD1, 1)
  PE_GLOAD(pv2, wA2, wB2, wC2, wD2, 2)
  PE_SWRITE(pv0, wA0, wB0, wC0, wD0, 0)
  __syncthreads();
  for (int s = 0; s < nsteps; s += 3) {
    // step s: ring slot 0 (its data went to LDS during step s - 1) is refilled with step s + 3; slot 1 goes to LDS for step s + 1
    if (s + 3 < nsteps) PE_GLOAD(pv0, wA0, wB0, wC0, wD0, s + 3)
    compute(s & 1);
    PE_SWRITE(pv1, wA1, wB1, wC1, wD1, (s + 1) & 1)
    __syncthreads();
    if (s + 4 < nsteps) PE_GLOAD(pv1, wA1, wB1, wC1, wD1, s + 4)
    compute((s + 1) & 1);
    PE_SWRITE(pv2, wA2, wB2, wC2, wD2, (s + 2) & 1)
    __syncthreads();
    if (s + 5 < nsteps) PE_GLOAD(pv2, wA2, wB2, wC2, wD2, s + 5)
    compute((s + 2) & 1);
    if (s + 3 < nsteps) PE_SWRITE(pv0, wA0, wB0, wC0, wD0, (s + 3) & 1)
    __syncthreads();
  }
  // ---- epilogue: + bias + position row, rows shifted past the CLS slot
  const int N = Np + 1;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + i * 16 + l15;
    if (m >= Mtot) continue;
    const int b = m / Np, rem = m - b * Np;
    float* orow = out + ((size_t)b * N + 1 + rem) * D;
    const float* prow = pos + (size_t)(1 + rem) * D;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + 4 * l4;
      if (n >= D) continue;
      const float4 bv = *reinterpret_cast<const float4*>(bias + n), pp = *reinterpret_cast<const float4*>(prow + n);
      const f32x4 a = acc[i][j];
      *reinterpret_cast<float4*>(orow + n) = make_float4(a[0] + bv.x + pp.x, a[1] + bv.y + pp.y, a[2] + bv.z + pp.z, a[3] + bv.w + pp.w);
    }
  }
}

// W [D, 3, p, p] fp32 -> Wp [D, kp] bf16 (kp = 3*(p/2)*32), k' order of the kernel; x3: pair layout [D, 2*kp] = [hi | lo]
__global__ void patch_pack_kernel(const float* __restrict__ W, int D, int p, bf16_t* __restrict__ out, int x3) {
  const int kp = 3 * (p / 2) * 32;
  const size_t total = (size_t)D * kp;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int n = (int)(idx / kp), k = (int)(idx - (size_t)n * kp);
    const int s = k >> 5, kk = k & 31, r = kk >> 4, j = kk & 15;
    const int c = s / (p / 2), si = s - c * (p / 2);
    const float v = j < p ? W[(((size_t)n * 3 + c) * p + (2 * si + r)) * p + j] : 0.f;
    const bf16_t hi = f2bf(v);
    if (x3) {
      out[(size_t)n * 2 * kp + k] = hi;
      out[(size_t)n * 2 * kp + kp + k] = f2bf(v - bf2f(hi));
    } else {
      out[idx] = hi;
    }
  }
}

int launch_patch_pack(const float* W, int D, int p, bf16_t* out, int x3, hipStream_t s) {
  if (p < 2 || p > 16 || (p & 1)) return 2;
  const size_t total = (size_t)D * 3 * (p / 2) * 32;
  hipLaunchKernelGGL(patch_pack_kernel, dim3((unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096)), dim3(256), 0, s, W, D, p, out, x3);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}

// img: fp32 [B,3,H,W] (u8 = 0) or uint8 [B,H,W,3] (u8 = 1).  Wp from launch_patch_pack.  out: the residual stream x [B, Np+1, D].
int launch_patch_embed(const void* img, int u8, int B, int H, int W, int p, const bf16_t* Wp, int x3, const float* bias, const float* pos,
                       float* out, int D, hipStream_t s) {
  if ((p != 14 && p != 16) || D % 4 != 0 || B <= 0 || (!u8 && (W & 1))) return 2;      // fp32 rows must keep 8-byte alignment
  const int gh = H / p, gw = W / p;
  if (gh <= 0 || gw <= 0) return 2;
  if ((size_t)B * 3 * H * W >= 0xffffffffull) return 2;          // 32-bit element offsets
  const int kp = 3 * (p / 2) * 32, ldw = x3 ? 2 * kp : kp;
  const int tiles = ((((B * gh * gw) + PE_M - 1) / PE_M + 7) / 8 * 8) * ((D + PE_N - 1) / PE_N);     // patch tiles padded to groups of 8 (one per XCD)
  const int lds = 2 * (x3 ? 4 : 2) * PE_PLANE;
#define PE_LAUNCH(X3_, U8_, P_) hipLaunchKernelGGL((patch_embed_kernel<X3_, U8_, P_>), dim3(tiles), dim3(256), lds, s, img, Wp, ldw, bias, pos, out, B, H, W, gh, gw, D)
  if (p == 14) {
    if (x3) { if (u8) PE_LAUNCH(true, true, 14); else PE_LAUNCH(true, false, 14); }
    else { if (u8) PE_LAUNCH(false, true, 14); else PE_LAUNCH(false, false, 14); }
  } else {
    if (x3) { if (u8) PE_LAUNCH(true, true, 16); else PE_LAUNCH(true, false, 16); }
    else { if (u8) PE_LAUNCH(false, true, 16); else PE_LAUNCH(false, false, 16); }
  }
#undef PE_LAUNCH
  return hipGetLastError() == hipSuccess ? 0 : 3;
}
