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
// v_mfma_f32_16x16x32_bf16, product computed transposed as in the GEMMs), two LDS stages: the pixel rows of step s+1 are in
// flight in registers and the weight slice by LDS-DMA while step s feeds the MFMAs.
//   X3 = false: bf16 operands (bf16 / fp8 modes);  X3 = true: split products on hi / lo planes (parity-gated bf16x3 mode):
//   pixel = hi + lo in registers, weight in the pair layout [Wh | Wl], Wl Ah + Wh Al + Wh Ah.
#include "dod_common.h"

#define PE_M 128
#define PE_N 128
#define PE_PLANE (128 * 64)        // 128 rows x 32 k' bf16: 8 KiB

__device__ __forceinline__ int pe_swz(int row, int chunk) { return chunk ^ (((row >> 3) & 1) << 1); }

template <bool X3, bool U8>
__global__ __launch_bounds__(256) void patch_embed_kernel(const void* __restrict__ img_, const bf16_t* __restrict__ Wp, int ldw,
                                                          const float* __restrict__ bias, const float* __restrict__ pos,
                                                          float* __restrict__ out, int B, int H, int W, int p, int gh, int gw, int D) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NPL = X3 ? 4 : 2;                      // planes per stage: A(h) [, A(l)], W(h) [, W(l)]
  constexpr int STAGE = NPL * PE_PLANE;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wu = __builtin_amdgcn_readfirstlane(wid);
  const int wm = wu >> 1, wn = wu & 1;
  const int Np = gh * gw, Mtot = B * Np;
  const int tiles_m = (Mtot + PE_M - 1) / PE_M;
  const int tm = blockIdx.x % tiles_m, tn = blockIdx.x / tiles_m;       // consecutive blocks walk M for one weight panel
  const int m0 = tm * PE_M, n0 = tn * PE_N;
  const int nsteps = 3 * (p / 2), kp = nsteps * 32;
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;

  // ---- A gather: thread -> (patch pl = tid >> 1, pixel row r = tid & 1) of the step's row pair
  const int pl = tid >> 1, r = tid & 1;
  size_t pix0;                                          // element offset of img[b][0][py*p + r][px*p] (channel 0, row pair 0)
  size_t cstride, rstride;                              // per channel / per pixel row
  {
    int m = m0 + pl; m = m < Mtot ? m : Mtot - 1;
    const int b = m / Np, rem = m - b * Np, py = rem / gw, px = rem - py * gw;
    if (U8) {                                           // HWC bytes: ((b*H + y)*W + x)*3 + c
      pix0 = (((size_t)b * H + (size_t)py * p + r) * W + (size_t)px * p) * 3;
      cstride = 1; rstride = (size_t)W * 3;
    } else {                                            // CHW floats
      pix0 = (((size_t)b * 3) * H + (size_t)py * p + r) * W + (size_t)px * p;
      cstride = (size_t)H * W; rstride = (size_t)W;
    }
  }
  float pv[16];
  auto gload = [&](int s) {
    const int c = s / (p / 2), si = s - c * (p / 2);
    const size_t o = pix0 + c * cstride + (size_t)(2 * si) * rstride;
    if (U8) {
      const unsigned char* q = (const unsigned char*)img_ + o;
#pragma unroll
      for (int j = 0; j < 16; ++j) pv[j] = j < p ? (float)q[3 * j] / 255.0f : 0.f;      // ToTensor
    } else {
      const float* q = (const float*)img_ + o;
#pragma unroll
      for (int j = 0; j < 16; ++j) pv[j] = j < p ? q[j] : 0.f;
    }
  };
  auto awrite = [&](int stage) {
    char* sa = smem + stage * STAGE;
    uint4 h0, h1;
    h0.x = pack2bf(pv[0], pv[1]); h0.y = pack2bf(pv[2], pv[3]); h0.z = pack2bf(pv[4], pv[5]); h0.w = pack2bf(pv[6], pv[7]);
    h1.x = pack2bf(pv[8], pv[9]); h1.y = pack2bf(pv[10], pv[11]); h1.z = pack2bf(pv[12], pv[13]); h1.w = pack2bf(pv[14], pv[15]);
    const int o0 = pl * 64 + pe_swz(pl, 2 * r) * 16, o1 = pl * 64 + pe_swz(pl, 2 * r + 1) * 16;
    *reinterpret_cast<uint4*>(sa + o0) = h0;
    *reinterpret_cast<uint4*>(sa + o1) = h1;
    if (X3) {
      auto lo2 = [](float a, float b, unsigned hi) { return pack2bf(a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u)); };
      uint4 l0, l1;
      l0.x = lo2(pv[0], pv[1], h0.x); l0.y = lo2(pv[2], pv[3], h0.y); l0.z = lo2(pv[4], pv[5], h0.z); l0.w = lo2(pv[6], pv[7], h0.w);
      l1.x = lo2(pv[8], pv[9], h1.x); l1.y = lo2(pv[10], pv[11], h1.y); l1.z = lo2(pv[12], pv[13], h1.z); l1.w = lo2(pv[14], pv[15], h1.w);
      *reinterpret_cast<uint4*>(sa + PE_PLANE + o0) = l0;
      *reinterpret_cast<uint4*>(sa + PE_PLANE + o1) = l1;
    }
  };
  // ---- W slice by LDS-DMA: per plane each wave issues two 16-row pieces
  const bf16_t *gW0, *gW1;
  {
    const int rl = wid * 32 + (lane >> 2);
    const int c = pe_swz(rl, lane & 3);
    int r0 = n0 + rl, r1 = n0 + rl + 16; r0 = r0 < D ? r0 : D - 1; r1 = r1 < D ? r1 : D - 1;
    gW0 = Wp + (size_t)r0 * ldw + c * 8; gW1 = Wp + (size_t)r1 * ldw + c * 8;
  }
  auto wstage = [&](int stage, int s) {
    char* d = smem + stage * STAGE + (X3 ? 2 : 1) * PE_PLANE + wu * 2048;
    __builtin_amdgcn_global_load_lds((gptr_t)(gW0 + s * 32), (lptr_t)(d), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr_t)(gW1 + s * 32), (lptr_t)(d + 1024), 16, 0, 0);
    if (X3) {
      __builtin_amdgcn_global_load_lds((gptr_t)(gW0 + kp + s * 32), (lptr_t)(d + PE_PLANE), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(gW1 + kp + s * 32), (lptr_t)(d + PE_PLANE + 1024), 16, 0, 0);
    }
  };

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

  gload(0);
  wstage(0, 0);
  awrite(0);
  __syncthreads();            // drains the DMA (vmcnt(0)) and publishes the A tile
  for (int s = 0; s < nsteps; ++s) {
    const int cur = s & 1;
    if (s + 1 < nsteps) { gload(s + 1); wstage(cur ^ 1, s + 1); }
    const char* st = smem + cur * STAGE;
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
    if (s + 1 < nsteps) awrite(cur ^ 1);
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
  if (p < 2 || p > 16 || (p & 1) || D % 4 != 0 || B <= 0) return 2;
  const int gh = H / p, gw = W / p;
  if (gh <= 0 || gw <= 0) return 2;
  const int kp = 3 * (p / 2) * 32, ldw = x3 ? 2 * kp : kp;
  const int tiles = (((B * gh * gw) + PE_M - 1) / PE_M) * ((D + PE_N - 1) / PE_N);
  const int lds = 2 * (x3 ? 4 : 2) * PE_PLANE;
  static bool attr_set[16] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 16 && !attr_set[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(patch_embed_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * PE_PLANE);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(patch_embed_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * PE_PLANE);
    attr_set[dev] = true;
  }
  if (x3) {
    if (u8) hipLaunchKernelGGL((patch_embed_kernel<true, true>), dim3(tiles), dim3(256), lds, s, img, Wp, ldw, bias, pos, out, B, H, W, p, gh, gw, D);
    else hipLaunchKernelGGL((patch_embed_kernel<true, false>), dim3(tiles), dim3(256), lds, s, img, Wp, ldw, bias, pos, out, B, H, W, p, gh, gw, D);
  } else {
    if (u8) hipLaunchKernelGGL((patch_embed_kernel<false, true>), dim3(tiles), dim3(256), lds, s, img, Wp, ldw, bias, pos, out, B, H, W, p, gh, gw, D);
    else hipLaunchKernelGGL((patch_embed_kernel<false, false>), dim3(tiles), dim3(256), lds, s, img, Wp, ldw, bias, pos, out, B, H, W, p, gh, gw, D);
  }
  return hipGetLastError() == hipSuccess ? 0 : 3;
}
