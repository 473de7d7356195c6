// bf16x3 split-product GEMM for the parity-gated mode:
//   C[M,N] = A[M,K] W[N,K]^T  with  A = Ah + Al, W = Wh + Wl (hi = bf16(x), lo = bf16(x - hi)),
//   computed as  Ah Wh^T + Ah Wl^T + Al Wh^T   (lo.lo, 2^-18 relative, dropped; fp32 accumulate).
// Operands in the PAIR layout: A2 [M, 2K] = [Ah | Al], W2 [N, 2K] = [Wh | Wl] (bf16, K contiguous in each half).
// Same nn.Linear sites as gemm_bf16.hip (modeling_dinov2.py:199-201, 246-252, 281-297) at fp32-class accuracy.
//
// Why a kernel of its own: run as ONE bf16 GEMM with K' = 3K over [Ah|Ah|Al] x [Wh|Wl|Wh] (the first version), a K step
// stages six tiles for three products -- the bf16 kernels are bound by the per-CU global->LDS path, so the split GEMM ran
// at a third of their rate (275 TFLOP/s algorithmic).  Here a K step stages FOUR tiles (Ah, Al, Wh, Wl) and issues the three
// products from them: a third fewer staged bytes per product, and the kernel turns MFMA-bound.
// Tile 256x256x32, 16 waves (4x4 of 64x64, v_mfma_f32_16x16x32_bf16), one workgroup per CU, 64-KiB K-tiles in a
// two-slot ring (tile t+1 streams in under the 48 MFMAs per wave of tile t), transposed product + the shared LDS-staged
// epilogue (gemm_epi.h), XCD-aware grouped tile order.
#include "dod_common.h"
#include "gemm_epi.h"
#include <cstdlib>

#define X3M 256
#define X3N 256
#define X3K 32
#define X3_PLANE (256 * X3K * 2)          // one 256-row x 32-k bf16 plane: 16 KiB
#define X3_STAGE (4 * X3_PLANE)           // Ah | Al | Wh | Wl = 64 KiB
#define X3_SLOTS 2

__device__ __forceinline__ int x3_swz(int row, int chunk) { return chunk ^ (((row >> 3) & 1) << 1); }   // 16x16x32 lane map, 64-B rows

// PLAIN = true: the same structure as an ordinary bf16 GEMM with BK = 64 -- the second plane of each operand holds k 32..63
// of the K-tile instead of the lo halves, two products per K-tile (experiment / DINODET_GEMM_TILE=x).
template <bool PLAIN>
__global__ __launch_bounds__(1024) void gemm_x3_256x256_kernel(const bf16_t* __restrict__ A2, int lda,
                                                               const bf16_t* __restrict__ W2, int ldw, int M, int N,
                                                               int K, GemmEpi e, int GM) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 2, wn = wid & 3;
  const int tiles_m = (M + X3M - 1) / X3M, tiles_n = (N + X3N - 1) / X3N;
  const int nwg = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  int tm, tn;
  {
    const int per_group = GM * tiles_n;
    const int grp = bid / per_group, first_m = grp * GM;
    const int gsz = (tiles_m - first_m) < GM ? (tiles_m - first_m) : GM;
    const int in_g = bid - grp * per_group;
    tm = first_m + in_g % gsz;
    tn = in_g / gsz;
  }
  const int m0 = tm * X3M, n0 = tn * X3N;
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  // per K-tile a wave issues ONE 16-row piece of each plane (16 waves x 16 rows = 256)
  const bf16_t* gA; const bf16_t* gW;
  {
    const int rl = wid * 16 + (lane >> 2);
    const int c = x3_swz(rl, lane & 3);
    int ra = m0 + rl; ra = ra < M ? ra : M - 1;
    int rw = n0 + rl; rw = rw < N ? rw : N - 1;
    gA = A2 + (size_t)ra * lda + c * 8;
    gW = W2 + (size_t)rw * ldw + c * 8;
  }
  const int wu = __builtin_amdgcn_readfirstlane(wid);
#define STAGE_X3(slot_, k0)                                                                                \
  {                                                                                                        \
    char* s_ = smem + (slot_) * X3_STAGE + wu * 1024;                                                      \
    __builtin_amdgcn_global_load_lds((gptr_t)(gA + (k0)), (lptr_t)(s_), 16, 0, 0);                         \
    __builtin_amdgcn_global_load_lds((gptr_t)(gA + (PLAIN ? 32 : K) + (k0)), (lptr_t)(s_ + X3_PLANE), 16, 0, 0); \
    __builtin_amdgcn_global_load_lds((gptr_t)(gW + (k0)), (lptr_t)(s_ + 2 * X3_PLANE), 16, 0, 0);          \
    __builtin_amdgcn_global_load_lds((gptr_t)(gW + (PLAIN ? 32 : K) + (k0)), (lptr_t)(s_ + 3 * X3_PLANE), 16, 0, 0); \
  }
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int KSTEP = PLAIN ? 2 * X3K : X3K;
  const int nk = K / KSTEP;
  const int l15 = lane & 15, l4 = lane >> 4;
  STAGE_X3(0, 0)
  int offA[4], offW[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { const int row = wm * 64 + i * 16 + l15; offA[i] = row * 64 + x3_swz(row, l4) * 16; }
#pragma unroll
  for (int j = 0; j < 4; ++j) { const int row = wn * 64 + j * 16 + l15; offW[j] = 2 * X3_PLANE + row * 64 + x3_swz(row, l4) * 16; }
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // tile kt landed (the only one in flight)
    __builtin_amdgcn_s_barrier();                         // ... for all waves, and all waves are done with tile kt-1
    asm volatile("" ::: "memory");
    if (kt + 1 < nk) STAGE_X3((kt + 1) & 1, (kt + 1) * KSTEP)
    const char* st = smem + (kt & 1) * X3_STAGE;
    bf16x8 wh[4], wl[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      wh[j] = *reinterpret_cast<const bf16x8*>(st + offW[j]);
      wl[j] = *reinterpret_cast<const bf16x8*>(st + offW[j] + X3_PLANE);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bf16x8 ah = *reinterpret_cast<const bf16x8*>(st + offA[i]);
      const bf16x8 al = *reinterpret_cast<const bf16x8*>(st + offA[i] + X3_PLANE);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (PLAIN) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[j], ah, acc[i][j], 0, 0, 0);   // k 0..31
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[j], al, acc[i][j], 0, 0, 0);   // k 32..63
        } else {
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[j], ah, acc[i][j], 0, 0, 0);   // small terms first
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[j], al, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[j], ah, acc[i][j], 0, 0, 0);
        }
      }
    }
  }
  // epilogue: two passes of 128 tile rows through a 128 x 256 fp32 LDS tile (pitch +16 B)
  constexpr int PITCH = X3N * 4 + 16;
  const ColParams cp = load_col_params<X3N>(e, n0, N, tid);
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row_l = wm * 32 + ii * 16 + l15;
        const int col = wn * 64 + j * 16 + 4 * l4;
        const f32x4 a = acc[pass * 2 + ii][j];
        *reinterpret_cast<float4*>(smem + row_l * PITCH + col * 4) = make_float4(a[0], a[1], a[2], a[3]);
      }
    __syncthreads();
    drain_tile<128, X3N, 1024>(smem, PITCH, e, cp, M, N, n0, tid,
                               [&](int row_l) { return m0 + (row_l >> 5) * 64 + pass * 32 + (row_l & 31); });
  }
}

static constexpr int LDSX3 = (128 * (X3N * 4 + 16)) > X3_SLOTS * X3_STAGE ? (128 * (X3N * 4 + 16)) : X3_SLOTS * X3_STAGE;

// A2 [M, lda >= 2K], W2 [N, ldw >= 2K]: pair layout; K % 32 == 0
int launch_gemm_x3(const bf16_t* A2, int lda, const bf16_t* W2, int ldw, int M, int N, int K, const GemmEpi& e,
                   hipStream_t s) {
  if (M <= 0 || N <= 0 || K <= 0) return 1;
  if (K % X3K != 0 || N % 4 != 0 || lda % 8 != 0 || ldw % 8 != 0 || lda < 2 * K || ldw < 2 * K) return 2;
  if (e.out_f32 && e.ldc % 4 != 0) return 2;
  if (e.resid && e.ldr % 4 != 0) return 2;
  if (!e.out_f32 && !e.out_bf16) return 2;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_x3_256x256_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDSX3);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_x3_256x256_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDSX3);
    attr_set = true;
  }
  static const char* gme = getenv("DINODET_GEMM_GM");
  const int gm = gme ? atoi(gme) : 4;
  const int tiles = ((M + X3M - 1) / X3M) * ((N + X3N - 1) / X3N);
  hipLaunchKernelGGL(gemm_x3_256x256_kernel<false>, dim3(tiles), dim3(1024), LDSX3, s, A2, lda, W2, ldw, M, N, K, e, gm);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}

// plain bf16 GEMM on the same structure (256x256x64, 16 waves, two 64-KiB slots): K % 64 == 0
int launch_gemm_bf16_k64(const bf16_t* A, int lda, const bf16_t* W, int ldw, int M, int N, int K, const GemmEpi& e, hipStream_t s) {
  if (M <= 0 || N <= 0 || K <= 0 || K % 64 != 0) return 2;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_x3_256x256_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDSX3);
    attr_set = true;
  }
  static const char* gme = getenv("DINODET_GEMM_GM");
  const int gm = gme ? atoi(gme) : 4;
  const int tiles = ((M + X3M - 1) / X3M) * ((N + X3N - 1) / X3N);
  hipLaunchKernelGGL(gemm_x3_256x256_kernel<true>, dim3(tiles), dim3(1024), LDSX3, s, A, lda, W, ldw, M, N, K, e, gm);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}
