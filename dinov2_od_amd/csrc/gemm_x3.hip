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
#include <cstdio>
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
template <bool PLAIN, int LN>
__global__ __launch_bounds__(1024) void gemm_x3_256x256_kernel(const bf16_t* __restrict__ A2, int lda,
                                                               const bf16_t* __restrict__ W2, int ldw, int M, int N,
                                                               int K, GemmEpi e, int GM) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 2, wn = wid & 3;
  const int tiles_m = (M + X3M - 1) / X3M, tiles_n = (N + X3N - 1) / X3N;
  int tm, tn;
  tile_map(blockIdx.x, tiles_m, tiles_n, GM, &tm, &tn);
  const int m0 = tm * X3M, n0 = tn * X3N;
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  // per K-tile a wave issues ONE 16-row piece of each plane (16 waves x 16 rows = 256), through buffer descriptors: the base in
  // SGPRs, one 32-bit byte offset per lane computed once, the K position as the scalar offset (no per-DMA 64-bit address math)
  unsigned vA, vW;
  {
    const int rl = wid * 16 + (lane >> 2);
    const int c = x3_swz(rl, lane & 3);
    int ra = m0 + rl; ra = ra < M ? ra : M - 1;
    int rw = n0 + rl; rw = rw < N ? rw : N - 1;
    vA = (unsigned)(((size_t)ra * lda + c * 8) * 2);
    vW = (unsigned)(((size_t)rw * ldw + c * 8) * 2);
  }
  const size_t bytesA = (size_t)M * lda * 2, bytesW = (size_t)N * ldw * 2;
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)A2, 0, (int)(bytesA > 0xfffffff0u ? 0xfffffff0u : bytesA), 0x00020000);
  const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)W2, 0, (int)(bytesW > 0xfffffff0u ? 0xfffffff0u : bytesW), 0x00020000);
  const int wu = __builtin_amdgcn_readfirstlane(wid);
  const int pl1b = (PLAIN ? 32 : K) * 2;
#define STAGE_X3(slot_, k0)                                                                                \
  {                                                                                                        \
    char* s_ = smem + (slot_) * X3_STAGE + wu * 1024;                                                      \
    const int kb_ = (k0) * 2;                                                                              \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lptr_t)(s_), 16, vA, kb_, 0, 0);                         \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lptr_t)(s_ + X3_PLANE), 16, vA, kb_ + pl1b, 0, 0);       \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lptr_t)(s_ + 2 * X3_PLANE), 16, vW, kb_, 0, 0);          \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lptr_t)(s_ + 3 * X3_PLANE), 16, vW, kb_ + pl1b, 0, 0);   \
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
  const ColParams cp = load_col_params<X3N, LN>(e, n0, N, tid);
  const bool wide = drain8_ok(e, N);
  // the two passes written out (as a `#pragma unroll` loop the body outgrows the unroller's threshold and the accumulators spill)
#define X3_PASS(P)                                                                                                          \
  {                                                                                                                         \
    __builtin_amdgcn_s_barrier();                                                                                           \
    asm volatile("" ::: "memory");                                                                                          \
    _Pragma("unroll") for (int ii = 0; ii < 2; ++ii)                                                                        \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                       \
        const int row_l = wm * 32 + ii * 16 + l15;                                                                          \
        const int col = wn * 64 + j * 16 + 4 * l4;                                                                          \
        const f32x4 a = acc[(P) * 2 + ii][j];                                                                               \
        *reinterpret_cast<float4*>(smem + row_l * PITCH + col * 4) = make_float4(a[0], a[1], a[2], a[3]);                   \
      }                                                                                                                     \
    auto rowmap = [&](int row_l) { return m0 + (row_l >> 5) * 64 + (P) * 32 + (row_l & 31); };                              \
    stage_row_stats<128, X3N, LN>(smem, PITCH, e, M, tid, rowmap, K, n0);                                                          \
    __syncthreads();                                                                                                        \
    if (wide) drain_tile_bf16x8<128, X3N, 1024, LN>(smem, PITCH, e, M, N, n0, tid, rowmap);                                 \
    else drain_tile<128, X3N, 1024, LN>(smem, PITCH, e, cp, M, N, n0, tid, rowmap);                                         \
  }
  X3_PASS(0)
  X3_PASS(1)
#undef X3_PASS
}

// tile-order mode of the 256x256 kernels (gemm_epi.h tile_map): group depth 4, time-ordered ("chunked") map.  Tuning builds: DINODET_GEMM_GM =
// group depth, DINODET_GEMM_ORDER = bit 0 reverse, bit 1 chunked, DINODET_GEMM_STAGGER = "groups,step_us" (read per call)
int gemm_tile_mode() {
  const char* g = DOD_TUNE_ENV("DINODET_GEMM_GM");
  const char* o = DOD_TUNE_ENV("DINODET_GEMM_ORDER");
  int gm = g ? atoi(g) : 4;
  if (gm < 1 || gm > 64) gm = 4;
  const int ord = o ? atoi(o) : 2;        // default: time-ordered map (measured +5 % on QKV / fc2, +3-5 % on out-proj at M = 87680)
  int st = 0;
  if (const char* sg = DOD_TUNE_ENV("DINODET_GEMM_STAGGER")) {     // "groups,step_us" (tuning switch)
    int G = 0; float us = 0.f;
    if (sscanf(sg, "%d,%f", &G, &us) == 2 && G >= 2 && G <= 16 && us > 0.f) {
      int step = (int)(us / 0.16f + 0.5f); step = step < 1 ? 1 : (step > 0xfff ? 0xfff : step);
      st = ((G - 1) << 12) | (step << 16);
    }
  }
  return gm | ((ord & 1) ? 0x100 : 0) | ((ord & 2) ? 0x200 : 0) | st;
}

static constexpr int LDSX3 = (128 * (X3N * 4 + 16)) > X3_SLOTS * X3_STAGE ? (128 * (X3N * 4 + 16)) : X3_SLOTS * X3_STAGE;

static void x3_attr() {      // > 64 KiB of dynamic LDS: once per device
  static bool attr_set[16] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 16 && !attr_set[dev]) {
#define ATTR_(LN_) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_x3_256x256_kernel<false, LN_>), hipFuncAttributeMaxDynamicSharedMemorySize, LDSX3); \
                   (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_x3_256x256_kernel<true, LN_>), hipFuncAttributeMaxDynamicSharedMemorySize, LDSX3);
    ATTR_(LN_NONE) ATTR_(LN_CONS) ATTR_(LN_PROD)
#undef ATTR_
    attr_set[dev] = true;
  }
}

// A2 [M, lda >= 2K], W2 [N, ldw >= 2K]: pair layout; K % 32 == 0
int launch_gemm_x3(const bf16_t* A2, int lda, const bf16_t* W2, int ldw, int M, int N, int K, const GemmEpi& e,
                   hipStream_t s) {
  if (M <= 0 || N <= 0 || K <= 0) return 1;
  if (K % X3K != 0 || N % 4 != 0 || lda % 8 != 0 || ldw % 8 != 0 || lda < 2 * K || ldw < 2 * K) return 2;
  if (e.out_f32 && e.ldc % 4 != 0) return 2;
  if (e.resid && e.ldr % 4 != 0) return 2;
  if (!e.out_f32 && !e.out_bf16) return 2;
  { const int t = gemm_tail_split(1, A2, lda, W2, ldw, M, N, K, e, s); if (t >= 0) return t; }
  {
    // the 8-wave ping-pong kernel (gemm_pp.hip) wherever a workgroup's K loop or column count is long enough to pay for its
    // 512-thread epilogue: measured at M = 87680 against the 16-wave kernel below -- QKV 400 vs 378, fc1 376 vs 363, fc2 406 vs 371
    // TFLOP/s algorithmic; out-proj (N = K = 768) 276 vs 293, so that one stays.  (tuning builds: DINODET_X3_TILE = p / w forces either)
    const char* v = DOD_TUNE_ENV("DINODET_X3_TILE");
    // (round 4, after both kernels' epilogues were rewritten: the ping-pong kernel wins out-proj too -- 337.6 vs 345.1 us at M = 87 680,
    // 183.0 vs 185.8 at 43 840, 58.0 vs 59.6 at 10 960 -- so every large-M split product takes it)
    const bool pp = v ? v[0] == 'p' : (M >= 4096);
    if (pp) return launch_gemm_x3_pp(A2, lda, W2, ldw, M, N, K, e, s);
  }
  x3_attr();
  const int gm = gemm_tile_mode();
  const int tiles = ((M + X3M - 1) / X3M) * ((N + X3N - 1) / X3N);
#define GO_(LN_) hipLaunchKernelGGL((gemm_x3_256x256_kernel<false, LN_>), dim3(tiles), dim3(1024), LDSX3, s, A2, lda, W2, ldw, M, N, K, e, gm);
  LN_DISPATCH(e, GO_)
#undef GO_
  return hipGetLastError() == hipSuccess ? 0 : 3;
}

// plain bf16 GEMM on the same structure (256x256x64, 16 waves, two 64-KiB slots): K % 64 == 0
int launch_gemm_bf16_k64(const bf16_t* A, int lda, const bf16_t* W, int ldw, int M, int N, int K, const GemmEpi& e, hipStream_t s) {
  if (M <= 0 || N <= 0 || K <= 0 || K % 64 != 0) return 2;
  x3_attr();
  const int gm = gemm_tile_mode();
  const int tiles = ((M + X3M - 1) / X3M) * ((N + X3N - 1) / X3N);
#define GO_(LN_) hipLaunchKernelGGL((gemm_x3_256x256_kernel<true, LN_>), dim3(tiles), dim3(1024), LDSX3, s, A, lda, W, ldw, M, N, K, e, gm);
  LN_DISPATCH(e, GO_)
#undef GO_
  return hipGetLastError() == hipSuccess ? 0 : 3;
}
