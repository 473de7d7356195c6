// bf16 MFMA GEMM with fused epilogues for the DINOv2 blocks (K4, K6, K7, K9, K14 and the patch-embed K1).
//   C[M,N] = A[M,K] * W[N,K]^T   A, W bf16 row-major (K contiguous), fp32 accumulate.
// Reference ops replaced: nn.Linear in Dinov2SelfAttention / Dinov2SelfOutput / Dinov2MLP
// (site-packages/transformers/models/dinov2/modeling_dinov2.py:199-201, 246-252, 281-297),
// LayerScale + residual (:367-370, :377-380), Conv2d patch projection (:139,148).
//
// Tiling (gfx950): 128x128x64 block tile, 256 threads = 4 waves (2x2), each wave a 64x64 sub-tile
// as 2x2 v_mfma_f32_32x32x16_bf16 accumulators (64 acc VGPRs).  The product is computed
// TRANSPOSED (D = W_tile * A_tile^T) so that a lane owns one output row m and its registers
// run along n: 4 consecutive n per register quad -> 8-byte (bf16) / 16-byte (fp32) row-major
// stores and float4 bias/scale/residual reads in the epilogue.
// Staging: global_load_lds_dwordx4 (LDS-DMA, no VGPR round trip): one wave-instruction lands 8 rows x
// 128 B = 1 KiB lane-linear in LDS; the bank swizzle (16-B chunk ^= (row>>1)&7, conflict-free for
// the ds_read_b128 lane groups on 128-B rows) is applied on the per-lane SOURCE address and again on
// the read (an involution), never on the LDS destination.  Two LDS stages (64 KiB, 2 workgroups/CU):
// tile t+1 streams in while tile t feeds the MFMAs; one barrier per K-tile.
// Tile order: XCD-aware (blocks b, b+8, .. share an L2) and grouped 8 m-tiles deep inside each XCD's
// contiguous run, so the ~64 tiles resident on an XCD share 8 A panels and 8 W panels (3 MiB < 4 MiB L2).
#include "dod_common.h"
#include "gemm_epi.h"
#include <atomic>
#include <cstdlib>

#define BM 128
#define BN 128
#define BK 64
#define STAGE_BYTES (2 * BM * BK * 2)   // A + W tile, bf16

__device__ __forceinline__ int swz128(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

// SLOTS: K-tiles of the LDS ring.  2 (64 KiB, two workgroups per CU): tile t+1 streams in while tile t feeds the MFMAs.  3 (96 KiB, one
// workgroup per CU; round 4): TWO tiles in flight behind a counted vmcnt -- for grids of at most one workgroup per CU (the decoder's
// query-side linears, the cut-off last round of a block GEMM), where nothing else on the CU hides the DMA latency and a K step of the
// 2-slot form lasts one memory round trip (~1.1 us for 0.15 us of MFMAs).
template <int LN, int SLOTS>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(const bf16_t* __restrict__ A, int lda,
                                                        const bf16_t* __restrict__ W, int ldw,
                                                        int M, int N, int K, GemmEpi e) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  // XCD-aware remap (bijective for any grid): blocks b, b+8, ... share an XCD/L2; give each XCD a
  // contiguous run of tiles that walk M for a fixed weight panel.
  const int nwg = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  int tm, tn;
  {
    const int GM = 8;
    const int per_group = GM * tiles_n;
    const int grp = bid / per_group, first_m = grp * GM;
    const int gsz = (tiles_m - first_m) < GM ? (tiles_m - first_m) : GM;
    const int in_g = bid - grp * per_group;
    tm = first_m + in_g % gsz;
    tn = in_g / gsz;
  }
  const int m0 = tm * BM, n0 = tn * BN;

  // staging: per K-tile each wave issues 4 LDS-DMA pieces of A and 4 of W (piece = 8 rows x 128 B).
  // lane -> (row = piece*8 + lane>>3, LDS chunk slot = lane&7); it fetches global chunk slot ^ ((row>>1)&7).
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const bf16_t* gA0; const bf16_t* gA1; const bf16_t* gA2; const bf16_t* gA3;
  const bf16_t* gW0; const bf16_t* gW1; const bf16_t* gW2; const bf16_t* gW3;
  {
    auto src = [&](const bf16_t* base, int ld, int r0, int piece, int lim) {
      const int rl = piece * 8 + (lane >> 3);
      int r = r0 + rl; r = r < lim ? r : lim - 1;
      const int c = (lane & 7) ^ ((rl >> 1) & 7);
      return base + (size_t)r * ld + c * 8;
    };
    const int p0 = wid * 4;
    gA0 = src(A, lda, m0, p0, M); gA1 = src(A, lda, m0, p0 + 1, M); gA2 = src(A, lda, m0, p0 + 2, M); gA3 = src(A, lda, m0, p0 + 3, M);
    gW0 = src(W, ldw, n0, p0, N); gW1 = src(W, ldw, n0, p0 + 1, N); gW2 = src(W, ldw, n0, p0 + 2, N); gW3 = src(W, ldw, n0, p0 + 3, N);
  }
  const int wbase = __builtin_amdgcn_readfirstlane(wid) * 4096;     // this wave's 4 pieces, wave-uniform
#define STAGE(stage, k0)                                                                                   \
  {                                                                                                        \
    char* sA_ = smem + (stage) * STAGE_BYTES + wbase;                                                      \
    char* sW_ = sA_ + BM * BK * 2;                                                                         \
    __builtin_amdgcn_global_load_lds((gptr_t)(gA0 + (k0)), (lptr_t)(sA_), 16, 0, 0);                       \
    __builtin_amdgcn_global_load_lds((gptr_t)(gA1 + (k0)), (lptr_t)(sA_ + 1024), 16, 0, 0);                \
    __builtin_amdgcn_global_load_lds((gptr_t)(gA2 + (k0)), (lptr_t)(sA_ + 2048), 16, 0, 0);                \
    __builtin_amdgcn_global_load_lds((gptr_t)(gA3 + (k0)), (lptr_t)(sA_ + 3072), 16, 0, 0);                \
    __builtin_amdgcn_global_load_lds((gptr_t)(gW0 + (k0)), (lptr_t)(sW_), 16, 0, 0);                       \
    __builtin_amdgcn_global_load_lds((gptr_t)(gW1 + (k0)), (lptr_t)(sW_ + 1024), 16, 0, 0);                \
    __builtin_amdgcn_global_load_lds((gptr_t)(gW2 + (k0)), (lptr_t)(sW_ + 2048), 16, 0, 0);                \
    __builtin_amdgcn_global_load_lds((gptr_t)(gW3 + (k0)), (lptr_t)(sW_ + 3072), 16, 0, 0);                \
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = K / BK;
  STAGE(0, 0)
  if (SLOTS > 2 && nk > 1) STAGE(1, BK)
  const int lr = lane & 31, lh = lane >> 5;
  int slot = 0;
  for (int kt = 0; kt < nk; ++kt) {
    // tile kt has landed (8 DMA instructions per wave and tile; SLOTS - 2 younger tiles may stay in flight); the barrier also says every
    // wave has finished reading tile kt - 1, whose slot the next DMA refills
    if (SLOTS > 2 && kt + 1 < nk) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (kt + SLOTS - 1 < nk) {
      const int ns = slot + SLOTS - 1 >= SLOTS ? slot - 1 : slot + SLOTS - 1;
      STAGE(ns, (kt + SLOTS - 1) * BK)
    }
    const char* sA = smem + slot * STAGE_BYTES;
    const char* sW = sA + BM * BK * 2;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      bf16x8 af[2], wf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int rowa = wm * 64 + i * 32 + lr;
        af[i] = *reinterpret_cast<const bf16x8*>(sA + rowa * 128 + swz128(rowa, kk * 2 + lh) * 16);
        const int roww = wn * 64 + i * 32 + lr;
        wf[i] = *reinterpret_cast<const bf16x8*>(sW + roww * 128 + swz128(roww, kk * 2 + lh) * 16);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
    }
    slot = slot + 1 == SLOTS ? 0 : slot + 1;
  }
  __syncthreads();

  // all ring reads are done (last barrier of the loop); reuse the ring as the fp32 output tile
  constexpr int PITCH = BN * 4 + 16;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
      stage_acc(smem, PITCH, wm * 64 + i * 32 + lr, wn * 64 + j * 32, acc[i][j], lh);
  const ColParams cp = load_col_params<BN, LN>(e, n0, N, tid);
  auto rowmap = [&](int row_l) { return m0 + row_l; };
  stage_row_stats<BM, BN, LN>(smem, PITCH, e, M, tid, rowmap, K, n0);
  __syncthreads();
  drain_tile<BM, BN, 256, LN>(smem, PITCH, e, cp, M, N, n0, tid, rowmap);
}

// (The 256x256 8-wave, 256x256x64, 256x128 32x32x16 and 16-wave 256x256x32 variants of rounds 1-2 -- each measured against the kernels
// below and superseded, DESIGN.md section 4 -- were removed in round 4: git history has them.)
#define B4K 32
// ============================================================================================
// Mid variant: 256x128x32 tile, 512 threads = 8 waves (4 along M x 2 along N, 64x64 each), THREE-slot
// LDS-DMA ring of 24-KiB K-tiles (72 KiB) so that TWO workgroups share a CU: one workgroup's prologue
// (first-tile latency) and LDS-staged epilogue (store drain) overlap the other's MFMA loop -- with one
// workgroup per CU those phases run serially and cost ~1/3 of a K=768 GEMM.  Two K-tiles in flight per
// workgroup behind vmcnt(3) (3 pieces per wave per K-tile), one raw barrier per K-tile.
#define B5M 256
#define B5N 128
#define B5_STAGE ((B5M + B5N) * B4K * 2)   // 24 KiB
#define B5_SLOTS 3

#define STAGE5(slot_, k0)                                                                                  \
  {                                                                                                        \
    char* sA_ = smem + (slot_) * B5_STAGE + wu * 2048;                                                     \
    char* sW_ = smem + (slot_) * B5_STAGE + B5M * B4K * 2 + wu * 1024;                                     \
    __builtin_amdgcn_global_load_lds((gptr_t)(gA0 + (k0)), (lptr_t)(sA_), 16, 0, 0);                       \
    __builtin_amdgcn_global_load_lds((gptr_t)(gA1 + (k0)), (lptr_t)(sA_ + 1024), 16, 0, 0);                \
    __builtin_amdgcn_global_load_lds((gptr_t)(gW0 + (k0)), (lptr_t)(sW_), 16, 0, 0);                       \
  }

// ============================================================================================
// 256x128x32 with v_mfma_f32_16x16x32_bf16 (4x4 accumulators of 16x16 per wave) -- same tile, ring and epilogue as
// the 32x32x16 kernel; the chip can hold a higher clock on this shape (cdna guide 5.4 rule 28: build both, keep the
// faster by wall on random data).  Fragment rows are lane&15 and the 16-B k-chunk is lane>>4, so the 64-B-row
// swizzle becomes chunk ^= ((row>>3)&1)<<1 (conflict-free for the ds_read_b128 lane groups with that lane map).
__device__ __forceinline__ int swz64m16(int row, int chunk) { return chunk ^ (((row >> 3) & 1) << 1); }

#ifdef DINODET_TUNING
// tuning builds only (tools/gemm_timeline.py): when non-null, thread 0 of every workgroup stores
// {s_memrealtime at start, after the K loop, at exit (stores drained), blockIdx}
__device__ unsigned long long* g_gemm_stamps = nullptr;
#endif

template <int LN>
__global__ __launch_bounds__(512, 4) void gemm_bf16_256x128_m16_kernel(const bf16_t* __restrict__ A, int lda,
                                                                       const bf16_t* __restrict__ W, int ldw,
                                                                       int M, int N, int K, GemmEpi e, int GM) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int tiles_m = (M + B5M - 1) / B5M, tiles_n = (N + B5N - 1) / B5N;
  const int nwg = tiles_m * tiles_n;
#ifdef DINODET_TUNING
  unsigned long long* stamps = g_gemm_stamps;
  unsigned long long t_start = 0, t_loop = 0;
  if (stamps) t_start = __builtin_amdgcn_s_memrealtime();
#endif
  int tm, tn;
  tile_map(blockIdx.x, tiles_m, tiles_n, GM, &tm, &tn);
  const int m0 = tm * B5M, n0 = tn * B5N;
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const bf16_t* gA0; const bf16_t* gA1; const bf16_t* gW0;
  {
    auto src = [&](const bf16_t* base, int ld, int r0, int piece, int lim) {
      const int rl = piece * 16 + (lane >> 2);
      int r = r0 + rl; r = r < lim ? r : lim - 1;
      const int c = swz64m16(rl, lane & 3);
      return base + (size_t)r * ld + c * 8;
    };
    gA0 = src(A, lda, m0, wid * 2, M); gA1 = src(A, lda, m0, wid * 2 + 1, M);
    gW0 = src(W, ldw, n0, wid, N);
  }
  const int wu = __builtin_amdgcn_readfirstlane(wid);
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nk = K / B4K;
  const int l15 = lane & 15, l4 = lane >> 4;
  STAGE5(0, 0)
  if (nk > 1) STAGE5(1, B4K)
  int offA[4], offW[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { const int row = wm * 64 + i * 16 + l15; offA[i] = row * 64 + swz64m16(row, l4) * 16; }
#pragma unroll
  for (int j = 0; j < 4; ++j) { const int row = wn * 64 + j * 16 + l15; offW[j] = B5M * B4K * 2 + row * 64 + swz64m16(row, l4) * 16; }
  int slot = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (kt + 2 < nk) {
      const int ns = slot >= 1 ? slot - 1 : 2;
      STAGE5(ns, (kt + 2) * B4K)
    }
    const char* st = smem + slot * B5_STAGE;
    bf16x8 af[4], wf[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const bf16x8*>(st + offA[i]);
#pragma unroll
    for (int j = 0; j < 4; ++j) wf[j] = *reinterpret_cast<const bf16x8*>(st + offW[j]);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
    slot = slot == 2 ? 0 : slot + 1;
  }
#ifdef DINODET_TUNING
  if (stamps) t_loop = __builtin_amdgcn_s_memrealtime();
#endif
  // epilogue: D = W A^T: lane&15 = m within the 16-row block, (lane>>4)*4 + reg = n within the 16-col block
  constexpr int PITCH = B5N * 4 + 16;
  const ColParams cp = load_col_params<B5N, LN>(e, n0, N, tid);
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
    auto rowmap = [&](int row_l) { return m0 + (row_l >> 5) * 64 + pass * 32 + (row_l & 31); };
    stage_row_stats<128, B5N, LN>(smem, PITCH, e, M, tid, rowmap, K, n0);
    __syncthreads();
    drain_tile<128, B5N, 512, LN>(smem, PITCH, e, cp, M, N, n0, tid, rowmap);
  }
#ifdef DINODET_TUNING
  if (stamps && tid == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long* o = stamps + (size_t)blockIdx.x * 4;
    o[0] = t_start; o[1] = t_loop; o[2] = __builtin_amdgcn_s_memrealtime(); o[3] = blockIdx.x;
  }
#endif
}

// dynamic LDS: the staging ring, or the padded fp32 epilogue tile if larger
static constexpr int LDS128 = (BM * (BN * 4 + 16)) > 2 * STAGE_BYTES ? (BM * (BN * 4 + 16)) : 2 * STAGE_BYTES;
static constexpr int LDS128R = 3 * STAGE_BYTES;
static constexpr int LDS5 = (128 * (B5N * 4 + 16)) > B5_SLOTS * B5_STAGE ? (128 * (B5N * 4 + 16)) : B5_SLOTS * B5_STAGE;

static std::atomic<long> g_rem_cuts{0};
long gemm_rem_cut_count() { return g_rem_cuts.load(); }

int launch_gemm_bf16(const bf16_t* A, int lda, const bf16_t* W, int ldw, int M, int N, int K,
                     const GemmEpi& e, hipStream_t s) {
  if (M <= 0 || N <= 0 || K <= 0) return 1;
  if (K % BK != 0 || N % 4 != 0 || lda % 8 != 0 || ldw % 8 != 0 || e.ldc % 4 != 0) return 2;
  if (e.resid && e.ldr % 4 != 0) return 2;
  if (!e.out_f32 && !e.out_bf16) return 2;
  static bool attr_set[16] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 16 && !attr_set[dev]) {
#define ATTR_(LN_) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_kernel<LN_, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS128); \
                   (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_kernel<LN_, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS128R); \
                   (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_256x128_m16_kernel<LN_>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS5);
    ATTR_(LN_NONE) ATTR_(LN_CONS) ATTR_(LN_PROD)
#undef ATTR_
    attr_set[dev] = true;
  }
  const char* force = DOD_TUNE_ENV("DINODET_GEMM_TILE");     // tuning builds: "q" ping-pong, "x" 16-wave k64, "8" 256x128, "1" 128x128
  if (!force && M >= 2048 && N >= 512 && e.act != ACT_GELU && e.act != ACT_SIGMOID && e.rows_per_img == 0 && K % 64 == 0) {   // the 256x256-tile shapes
    const int t = gemm_tail_split(0, A, lda, W, ldw, M, N, K, e, s);      // short last round (M >= 8192), or a grid of a few dozen tiles (M >= 2048)
    if (t >= 0) return t;
  }
  if (force && force[0] == 'q' && K % 64 == 0) return launch_gemm_bf16_ppm(A, lda, W, ldw, M, N, K, e, s);
  if (force && force[0] == 'x') return launch_gemm_bf16_k64(A, lda, W, ldw, M, N, K, e, s);
  // grouped-order depth of the 256x128 kernel (m-tiles per group inside an XCD's run), measured: 8 for N = 3072, 4 for 2304, 2 for 768
  const int gm5 = (N >= 3072 ? 8 : (N >= 2048 ? 4 : 2)) | (gemm_tile_mode() & 0x300);
  const int tiles5 = ((M + B5M - 1) / B5M) * ((N + B5N - 1) / B5N);
#define GO_M16(LN_) hipLaunchKernelGGL(gemm_bf16_256x128_m16_kernel<LN_>, dim3(tiles5), dim3(512), LDS5, s, A, lda, W, ldw, M, N, K, e, gm5);
  if (!force && M >= 4096 && N >= 512 && K % 64 == 0 && e.act != ACT_SIGMOID && e.rows_per_img == 0) {
    // Round-aware choice between the 256x256 tiles (one workgroup per CU: the 16-wave `k64` kernel of gemm_x3.hip -- 256x256x64, two 64-KiB
    // slots: QKV 760 / out-proj 460 / fc2 765 TFLOP/s at M = 87 680 -- or the 8-wave ping-pong kernel of gemm_pp.hip, 3-7 % faster on
    // N >= 1536 at every M measured) and the 256x128 tile (`m16`, two co-resident workgroups per CU that run at about half speed each, so
    // one of its tiles costs ~0.55 of a 256x256 tile only while a CU holds a single one).  Cost in units of a 256x256 tile time = rounds
    // of CUs x tile cost; a GELU epilogue costs the one-workgroup-per-CU kernels ~8 % (nothing overlaps it).
    // tools/bench_pp.py --rows {4112, 8224, 16448, 21920, 43840, 87680}: the rule reproduces the faster kernel in 27 of 28 cases
    // (M = 8224, batch 32 at 224x224: QKV 46 vs 53 us, out-proj 27 vs 32, fc1 66 vs 78, fc2 59 vs 75).
    static int cus[16] = {};
    if (dev >= 0 && dev < 16 && !cus[dev]) { int c = 0; (void)hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev); cus[dev] = c > 0 ? c : 256; }
    const int cu = (dev >= 0 && dev < 16) ? cus[dev] : 256;
    const long tm = (M + 255) / 256, t_big = tm * ((N + 255) / 256), t_small = tm * ((N + 127) / 128);
    const double c_big = (double)((t_big + cu - 1) / cu) * (e.act == ACT_GELU ? 1.08 : 1.0);
    const double c_small = (double)((t_small + cu - 1) / cu) * (K >= 2048 ? 0.62 : 0.55);      // long K: the smaller tile's lower FLOP per staged byte shows
    // Round 4: with its epilogue passes written out the 16-wave kernel holds its accumulators in registers (it had spilled 34 of them since
    // round 1) and is the fastest choice for EVERY wide GEMM (N >= 1536: QKV, fc1 incl. its GELU epilogue) at every M measured
    // (tools/bench_pp.py, rows 8 224 / 10 960 / 43 840 / 87 680: QKV 46.9 / 53.2 / 173.0 / 338.2 us against 51.8 / 56.8 / 186.6 / 345.5 for the
    // round-3 choice, fc1 59.9 / 87.7 / 264.2 / 518.9 against 63.8 / 104.7 / 297.9 / 548.4); the narrow ones (N = 768) keep the round rule
    // below, and the long-K narrow one (fc2) the ping-pong kernel behind its tail split.
    if (N >= 1536 && K < 2048) {
      // A last round of a few tiles costs these short-K GEMMs a whole tile time when the grid is only one or two rounds deep (8 images of
      // 518^2: fc1 = 43 x 12 = 516 tiles = two rounds + 4; 32 images of 224^2: QKV = 33 x 9 = 297 = one round + 41).  The m-tiles of that
      // round are cut off and run as a launch of their own, which the small-M rules below spread over the chip as 128-row tiles.
      const long rounds = t_big / cu, rem = t_big % cu;
      static const int rem_div = [] { const char* v = DOD_TUNE_ENV("DINODET_GEMM_REMCUT"); return v ? atoi(v) : 6; }();      // 0: off
      // (rem >= CUs / 16: a handful of lone tiles finish at well under a tile time -- and a second stream's kernels fill the idle CUs -- so cutting
      // them costs more than it saves: 2 x 4 images of 518^2, fc1 = 264 tiles, 1 737 against 1 794 images/s with the cut)
      if (rounds >= 1 && rounds <= 2 && rem * 16 >= cu && rem_div > 0 && rem * rem_div <= cu && gemm_epi_rows_ok(e)) {
        const int tn = (N + 255) / 256;
        const int Mmain = (int)((rounds * cu) / tn) * 256, R = M - Mmain;
        if (Mmain >= 2048 && R > 0 && R < 2048) {
          const int rc = launch_gemm_bf16_k64(A, lda, W, ldw, Mmain, N, K, e, s);
          if (rc) return rc;
          ++g_rem_cuts;
          return launch_gemm_bf16(A + (size_t)Mmain * lda, lda, W, ldw, R, N, K, gemm_epi_rows(e, Mmain), s);
        }
      }
      return launch_gemm_bf16_k64(A, lda, W, ldw, M, N, K, e, s);
    }
    static const double m16_bias = [] { const char* v = DOD_TUNE_ENV("DINODET_GEMM_M16_BIAS"); return v ? atof(v) : 1.02; }();      // tuning builds: > 1 favours the 256x256 tiles
    if (c_big <= c_small * m16_bias) {
      if (K >= 2048) return launch_gemm_bf16_ppm(A, lda, W, ldw, M, N, K, e, s);
      return launch_gemm_bf16_k64(A, lda, W, ldw, M, N, K, e, s);
    }
    LN_DISPATCH(e, GO_M16)
    return hipGetLastError() == hipSuccess ? 0 : 3;
  }
  // Below that: 256x128x32 on v_mfma_f32_16x16x32_bf16 (two workgroups per CU) from 1 024 rows up, 128x128 for small M (decoder memory at small
  // batch, tests) -- and for a grid of a few dozen 256x128 tiles (the decoder's query-side linears: B*Q = 3 200 rows at batch 32 -> 78 workgroups
  // walking K' = 3K = 2 304), which leaves most CUs idle while one workgroup per CU pulls its operands at the per-CU staging rate whatever its
  // tile: the 128x128 kernel puts the same bytes on twice as many CUs (tools/bench_small_m.py, M = 3200: N = 768 41.5 -> 30.9 us, N = 1024
  // 42.7 -> 31.9, K = 3072 53.9 -> 38.7; N = 2304 (234 tiles) 47.8 vs 48.4: stays).
  const bool few = (long)tiles5 * 2 <= 256 + 64;
  const bool m16 = force ? (force[0] == '8') : (M >= 1024 && N >= 128 && !few);
  if (m16) {
    LN_DISPATCH(e, GO_M16)
  } else {
    const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    // a grid of at most one workgroup per CU has nothing beside it to hide its DMA latency: the 3-slot ring (two K-tiles in flight)
    static const int ring = [] { const char* v = DOD_TUNE_ENV("DINODET_GEMM_RING"); return v ? atoi(v) : 1; }();
    int cu128 = 256;
    { static int cus[16] = {}; if (dev >= 0 && dev < 16) { if (!cus[dev]) { int c = 0; (void)hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev); cus[dev] = c > 0 ? c : 256; } cu128 = cus[dev]; } }
#define GO_128(LN_) hipLaunchKernelGGL((gemm_bf16_kernel<LN_, 2>), dim3(tiles), dim3(256), LDS128, s, A, lda, W, ldw, M, N, K, e);
#define GO_128R(LN_) hipLaunchKernelGGL((gemm_bf16_kernel<LN_, 3>), dim3(tiles), dim3(256), LDS128R, s, A, lda, W, ldw, M, N, K, e);
    if (ring && tiles <= cu128 && K >= 3 * BK) { LN_DISPATCH(e, GO_128R) }
    else { LN_DISPATCH(e, GO_128) }
#undef GO_128
#undef GO_128R
  }
  return hipGetLastError() == hipSuccess ? 0 : 3;
}
#undef GO_M16

#ifdef DINODET_TUNING
extern "C" int dod_debug_gemm_stamps(void* dev_buf) {
  unsigned long long* p = (unsigned long long*)dev_buf;
  return hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_stamps), &p, sizeof(p)) == hipSuccess ? 0 : 4;
}

// ---- tuning only (tools/mfma_peak.py): register-only MFMA loop, no memory traffic -----------------------------
// Each wave runs `iters` rounds of 8 independent v_mfma_f32_16x16x32_bf16 (or 4 x 32x32x16) accumulator chains.
// out[block*4 + {0,1,2}] = {s_memtime cycles of wave 0, s_memrealtime ticks (100 MHz) of wave 0, checksum}.
template <int SHAPE>
__global__ __launch_bounds__(256, 2) void mfma_peak_kernel(int iters, unsigned long long* out) {
  bf16x8 a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = (bf16_t)(0x3c00 + threadIdx.x + i); b[i] = (bf16_t)(0x3c10 + threadIdx.x * 3 + i); }
  if (iters < 0) {   // negative iteration count: random sign / mantissa operands (+-0.5..1): the datapath toggles as on real data
    iters = -iters;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const unsigned r1 = (threadIdx.x * 8u + i + 1u) * 2654435761u, r2 = (threadIdx.x * 8u + i + 7u) * 40503u * 2246822519u;
      a[i] = (bf16_t)(0x3f00u | ((r1 >> 9) & 0xffu) | (((r1 >> 20) & 1u) << 15));
      b[i] = (bf16_t)(0x3f00u | ((r2 >> 9) & 0xffu) | (((r2 >> 20) & 1u) << 15));
    }
  }
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  float chk = 0.f;
  if (SHAPE == 16) {
    f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) chk += acc[i][0];
  } else if (SHAPE == 1) {   // v_mfma_f32_32x32x2_f32, ONE dependent chain per wave (what a single-accumulator kernel issues)
    f32x16 acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc1[r] = 0.f;
    const float fa = bf2f(a[0]) * (1.0f + 0.37f * bf2f(a[1])), fb = bf2f(b[0]) * (1.0f + 0.41f * bf2f(b[1]));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc1, 0, 0, 0);
    }
    chk += acc1[0];
  } else if (SHAPE == 2) {   // v_mfma_f32_32x32x2_f32 (exact fp32), 4 chains
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const float fa = bf2f(a[0]) * (1.0f + 0.37f * bf2f(a[1])), fb = bf2f(b[0]) * (1.0f + 0.41f * bf2f(b[1]));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) chk += acc[i][0];
  } else {
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) chk += acc[i][0];
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) {
    out[(size_t)blockIdx.x * 4 + 0] = c1 - c0;
    out[(size_t)blockIdx.x * 4 + 1] = r1 - r0;
    out[(size_t)blockIdx.x * 4 + 2] = (unsigned long long)__float_as_uint(chk);
  }
}
extern "C" int dod_debug_mfma_peak(int shape, int iters, int blocks, void* dev_out, void* stream) {
  if (!dev_out || iters == 0 || blocks <= 0 || (shape != 16 && shape != 32 && shape != 2 && shape != 1)) return 1;
  if (shape == 1) hipLaunchKernelGGL(mfma_peak_kernel<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, iters, (unsigned long long*)dev_out);
  else if (shape == 2) hipLaunchKernelGGL(mfma_peak_kernel<2>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, iters, (unsigned long long*)dev_out);
  else if (shape == 16) hipLaunchKernelGGL(mfma_peak_kernel<16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, iters, (unsigned long long*)dev_out);
  else hipLaunchKernelGGL(mfma_peak_kernel<32>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, iters, (unsigned long long*)dev_out);
  return hipGetLastError() == hipSuccess ? 0 : 4;
}

// MFMA / VALU co-issue probe (tools/mfma_peak.py): per iteration 4 independent v_mfma_f32_32x32x16_bf16 (mode & 1) and
// `nvalu` independent v_fma_f32 chains stepped once each (mode & 2), interleaved one MFMA : nvalu/4 VALU by the asm order.
template <int NV, int mode>
__global__ __launch_bounds__(256, 2) void mfma_valu_probe_kernel(int iters, unsigned long long* out) {
  bf16x8 a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = (bf16_t)(0x3c00 + threadIdx.x + i); b[i] = (bf16_t)(0x3c10 + threadIdx.x * 3 + i); }
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float v[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = 1.0f + 0.001f * (threadIdx.x + i);
  const float m = 0.999f, c = 0.001f;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (mode & 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[q]) : "v"(a), "v"(b));
      if (mode & 2) {
#pragma unroll
        for (int i = 0; i < NV / 4; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[q * (NV / 4) + i]) : "v"(m), "v"(c));
      }
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  float chk = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) chk += acc[i][0];
#pragma unroll
  for (int i = 0; i < NV; ++i) chk += v[i];
  if (threadIdx.x == 0) { out[(size_t)blockIdx.x * 4] = c1 - c0; out[(size_t)blockIdx.x * 4 + 2] = (unsigned long long)__float_as_uint(chk); }
}
template <int NV>
static int probe_launch(int mode, int iters, int blocks, void* dev_out, void* stream) {
  unsigned long long* o = (unsigned long long*)dev_out;
  if (mode == 1) hipLaunchKernelGGL((mfma_valu_probe_kernel<NV, 1>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, iters, o);
  else if (mode == 2) hipLaunchKernelGGL((mfma_valu_probe_kernel<NV, 2>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, iters, o);
  else if (mode == 3) hipLaunchKernelGGL((mfma_valu_probe_kernel<NV, 3>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, iters, o);
  else return 1;
  return hipGetLastError() == hipSuccess ? 0 : 4;
}
extern "C" int dod_debug_mfma_valu_probe(int nvalu, int mode, int iters, int blocks, void* dev_out, void* stream) {
  if (!dev_out || iters <= 0 || blocks <= 0) return 1;
  if (nvalu == 16) return probe_launch<16>(mode, iters, blocks, dev_out, stream);
  if (nvalu == 28) return probe_launch<28>(mode, iters, blocks, dev_out, stream);
  if (nvalu == 56) return probe_launch<56>(mode, iters, blocks, dev_out, stream);
  return 1;
}
#endif  // DINODET_TUNING
