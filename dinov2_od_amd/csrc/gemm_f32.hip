// Exact-fp32 MFMA GEMM (v_mfma_f32_32x32x2_f32: bit-for-bit a k-ordered fmaf chain) with the same
// fused epilogues as the bf16 kernel.  Used for (a) the strict-parity mode of the backbone, the mode
// gated at 1e-3 against the reference's fp32 CPU forward, (b) every decoder/head linear in both
// modes (K11-K13, K17-K19: deformable_attention.py:86-94,181,232-238,264-266; detr_decoder.py:80-81),
// whose sampling-coordinate math must stay fp32 (floor() at deformable_attention.py:114-115), and
// (c) every product of the native training step (dec_train.hip; train.py:1079-1109).
//   C[M,N] = A[M,K] * W[N,K]^T, arbitrary M, N, K (guarded).
// Tile 64x64x16, 256 threads = 4 waves (2x2), one 32x32 accumulator per wave, computed transposed
// (D = W_tile * A_tile^T) so a lane owns an output row and register quads run along n.
// LDS holds both operands k-major ([k][row], stride 66 floats) so the one-float-per-lane MFMA operands
// are conflict-free ds_read_b32 (lanes 0-31 consecutive rows at k, lanes 32-63 at k+1).
// Global loads run a four-deep register ring ahead of the LDS stage.  Measured on the training step's products (ViT-B 224x224,
// batch 16): 4 112 x 3 072 x 768 (3 120 workgroups) 91 TFLOP/s = 58 % of the fp32 MFMA peak; one workgroup per CU (1 600 x 768 x
// 768, 300 workgroups) 40 TFLOP/s with or without the ring, and slower (51 vs 48 us; the large grids 255 vs 212 us) with four
// k-tiles per LDS stage and barrier -- a lone workgroup's k-tile takes ~1 us whatever is in flight behind it, co-resident
// workgroups are what overlap it.
#include "dod_common.h"
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>

#define FBM 64
#define FBN 64
#define FBK 16
#define FLD 66   // 4*FLD == 8 (mod 32): the transposing ds_write_b32 pattern below is conflict-free
#define FPD 4    // k-tiles in flight per thread (register ring)

namespace {

// One operand's 64-row x 16-k tile: global -> registers -> LDS [k][row].
//   KM = false: stored [rows, K] (k contiguous): thread -> (row = tid / 4, k = 4 * (tid % 4)), transposing LDS write
//   KM = true : stored [K, rows] (rows contiguous): thread -> (k = tid / 16, row = 4 * (tid % 16)), straight LDS write
// VEC: base, pitch and batch strides allow aligned float4 loads (the pitch then covers a partial last quad); otherwise four scalar
// loads.  Both forms are branch-free -- clamped address, value selected to zero -- so the loads of the ring stay in flight
// across iterations (a divergent branch around a load makes the compiler drain vmcnt at the join).
template <bool KM, bool VEC>
struct TileLoad {
  const float* base; int ld, rows, kend, r0; int a, b;
  __device__ __forceinline__ void init(const float* base_, int ld_, int rows_, int kend_, int r0_, int tid) {
    base = base_; ld = ld_; rows = rows_; kend = kend_; r0 = r0_;
    if (KM) { a = tid >> 4; b = (tid & 15) * 4; } else { a = tid >> 2; b = (tid & 3) * 4; }
  }
  __device__ __forceinline__ float4 load(int k0) const {
    // (o, c): the strided and the contiguous coordinate of this thread's quad; (on, cn): their extents
    const int o = KM ? k0 + a : r0 + a, c = KM ? r0 + b : k0 + b;
    const int on = KM ? kend : rows, cn = KM ? rows : kend;
    const bool oin = o < on;
    const float* row = base + (size_t)(oin ? o : 0) * ld;
    float4 v;
    if (VEC) {
      v = *reinterpret_cast<const float4*>(row + (c < cn ? c : 0));
    } else {
      v.x = row[c < cn ? c : 0]; v.y = row[c + 1 < cn ? c + 1 : 0]; v.z = row[c + 2 < cn ? c + 2 : 0]; v.w = row[c + 3 < cn ? c + 3 : 0];
    }
    return v;               // unmasked: store() applies the guards, so nothing consumes the load before the LDS stage needs it
  }
  __device__ __forceinline__ void store(float* s, float4 v, int k0) const {
    const int o = KM ? k0 + a : r0 + a, c = KM ? r0 + b : k0 + b;
    const int on = KM ? kend : rows, cn = KM ? rows : kend;
    const bool oin = o < on;
    v.x = oin && c < cn ? v.x : 0.f; v.y = oin && c + 1 < cn ? v.y : 0.f; v.z = oin && c + 2 < cn ? v.z : 0.f; v.w = oin && c + 3 < cn ? v.w : 0.f;
    if (KM) {
      float* d = s + a * FLD + b;                    // 8-byte aligned (FLD even, b % 4 == 0)
      *reinterpret_cast<float2*>(d) = make_float2(v.x, v.y);
      *reinterpret_cast<float2*>(d + 2) = make_float2(v.z, v.w);
    } else {
      float* d = s + b * FLD + a;
      d[0] = v.x; d[FLD] = v.y; d[2 * FLD] = v.z; d[3 * FLD] = v.w;
    }
  }
};

// acc (32x32 per wave, transposed) over the k-tiles [kt0, kt1) of the (m0, n0) tile
template <class LA, class LW>
__device__ __forceinline__ void f32_mainloop(const LA& la, const LW& lw, int kt0, int kt1,
                                             float (*sA)[FBK * FLD], float (*sW)[FBK * FLD], f32x16& acc) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int lr = lane & 31, lh = lane >> 5;
  const int nk = kt1 - kt0;
  float4 ra[FPD], rw[FPD];
  // Tiles past the end load a clamped address and are zeroed at the LDS store: the ring's loop body has no branch around a load,
  // which is what lets the compiler count vmcnt down instead of draining it (checked in the ISA: vmcnt(6) before each LDS store).
#pragma unroll
  for (int p = 0; p < FPD; ++p) { ra[p] = la.load((kt0 + p) * FBK); rw[p] = lw.load((kt0 + p) * FBK); }
  la.store(sA[0], ra[0], kt0 * FBK);
  lw.store(sW[0], rw[0], kt0 * FBK);
  __syncthreads();
  auto mma = [&](int st) {
    const float* a = &sA[st][wm * 32 + lr];
    const float* w = &sW[st][wn * 32 + lr];
#pragma unroll
    for (int s = 0; s < FBK / 2; ++s) {
      const float av = a[(2 * s + lh) * FLD];
      const float wv = w[(2 * s + lh) * FLD];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv, av, acc, 0, 0, 0);
    }
  };
  int kt = 0;
  for (; kt + FPD <= nk; kt += FPD) {             // full groups: tile kt + j sits in LDS stage j & 1 and leaves ring slot j free
#pragma unroll
    for (int j = 0; j < FPD; ++j) {
      ra[j] = la.load((kt0 + kt + j + FPD) * FBK);
      rw[j] = lw.load((kt0 + kt + j + FPD) * FBK);
      mma(j & 1);
      la.store(sA[(j + 1) & 1], ra[(j + 1) % FPD], (kt0 + kt + j + 1) * FBK);
      lw.store(sW[(j + 1) & 1], rw[(j + 1) % FPD], (kt0 + kt + j + 1) * FBK);
      __syncthreads();
    }
  }
#pragma unroll
  for (int j = 0; j < FPD - 1; ++j) {             // the last nk % FPD tiles are already in the ring
    if (kt + j < nk) {                            // uniform
      mma(j & 1);
      if (kt + j + 1 < nk) {
        la.store(sA[(j + 1) & 1], ra[j + 1], (kt0 + kt + j + 1) * FBK);
        lw.store(sW[(j + 1) & 1], rw[j + 1], (kt0 + kt + j + 1) * FBK);
      }
      __syncthreads();
    }
  }
}

inline bool vec_ok(const void* p, long long ld, long long s1 = 0, long long s2 = 0) {
  return ((uintptr_t)p & 15) == 0 && ld >= 4 && ld % 4 == 0 && s1 % 4 == 0 && s2 % 4 == 0;
}

}  // namespace

// K split across workgroups (round 4).  A 64x64 tile walking K = 768 is 384 dependent v_mfma_f32_32x32x2_f32 per wave -- 24.6 k matrix-pipe cycles,
// 13-16 us at the clock these kernels hold, whatever else happens -- and the decoder's small linears ([B*Q, 768] x [768 | 50 | 91, 768]^T at
// B*Q = 400 .. 3 200 rows) put 2-156 such tiles on 256 CUs.  grid.y = S slices of the k-tiles; every slice stores its 64x64 partial
// ([slice][tile][16 registers][256 threads]: coalesced), the LAST workgroup to arrive at a tile (one atomic counter per tile, reset by that
// workgroup) sums the S partials IN SLICE ORDER -- so the result does not depend on which one was last -- and runs the epilogue.
// S is a function of (N, K) alone (launch_gemm_f32): a row's bits do not depend on the batch it is computed in.
template <bool VEC>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, int lda,
                                                       const float* __restrict__ W, int ldw,
                                                       int M, int N, int K, GemmEpi e, float* __restrict__ part, unsigned* __restrict__ counters) {
  __shared__ __attribute__((aligned(16))) float f32_smem[4 * FBK * FLD];     // sA[2] | sW[2]; the split-K epilogue's [64][65] tile
  __shared__ unsigned s_last;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int tiles_m = (M + FBM - 1) / FBM;
  const int tn = blockIdx.x / tiles_m, tm = blockIdx.x - tn * tiles_m;
  const int m0 = tm * FBM, n0 = tn * FBN;
  TileLoad<false, VEC> la, lw;
  la.init(A, lda, M, K, m0, tid);
  lw.init(W, ldw, N, K, n0, tid);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int S = gridDim.y, nk = (K + FBK - 1) / FBK;
  const int kt0 = (int)((long long)blockIdx.y * nk / S), kt1 = (int)((long long)(blockIdx.y + 1) * nk / S);
  f32_mainloop(la, lw, kt0, kt1, reinterpret_cast<float (*)[FBK * FLD]>(f32_smem),
               reinterpret_cast<float (*)[FBK * FLD]>(f32_smem + 2 * FBK * FLD), acc);
  if (S > 1) {
    const size_t tiles = gridDim.x;
    float* mine = part + ((size_t)blockIdx.y * tiles + blockIdx.x) * (16 * 256) + tid;
    // No device-scope FENCE anywhere here: on this chip (eight XCDs, an L2 each) `__threadfence()` writes back and invalidates the whole L2 of the
    // XCD -- inside a forward whose L2s are full of dirty activations that cost 17 % of the 8-image step.  The partials instead travel as
    // agent-scope atomic stores / loads (sc1: written through to, and read from, the level all XCDs share); a wave's stores have completed
    // there when its vmcnt reaches 0, which every wave waits for before the barrier that precedes the tile's counter increment.
#pragma unroll
    for (int r = 0; r < 16; ++r) __hip_atomic_store(mine + r * 256, acc[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      const unsigned old = __hip_atomic_fetch_add(&counters[blockIdx.x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_last = old == (unsigned)(S - 1);
      if (old == (unsigned)(S - 1)) __hip_atomic_store(&counters[blockIdx.x], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // every slice counted: ready for the next launch on this stream
    }
    __syncthreads();
    if (!s_last) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int sl = 0; sl < S; ++sl) {
      const float* p = part + ((size_t)sl * tiles + blockIdx.x) * (16 * 256) + tid;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] += __hip_atomic_load(p + r * 256, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }

  const int lr = lane & 31, lh = lane >> 5;
  const int m = m0 + wm * 32 + lr;
  if (m >= M) return;
  size_t orow = (size_t)m;
  const float* posrow = nullptr;
  if (e.rows_per_img > 0) {
    const int b = m / e.rows_per_img, p = m - b * e.rows_per_img;
    orow = (size_t)b * e.out_rows_per_img + 1 + p;
    posrow = e.pos + (size_t)(1 + p) * N;
  }
#pragma unroll
  for (int g = 0; g < 4; ++g) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int n = n0 + wn * 32 + 8 * g + 4 * lh + t;
      if (n >= N) continue;
      float v = acc[4 * g + t];
      if (e.bias) v += e.bias[n];
      if (e.act == ACT_GELU) v = gelu_erf(v);
      else if (e.act == ACT_RELU) v = fmaxf(v, 0.f);
      else if (e.act == ACT_SIGMOID) v = sigmoidf_(v);
      if (e.scale) v *= e.scale[n];
      if (posrow) v += posrow[n];
      if (e.resid) v += e.resid[orow * e.ldr + n];
      if (e.out_f32) e.out_f32[orow * e.ldc + n] = v;
      else e.out_bf16[orow * e.ldc + n] = f2bf(v);
    }
  }
}

// ---- scratch of the K split: F32K_SLOTS slabs per device, one per launching stream (least recently used first, as gemm_pp.hip's tail scratch:
// a forward may run as concurrent micro-batches on separate streams); a slab = F32K_TILES counters (zero between launches) + partials of at most
// F32K_PARTS (tile, slice) pairs.  Allocated by gemm_f32_ksplit_reserve() outside any stream capture, never freed (captured graphs keep the addresses).
#define F32K_SLOTS 4
#define F32K_TILES 4096
#define F32K_PARTS 1536
static std::mutex g_f32k_mu;
static char* g_f32k[16] = {};
static hipStream_t g_f32k_owner[16][F32K_SLOTS] = {};
static unsigned long long g_f32k_used[16][F32K_SLOTS] = {};
static unsigned long long g_f32k_clock = 0;
static constexpr size_t F32K_SLAB = (size_t)F32K_TILES * 4 + (size_t)F32K_PARTS * 16 * 256 * 4;
int gemm_f32_ksplit_reserve() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 3;
  std::lock_guard<std::mutex> lk(g_f32k_mu);
  if (g_f32k[dev]) return 0;
  char* blk = nullptr;
  if (hipMalloc((void**)&blk, F32K_SLAB * F32K_SLOTS) != hipSuccess) return 3;
  if (hipMemset(blk, 0, F32K_SLAB * F32K_SLOTS) != hipSuccess) { (void)hipFree(blk); return 3; }
  g_f32k[dev] = blk;
  return 0;
}
static char* f32k_slab(int dev, hipStream_t s) {
  std::lock_guard<std::mutex> lk(g_f32k_mu);
  if (dev < 0 || dev >= 16 || !g_f32k[dev]) return nullptr;
  int pick = -1;
  for (int i = 0; i < F32K_SLOTS; ++i)
    if (g_f32k_used[dev][i] && g_f32k_owner[dev][i] == s) { pick = i; break; }
  if (pick < 0) {
    pick = 0;
    for (int i = 1; i < F32K_SLOTS; ++i)
      if (g_f32k_used[dev][i] < g_f32k_used[dev][pick]) pick = i;
    g_f32k_owner[dev][pick] = s;
  }
  g_f32k_used[dev][pick] = ++g_f32k_clock;
  return g_f32k[dev] + (size_t)pick * F32K_SLAB;
}
static std::atomic<long> g_f32k_count{0};
long gemm_f32_ksplit_count() { return g_f32k_count.load(); }

int launch_gemm_f32(const float* A, int lda, const float* W, int ldw, int M, int N, int K,
                    const GemmEpi& e, hipStream_t s, bool allow_ksplit) {
  if (M <= 0 || N <= 0 || K <= 0) return 1;
  if (!e.out_f32 && !e.out_bf16) return 2;
  const int tiles_n = (N + FBN - 1) / FBN;
  const int tiles = ((M + FBM - 1) / FBM) * tiles_n;
  // Slices: from (N, K) alone -- the row count must not change a row's bits (micro-batches, single-image launches).  The callers that allow the
  // split send at most ~16 m-tiles of N >= 128 here (larger row counts take the bf16 split-3 form, dod_api.hip qlinear) and any number for the
  // narrow heads (N < 128): aim at two workgroups per CU for 16 m-tiles, at least 8 k-tiles per slice, at most 8 slices.
  int S = 1;
  float* part = nullptr; unsigned* counters = nullptr;
  static const int mode = [] { const char* v = DOD_TUNE_ENV("DINODET_F32_KSPLIT"); return v ? atoi(v) : 1; }();      // tuning builds: 0 = off
  { const int opt = dod_option(DOD_OPT_F32_KSPLIT); if (opt == 0) allow_ksplit = false; else if (opt == 1) allow_ksplit = true; }      // test hook
  if (allow_ksplit && mode && e.rows_per_img == 0) {
    const int nk = (K + FBK - 1) / FBK;
    const int target = mode > 1 ? mode : 512;      // workgroups aimed at for 16 m-tiles (tuning builds: DINODET_F32_KSPLIT = that number)
    int want = (target + 8 * tiles_n) / (16 * tiles_n);      // rounded: N = 768 (12 n-tiles) -> 3
    want = want > 8 ? 8 : want;
    want = want > nk / 8 ? nk / 8 : want;
    if (want >= 2 && tiles <= F32K_TILES && (long)tiles * want <= F32K_PARTS) {
      int dev = 0;
      (void)hipGetDevice(&dev);
      char* slab = f32k_slab(dev, s);
      if (slab) { S = want; counters = reinterpret_cast<unsigned*>(slab); part = reinterpret_cast<float*>(slab + (size_t)F32K_TILES * 4); ++g_f32k_count; }
    }
  }
  if (vec_ok(A, lda) && vec_ok(W, ldw)) hipLaunchKernelGGL(gemm_f32_kernel<true>, dim3(tiles, S), dim3(256), 0, s, A, lda, W, ldw, M, N, K, e, part, counters);
  else hipLaunchKernelGGL(gemm_f32_kernel<false>, dim3(tiles, S), dim3(256), 0, s, A, lda, W, ldw, M, N, K, e, part, counters);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}

// ------------------------------------------------------------------------------------------------ batched / transposed-operand form
template <bool AKM, bool WKM, bool VEC>
__global__ __launch_bounds__(256) void gemm_f32x_kernel(GemmF32X g) {
  __shared__ __attribute__((aligned(16))) float f32_smem[4 * FBK * FLD];     // sA[2] | sW[2]; the split-K epilogue's [64][65] tile
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int tiles_m = (g.M + FBM - 1) / FBM;
  const int tn = blockIdx.x / tiles_m, tm = blockIdx.x - tn * tiles_m;
  const int m0 = tm * FBM, n0 = tn * FBN;
  const int zb = blockIdx.y / g.hb, zh = blockIdx.y - zb * g.hb;
  const int nkt = (g.K + FBK - 1) / FBK;
  int kt0 = 0, kt1 = nkt;
  if (g.ksplit > 1) {
    const int per = (nkt + g.ksplit - 1) / g.ksplit;
    kt0 = blockIdx.z * per; kt1 = kt0 + per < nkt ? kt0 + per : nkt;
    if (kt0 >= kt1) return;                                   // uniform: before any barrier
  }
  const int kend = kt1 * FBK < g.K ? kt1 * FBK : g.K;
  TileLoad<AKM, VEC> la;
  TileLoad<WKM, VEC> lw;
  la.init(g.A + zb * g.a_sb + zh * g.a_sh, g.lda, g.M, kend, m0, tid);
  lw.init(g.W + zb * g.w_sb + zh * g.w_sh, g.ldw, g.N, kend, n0, tid);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  f32_mainloop(la, lw, kt0, kt1, reinterpret_cast<float (*)[FBK * FLD]>(f32_smem),
               reinterpret_cast<float (*)[FBK * FLD]>(f32_smem + 2 * FBK * FLD), acc);

  const int lr = lane & 31, lh = lane >> 5;
  float* cz = g.C + zb * g.c_sb + zh * g.c_sh;
  if (g.ksplit > 1) {
    // atomic accumulate of the slice's partial tile, staged through LDS so that one wave instruction covers 64 consecutive
    // columns of one row (a lane-owns-a-row scatter of 4-byte atomics ran at ~40 per ns: 80 us for a 384 x 768 gradient)
    __syncthreads();                                   // every wave is done reading the last stage
    float* T = f32_smem;                               // [64][65]
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int t = 0; t < 4; ++t) T[(wm * 32 + lr) * 65 + wn * 32 + 8 * q + 4 * lh + t] = acc[4 * q + t] * g.alpha;
    __syncthreads();
    const int n = n0 + lane;
    const float bv = (g.bias && blockIdx.z == 0 && n < g.N) ? g.bias[n] : 0.f;
    for (int rr = 0; rr < 16; ++rr) {
      const int row = wid * 16 + rr, m = m0 + row;
      if (m < g.M && n < g.N) unsafeAtomicAdd(cz + (size_t)m * g.ldc + n, T[row * 65 + lane] + bv);
    }
    return;
  }
  const int m = m0 + wm * 32 + lr;
  if (m >= g.M) return;
  float* crow = cz + (size_t)m * g.ldc;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int n = n0 + wn * 32 + 8 * q + 4 * lh + t;
      if (n >= g.N) continue;
      float v = acc[4 * q + t] * g.alpha;
      if (g.bias) v += g.bias[n];
      if (g.act == ACT_GELU) v = gelu_erf(v);
      else if (g.act == ACT_RELU) v = fmaxf(v, 0.f);
      else if (g.act == ACT_SIGMOID) v = sigmoidf_(v);
      crow[n] = g.accumulate ? crow[n] + v : v;
    }
  }
}

int launch_gemm_f32x(const GemmF32X& g_, hipStream_t s) {
  GemmF32X g = g_;
  if (g.M <= 0 || g.N <= 0 || g.K <= 0 || !g.A || !g.W || !g.C) return 1;
  if (g.batch < 1) g.batch = 1;
  if (g.hb < 1) g.hb = 1;
  if (g.batch % g.hb || g.batch > 65535) return 2;
  if (g.ksplit < 1) g.ksplit = 1;
  if (g.ksplit > 1 && g.act != ACT_NONE) return 2;
  const int nkt = (g.K + FBK - 1) / FBK;
  if (g.ksplit > nkt) g.ksplit = nkt;
  const int tiles = ((g.M + FBM - 1) / FBM) * ((g.N + FBN - 1) / FBN);
  const bool vec = vec_ok(g.A, g.lda, g.a_sb, g.a_sh) && vec_ok(g.W, g.ldw, g.w_sb, g.w_sh);
  const dim3 grid(tiles, g.batch, g.ksplit);
#define F32X_LAUNCH(AK, WK) do { if (vec) hipLaunchKernelGGL((gemm_f32x_kernel<AK, WK, true>), grid, dim3(256), 0, s, g); \
                                 else hipLaunchKernelGGL((gemm_f32x_kernel<AK, WK, false>), grid, dim3(256), 0, s, g); } while (0)
  if (g.a_kmajor && g.w_kmajor) F32X_LAUNCH(true, true);
  else if (g.a_kmajor) F32X_LAUNCH(true, false);
  else if (g.w_kmajor) F32X_LAUNCH(false, true);
  else F32X_LAUNCH(false, false);
#undef F32X_LAUNCH
  return hipGetLastError() == hipSuccess ? 0 : 3;
}
