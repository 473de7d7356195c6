// bf16 MFMA GEMM, 256x256x64 tile, 8 waves in two groups that run half a phase apart ("ping-pong"): while one group's
// waves issue MFMAs the other group's read their next fragments from LDS and issue the staging DMA, so each SIMD's
// matrix pipe always has one of its two waves feeding it.  Same nn.Linear sites as gemm_bf16.hip
// (modeling_dinov2.py:199-201, 246-252, 281-297); C[M,N] = A[M,K] W[N,K]^T, fp32 accumulate, shared epilogue (gemm_epi.h).
//
// Structure (guide: cdna_hip_programming.md section 5, "256^2 8-phase template", re-derived for this LDS image):
//   * wave (g, q), g = wid >> 2 the group, q = wid & 3: output rows g*128..+127, columns q*64..+63 -> 8 x 4 accumulator
//     blocks of 16x16 (128 registers), product computed transposed (lane owns an output row, registers run along n).
//   * a K-tile (64 k) is 4 PLANES of 16 KiB in LDS: A k0..31 | A k32..63 | W k0..31 | W k32..63, each [256 rows][64 B],
//     16-B chunk swizzled by row bit 3 (the conflict-free image of gemm_x3.hip); two K-tile buffers = 128 KiB.
//   * a K-tile is consumed in 4 phases (k half kh = p >> 1, row half mh = p & 1), 16 MFMAs each.  Per phase a wave does
//       LOAD:    ds_read the phase's fragments (W fragments once per k half), issue ONE plane of prefetch (2 LDS-DMA),
//                wait for its LDS reads (and, twice per K-tile, the counted vmcnt), barrier
//       COMPUTE: 16 MFMAs, barrier
//     Group 1 starts one barrier late, so its LOAD runs beside group 0's COMPUTE and vice versa.
//   * prefetch: planes are refilled as soon as every wave has read them -- the k-half-0 planes of tile t+2 during phases
//     2, 3 of tile t, the k-half-1 planes during phases 0, 1 of tile t+1 -- so each plane has 8-10 half-phases (~1 us) of
//     flight; s_waitcnt vmcnt(8) at the end of LOAD 1 and LOAD 3 (never 0 inside the loop), raw s_barrier.
//   * WAR: the LDS reads of a phase are retired (lgkmcnt(0)) BEFORE the barrier that ends the LOAD segment; a plane is
//     re-staged only after the barrier following group 1's last read of it.  RAW: a plane is read one barrier (at least)
//     after the barrier that follows every wave's vmcnt wait for it.
#include <atomic>
#include <mutex>
#include "dod_common.h"
#include "gemm_epi.h"
#include <cstdlib>
#include <cstring>

#define PPM 256
#define PPN 256
#define PPK 64
#define PP_PLANE (256 * 64)               // one 256-row x 32-k bf16 plane: 16 KiB
#define PP_TILE (4 * PP_PLANE)            // A k0 | A k1 | W k0 | W k1 = 64 KiB

__device__ __forceinline__ int pp_swz(int row, int chunk) { return chunk ^ (((row >> 3) & 1) << 1); }   // 16x16x32 lane map, 64-B rows

// (The first, k-phased form of this kernel -- 4 phases per K-tile along k -- was superseded by the m-phased kernels below and removed in
// round 4; the structure notes above describe the skeleton both share.)

// ---------------------------------------------------------------------------------------------------------------------------
// The same ping-pong structure with phases along m (quarter q = rows 32q..32q+31 of the wave's 128): every phase needs all
// four planes, the W fragments (all 4 n-blocks of both W planes: 32 registers) are read once per K-tile in phase 0.
//   X3 = false: plain bf16, planes = A k0..31 | A k32..63 | W k0..31 | W k32..63, 2 MFMAs per 16x16 block and K-tile (BK = 64)
//   X3 = true : the split product of gemm_x3.hip on PAIR-layout operands (A2 [M, 2K] = [Ah | Al], W2 [N, 2K] = [Wh | Wl]):
//               planes = Ah | Al | Wh | Wl of a 32-k K-tile, 3 MFMAs per block (Wl Ah + Wh Al + Wh Ah, lo*lo dropped):
//               24 fragment reads for 96 MFMAs per wave and K-tile (the 16-wave kernel: 32 for 96)
// Prefetch per K-tile u: LOAD 0 stages A-plane 0 of tile u+1, LOAD 1 A-plane 1 of u+1 (their buffer has been idle since tile
// u-1), LOAD 2 / 3 the two W planes of tile u+2 (tile u's W planes are dead after every wave's LOAD 0); ONE counted wait per
// K-tile, vmcnt(4) at the end of LOAD 3 (tile u+1 complete, the two W planes of u+2 still in flight).
// tuning only (tools/pp_timeline.py): when non-null, lane 0 of wave 0 of every workgroup stores {s_memtime at entry, after the prologue,
// after the K loop, at exit (stores drained), s_memrealtime at entry, at exit, blockIdx, 0}
#ifdef DINODET_TUNING
__device__ unsigned long long* g_pp_stamps = nullptr;
extern "C" int dod_debug_pp_stamps(void* dev_buf) {
  unsigned long long* p = (unsigned long long*)dev_buf;
  return hipMemcpyToSymbol(HIP_SYMBOL(g_pp_stamps), &p, sizeof(p)) == hipSuccess ? 0 : 3;
}
#define PP_STAMPS_DECL unsigned long long* const stamps = g_pp_stamps;
#else
#define PP_STAMPS_DECL unsigned long long* const stamps = nullptr;      // release build: the stamp code folds away
#endif

// DC = true: the two LDS-DMA instructions of a phase are issued from inside the COMPUTE segment (between its MFMAs, whose issue slots
// have slack: an MFMA holds the issue port for half its 16 cycles) instead of the LOAD segment, whose length -- not the MFMAs' --
// paces the ping-pong once it exceeds the partner's COMPUTE; the counted wait of LOAD 3 then leaves vmcnt(2) (one plane) in flight.
template <bool X3, bool DC, int LN>
__global__ __launch_bounds__(512) void gemm_ppm_256x256_kernel(const bf16_t* __restrict__ A, int lda,
                                                               const bf16_t* __restrict__ W, int ldw, int M, int N,
                                                               int K, GemmEpi e_, int GM) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wu = __builtin_amdgcn_readfirstlane(wid);
  const int grp_ = wu >> 2, wq = wu & 3;
  const int tiles_m = (M + PPM - 1) / PPM, tiles_n = (N + PPN - 1) / PPN;
  PP_STAMPS_DECL
  unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, tr0 = 0;
  if (stamps) { ts0 = __builtin_amdgcn_s_memtime(); tr0 = __builtin_amdgcn_s_memrealtime(); }
  int tm, tn;
  tile_map(blockIdx.x, tiles_m, tiles_n, GM, &tm, &tn);
  if (tm < 0) return;                              // padding of the weight-resident map's grid
  const int m0 = tm * PPM, n0 = tn * PPN;
  // K slice (gemm_tail_split): this workgroup computes k in [kbase, kbase + Kl) and writes its fp32 partial to its own slab
  const int Kl = e_.ksplit > 1 ? e_.kslice_len : K;
  const int kbase = e_.ksplit > 1 ? (int)blockIdx.y * Kl : 0;
  GemmEpi e = e_;
  if (e_.ksplit > 1) e.out_f32 = e_.out_f32 + (size_t)blockIdx.y * (size_t)e_.kslice_stride;
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  // staging through buffer descriptors: wave-uniform base in SGPRs, one 32-bit byte offset per lane (computed once), the
  // K position as the scalar offset -- no 64-bit address arithmetic per DMA and half the address payload of global_load_lds
  unsigned vA0, vA1, vW0, vW1;
  {
    const int rl = wid * 32 + (lane >> 2);
    const int c = pp_swz(rl, lane & 3);
    int ra0 = m0 + rl, ra1 = m0 + rl + 16; ra0 = ra0 < M ? ra0 : M - 1; ra1 = ra1 < M ? ra1 : M - 1;
    int rw0 = n0 + rl, rw1 = n0 + rl + 16; rw0 = rw0 < N ? rw0 : N - 1; rw1 = rw1 < N ? rw1 : N - 1;
    vA0 = (unsigned)(((size_t)ra0 * lda + c * 8) * 2); vA1 = (unsigned)(((size_t)ra1 * lda + c * 8) * 2);
    vW0 = (unsigned)(((size_t)rw0 * ldw + c * 8) * 2); vW1 = (unsigned)(((size_t)rw1 * ldw + c * 8) * 2);
  }
  const size_t bytesA = (size_t)M * lda * 2, bytesW = (size_t)N * ldw * 2;
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)(bytesA > 0xfffffff0u ? 0xfffffff0u : bytesA), 0x00020000);
  const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, (int)(bytesW > 0xfffffff0u ? 0xfffffff0u : bytesW), 0x00020000);
  char* const sdst = smem + wu * 2048;
  constexpr int KT = X3 ? 32 : 64;                 // k per K-tile
  const int pl1 = X3 ? K : 32;                      // element offset of an operand's second plane
#define PPM_STAGE_H(t, pl, hh)                                                                                 \
  {                                                                                                            \
    char* d_ = sdst + ((t) & 1) * PP_TILE + (pl) * PP_PLANE + (hh) * 1024;                                     \
    const int ko_ = (kbase + (t) * KT + (((pl) & 1) ? pl1 : 0)) * 2;                                                   \
    if ((pl) < 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lptr_t)(d_), 16, (hh) ? vA1 : vA0, ko_, 0, 0); \
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lptr_t)(d_), 16, (hh) ? vW1 : vW0, ko_, 0, 0);          \
  }
#define PPM_STAGE(t, pl) { PPM_STAGE_H(t, pl, 0) PPM_STAGE_H(t, pl, 1) }
  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nk = Kl / KT;
  const int l15 = lane & 15, l4 = lane >> 4;
  int offA[8], offW[4];
#pragma unroll
  for (int i = 0; i < 8; ++i) { const int row = grp_ * 128 + i * 16 + l15; offA[i] = row * 64 + pp_swz(row, l4) * 16; }
#pragma unroll
  for (int j = 0; j < 4; ++j) { const int row = wq * 64 + j * 16 + l15; offW[j] = 2 * PP_PLANE + row * 64 + pp_swz(row, l4) * 16; }

  PPM_STAGE(0, 0) PPM_STAGE(0, 1) PPM_STAGE(0, 2) PPM_STAGE(0, 3)
  if (nk > 1) { PPM_STAGE(1, 2) PPM_STAGE(1, 3) }
  if (nk > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (grp_ == 1) __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  if (stamps) ts1 = __builtin_amdgcn_s_memtime();
  bf16x8 w0[4], w1[4], a0[2], a1[2];       // plane-0 / plane-1 fragments
#define PPM_PHASE(st, q, PF_ON, PF_T, PF_PL, VW)                                                               \
  {                                                                                                            \
    if ((q) == 0) {                                                                                            \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                          \
        w0[j] = *reinterpret_cast<const bf16x8*>((st) + offW[j]);                                              \
        w1[j] = *reinterpret_cast<const bf16x8*>((st) + offW[j] + PP_PLANE);                                   \
      }                                                                                                        \
    }                                                                                                          \
    _Pragma("unroll") for (int ii = 0; ii < 2; ++ii) {                                                         \
      a0[ii] = *reinterpret_cast<const bf16x8*>((st) + offA[2 * (q) + ii]);                                    \
      a1[ii] = *reinterpret_cast<const bf16x8*>((st) + offA[2 * (q) + ii] + PP_PLANE);                         \
    }                                                                                                          \
    if (!DC && (PF_ON)) PPM_STAGE(PF_T, PF_PL)                                                                 \
    if ((VW) == 4) { if (DC) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); } \
    else if ((VW) == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                       \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
    __builtin_amdgcn_s_barrier();                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
    __builtin_amdgcn_s_setprio(1);                                                                             \
    _Pragma("unroll") for (int ii = 0; ii < 2; ++ii)                                                           \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                          \
        if (DC && (PF_ON) && j == 1) {                                                                         \
          __builtin_amdgcn_sched_barrier(0);                                                                   \
          PPM_STAGE_H(PF_T, PF_PL, ii)                                                                         \
          __builtin_amdgcn_sched_barrier(0);                                                                   \
        }                                                                                                      \
        f32x4 c_ = acc[2 * (q) + ii][j];                                                                       \
        if (X3) {                                                                                              \
          c_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1[j], a0[ii], c_, 0, 0, 0);   /* Wl Ah: small terms first */ \
          c_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0[j], a1[ii], c_, 0, 0, 0);   /* Wh Al */              \
          c_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0[j], a0[ii], c_, 0, 0, 0);   /* Wh Ah */              \
        } else {                                                                                               \
          c_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0[j], a0[ii], c_, 0, 0, 0);   /* k 0..31 */            \
          c_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1[j], a1[ii], c_, 0, 0, 0);   /* k 32..63 */           \
        }                                                                                                      \
        acc[2 * (q) + ii][j] = c_;                                                                             \
      }                                                                                                        \
    __builtin_amdgcn_s_setprio(0);                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
    __builtin_amdgcn_s_barrier();                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
  }

  int u = 0;
  for (; u + 2 < nk; ++u) {
    const char* st = smem + (u & 1) * PP_TILE;
    PPM_PHASE(st, 0, true, u + 1, 0, -1)
    PPM_PHASE(st, 1, true, u + 1, 1, -1)
    PPM_PHASE(st, 2, true, u + 2, 2, -1)
    PPM_PHASE(st, 3, true, u + 2, 3, 4)           // tile u + 1 complete; the two W planes of u + 2 in flight
  }
  if (u + 1 < nk) {
    const char* st = smem + (u & 1) * PP_TILE;
    PPM_PHASE(st, 0, true, u + 1, 0, -1)
    PPM_PHASE(st, 1, true, u + 1, 1, -1)
    PPM_PHASE(st, 2, false, 0, 0, -1)
    PPM_PHASE(st, 3, false, 0, 0, 0)
    ++u;
  }
  {
    const char* st = smem + (u & 1) * PP_TILE;
    PPM_PHASE(st, 0, false, 0, 0, -1)
    PPM_PHASE(st, 1, false, 0, 0, -1)
    PPM_PHASE(st, 2, false, 0, 0, -1)
    PPM_PHASE(st, 3, false, 0, 0, -1)
  }
  if (grp_ == 0) __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (stamps) ts2 = __builtin_amdgcn_s_memtime();

  constexpr int PITCH = PPN * 4 + 16;
  const ColParams cp = load_col_params<PPN, LN>(e, n0, N, tid);
  const bool wide = drain8_ok(e, N);
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    if (pass) __syncthreads();
#pragma unroll
    for (int ii = 0; ii < 4; ++ii)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row_l = grp_ * 64 + ii * 16 + l15;
        const int col = wq * 64 + j * 16 + 4 * l4;
        const f32x4 a = acc[pass * 4 + ii][j];
        *reinterpret_cast<float4*>(smem + row_l * PITCH + col * 4) = make_float4(a[0], a[1], a[2], a[3]);
      }
    auto rowmap = [&](int row_l) { return m0 + (row_l >> 6) * 128 + pass * 64 + (row_l & 63); };
    stage_row_stats<128, PPN, LN>(smem, PITCH, e, M, tid, rowmap, K, n0);
    __syncthreads();
    if (wide) drain_tile_bf16x8<128, PPN, 512, LN>(smem, PITCH, e, M, N, n0, tid, rowmap);
    else drain_tile<128, PPN, 512, LN>(smem, PITCH, e, cp, M, N, n0, tid, rowmap);
  }
  if (stamps && tid == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long* o = stamps + (size_t)blockIdx.x * 8;
    o[0] = ts0; o[1] = ts1; o[2] = ts2; o[3] = __builtin_amdgcn_s_memtime(); o[4] = tr0; o[5] = __builtin_amdgcn_s_memrealtime(); o[6] = blockIdx.x; o[7] = 0;
  }
}


// ---------------------------------------------------------------------------------------------------------------------------
// H2 GEMM (the fp16x2 precision mode): the ping-pong structure above on H2-format operands (dod_common.h) --
// x.w ~ fp16(x) fp16(w) on v_mfma_f32_32x32x16_f16 plus BOTH cross terms as one block-scaled e4m3 MFMA
// (v_mfma_scale_f32_32x32x64_f8f6f4; per lane the k-slots [e4m3(hx) | e4m3(lx 2^11)] . [e4m3(lw 2^(e+11)) | e4m3(hw 2^e)]).
//   * planes of a 32-k K-tile: A fp16 | A e4m3 (main, remainder) | W fp16, each [256 rows][64 B], and W e4m3 remainder
//     [256 rows][32 B]: 56 KiB per K-tile, two buffers;
//   * 32x32 MFMA lane map: lane (r = lane & 31, g = lane >> 5) reads the 32 bytes (chunks 2g, 2g+1) of row r of a 64-byte plane:
//     fp16 k 16g..16g+15 (two 32x32x16 products: any k order both operands share is a valid dot product), or the e4m3 main and
//     remainder bytes of the same 16 k; from the 32-byte plane chunk g.  chunk ^= (row >> 2) & 3 (64-byte rows) and
//     chunk ^= (row >> 3) & 1 (32-byte rows) make every ds_read_b128 lane group conflict-free;
//   * e4m3(hw 2^e) is converted from the fp16 weight fragments once per K-tile (8 v_cvt_scalef32_pk_fp8_f16 per 32 x 16 block,
//     which divides by its scale operand: 2^-e, built from the row's exponent byte);
//   * TWO phases per K-tile (row blocks 2p, 2p+1 of the wave's four): 12 MFMAs = 4 * 64 + 8 * 32 = 512 matrix-pipe cycles per
//     COMPUTE segment; per 32x32 block and K-tile 128 cycles against 192 for the three bf16 products;
//   * E8M0 scale operands: weights 2^-(e_n + 11) (one byte per n-block, constant over K), activations 2^0.
// Rings: THREE A buffers (2 x 16 KiB planes each), TWO W buffers (16 + 8 KiB): 144 KiB.  The A panel streams from beyond L2 (every
// m-tile's rows are fetched by only tiles_n workgroups), so its planes are staged TWO K-tiles ahead; the W planes, shared by every
// m-tile of the XCD's group and L2 / Infinity-Cache resident, one and a half.  Per K-tile u: LOAD 0 stages the A planes of tile
// u + 2 (4 pieces), LOAD 1 the W planes of tile u + 2 (3 pieces); one counted wait per K-tile, vmcnt(7) at the end of LOAD 1 (tile
// u + 1 complete, tile u + 2 in flight).  With A one K-tile ahead (half a K-tile of flight) the loop ran at 2600 cycles per K-tile
// against 2048 of MFMA work -- and at 2130 with every A row aliased to one row (cache hits): it waited for the A panel.
__device__ __forceinline__ int h2_swz(int row, int chunk) { return chunk ^ ((row >> 2) & 3); }
typedef __attribute__((ext_vector_type(8))) int i32x8_;
typedef __attribute__((ext_vector_type(4))) int i32x4_;
typedef __attribute__((ext_vector_type(2))) short s16x2_;
#define H2_ABUF (32 * 1024)        // A buffer t % 3: fp16 plane, then the e4m3 (main, remainder) plane
#define H2_A8 (16 * 1024)
#define H2_WBASE (3 * H2_ABUF)     // W buffer t & 1: fp16 plane (16 KiB), then the e4m3 remainder plane (8 KiB)
#define H2_WBUF (24 * 1024)
#define H2_WL (16 * 1024)
#define H2_LDS (H2_WBASE + 2 * H2_WBUF)

// 8 fp16 (one 16-byte fragment) -> 8 e4m3 bytes of (value / inv_scale), round to nearest even
__device__ __forceinline__ void h2_cvt8(const bf16x8& raw, float inv_scale, int& d0, int& d1) {
  const f16x8 v = __builtin_bit_cast(f16x8, raw);
  s16x2_ r0 = {0, 0}, r1 = {0, 0};
  r0 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(r0, f16x2{v[0], v[1]}, inv_scale, false);
  r0 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(r0, f16x2{v[2], v[3]}, inv_scale, true);
  r1 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(r1, f16x2{v[4], v[5]}, inv_scale, false);
  r1 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(r1, f16x2{v[6], v[7]}, inv_scale, true);
  d0 = __builtin_bit_cast(int, r0); d1 = __builtin_bit_cast(int, r1);
}

template <int LN>
__global__ __launch_bounds__(512) void gemm_h2_256x256_kernel(const char* __restrict__ A, int lda, const char* __restrict__ W, int ldw,
                                                              int M, int N, int K, GemmEpi e_, int GM) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wu = __builtin_amdgcn_readfirstlane(wid);
  const int grp_ = wu >> 2, wq = wu & 3;
  const int tiles_m = (M + PPM - 1) / PPM, tiles_n = (N + PPN - 1) / PPN;
  PP_STAMPS_DECL
  unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, tr0 = 0;
  if (stamps) { ts0 = __builtin_amdgcn_s_memtime(); tr0 = __builtin_amdgcn_s_memrealtime(); }
  int tm, tn;
  tile_map(blockIdx.x, tiles_m, tiles_n, GM, &tm, &tn);
  const int m0 = tm * PPM, n0 = tn * PPN;
  // K slice (gemm_tail_split): k-tiles [kb, kb + nk) of the operands, fp32 partial to this slice's slab
  const int Kl = e_.ksplit > 1 ? e_.kslice_len : K;
  const int kb = e_.ksplit > 1 ? (int)blockIdx.y * (Kl / 32) : 0;
  GemmEpi e = e_;
  if (e_.ksplit > 1) e.out_f32 = e_.out_f32 + (size_t)blockIdx.y * (size_t)e_.kslice_stride;
  typedef __attribute__((address_space(3))) void* lptr_t;
  // staging (buffer-descriptor LDS-DMA): 64-byte planes as two 16-row pieces per wave (lane -> row = lane >> 2, LDS chunk slot
  // lane & 3, fetching global chunk slot ^ swizzle(row)); the 32-byte W remainder plane as ONE 32-row piece (row = lane >> 1)
  unsigned vA0, vA1, vW0, vW1, vWl;
  {
    const int rl = wid * 32 + (lane >> 2);
    const int c = h2_swz(rl, lane & 3);              // rows rl and rl + 16 share (row >> 2) & 3
    int ra0 = m0 + rl, ra1 = m0 + rl + 16; ra0 = ra0 < M ? ra0 : M - 1; ra1 = ra1 < M ? ra1 : M - 1;
    int rw0 = n0 + rl, rw1 = n0 + rl + 16; rw0 = rw0 < N ? rw0 : N - 1; rw1 = rw1 < N ? rw1 : N - 1;
    vA0 = (unsigned)((size_t)ra0 * lda + c * 16); vA1 = (unsigned)((size_t)ra1 * lda + c * 16);
    vW0 = (unsigned)((size_t)rw0 * ldw + c * 16); vW1 = (unsigned)((size_t)rw1 * ldw + c * 16);
    const int r2 = wid * 32 + (lane >> 1);
    int rwl = n0 + r2; rwl = rwl < N ? rwl : N - 1;
    vWl = (unsigned)((size_t)rwl * ldw + 2 * (size_t)K + (((lane & 1) ^ ((r2 >> 3) & 1)) * 16));
  }
  const size_t bytesA = (size_t)M * lda, bytesW = (size_t)N * ldw;
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)(bytesA > 0xfffffff0u ? 0xfffffff0u : bytesA), 0x00020000);
  const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, (int)(bytesW > 0xfffffff0u ? 0xfffffff0u : bytesW), 0x00020000);
  const int K2 = 2 * K;
  // A planes of K-tile t into A buffer ab (4 pieces per wave); W planes of K-tile t into W buffer t & 1 (3 pieces)
#define H2_STAGE_A(t, ab)                                                                                      \
  {                                                                                                            \
    char* b_ = smem + (ab) * H2_ABUF + wu * 2048;                                                              \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lptr_t)(b_), 16, vA0, (kb + (t)) * 64, 0, 0);                       \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lptr_t)(b_ + 1024), 16, vA1, (kb + (t)) * 64, 0, 0);                \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lptr_t)(b_ + H2_A8), 16, vA0, K2 + (kb + (t)) * 64, 0, 0);          \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lptr_t)(b_ + H2_A8 + 1024), 16, vA1, K2 + (kb + (t)) * 64, 0, 0);   \
  }
#define H2_STAGE_W(t)                                                                                          \
  {                                                                                                            \
    char* b_ = smem + H2_WBASE + ((t) & 1) * H2_WBUF;                                                          \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lptr_t)(b_ + wu * 2048), 16, vW0, (kb + (t)) * 64, 0, 0);           \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lptr_t)(b_ + wu * 2048 + 1024), 16, vW1, (kb + (t)) * 64, 0, 0);    \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lptr_t)(b_ + H2_WL + wu * 1024), 16, vWl, (kb + (t)) * 32, 0, 0);   \
  }
  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int nk = Kl / 32;
  const int lr = lane & 31, lg = lane >> 5;
  int offA[4], offW[2], offWl[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) { const int row = grp_ * 128 + i * 32 + lr; offA[i] = row * 64 + h2_swz(row, 2 * lg) * 16; }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = wq * 64 + j * 32 + lr;
    offW[j] = row * 64 + h2_swz(row, 2 * lg) * 16;
    offWl[j] = H2_WL + row * 32 + ((lg ^ ((row >> 3) & 1)) * 16);
  }
  // E8M0 block scale of the weight operand (the two n-blocks' bytes in bits 0..7 / 8..15): 2^-(e + 11); and 2^-e as the divisor of
  // the in-kernel fp16 -> e4m3 conversion (a float whose exponent field is the row's byte 127 - e)
  int sc_w; float cinv[2];
  {
    int na = n0 + wq * 64 + lr, nb = na + 32;
    na = na < N ? na : N - 1; nb = nb < N ? nb : N - 1;
    const int ba = (int)e.h2_wexp[na], bb = (int)e.h2_wexp[nb];
    cinv[0] = __uint_as_float((unsigned)ba << 23); cinv[1] = __uint_as_float((unsigned)bb << 23);
    sc_w = (ba - 11 < 0 ? 0 : ba - 11) | ((bb - 11 < 0 ? 0 : bb - 11) << 8);
  }
  const int sc_a = 127;

  H2_STAGE_A(0, 0) H2_STAGE_W(0)
  if (nk > 1) { H2_STAGE_A(1, 1) H2_STAGE_W(1) }
  if (nk > 1) asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (grp_ == 1) __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (stamps) ts1 = __builtin_amdgcn_s_memtime();

  bf16x8 w16[2][2], wl[2], a16[2][2], a8[2][2];      // read as ushort vectors (an int-typed read would make hipcc drain the DMA ring)
  i32x8_ w8v[2];
  // one phase: sa / sw = this K-tile's A / W buffers; PF: 0 nothing to stage, 1 the A planes of K-tile PF_T into A buffer PF_AB, 2 the W
  // planes of K-tile PF_T; VW: vmcnt to wait for at the end of the LOAD segment (-1: none)
#define H2_PHASE(sa, sw, p, PF, PF_T, PF_AB, VW)                                                               \
  {                                                                                                            \
    if ((p) == 0) {                                                                                            \
      _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                          \
        w16[j][0] = *reinterpret_cast<const bf16x8*>((sw) + offW[j]);                                          \
        w16[j][1] = *reinterpret_cast<const bf16x8*>((sw) + (offW[j] ^ 16));                                   \
        wl[j] = *reinterpret_cast<const bf16x8*>((sw) + offWl[j]);                                             \
      }                                                                                                        \
    }                                                                                                          \
    _Pragma("unroll") for (int ii = 0; ii < 2; ++ii) {                                                         \
      a16[ii][0] = *reinterpret_cast<const bf16x8*>((sa) + offA[2 * (p) + ii]);                                \
      a16[ii][1] = *reinterpret_cast<const bf16x8*>((sa) + (offA[2 * (p) + ii] ^ 16));                         \
      a8[ii][0] = *reinterpret_cast<const bf16x8*>((sa) + offA[2 * (p) + ii] + H2_A8);                         \
      a8[ii][1] = *reinterpret_cast<const bf16x8*>((sa) + (offA[2 * (p) + ii] ^ 16) + H2_A8);                  \
    }                                                                                                          \
    if ((PF) == 1) H2_STAGE_A(PF_T, PF_AB)                                                                     \
    if ((PF) == 2) H2_STAGE_W(PF_T)                                                                            \
    if ((VW) == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");                                            \
    else if ((VW) == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                       \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
    __builtin_amdgcn_s_barrier();                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
    __builtin_amdgcn_s_setprio(1);                                                                             \
    _Pragma("unroll") for (int ii = 0; ii < 2; ++ii)                                                           \
      _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                          \
        f32x16 c_ = acc[2 * (p) + ii][j];                                                                      \
        c_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w16[j][0]), __builtin_bit_cast(f16x8, a16[ii][0]), c_, 0, 0, 0); \
        c_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w16[j][1]), __builtin_bit_cast(f16x8, a16[ii][1]), c_, 0, 0, 0); \
        acc[2 * (p) + ii][j] = c_;                                                                             \
      }                                                                                                        \
    if ((p) == 0) {       /* e4m3(hw 2^e) from the fp16 weight fragments: under the main-product MFMAs above */  \
      _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                          \
        int d0, d1, d2, d3;                                                                                    \
        h2_cvt8(w16[j][0], cinv[j], d0, d1);                                                                   \
        h2_cvt8(w16[j][1], cinv[j], d2, d3);                                                                   \
        const i32x4_ l_ = __builtin_bit_cast(i32x4_, wl[j]);                                                   \
        w8v[j] = i32x8_{l_[0], l_[1], l_[2], l_[3], d0, d1, d2, d3};                                           \
      }                                                                                                        \
    }                                                                                                          \
    _Pragma("unroll") for (int ii = 0; ii < 2; ++ii) {                                                         \
      const i32x8_ a8v = __builtin_shufflevector(__builtin_bit_cast(i32x4_, a8[ii][0]), __builtin_bit_cast(i32x4_, a8[ii][1]), 0, 1, 2, 3, 4, 5, 6, 7); \
      acc[2 * (p) + ii][0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(w8v[0], a8v, acc[2 * (p) + ii][0], 0, 0, 0, sc_w, 0, sc_a); \
      acc[2 * (p) + ii][1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(w8v[1], a8v, acc[2 * (p) + ii][1], 0, 0, 1, sc_w, 0, sc_a); \
    }                                                                                                          \
    __builtin_amdgcn_s_setprio(0);                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
    __builtin_amdgcn_s_barrier();                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
  }

  int u = 0, ua = 0;                            // ua = u % 3
  for (; u + 2 < nk; ++u) {
    const char* sa = smem + ua * H2_ABUF;
    const char* sw = smem + H2_WBASE + (u & 1) * H2_WBUF;
    const int ab2 = ua == 0 ? 2 : ua - 1;       // (u + 2) % 3
    H2_PHASE(sa, sw, 0, 1, u + 2, ab2, -1)
    H2_PHASE(sa, sw, 1, 2, u + 2, 0, 7)
    ua = ua == 2 ? 0 : ua + 1;
  }
  if (u + 1 < nk) {
    const char* sa = smem + ua * H2_ABUF;
    const char* sw = smem + H2_WBASE + (u & 1) * H2_WBUF;
    H2_PHASE(sa, sw, 0, 0, 0, 0, -1)
    H2_PHASE(sa, sw, 1, 0, 0, 0, 0)
    ua = ua == 2 ? 0 : ua + 1;
    ++u;
  }
  {
    const char* sa = smem + ua * H2_ABUF;
    const char* sw = smem + H2_WBASE + (u & 1) * H2_WBUF;
    H2_PHASE(sa, sw, 0, 0, 0, 0, -1)
    H2_PHASE(sa, sw, 1, 0, 0, 0, -1)
  }
  if (grp_ == 0) __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (stamps) ts2 = __builtin_amdgcn_s_memtime();

  constexpr int PITCH = PPN * 4 + 16;
  const ColParams cp = load_col_params<PPN, LN>(e, n0, N, tid);
  const bool wide = drain8_ok(e, N);
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    if (pass) __syncthreads();
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
      for (int j = 0; j < 2; ++j) stage_acc(smem, PITCH, grp_ * 64 + ii * 32 + lr, wq * 64 + j * 32, acc[pass * 2 + ii][j], lg);
    auto rowmap = [&](int row_l) { return m0 + (row_l >> 6) * 128 + pass * 64 + (row_l & 63); };
    stage_row_stats<128, PPN, LN>(smem, PITCH, e, M, tid, rowmap, K, n0);
    __syncthreads();
    if (wide) drain_tile_bf16x8<128, PPN, 512, LN>(smem, PITCH, e, M, N, n0, tid, rowmap);
    else drain_tile<128, PPN, 512, LN>(smem, PITCH, e, cp, M, N, n0, tid, rowmap);
  }
  if (stamps && tid == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long* o = stamps + (size_t)blockIdx.x * 8;
    o[0] = ts0; o[1] = ts1; o[2] = ts2; o[3] = __builtin_amdgcn_s_memtime(); o[4] = tr0; o[5] = __builtin_amdgcn_s_memrealtime(); o[6] = blockIdx.x; o[7] = 0;
  }
}

// (An fp8 kernel on this skeleton -- round 2: faster than the 256x128 kernel of gemm_fp8.hip in isolation at K >= 1536, slower inside the ViT-g
// forward, DESIGN.md section 9 -- was an opt-in and was removed in round 4.)

static constexpr int LDSPP = (128 * (PPN * 4 + 16)) > 2 * PP_TILE ? (128 * (PPN * 4 + 16)) : 2 * PP_TILE;

#ifdef DINODET_TUNING
// weight-resident tile map (gemm_epi.h tile_map, mode bit 0x400): DINODET_GEMM_WRES = 1 (read per launch)
static int wres_grid(int M, int N, int* gm) {
  const int tiles_m = (M + PPM - 1) / PPM, tiles_n = (N + PPN - 1) / PPN;
  const char* v = getenv("DINODET_GEMM_WRES");
  if (v && v[0] == '1' && tiles_n % 2 == 0 && tiles_m >= 32) { *gm = (*gm & ~0x300) | 0x400; return 8 * ((tiles_m + 3) / 4) * (tiles_n / 2); }
  return tiles_m * tiles_n;
}
#else
static int wres_grid(int M, int N, int*) { return ((M + PPM - 1) / PPM) * ((N + PPN - 1) / PPN); }
#endif

static void ppm_attr() {      // > 64 KiB of dynamic LDS: once per device
  static bool attr_set[16] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 16 && !attr_set[dev]) {
#define ATTR_(LN_) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_ppm_256x256_kernel<false, false, LN_>), hipFuncAttributeMaxDynamicSharedMemorySize, LDSPP); \
                   (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_ppm_256x256_kernel<true, false, LN_>), hipFuncAttributeMaxDynamicSharedMemorySize, LDSPP);
    ATTR_(LN_NONE) ATTR_(LN_CONS) ATTR_(LN_PROD)
#undef ATTR_
    attr_set[dev] = true;
  }
}

// plain bf16 on the m-phased ping-pong kernel: K % 64 == 0
int launch_gemm_bf16_ppm(const bf16_t* A, int lda, const bf16_t* W, int ldw, int M, int N, int K, const GemmEpi& e, hipStream_t s) {
  if (M <= 0 || N <= 0 || K <= 0 || K % PPK != 0) return 2;
  if (N % 4 != 0 || lda % 8 != 0 || ldw % 8 != 0 || e.ldc % 4 != 0) return 2;
  if (e.resid && e.ldr % 4 != 0) return 2;
  if (!e.out_f32 && !e.out_bf16) return 2;
  ppm_attr();
  int gm = gemm_tile_mode();
  const int tiles = wres_grid(M, N, &gm);
#define GO_(LN_) hipLaunchKernelGGL((gemm_ppm_256x256_kernel<false, false, LN_>), dim3(tiles), dim3(512), LDSPP, s, A, lda, W, ldw, M, N, K, e, gm);
  LN_DISPATCH(e, GO_)
#undef GO_
  return hipGetLastError() == hipSuccess ? 0 : 3;
}

// split product on pair-layout operands (same contract as launch_gemm_x3): K % 32 == 0
int launch_gemm_x3_pp(const bf16_t* A2, int lda, const bf16_t* W2, int ldw, int M, int N, int K, const GemmEpi& e, hipStream_t s) {
  if (M <= 0 || N <= 0 || K <= 0) return 1;
  if (K % 32 != 0 || N % 4 != 0 || lda % 8 != 0 || ldw % 8 != 0 || lda < 2 * K || ldw < 2 * K) return 2;
  if (e.out_f32 && e.ldc % 4 != 0) return 2;
  if (e.resid && e.ldr % 4 != 0) return 2;
  if (!e.out_f32 && !e.out_bf16) return 2;
  ppm_attr();
  int gm = gemm_tile_mode();
  const int tiles = wres_grid(M, N, &gm);
#define GO_(LN_) hipLaunchKernelGGL((gemm_ppm_256x256_kernel<true, false, LN_>), dim3(tiles), dim3(512), LDSPP, s, A2, lda, W2, ldw, M, N, K, e, gm);
  LN_DISPATCH(e, GO_)
#undef GO_
  return hipGetLastError() == hipSuccess ? 0 : 3;
}

// H2 operands (dod_common.h): A [M, K] activation rows at pitch lda bytes (>= 4K), W [N, K] weight rows at pitch ldw bytes (>= 3K)
static constexpr int LDSH2 = (128 * (PPN * 4 + 16)) > H2_LDS ? (128 * (PPN * 4 + 16)) : H2_LDS;
static void h2_attr() {
  static bool attr_set[16] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 16 && !attr_set[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_h2_256x256_kernel<LN_NONE>), hipFuncAttributeMaxDynamicSharedMemorySize, LDSH2);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_h2_256x256_kernel<LN_CONS>), hipFuncAttributeMaxDynamicSharedMemorySize, LDSH2);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_h2_256x256_kernel<LN_PROD>), hipFuncAttributeMaxDynamicSharedMemorySize, LDSH2);
    attr_set[dev] = true;
  }
}
int launch_gemm_h2(const void* A, int lda, const void* W, int ldw, int M, int N, int K, const GemmEpi& e, hipStream_t s) {
  if (M <= 0 || N <= 0 || K <= 0) return 1;
  if (K % 32 != 0 || N % 4 != 0 || lda % 16 != 0 || ldw % 16 != 0 || lda < 4 * K || ldw < 3 * K || !e.h2_wexp) return 2;
  if (e.out_f32 && e.ldc % 4 != 0) return 2;
  if (e.resid && e.ldr % 4 != 0) return 2;
  if (!e.out_f32 && !e.out_bf16) return 2;
  { const int No = e.glu ? N / 2 : N;     // columns of the output row (GemmEpi::glu: the gate halves them)
    if (e.out_h2 && (No % 32 != 0 || e.ldc < 2 * No || e.ldc % 8 != 0 || e.out_split != 0)) return 2;
    if (e.glu && e.out_h2 && (N % 8 != 0)) return 2; }
  { const int t = gemm_tail_split(2, A, lda, W, ldw, M, N, K, e, s); if (t >= 0) return t; }
  h2_attr();
  const int gm = gemm_tile_mode();
  const int tiles = ((M + PPM - 1) / PPM) * ((N + PPN - 1) / PPN);
#define GO_(LN_) hipLaunchKernelGGL(gemm_h2_256x256_kernel<LN_>, dim3(tiles), dim3(512), LDSH2, s, (const char*)A, lda, (const char*)W, ldw, M, N, K, e, gm);
  LN_DISPATCH(e, GO_)
#undef GO_
  return hipGetLastError() == hipSuccess ? 0 : 3;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Wave-quantisation tail (dod_common.h gemm_tail_split).  One 256x256 tile per CU and round: 343 x 3 = 1029 tiles of an N = 768 GEMM at
// M = 87680 take FIVE rounds on 256 CUs for 4.02 rounds of work (tools/pp_timeline.py: kernel span = 5 tile times).  The rows of the
// short last round are cut off, their tiles K-split S ways in ONE launch (grid.y = slice: S CUs per tile, fp32 partial slabs), and
// a reduce + epilogue launch sums the slabs into an LDS tile and drains it with the GEMMs' own epilogue code.
#define KSR_ROWS 16      // rows per workgroup of the reduce launch: 16 x 256 outputs, so that a 384-row remainder still spreads over 72 CUs
template <int LN>
__global__ __launch_bounds__(512) void gemm_ksplit_reduce_kernel(const float* __restrict__ part, long long slice_stride, int S, int R, int N,
                                                                 GemmEpi e, int m_base, int M, int K) {
  __shared__ __attribute__((aligned(16))) char smem[KSR_ROWS * (PPN * 4 + 16)];
  constexpr int PITCH = PPN * 4 + 16;
  const int tid = threadIdx.x;
  const int r0 = blockIdx.x * KSR_ROWS, n0 = blockIdx.y * PPN;
  const int c4 = tid & 63;                                     // 64 float4 columns of the 256-column tile
#pragma unroll
  for (int rl = tid >> 6; rl < KSR_ROWS; rl += 8) {
    const int r = r0 + rl, n = n0 + 4 * c4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < R && n < N) {
      const float* p = part + (size_t)r * N + n;
      for (int sl = 0; sl < S; ++sl) {
        const float4 v = *reinterpret_cast<const float4*>(p + (size_t)sl * slice_stride);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
    }
    *reinterpret_cast<float4*>(smem + rl * PITCH + c4 * 16) = acc;
  }
  auto rowmap = [&](int row_l) { return m_base + r0 + row_l; };
  stage_row_stats<KSR_ROWS, PPN, LN>(smem, PITCH, e, M, tid, rowmap, K, n0);
  __syncthreads();
  const ColParams cp = load_col_params<PPN, LN>(e, n0, N, tid);
  if (drain8_ok(e, N)) drain_tile_bf16x8<KSR_ROWS, PPN, 512, LN>(smem, PITCH, e, M, N, n0, tid, rowmap);
  else drain_tile<KSR_ROWS, PPN, 512, LN>(smem, PITCH, e, cp, M, N, n0, tid, rowmap);
}

// scratch: TAIL_SLOTS slabs per device, handed to launching streams least-recently-used first (a forward may run as concurrent
// micro-batches on separate streams -- engine.py: two -- and a process creates many short-lived streams over time: warm-up, capture,
// side streams).  Streams that launch concurrently must not share a slab: up to TAIL_SLOTS of them never do.
#define TAIL_SLOTS 4
#define TAIL_GENS 4                            // the scratch only ever GROWS by adding a block: captured hipGraphs keep slab addresses
static std::mutex g_tail_mu;                   // autograd / DataLoader threads may launch GEMMs beside the main thread
static float* g_tail_scratch[16][TAIL_GENS] = {};
static int g_tail_gen[16] = {};                // blocks allocated so far (the newest one is handed out)
static size_t g_tail_bytes[16] = {};           // bytes per slab of the newest block
static hipStream_t g_tail_owner[16][TAIL_SLOTS] = {};
static unsigned long long g_tail_used[16][TAIL_SLOTS] = {};
static unsigned long long g_tail_clock = 0;
int gemm_tail_reserve(size_t bytes) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 3;
  std::lock_guard<std::mutex> lk(g_tail_mu);
  if (g_tail_bytes[dev] >= bytes) return 0;
  if (g_tail_gen[dev] >= TAIL_GENS) return 3;  // never free a block a captured graph may still reference: refuse instead
  float* blk = nullptr;
  if (hipMalloc((void**)&blk, bytes * TAIL_SLOTS) != hipSuccess) return 3;
  g_tail_scratch[dev][g_tail_gen[dev]++] = blk;
  for (int i = 0; i < TAIL_SLOTS; ++i) { g_tail_used[dev][i] = 0; g_tail_owner[dev][i] = nullptr; }
  g_tail_bytes[dev] = bytes;
  return 0;
}
static float* tail_slab(int dev, hipStream_t s) {
  std::lock_guard<std::mutex> lk(g_tail_mu);
  if (!g_tail_gen[dev]) return nullptr;
  int pick = -1;
  for (int i = 0; i < TAIL_SLOTS; ++i)
    if (g_tail_used[dev][i] && g_tail_owner[dev][i] == s) { pick = i; break; }
  if (pick < 0) {
    pick = 0;
    for (int i = 1; i < TAIL_SLOTS; ++i)
      if (g_tail_used[dev][i] < g_tail_used[dev][pick]) pick = i;
    g_tail_owner[dev][pick] = s;
  }
  g_tail_used[dev][pick] = ++g_tail_clock;
  return (float*)((char*)g_tail_scratch[dev][g_tail_gen[dev] - 1] + (size_t)pick * g_tail_bytes[dev]);
}

// mode of the split (test option DOD_OPT_TAILSPLIT): -1 / 1 = the shipped heuristic, 0 = off, 2 = every qualifying shape
static int tail_mode() {
  const int m = dod_option(DOD_OPT_TAILSPLIT);
  if (m >= 0) return m;
  static const char* env = DOD_TUNE_ENV("DINODET_GEMM_TAILSPLIT");
  return env ? (env[0] == '0' ? 0 : (env[0] == '2' ? 2 : 1)) : 1;
}

static thread_local bool t_in_tail_split = false;
static std::atomic<long> g_tail_splits{0};
long gemm_tail_split_count() { return g_tail_splits.load(); }
int gemm_tail_split(int kind, const void* A, int lda, const void* W, int ldw, int M, int N, int K, const GemmEpi& e, hipStream_t s) {
  if (t_in_tail_split || e.ksplit > 1 || e.rows_per_img != 0 || e.a_scale || e.out_split > 0) return -1;
  const int mode = tail_mode();
  if (mode == 0) return -1;
  const int ktile = kind == 0 ? 64 : 32;
  if (N % 4 != 0 || K % ktile != 0) return -1;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return -1;
  static int cus[16] = {};
  if (!cus[dev]) { int c = 0; (void)hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev); cus[dev] = c > 0 ? c : 256; }
  const int CU = cus[dev];
  const int tiles_n = (N + PPN - 1) / PPN, tiles_m = (M + PPM - 1) / PPM, tiles = tiles_m * tiles_n;
  const int nk = K / ktile;
  const bool force = mode == 2;
  int Mmain, R, tiles_r, S = 0;
  // plain bf16 (kind 0) has a 256x128 tile for underfilled grids and does NOT take this K-split by default.  Round 3 tried it on the
  // decoder's query-side linears (M = B*Q = 3 200 rows at batch 32: 39 tiles of 256x256 walking K' = 3K = 2 304, 48.5 us on 78 CUs as
  // 256x128 tiles): six slices per tile (234 workgroups) + the reduce launch write and re-read 59 MB of fp32 partials, and the forward
  // gains nothing (`bench.py --workload vitb224`: 7 539 vs 7 571 images/s, vitb518 2 241.7 both ways).  Tuning builds: DINODET_GEMM_KSPLIT0 = n
  // enables it for grids of at most CUs * n / 12 tiles.
  static const int k0frac = [] { const char* v = DOD_TUNE_ENV("DINODET_GEMM_KSPLIT0"); return v ? atoi(v) : 0; }();
  if (M >= 2048 && (kind >= 1 ? tiles * 2 <= CU + CU / 8 : tiles * 12 <= CU * k0frac)) {
    // (a) an UNDERFILLED single round (the compensated kernels have no smaller tile): 99 tiles of an N = 768 GEMM at M = 8224 leave 157
    // CUs idle -- every tile is K-split so that tiles x S fills the chip (no main launch)
    Mmain = 0; R = M; tiles_r = tiles;
    for (int c : {8, 6, 4, 3, 2})
      if (nk % c == 0 && nk / c >= 3 && tiles_r * c <= CU) { S = c; break; }
  } else {
    // (b) a short LAST round.  Measured at M = 87680 (343 x 3 tiles, 4.02 rounds; tools/bench_h2.py, tools/bench_pp.py with
    // DINODET_GEMM_TAILSPLIT=0 / 2): the unsplit kernels last ~4.3 tile times, not 5 -- the five tiles of the last round run alone on
    // the chip -- so the split pays where a tile is long: fc2 (K = 3072) 976 vs 1047 us (split product), 866 vs 944 (H2), 474 vs 491
    // (plain bf16); out-proj (K = 768) 367 vs 373, 346 vs 350, 218 vs 217: not worth a second and third launch.  Enabled at K >= 2048
    // (DINODET_GEMM_TAILSPLIT=2 forces every qualifying shape).
    // (with the engine's two concurrent micro-batches the other stream's kernels fill the last-round bubble anyway: fp16x2 at 2 x 32
    // images 1 198 images/s with the split, 1 208 without; so case (b) is kept for the launches that only a single-stream forward of a
    // large batch makes, M >= 65536 rows)
    const int rounds = tiles / CU, rem = tiles % CU;
    // (round 4) a grid one or two rounds deep whose last round is 6-16 % full -- the compensated QKV at 32 images of 224^2: 33 x 9 = 297 tiles --
    // loses most of a tile time to it at any K (bench.py --workload vitb224: bf16x3 4 953 -> 5 015 images/s, fp16x2 5 258 -> 5 318; a fuller or
    // emptier last round, e.g. 2 x 4 images of 518^2, gains nothing).  The plain bf16 kernels cut those rows off instead (gemm_bf16.hip).
    const bool shallow = kind >= 1 && M >= 4096 && rounds >= 1 && rounds <= 2 && rem * 16 >= CU && rem * 6 <= CU;
    if (M < 8192 && !shallow) return -1;
    if (!force && !shallow && (K < 2048 || M < 65536)) return -1;
    if (rounds < 1 || rounds > 6 || rem == 0) return -1;
    const int m_main = (rounds * CU) / tiles_n;
    Mmain = m_main * PPM; R = M - Mmain;
    if (R <= 0) return -1;
    tiles_r = ((R + PPM - 1) / PPM) * tiles_n;
    for (int c : {8, 6, 4, 3, 2})
      if (nk % c == 0 && nk / c >= 3 && tiles_r * c <= CU) { S = c; break; }
    // worth it when the last round shrinks from one tile time to 1/S of one (+ ~1/4 for the slab round trip and the two extra
    // launches) and that is >= 4 % of the kernel: 1029 tiles (4 rounds + 5: S = 8) and 344 tiles (1 round + 88: S = 2) both qualify
    if (S >= 2 && (1.0 - 1.0 / S - 0.25) < 0.04 * (rounds + 1)) S = 0;
  }
  if (S < 2) return -1;
  const size_t slab = (size_t)R * N;
  if (g_tail_bytes[dev] < slab * S * 4) return -1;           // reserved outside stream capture (gemm_tail_reserve)
  float* scratch = tail_slab(dev, s);
  if (!scratch) return -1;
  // ---- main rows: the caller's own dispatch
  if (Mmain > 0) {
    t_in_tail_split = true;
    int rc;
    if (kind == 0) rc = launch_gemm_bf16((const bf16_t*)A, lda, (const bf16_t*)W, ldw, Mmain, N, K, e, s);
    else if (kind == 1) rc = launch_gemm_x3((const bf16_t*)A, lda, (const bf16_t*)W, ldw, Mmain, N, K, e, s);
    else rc = launch_gemm_h2(A, lda, W, ldw, Mmain, N, K, e, s);
    t_in_tail_split = false;
    if (rc) return rc;
  }
  // ---- remainder rows, K-split: fp32 partial slabs
  GemmEpi es; memset(&es, 0, sizeof es);
  es.out_f32 = scratch; es.ldc = N; es.h2_wexp = e.h2_wexp;
  es.ksplit = S; es.kslice_len = K / S; es.kslice_stride = (long long)slab;
  const int gm = gemm_tile_mode() & 0xfff;          // no start stagger
  if (kind == 2) {
    h2_attr();
    const char* Ar = (const char*)A + (size_t)Mmain * lda;
    hipLaunchKernelGGL(gemm_h2_256x256_kernel<LN_NONE>, dim3(tiles_r, S), dim3(512), LDSH2, s, Ar, lda, (const char*)W, ldw, R, N, K, es, gm);
  } else {
    ppm_attr();
    const bf16_t* Ar = (const bf16_t*)A + (size_t)Mmain * lda;
    if (kind == 1) hipLaunchKernelGGL((gemm_ppm_256x256_kernel<true, false, LN_NONE>), dim3(tiles_r, S), dim3(512), LDSPP, s, Ar, lda, (const bf16_t*)W, ldw, R, N, K, es, gm);
    else hipLaunchKernelGGL((gemm_ppm_256x256_kernel<false, false, LN_NONE>), dim3(tiles_r, S), dim3(512), LDSPP, s, Ar, lda, (const bf16_t*)W, ldw, R, N, K, es, gm);
  }
  if (hipGetLastError() != hipSuccess) return 3;
  // ---- reduce + the caller's epilogue on global rows Mmain..M-1
  ++g_tail_splits;
#define GO_(LN_) hipLaunchKernelGGL(gemm_ksplit_reduce_kernel<LN_>, dim3((R + KSR_ROWS - 1) / KSR_ROWS, tiles_n), dim3(512), 0, s, scratch, (long long)slab, S, R, N, e, Mmain, M, K);
  LN_DISPATCH(e, GO_)
#undef GO_
  return hipGetLastError() == hipSuccess ? 0 : 3;
}
