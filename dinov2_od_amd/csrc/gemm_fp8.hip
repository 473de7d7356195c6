// fp8 (OCP e4m3) MFMA GEMM for the ViT-g/14 fp8 configuration (BASELINE.json configs[4]):
//   C[M,N] = (Aq[M,K] * Wq[N,K]^T) * a_scale[m] * w_scale[n]  (+ the usual fused epilogue),
// Aq / Wq one byte per element, K contiguous; per-token scales from the producing LayerNorm / SwiGLU kernel, per-output-
// feature scales from weight packing.  Replaces the same nn.Linear sites as gemm_bf16.hip (modeling_dinov2.py:199-201,
// 281-314) when the handle's precision is fp8.
//
// Same skeleton as the 256x128 bf16 kernel: 8 waves (4 x 2) of 64x64, transposed product (a lane owns an output row),
// LDS-DMA into a 3-slot ring, two workgroups per CU, one raw barrier per K-tile -- but a K-tile is 64 BYTES = 64 fp8
// elements per row (the same 24 KiB of staging now carries twice the K), and one v_mfma_f32_32x32x64_f8f6f4 (2x the
// bf16 MAC rate) consumes a whole K-tile per 32x32 block: lane (row r = lane&31, half h = lane>>5) supplies bytes
// [32h, 32h+32) of row r, i.e. the two 16-B chunks 2h and 2h+1 (swizzled as chunk ^ (r>>2)&3, conflict-free b128 reads).
// Both operands use the same byte->lane map, so the k order inside the instruction is irrelevant to the dot product.
#include "dod_common.h"
#include "gemm_epi.h"
#include <cstdlib>
#include <type_traits>

#define F8M 256
#define F8N 128
#define F8K 64                                  // bytes (= elements) per row per K-tile
#define F8_STAGE ((F8M + F8N) * F8K)            // 24 KiB
#define F8_SLOTS 3

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// SwiGLU epilogue of the weights_in GEMM that QUANTISES what it gates (GemmEpi::out_bs set): interleaved pair columns (x1, x2) -> gated value,
// 64 gated columns per tile row = two blocks of 32 = 16 consecutive threads each; block maximum by four shuffles, e8m0 byte + e4m3 bytes
// straight from the fp32 gate -- the bf16 hidden rows and the row-quantisation pass over them (97 us per ViT-g block) disappear; the next
// GEMM (weights_out) takes the block scales in its MFMAs (gemm_fp8mx_256x128_kernel).  e.out_bf16 = e4m3 rows, pitch e.ldc BYTES (= N / 2).
template <int ROWS, int COLS, int NT, class RowMap>
__device__ __forceinline__ void drain_glu_mx(const char* sm, int pitch, const GemmEpi& e, const ColParams& cp, int M, int N, int n0, int tid,
                                             RowMap rowmap) {
  constexpr int C4 = COLS / 4;
  const int c4 = tid % C4, n = n0 + 4 * c4, F = N >> 1;
  unsigned char* q8 = reinterpret_cast<unsigned char*>(e.out_bf16);
#pragma unroll
  for (int row_l = tid / C4; row_l < ROWS; row_l += NT / C4) {
    const int m = rowmap(row_l);
    const bool ok = m < M && n < N;
    float4 v = *reinterpret_cast<const float4*>(sm + row_l * pitch + c4 * 16);
    const float sa = e.a_scale ? e.a_scale[m < M ? m : M - 1] : 1.0f;
    v.x = v.x * (sa * cp.wscale.x) + cp.bias.x; v.y = v.y * (sa * cp.wscale.y) + cp.bias.y;
    v.z = v.z * (sa * cp.wscale.z) + cp.bias.z; v.w = v.w * (sa * cp.wscale.w) + cp.bias.w;
    const float g0 = ok ? silu_mul(v.x, v.y) : 0.f, g1 = ok ? silu_mul(v.z, v.w) : 0.f;
    float amax = fmaxf(fabsf(g0), fabsf(g1));              // every lane takes part in the shuffles (rows past M contribute zeros)
    amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
    amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
    amax = fmaxf(amax, __shfl_xor(amax, 4, 64));
    amax = fmaxf(amax, __shfl_xor(amax, 8, 64));
    if (!ok) continue;
    const unsigned eb = mx_ebyte(amax);
    const float inv = mx_inv_scale(eb);
    const int gc = n >> 1;                                  // gated column
    *reinterpret_cast<unsigned short*>(q8 + (size_t)m * e.ldc + gc) = (unsigned short)(pack4_fp8(g0 * inv, g1 * inv, 0.f, 0.f) & 0xffffu);
    if ((c4 & 15) == 0) e.out_bs[(size_t)m * (F >> 5) + mx_scale_off(F, gc >> 5)] = (unsigned char)eb;
  }
}

__global__ __launch_bounds__(512, 4) void gemm_fp8_256x128_kernel(const unsigned char* __restrict__ A, int lda,
                                                                  const unsigned char* __restrict__ W, int ldw,
                                                                  int M, int N, int K, GemmEpi e, int GM) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int tiles_m = (M + F8M - 1) / F8M, tiles_n = (N + F8N - 1) / F8N;
  const int nwg = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {   // XCD-aware bijective remap, then GM m-tiles deep groups inside each XCD's run (as in gemm_bf16.hip)
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
  const int m0 = tm * F8M, n0 = tn * F8N;
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  // per K-tile a wave issues 2 pieces of A and 1 of W; piece = 16 rows x 64 B, lane -> (row = lane>>2, chunk = lane&3)
  const unsigned char* gA0; const unsigned char* gA1; const unsigned char* gW0;
  {
    auto src = [&](const unsigned char* base, int ld, int r0, int piece, int lim) {
      const int rl = piece * 16 + (lane >> 2);
      int r = r0 + rl; r = r < lim ? r : lim - 1;
      return base + (size_t)r * ld + swz64(rl, lane & 3) * 16;
    };
    gA0 = src(A, lda, m0, wid * 2, M); gA1 = src(A, lda, m0, wid * 2 + 1, M);
    gW0 = src(W, ldw, n0, wid, N);
  }
  const int wu = __builtin_amdgcn_readfirstlane(wid);
#define STAGE8(slot_, k0)                                                                                  \
  {                                                                                                        \
    char* sA_ = smem + (slot_) * F8_STAGE + wu * 2048;                                                     \
    char* sW_ = smem + (slot_) * F8_STAGE + F8M * F8K + wu * 1024;                                         \
    __builtin_amdgcn_global_load_lds((gptr_t)(gA0 + (k0)), (lptr_t)(sA_), 16, 0, 0);                       \
    __builtin_amdgcn_global_load_lds((gptr_t)(gA1 + (k0)), (lptr_t)(sA_ + 1024), 16, 0, 0);                \
    __builtin_amdgcn_global_load_lds((gptr_t)(gW0 + (k0)), (lptr_t)(sW_), 16, 0, 0);                       \
  }
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int nk = K / F8K;
  const int lr = lane & 31, lh = lane >> 5;
  STAGE8(0, 0)
  if (nk > 1) STAGE8(1, F8K)
  int offA[2], offW[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) { const int row = wm * 64 + i * 32 + lr; offA[i] = row * 64 + swz64(row, 2 * lh) * 16; }
#pragma unroll
  for (int j = 0; j < 2; ++j) { const int row = wn * 64 + j * 32 + lr; offW[j] = F8M * F8K + row * 64 + swz64(row, 2 * lh) * 16; }
  int slot = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (kt + 2 < nk) {
      const int ns = slot >= 1 ? slot - 1 : 2;
      STAGE8(ns, (kt + 2) * F8K)
    }
    const char* st = smem + slot * F8_STAGE;
    i32x8 af[2], wf[2];
#pragma unroll
    // the fragments are read as ushort vectors, like the bf16 kernels: typed as int the reads "may alias" the pending
    // LDS-DMA writes and hipcc drains the ring (vmcnt(0)) in every iteration
    for (int i = 0; i < 2; ++i) {
      const i32x4 lo = __builtin_bit_cast(i32x4, *reinterpret_cast<const bf16x8*>(st + offA[i]));
      const i32x4 hi = __builtin_bit_cast(i32x4, *reinterpret_cast<const bf16x8*>(st + (offA[i] ^ 16)));   // chunk 2h+1 = (2h ^ s) ^ 1
      af[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const i32x4 lo = __builtin_bit_cast(i32x4, *reinterpret_cast<const bf16x8*>(st + offW[j]));
      const i32x4 hi = __builtin_bit_cast(i32x4, *reinterpret_cast<const bf16x8*>(st + (offW[j] ^ 16)));
      wf[j] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)   // cbsz = blgp = 0: both operands e4m3; scale operands 0 -> the unscaled instruction form
        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wf[j], af[i], acc[i][j], 0, 0, 0, 0, 0, 0);
    slot = slot == 2 ? 0 : slot + 1;
  }
  // epilogue: two passes of 128 tile rows through a 128 x 128 fp32 LDS tile; dequant scales applied in drain_tile
  constexpr int PITCH = F8N * 4 + 16;
  const ColParams cp = load_col_params<F8N>(e, n0, N, tid);
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int j = 0; j < 2; ++j)
      stage_acc(smem, PITCH, wm * 32 + lr, wn * 64 + j * 32, acc[pass][j], lh);
    __syncthreads();
    if (e.out_bs) drain_glu_mx<128, F8N, 512>(smem, PITCH, e, cp, M, N, n0, tid, [&](int row_l) { return m0 + (row_l >> 5) * 64 + pass * 32 + (row_l & 31); });
    else drain_tile<128, F8N, 512>(smem, PITCH, e, cp, M, N, n0, tid,
                                   [&](int row_l) { return m0 + (row_l >> 5) * 64 + pass * 32 + (row_l & 31); });
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// The same kernel with BLOCK-SCALED activations (MX style): A carries one e8m0 byte per 32 elements along K (value 2^(byte - 127)) instead
// of one fp32 scale per row, so that a producer can quantise the tile it holds -- the attention kernel its 64 context columns of a head, the
// SwiGLU epilogue its 64 gated columns -- without knowing the row's maximum over ALL columns, which took a separate pass over the row
// (quant_rows_fp8_kernel: 7 % of the ViT-g forward).  v_mfma_scale_f32_32x32x64_f8f6f4 takes exactly that: a lane supplies the 32 bytes of
// one block of its row and, in the scale operand, that block's byte (op_sel picks one of the four bytes of the VGPR).  W keeps its
// per-output-feature fp32 scale (epilogue) and contributes the constant block scale 2^0.
// Scale layout As [M][2][K / 64] (row pitch K / 32 bytes): the bytes of a (row, 32-byte half h of the K-tile) for FOUR consecutive K-tiles are
// one aligned dword, so the eight waves bring the scales of a group of four K-tiles with ONE 4-byte-per-lane LDS-DMA each (256 rows x 2
// halves x 4 B = 2 KiB, double-buffered behind the ring) -- no register load competes with the ring's vmcnt.  K % 256 == 0.
// Which bytes a lane's scale covers was MEASURED (tools/mx_probe.py: one element doubled, one block switched on): the instruction's K block b is
// the FIRST 16 bytes (b = 0) or the LAST 16 bytes (b = 1) of every lane's 32, from both lane halves, and its scale is read from lanes 32b ..
// 32b + 31.  So lane half g takes the 16-byte chunks g and g + 2 of a row's 64-byte K-tile (not 2g and 2g + 1 as the unscaled kernel, where
// any byte order both operands share is a dot product): chunks 0, 1 -- the row's first 32 elements -- are then the instruction's block 0.
// vmcnt per wave and K-tile: 3 ring pieces, plus the scale piece of the NEXT group issued in the first iteration of a group; at the top of
// iteration kt everything issued in iteration kt - 1 may still fly: 3, or 4 when kt - 1 opened a group that has a successor.
// WBS (round 4): W block-scaled too (GemmEpi::w_bs, the same [rows][2][K / 64] layout: quant_mx_fp8_kernel packs the weights): its 128 rows
// x 2 halves are 256 scale dwords per group of four K-tiles -- every wave issues one more 4-byte-per-lane LDS-DMA per group (waves 4..7
// repeat waves 0..3: the same bytes to the same LDS words, so that every wave's vmcnt counts the same pieces) -- and the weight fragment's scale
// operand takes this K-tile's byte instead of the constant 2^0.
#define F8_SC_BYTES 2048
#define F8_SCW_BYTES 1024
template <bool WBS>
__global__ __launch_bounds__(512, 4) void gemm_fp8mx_256x128_kernel(const unsigned char* __restrict__ A, int lda,
                                                                    const unsigned char* __restrict__ W, int ldw,
                                                                    int M, int N, int K, GemmEpi e, int GM) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int tiles_m = (M + F8M - 1) / F8M, tiles_n = (N + F8N - 1) / F8N;
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
  const int m0 = tm * F8M, n0 = tn * F8N;
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const unsigned char* gA0; const unsigned char* gA1; const unsigned char* gW0; const unsigned char* gS; const unsigned char* gSW = nullptr;
  {
    auto src = [&](const unsigned char* base, int ld, int r0, int piece, int lim) {
      const int rl = piece * 16 + (lane >> 2);
      int r = r0 + rl; r = r < lim ? r : lim - 1;
      return base + (size_t)r * ld + swz64(rl, lane & 3) * 16;
    };
    gA0 = src(A, lda, m0, wid * 2, M); gA1 = src(A, lda, m0, wid * 2 + 1, M);
    gW0 = src(W, ldw, n0, wid, N);
    // scale piece of this wave: LDS dword index wid * 64 + lane = tile row * 2 + half
    const int sidx = wid * 64 + lane;
    int r = m0 + (sidx >> 1); r = r < M ? r : M - 1;
    gS = e.a_bs + (size_t)r * (K >> 5) + (size_t)(sidx & 1) * (K >> 6);
    if (WBS) {
      const int sw = (wid & 3) * 64 + lane;
      int rw = n0 + (sw >> 1); rw = rw < N ? rw : N - 1;
      gSW = e.w_bs + (size_t)rw * (K >> 5) + (size_t)(sw & 1) * (K >> 6);
    }
  }
  const int wu = __builtin_amdgcn_readfirstlane(wid);
  char* const sc_base = smem + F8_SLOTS * F8_STAGE;                   // two 2-KiB scale buffers behind the ring
#define STAGE8M(slot_, k0)                                                                                 \
  {                                                                                                        \
    char* sA_ = smem + (slot_) * F8_STAGE + wu * 2048;                                                     \
    char* sW_ = smem + (slot_) * F8_STAGE + F8M * F8K + wu * 1024;                                         \
    __builtin_amdgcn_global_load_lds((gptr_t)(gA0 + (k0)), (lptr_t)(sA_), 16, 0, 0);                       \
    __builtin_amdgcn_global_load_lds((gptr_t)(gA1 + (k0)), (lptr_t)(sA_ + 1024), 16, 0, 0);                \
    __builtin_amdgcn_global_load_lds((gptr_t)(gW0 + (k0)), (lptr_t)(sW_), 16, 0, 0);                       \
  }
#define STAGE8S(g_)                                                                                                                            \
  {                                                                                                                                            \
    __builtin_amdgcn_global_load_lds((gptr_t)(gS + 4 * (g_)), (lptr_t)(sc_base + ((g_) & 1) * F8_SC_BYTES + wu * 256), 4, 0, 0);               \
    if (WBS) __builtin_amdgcn_global_load_lds((gptr_t)(gSW + 4 * (g_)), (lptr_t)(sc_base + 2 * F8_SC_BYTES + ((g_) & 1) * F8_SCW_BYTES + (wu & 3) * 256), 4, 0, 0); \
  }
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int nk = K / F8K, ng = nk >> 2;
  const int lr = lane & 31, lh = lane >> 5;
  STAGE8S(0)
  STAGE8M(0, 0)
  STAGE8M(1, F8K)                      // nk >= 4
  int offA[2], offW[2], offS[2], offSW[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = wm * 64 + i * 32 + lr;
    offA[i] = row * 64 + swz64(row, lh) * 16;
    offS[i] = (row * 2 + lh) * 4;
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) { const int row = wn * 64 + j * 32 + lr; offW[j] = F8M * F8K + row * 64 + swz64(row, lh) * 16; offSW[j] = (row * 2 + lh) * 4; }
  int slot = 0;
  typedef unsigned short us2 __attribute__((ext_vector_type(2)));
  int sca[2] = {0, 0}, scw[2] = {0x7f7f7f7f, 0x7f7f7f7f};
  for (int kt = 0; kt < nk; ++kt) {
    const int u = kt & 3, g = kt >> 2;                 // wave-uniform
    const bool has_next = g + 1 < ng;
    if (kt + 1 < nk) {
      if (u == 1 && has_next) { if (WBS) asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
      else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (u == 0 && has_next) STAGE8S(g + 1)
    if (kt + 2 < nk) {
      const int ns = slot >= 1 ? slot - 1 : 2;
      STAGE8M(ns, (kt + 2) * F8K)
    }
    const char* st = smem + slot * F8_STAGE;
    if (u == 0) {       // this group's scale dwords (read as ushort pairs: an int-typed read "may alias" the pending LDS-DMA writes)
#pragma unroll
      for (int i = 0; i < 2; ++i) sca[i] = __builtin_bit_cast(int, *reinterpret_cast<const us2*>(sc_base + (g & 1) * F8_SC_BYTES + offS[i]));
      if (WBS) {
#pragma unroll
        for (int j = 0; j < 2; ++j) scw[j] = __builtin_bit_cast(int, *reinterpret_cast<const us2*>(sc_base + 2 * F8_SC_BYTES + (g & 1) * F8_SCW_BYTES + offSW[j]));
      }
    }
    i32x8 af[2], wf[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const i32x4 lo = __builtin_bit_cast(i32x4, *reinterpret_cast<const bf16x8*>(st + offA[i]));
      const i32x4 hi = __builtin_bit_cast(i32x4, *reinterpret_cast<const bf16x8*>(st + (offA[i] ^ 32)));
      af[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const i32x4 lo = __builtin_bit_cast(i32x4, *reinterpret_cast<const bf16x8*>(st + offW[j]));
      const i32x4 hi = __builtin_bit_cast(i32x4, *reinterpret_cast<const bf16x8*>(st + (offW[j] ^ 32)));
      wf[j] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int sb = (int)((unsigned)sca[i] >> (8 * u));      // this K-tile's byte into byte 0 (op_sel 0)
#pragma unroll
      for (int j = 0; j < 2; ++j)   // W: its block's byte, or the constant block scale 2^0 (byte 127)
        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wf[j], af[i], acc[i][j], 0, 0, 0, WBS ? (int)((unsigned)scw[j] >> (8 * u)) : 127, 0, sb);
    }
    slot = slot == 2 ? 0 : slot + 1;
  }
#undef STAGE8M
#undef STAGE8S
  constexpr int PITCH = F8N * 4 + 16;
  const ColParams cp = load_col_params<F8N>(e, n0, N, tid);
  const bool wide = drain8_ok(e, N);      // both operands block-scaled, bf16 output: nothing to dequantise -> 16-byte stores
  // the two passes written out: as a `#pragma unroll` loop the body outgrows the unroller's threshold, acc[pass] becomes a scratch array
  // and the kernel spills (104 bytes of scratch per lane); written out, every variant holds its accumulators in registers (0 spills)
#define F8MX_PASS(P)                                                                                                           \
  {                                                                                                                            \
    __builtin_amdgcn_s_barrier();                                                                                              \
    asm volatile("" ::: "memory");                                                                                             \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) stage_acc(smem, PITCH, wm * 32 + lr, wn * 64 + j * 32, acc[P][j], lh);       \
    auto rowmap = [&](int row_l) { return m0 + (row_l >> 5) * 64 + (P) * 32 + (row_l & 31); };                                 \
    __syncthreads();                                                                                                           \
    if (e.out_bs) drain_glu_mx<128, F8N, 512>(smem, PITCH, e, cp, M, N, n0, tid, rowmap);                                      \
    else if (wide) drain_tile_bf16x8<128, F8N, 512>(smem, PITCH, e, M, N, n0, tid, rowmap);                                    \
    else drain_tile<128, F8N, 512>(smem, PITCH, e, cp, M, N, n0, tid, rowmap);                                                 \
  }
  F8MX_PASS(0)
  F8MX_PASS(1)
#undef F8MX_PASS
}

// ---------------------------------------------------------------------------------------------------------------------------
// 256x256 tile, 16 waves (4 x 4, 64x64 each), ONE workgroup per CU, both operands block-scaled (round 4).  Same per-wave arithmetic as the 256x128
// kernel above (2 x 2 blocks of v_mfma_scale_f32_32x32x64_f8f6f4, the same fragment / scale lane maps and scale groups); taken from 4 096 rows up.
// Built for its staged bytes -- (256 + 256) x 64 B per 256 x 256 x 64 MACs, a third fewer per MAC than two co-resident 256x128 workgroups -- and
// credited by the measurement with something else: alone it was 2 % slower, under two concurrent micro-batches 5.5-9 % faster, because
// one-workgroup-per-CU grids of two streams interleave workgroup by workgroup (profiles/r04_fp8_256x256_ab.txt).  First form: 64-byte LDS rows,
// three 32-KiB slots, a barrier per K-tile (commit 31578a1); this one has 128-BYTE rows: one K step = two 64-byte K-tiles (eight MFMAs per wave
// and barrier instead of four), two 64-KiB slots (tile t + 1 streams in while tile t feeds the MFMAs: the bf16 16-wave kernel's protocol,
// gemm_x3.hip PLAIN) -- fp8 GEMM class 72.6 -> 68.8 ms per ViT-g step, 330.7 -> 343.4 images/s on one box (tools/experiments/r4_exp32.sh).
// The fragment of K-tile kk of a step is chunks 4 kk + g and 4 kk + g + 2 of the row (g = the lane half; MX lane map as above); with the
// 128-byte-row swizzle (chunk ^= (row >> 1) & 7) those are the row's base offset XOR 64 kk and XOR 64 kk + 32.  A scale group (four K-tiles) is
// two steps: waves 0-7 bring the A scales, waves 8-15 the W scales (one 4-byte-per-lane piece per group each).
#define F8BN 256
#define F8CK 128
#define F8C_STAGE ((F8M + F8BN) * F8CK)         // 64 KiB
__global__ __launch_bounds__(1024, 1) void gemm_fp8mx_256x256x128_kernel(const unsigned char* __restrict__ A, int lda,
                                                                         const unsigned char* __restrict__ W, int ldw,
                                                                         int M, int N, int K, GemmEpi e, int GM) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 2, wn = wid & 3;
  const int tiles_m = (M + F8M - 1) / F8M, tiles_n = (N + F8BN - 1) / F8BN;
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
  const int m0 = tm * F8M, n0 = tn * F8BN;
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const unsigned char* gA0; const unsigned char* gA1; const unsigned char* gW0; const unsigned char* gW1; const unsigned char* gS;
  {
    auto src = [&](const unsigned char* base, int ld, int r0, int piece, int lim) {      // piece = 8 rows x 128 B; the swizzle goes on the source chunk
      const int rl = piece * 8 + (lane >> 3);
      int r = r0 + rl; r = r < lim ? r : lim - 1;
      return base + (size_t)r * ld + (((lane & 7) ^ ((rl >> 1) & 7)) * 16);
    };
    gA0 = src(A, lda, m0, wid * 2, M); gA1 = src(A, lda, m0, wid * 2 + 1, M);
    gW0 = src(W, ldw, n0, wid * 2, N); gW1 = src(W, ldw, n0, wid * 2 + 1, N);
    const int sidx = (wid & 7) * 64 + lane;
    if (wid < 8) { int r = m0 + (sidx >> 1); r = r < M ? r : M - 1; gS = e.a_bs + (size_t)r * (K >> 5) + (size_t)(sidx & 1) * (K >> 6); }
    else { int r = n0 + (sidx >> 1); r = r < N ? r : N - 1; gS = e.w_bs + (size_t)r * (K >> 5) + (size_t)(sidx & 1) * (K >> 6); }
  }
  const int wu = __builtin_amdgcn_readfirstlane(wid);
  char* const sc_base = smem + 2 * F8C_STAGE;
#define STAGE8C(slot_, k0)                                                                                 \
  {                                                                                                        \
    char* sA_ = smem + (slot_) * F8C_STAGE + wu * 2048;                                                    \
    char* sW_ = smem + (slot_) * F8C_STAGE + F8M * F8CK + wu * 2048;                                       \
    __builtin_amdgcn_global_load_lds((gptr_t)(gA0 + (k0)), (lptr_t)(sA_), 16, 0, 0);                       \
    __builtin_amdgcn_global_load_lds((gptr_t)(gA1 + (k0)), (lptr_t)(sA_ + 1024), 16, 0, 0);                \
    __builtin_amdgcn_global_load_lds((gptr_t)(gW0 + (k0)), (lptr_t)(sW_), 16, 0, 0);                       \
    __builtin_amdgcn_global_load_lds((gptr_t)(gW1 + (k0)), (lptr_t)(sW_ + 1024), 16, 0, 0);                \
  }
#define STAGE8CS(g_)                                                                                                                           \
  __builtin_amdgcn_global_load_lds((gptr_t)(gS + 4 * (g_)), (lptr_t)(sc_base + (wu >> 3) * (2 * F8_SC_BYTES) + ((g_) & 1) * F8_SC_BYTES + (wu & 7) * 256), 4, 0, 0);
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int nks = K / F8CK, ng = nks >> 1;           // K % 256 == 0: whole scale groups
  const int lr = lane & 31, lh = lane >> 5;
  STAGE8CS(0)
  STAGE8C(0, 0)
  int offA[2], offW[2], offS[2], offSW[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = wm * 64 + i * 32 + lr;
    offA[i] = row * 128 + ((lh ^ ((row >> 1) & 7)) * 16);
    offS[i] = (row * 2 + lh) * 4;
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = wn * 64 + j * 32 + lr;
    offW[j] = F8M * F8CK + row * 128 + ((lh ^ ((row >> 1) & 7)) * 16);
    offSW[j] = 2 * F8_SC_BYTES + (row * 2 + lh) * 4;
  }
  typedef unsigned short us2 __attribute__((ext_vector_type(2)));
  int sca[2] = {0, 0}, scw[2] = {0x7f7f7f7f, 0x7f7f7f7f};
  for (int ks = 0; ks < nks; ++ks) {
    const int g = ks >> 1, half = ks & 1;              // wave-uniform
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // tile ks (and, at a group's first step, its scale dwords) has landed
    __builtin_amdgcn_s_barrier();                      // ... for every wave; and every wave is done with tile ks - 1: its slot may be refilled
    asm volatile("" ::: "memory");
    if (half == 0 && g + 1 < ng) STAGE8CS(g + 1)
    if (ks + 1 < nks) STAGE8C((ks + 1) & 1, (ks + 1) * F8CK)
    const char* st = smem + (ks & 1) * F8C_STAGE;
    if (half == 0) {
#pragma unroll
      for (int i = 0; i < 2; ++i) sca[i] = __builtin_bit_cast(int, *reinterpret_cast<const us2*>(sc_base + (g & 1) * F8_SC_BYTES + offS[i]));
#pragma unroll
      for (int j = 0; j < 2; ++j) scw[j] = __builtin_bit_cast(int, *reinterpret_cast<const us2*>(sc_base + (g & 1) * F8_SC_BYTES + offSW[j]));
    }
#pragma unroll 1      // (unrolled, hipcc hoists both K-tiles' fragments above the first MFMA: 64 fragment + 64 accumulator registers, and spills)
    for (int kk = 0; kk < 2; ++kk) {
      const int u = half * 2 + kk;                     // K-tile within the scale group
      i32x8 af[2], wf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const i32x4 lo = __builtin_bit_cast(i32x4, *reinterpret_cast<const bf16x8*>(st + (offA[i] ^ (kk * 64))));
        const i32x4 hi = __builtin_bit_cast(i32x4, *reinterpret_cast<const bf16x8*>(st + (offA[i] ^ (kk * 64 + 32))));
        af[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const i32x4 lo = __builtin_bit_cast(i32x4, *reinterpret_cast<const bf16x8*>(st + (offW[j] ^ (kk * 64))));
        const i32x4 hi = __builtin_bit_cast(i32x4, *reinterpret_cast<const bf16x8*>(st + (offW[j] ^ (kk * 64 + 32))));
        wf[j] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int sb = (int)((unsigned)sca[i] >> (8 * u));
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wf[j], af[i], acc[i][j], 0, 0, 0, (int)((unsigned)scw[j] >> (8 * u)), 0, sb);
      }
    }
  }
#undef STAGE8C
#undef STAGE8CS
  constexpr int PITCH = F8BN * 4 + 16;
  const ColParams cp = load_col_params<F8BN>(e, n0, N, tid);
  const bool wide = drain8_ok(e, N);
#define F8C_PASS(P)                                                                                                            \
  {                                                                                                                            \
    __builtin_amdgcn_s_barrier();                                                                                              \
    asm volatile("" ::: "memory");                                                                                             \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) stage_acc(smem, PITCH, wm * 32 + lr, wn * 64 + j * 32, acc[P][j], lh);       \
    auto rowmap = [&](int row_l) { return m0 + (row_l >> 5) * 64 + (P) * 32 + (row_l & 31); };                                 \
    __syncthreads();                                                                                                           \
    if (e.out_bs) drain_glu_mx<128, F8BN, 1024>(smem, PITCH, e, cp, M, N, n0, tid, rowmap);                                    \
    else if (wide) drain_tile_bf16x8<128, F8BN, 1024>(smem, PITCH, e, M, N, n0, tid, rowmap);                                  \
    else drain_tile<128, F8BN, 1024>(smem, PITCH, e, cp, M, N, n0, tid, rowmap);                                               \
  }
  F8C_PASS(0)
  F8C_PASS(1)
#undef F8C_PASS
}
static constexpr int LDS8C = 2 * F8C_STAGE + 4 * F8_SC_BYTES;      // 136 KiB (>= the epilogue's 133 KiB fp32 tile)

static constexpr int LDS8MX = F8_SLOTS * F8_STAGE + 2 * F8_SC_BYTES + 2 * F8_SCW_BYTES;
static constexpr int LDS8 = (128 * (F8N * 4 + 16)) > F8_SLOTS * F8_STAGE ? (128 * (F8N * 4 + 16)) : F8_SLOTS * F8_STAGE;

static void fp8_attr() {      // > 64 KiB of dynamic LDS: once per DEVICE (a process may drive several)
  static bool attr_set[16] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 16 && !attr_set[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_fp8_256x128_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS8);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_fp8mx_256x128_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS8MX);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_fp8mx_256x128_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS8MX);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_fp8mx_256x256x128_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS8C);
    attr_set[dev] = true;
  }
}

int launch_gemm_fp8(const unsigned char* A, int lda, const unsigned char* W, int ldw, int M, int N, int K,
                    const GemmEpi& e, hipStream_t s) {
  if (M <= 0 || N <= 0 || K <= 0) return 1;
  if (K % F8K != 0 || N % 4 != 0 || lda % 16 != 0 || ldw % 16 != 0 || e.ldc % 4 != 0) return 2;
  if ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(W)) & 15) return 2;
  if (e.resid && e.ldr % 4 != 0) return 2;
  if (!e.out_f32 && !e.out_bf16) return 2;
  if ((!e.w_scale && !e.w_bs) || (!e.a_scale && !e.a_bs) || (e.w_bs && !e.a_bs)) return 2;
  if (e.out_bs && (!e.glu || !e.out_bf16 || N % 128 != 0 || e.scale || e.resid || e.act != ACT_NONE)) return 2;
  if (e.a_bs) {          // block-scaled activations
    if (K % 256 != 0) return 2;
    if ((reinterpret_cast<uintptr_t>(e.a_bs) | reinterpret_cast<uintptr_t>(e.w_bs)) & 3) return 2;      // the scale bytes arrive by 4-byte LDS-DMA
    fp8_attr();
    const int gm = N >= 3072 ? 8 : (N >= 2048 ? 4 : 2);
    const int tiles = ((M + F8M - 1) / F8M) * ((N + F8N - 1) / F8N);
    // both operands block-scaled, a grid of several rounds of 256x256 tiles: the one-workgroup-per-CU tile (fewer staged bytes per MAC)
    static const int big = [] { const char* v = DOD_TUNE_ENV("DINODET_FP8_TILE"); return v ? atoi(v) : 1; }();      // tuning builds: 0 = 256x128 everywhere
    if (e.w_bs && big && !(big == 2 && e.out_bs) && M >= 4096 && N >= 512) {      // (tuning value 2: not for the gated weights_in)
      const int tiles_b = ((M + F8M - 1) / F8M) * ((N + F8BN - 1) / F8BN);
      hipLaunchKernelGGL(gemm_fp8mx_256x256x128_kernel, dim3(tiles_b), dim3(1024), LDS8C, s, A, lda, W, ldw, M, N, K, e, gm);
      return hipGetLastError() == hipSuccess ? 0 : 3;
    }
    if (e.w_bs) hipLaunchKernelGGL(gemm_fp8mx_256x128_kernel<true>, dim3(tiles), dim3(512), LDS8MX, s, A, lda, W, ldw, M, N, K, e, gm);
    else hipLaunchKernelGGL(gemm_fp8mx_256x128_kernel<false>, dim3(tiles), dim3(512), LDS8MX, s, A, lda, W, ldw, M, N, K, e, gm);
    return hipGetLastError() == hipSuccess ? 0 : 3;
  }
  fp8_attr();
  const int gm = N >= 3072 ? 8 : (N >= 2048 ? 4 : 2);
  const int tiles = ((M + F8M - 1) / F8M) * ((N + F8N - 1) / F8N);
  hipLaunchKernelGGL(gemm_fp8_256x128_kernel, dim3(tiles), dim3(512), LDS8, s, A, lda, W, ldw, M, N, K, e, gm);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}

// ---------------------------------------------------------------------------------------------- row quantisation
// One wave per row: amax by wave reduction, scale = amax / 448 (e4m3 max finite), q = rne(x / scale).
template <bool IN_BF16>
__global__ __launch_bounds__(256) void quant_rows_fp8_kernel(const void* __restrict__ xin, int ld, int rows, int cols,
                                                             unsigned char* __restrict__ q, int ldq,
                                                             float* __restrict__ scale) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  auto ld4 = [&](int c) -> float4 {
    if (IN_BF16) {
      const uint2 u = *reinterpret_cast<const uint2*>((const bf16_t*)xin + (size_t)row * ld + c);
      return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                         __uint_as_float(u.y & 0xffff0000u));
    }
    return *reinterpret_cast<const float4*>((const float*)xin + (size_t)row * ld + c);
  };
  float amax = 0.f;
  for (int c = lane * 4; c < cols; c += 256) {
    const float4 v = ld4(c);
    amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
  }
  amax = wave_max(amax);
  const float sc = amax > 0.f ? amax / 448.0f : 1.0f;
  const float inv = 1.0f / sc;
  if (lane == 0) scale[row] = sc;
  for (int c = lane * 4; c < cols; c += 256) {
    const float4 v = ld4(c);
    *reinterpret_cast<unsigned*>(q + (size_t)row * ldq + c) = pack4_fp8(v.x * inv, v.y * inv, v.z * inv, v.w * inv);
  }
}

// Block-scaled quantisation of whole rows (dod_common.h "MX"): one wave per row, a block of 32 elements = 8 lanes x 4.  The forward's
// producers (attention epilogue, SwiGLU epilogue) quantise their own tiles with the same two helpers; this kernel is the operator-level
// form (tests, tools) and the fallback where a producer cannot (cols % 32 == 0).
template <bool IN_BF16>
__global__ __launch_bounds__(256) void quant_mx_fp8_kernel(const void* __restrict__ xin, int ld, int rows, int cols,
                                                           unsigned char* __restrict__ q, int ldq, unsigned char* __restrict__ bs) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  for (int c0 = 0; c0 < cols; c0 += 256) {
    const int c = c0 + lane * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < cols) {
      if (IN_BF16) {
        const uint2 u = *reinterpret_cast<const uint2*>((const bf16_t*)xin + (size_t)row * ld + c);
        v = make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
      } else v = *reinterpret_cast<const float4*>((const float*)xin + (size_t)row * ld + c);
    }
    float amax = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
    amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
    amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
    amax = fmaxf(amax, __shfl_xor(amax, 4, 64));
    const unsigned eb = mx_ebyte(amax);
    const float inv = mx_inv_scale(eb);
    if (c < cols) {
      *reinterpret_cast<unsigned*>(q + (size_t)row * ldq + c) = pack4_fp8(v.x * inv, v.y * inv, v.z * inv, v.w * inv);
      if ((lane & 7) == 0) bs[(size_t)row * (cols >> 5) + mx_scale_off(cols, c >> 5)] = (unsigned char)eb;
    }
  }
}
int launch_quant_mx_fp8(const void* x, int in_bf16, int ld, int rows, int cols, unsigned char* q, int ldq, unsigned char* bs, hipStream_t s) {
  if (rows <= 0 || cols <= 0) return 1;
  if (cols % 64 != 0 || ld % 4 != 0 || ldq % 4 != 0) return 2;
  const dim3 grid((rows + 3) / 4), block(256);
  if (in_bf16) hipLaunchKernelGGL(quant_mx_fp8_kernel<true>, grid, block, 0, s, x, ld, rows, cols, q, ldq, bs);
  else hipLaunchKernelGGL(quant_mx_fp8_kernel<false>, grid, block, 0, s, x, ld, rows, cols, q, ldq, bs);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}

// bf16 rows of at most 4096 columns (the ViT-g forward's two per-block passes: the attention output, 1536 wide, and the gated SwiGLU output,
// 4096 wide): the row stays in registers between the amax and the quantisation -- ONE read, in 16-byte loads (8 bf16 per lane and load,
// NV <= 8 of them), 8-byte stores -- instead of two passes of 8-byte loads.  Same arithmetic as the generic kernel: the same bits.
template <int NV>
__global__ __launch_bounds__(256) void quant_rows_fp8_reg_kernel(const bf16_t* __restrict__ xin, int ld, int rows, int cols,
                                                                 unsigned char* __restrict__ q, int ldq, float* __restrict__ scale) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const bf16_t* xr = xin + (size_t)row * ld;
  uint4 v[NV];
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * 8;
    v[i] = make_uint4(0u, 0u, 0u, 0u);
    if (c < cols) v[i] = *reinterpret_cast<const uint4*>(xr + c);
    const unsigned w[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
    for (int j = 0; j < 4; ++j) amax = fmaxf(amax, fmaxf(fabsf(__uint_as_float(w[j] << 16)), fabsf(__uint_as_float(w[j] & 0xffff0000u))));
  }
  amax = wave_max(amax);
  const float sc = amax > 0.f ? amax / 448.0f : 1.0f;
  const float inv = 1.0f / sc;
  if (lane == 0) scale[row] = sc;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * 8;
    if (c >= cols) continue;
    uint2 o;
    o.x = pack4_fp8(__uint_as_float(v[i].x << 16) * inv, __uint_as_float(v[i].x & 0xffff0000u) * inv, __uint_as_float(v[i].y << 16) * inv, __uint_as_float(v[i].y & 0xffff0000u) * inv);
    o.y = pack4_fp8(__uint_as_float(v[i].z << 16) * inv, __uint_as_float(v[i].z & 0xffff0000u) * inv, __uint_as_float(v[i].w << 16) * inv, __uint_as_float(v[i].w & 0xffff0000u) * inv);
    *reinterpret_cast<uint2*>(q + (size_t)row * ldq + c) = o;
  }
}

int launch_quant_rows_fp8(const void* x, int in_bf16, int ld, int rows, int cols, unsigned char* q, int ldq, float* scale,
                          hipStream_t s) {
  if (rows <= 0 || cols <= 0) return 1;
  if (cols % 4 != 0 || ld % 4 != 0 || ldq % 4 != 0) return 2;
  const int blocks = (rows + 3) / 4;
  if (in_bf16 && cols % 8 == 0 && cols <= 4096 && ld % 8 == 0 && ldq % 8 == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(q)) & 15) == 0) {
    const int nv = (cols + 511) / 512;
#define QR_GO(N) hipLaunchKernelGGL(quant_rows_fp8_reg_kernel<N>, dim3(blocks), dim3(256), 0, s, (const bf16_t*)x, ld, rows, cols, q, ldq, scale)
    if (nv <= 1) QR_GO(1); else if (nv <= 2) QR_GO(2); else if (nv <= 3) QR_GO(3); else if (nv <= 4) QR_GO(4); else QR_GO(8);
#undef QR_GO
    return hipGetLastError() == hipSuccess ? 0 : 3;
  }
  if (in_bf16) hipLaunchKernelGGL(quant_rows_fp8_kernel<true>, dim3(blocks), dim3(256), 0, s, x, ld, rows, cols, q, ldq, scale);
  else hipLaunchKernelGGL(quant_rows_fp8_kernel<false>, dim3(blocks), dim3(256), 0, s, x, ld, rows, cols, q, ldq, scale);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}

// SwiGLU -> fp8 rows (ViT-g MLP, modeling_dinov2.py:310-314): one wave per row; silu(a)*b is kept in registers between
// the amax pass and the quantisation pass (Fh <= 4096: 16 float4 per lane), so the 4*Fh input bytes are read once
// (the two-pass form read them twice and cost 300 us per layer at 43 840 x 4096: 10 % of the ViT-g fp8 step).
#define SW8_MAXC 16
__global__ __launch_bounds__(256) void swiglu_fp8_kernel(const bf16_t* __restrict__ in, int rows, int Fh,
                                                         unsigned char* __restrict__ q, float* __restrict__ scale) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const bf16_t* pa = in + (size_t)row * 2 * Fh;
  const bf16_t* pb = pa + Fh;
  auto val4 = [&](int c) -> float4 {
    const uint2 ua = *reinterpret_cast<const uint2*>(pa + c), ub = *reinterpret_cast<const uint2*>(pb + c);
    const float a0 = __uint_as_float(ua.x << 16), a1 = __uint_as_float(ua.x & 0xffff0000u), a2 = __uint_as_float(ua.y << 16), a3 = __uint_as_float(ua.y & 0xffff0000u);
    const float b0 = __uint_as_float(ub.x << 16), b1 = __uint_as_float(ub.x & 0xffff0000u), b2 = __uint_as_float(ub.y << 16), b3 = __uint_as_float(ub.y & 0xffff0000u);
    return make_float4(a0 / (1.0f + expf(-a0)) * b0, a1 / (1.0f + expf(-a1)) * b1, a2 / (1.0f + expf(-a2)) * b2, a3 / (1.0f + expf(-a3)) * b3);
  };
  const bool cached = Fh <= SW8_MAXC * 256;      // wave-uniform
  float4 v[SW8_MAXC];
  float amax = 0.f;
  if (cached) {
#pragma unroll
    for (int i = 0; i < SW8_MAXC; ++i) {
      const int c = lane * 4 + 256 * i;
      if (c < Fh) {
        v[i] = val4(c);
        amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v[i].x), fabsf(v[i].y)), fmaxf(fabsf(v[i].z), fabsf(v[i].w))));
      }
    }
  } else {
    for (int c = lane * 4; c < Fh; c += 256) {
      const float4 t = val4(c);
      amax = fmaxf(amax, fmaxf(fmaxf(fabsf(t.x), fabsf(t.y)), fmaxf(fabsf(t.z), fabsf(t.w))));
    }
  }
  amax = wave_max(amax);
  const float sc = amax > 0.f ? amax / 448.0f : 1.0f;
  const float inv = 1.0f / sc;
  if (lane == 0) scale[row] = sc;
  if (cached) {
#pragma unroll
    for (int i = 0; i < SW8_MAXC; ++i) {
      const int c = lane * 4 + 256 * i;
      if (c < Fh) *reinterpret_cast<unsigned*>(q + (size_t)row * Fh + c) = pack4_fp8(v[i].x * inv, v[i].y * inv, v[i].z * inv, v[i].w * inv);
    }
  } else {
    for (int c = lane * 4; c < Fh; c += 256) {
      const float4 t = val4(c);
      *reinterpret_cast<unsigned*>(q + (size_t)row * Fh + c) = pack4_fp8(t.x * inv, t.y * inv, t.z * inv, t.w * inv);
    }
  }
}
int launch_swiglu_fp8(const bf16_t* in, int rows, int Fh, unsigned char* q, float* scale, hipStream_t s) {
  if (rows <= 0 || Fh <= 0) return 1;
  if (Fh % 4 != 0) return 2;
  hipLaunchKernelGGL(swiglu_fp8_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, in, rows, Fh, q, scale);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}
