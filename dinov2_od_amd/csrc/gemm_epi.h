// Shared GEMM epilogue (bf16 and fp8 MFMA kernels): accumulators -> LDS tile -> whole-row HBM stores.
#pragma once
#include "dod_common.h"

// 64-byte LDS rows read by 32-row MFMA lane groups: 16-B chunk ^= (row >> 2) & 3 (conflict-free ds_read_b128)
__device__ __forceinline__ int swz64(int row, int chunk) { return chunk ^ ((row >> 2) & 3); }

// ---- workgroup -> output tile map (256x256 kernels) -----------------------------------------------------------------------
// mode bits: low 8 = GM (m-tiles per group), 0x100 = reverse the m order, 0x200 = time-ordered ("chunked") map.
// Classic map: each XCD (blocks b, b+8, ... share an L2) gets ONE contiguous run of the tile list, walked in groups of GM m-tiles
// x all n-tiles: the 8 XCDs sweep 8 different eighths of M at the same time.
// Chunked map: the tile list is cut into chunks of 8 groups; group k of a chunk runs on XCD k, chunks run one after the other, so
// the whole chip moves through M together -- in the order the producer kernel wrote the A rows (or, with the reverse bit, most
// recently written rows first: those are the ones still in the 256-MiB Infinity Cache).  Both maps are bijections for any grid.
// (tuning builds) Start stagger (mode bits 16..27 = delay step in units of 0.16 us, bits 12..15 = groups - 1): the workgroups of the first round
// (one per CU) in group k = (b / 8) % groups start k steps late, so that the CUs' HBM-bound epilogues -- which otherwise all run at
// the same instants, between compute-only K loops -- interleave with the other groups' K loops.
#ifdef DINODET_TUNING
__device__ __forceinline__ void stagger_start(int b, int mode) {
  const int sd = (mode >> 16) & 0xfff;
  if (sd == 0 || b >= 256) return;
  const int k = (b >> 3) % (((mode >> 12) & 0xf) + 1);
  if (k == 0) return;
  const unsigned long long until = __builtin_amdgcn_s_memrealtime() + (unsigned long long)k * sd * 16;
  while (__builtin_amdgcn_s_memrealtime() < until) __builtin_amdgcn_s_sleep(16);
}

#endif

__device__ __forceinline__ void tile_map(int b, int tiles_m, int tiles_n, int mode, int* tm_out, int* tn_out) {
#ifdef DINODET_TUNING
  stagger_start(b, mode);
#endif
  const int GM = mode & 0xff;
  const int nwg = tiles_m * tiles_n;
  int tm, tn;
#ifdef DINODET_TUNING
  if (mode & 0x400) {
    // weight-resident map (tiles_n even; grid = 8 * ceil(tiles_m / 4) * tiles_n / 2, padded: tm = -1 -> the workgroup exits): XCD x owns
    // n-half x & 1 -- half the weight matrix stays in its L2 -- and every fourth m-tile; it walks n fastest, so the tiles that share an
    // A panel run at the same time and the panel streams through the L2 once per half
    const int xcd = b & 7, idx = b >> 3, hn = tiles_n >> 1;
    tm = (idx / hn) * 4 + (xcd >> 1);
    tn = (xcd & 1) * hn + idx % hn;
    if (tm >= tiles_m) tm = -1;
  } else
#endif
  if (mode & 0x200) {
    const int per_group = GM * tiles_n;
    const int nfull = tiles_m / (8 * GM);                  // full chunks
    const int body = nfull * 8 * per_group;
    if (b < body) {
      const int xcd = b & 7, idx = b >> 3;
      const int chunk = idx / per_group, in_g = idx - chunk * per_group;
      tm = (chunk * 8 + xcd) * GM + in_g % GM;
      tn = in_g / GM;
    } else {                                               // ragged tail: fewer than 8 GM m-tiles
      const int t = b - body, R = tiles_m - nfull * 8 * GM;
      tm = nfull * 8 * GM + t % R;
      tn = t / R;
    }
  } else {
    int bid = b;
    {
      const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
      bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int per_group = GM * tiles_n;
    const int grp = bid / per_group, first_m = grp * GM;
    const int gsz = (tiles_m - first_m) < GM ? (tiles_m - first_m) : GM;
    const int in_g = bid - grp * per_group;
    tm = first_m + in_g % gsz;
    tn = in_g / gsz;
  }
  if ((mode & 0x100) && tm >= 0) tm = tiles_m - 1 - tm;
  *tm_out = tm; *tn_out = tn;
}

__device__ __forceinline__ float silu_mul(float a, float b) { return a / (1.0f + expf(-a)) * b; }     // Dinov2SwiGLUFFN: silu(x1) * x2

// ---- epilogue, staged through LDS so that HBM sees whole rows ---------------------------------------
// Phase 1 (stage_acc): a lane owns output row m and its register quads 4 consecutive n: it applies
// bias / activation / LayerScale (float4 per-n parameter reads) and writes float4s into an fp32 LDS
// tile [rows][cols] with a row pitch of cols*4 + 16 bytes (conflict-free ds_write_b128).
// Phase 2 (drain_tile): every wave instruction moves ONE whole tile row: 16 B per lane from LDS, the
// coalesced fp32 residual / position rows from HBM, and a coalesced store (1 KiB fp32 or 512 B bf16 per
// instruction) -- instead of 32 rows x 16-B fragments per store instruction straight from the
// accumulator layout, which made the epilogue cost 2x its HBM time.
__device__ __forceinline__ void stage_acc(char* sm, int pitch, int row_l, int col_l, const f32x16& a, int lh) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int nl = col_l + 8 * g + 4 * lh;
    *reinterpret_cast<float4*>(sm + row_l * pitch + nl * 4) = make_float4(a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]);
  }
}

// LN (compile time; GemmEpi::ln_*): LN_NONE, LN_CONS = the folded-LayerNorm consumer (QKV, fc1 / weights_in), LN_PROD = the producer (the
// in-place residual epilogue of out-proj / fc2).  A kernel template parameter, so that the kernels without a fold compile exactly as before.
enum { LN_NONE = 0, LN_CONS = 1, LN_PROD = 2 };
// Folded LayerNorm: the per-row scalars the drain needs -- (mean, rstd) for the consumer, the shift for the producer -- ride in the 16 bytes of
// padding behind each row of the staged tile: ONE vector-memory instruction per pass (threads 0..ROWS-1) instead of one per row visit.  The
// epilogues are paced by their vector-memory INSTRUCTIONS, not their bytes: fetched per visit, the statistics cost the QKV GEMM +42 us and
// fc1 +32 us (profiles/r04_lnfold_ab.txt).  Call between staging the accumulators and the barrier in front of the drain.
// The consumer also FINISHES the statistics (GemmEpi::ln_part_in): mean = k + S / D, var = Q / D - (S / D)^2 from the producer's group sums of
// (x - k), (x - k)^2 -- S / D is the change of the mean, a fraction of the row's spread, so nothing cancels -- D = this GEMM's K; every tile
// recomputes its rows' pair (a dozen 8-byte loads per row and pass), the tiles of column 0 publish it for the next producer's shift.
template <int ROWS, int COLS, int LN, typename RowMap>
__device__ __forceinline__ void stage_row_stats(char* sm, int pitch, const GemmEpi& e, int M, int tid, RowMap rowmap, int D = 0, int n0 = 0) {
  if (LN == LN_NONE) return;
  if (tid < ROWS) {
    const float2* src = LN == LN_CONS ? e.ln_stats : e.ln_shift;
    const int m = rowmap(tid);
    float2 st = make_float2(0.f, 0.f);
    if (src && m < M) {
      st = src[m];
      if (LN == LN_CONS && e.ln_part_in) {
        const float2* p = e.ln_part_in + (size_t)m * e.ln_npart;
        float S = 0.f, Q = 0.f;
        for (int g = 0; g < e.ln_npart; ++g) { const float2 pg = p[g]; S += pg.x; Q += pg.y; }
        const float dm = S / (float)D;
        st = make_float2(st.x + dm, 1.0f / sqrtf(fmaxf(Q / (float)D - dm * dm, 0.f) + e.ln_eps));
        if (n0 == 0 && e.ln_stats_out) e.ln_stats_out[m] = st;
      }
    }
    *reinterpret_cast<float2*>(sm + tid * pitch + COLS * 4) = st;
  }
}

// Per-thread epilogue parameters: with COLS/4 dividing the thread count, a thread drains the SAME four
// columns of every row it visits, so bias / LayerScale are two float4 registers loaded once per tile.
struct ColParams { float4 bias, scale, wscale; };   // wscale: per-output-feature dequant scale of an fp8 weight
__host__ __device__ __forceinline__ int ln_mode_of(const GemmEpi& e) { return e.ln_part ? LN_PROD : (e.ln_stats ? LN_CONS : LN_NONE); }
#define LN_DISPATCH(e_, GO_)                                          \
  switch (ln_mode_of(e_)) {                                           \
    case LN_PROD: { GO_(LN_PROD) } break;                             \
    case LN_CONS: { GO_(LN_CONS) } break;                             \
    default: { GO_(LN_NONE) } break;                                  \
  }

template <int COLS, int LN = LN_NONE>
__device__ __forceinline__ ColParams load_col_params(const GemmEpi& e, int n0, int N, int tid) {
  ColParams c;
  const int n = n0 + 4 * (tid % (COLS / 4));
  const bool ok = n < N;
  c.bias = (e.bias && ok) ? *reinterpret_cast<const float4*>(e.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
  const float* sc = LN == LN_CONS ? e.ln_c : e.scale;      // folded LayerNorm (never together with a LayerScale): the column sums c[n]
  c.scale = (sc && ok) ? *reinterpret_cast<const float4*>(sc + n) : make_float4(1.f, 1.f, 1.f, 1.f);
  c.wscale = (e.w_scale && ok) ? *reinterpret_cast<const float4*>(e.w_scale + n) : make_float4(1.f, 1.f, 1.f, 1.f);
  return c;
}

// Sum over a 32-lane half of the wave (= the 128 output columns of a row's statistics group) on DPP adds -- no LDS crossbar, no waits:
// quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror leave every lane with the sum of its 16-lane row; row_bcast:15 into rows 1
// and 3 adds the row before.  COMPLETE IN LANES 16..31 AND 48..63 ONLY (the writer lane of a group is its lane 16).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_pick(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, true));
}
__device__ __forceinline__ float half32_sum_hi(float v) {
  v += dpp_pick<0xB1, 0xf>(v);
  v += dpp_pick<0x4E, 0xf>(v);
  v += dpp_pick<0x141, 0xf>(v);
  v += dpp_pick<0x140, 0xf>(v);
  v += dpp_pick<0x142, 0xa>(v);
  return v;
}
// the new residual row in the operand format of the GEMM that reads it next (GemmEpi::ln_op; the writers of rowops.hip's LayerNorm)
__device__ __forceinline__ void ln_store_op(const GemmEpi& e, size_t m, int n, int N, const float4& v) {
  if (e.ln_op_kind == LNOP_H2) {
    uint2 f16; unsigned hi8, lo8;
    h2_quad(v, 1.0f, f16, hi8, lo8);
    char* row = reinterpret_cast<char*>(e.ln_op) + m * (size_t)e.ln_op_ld * 2;
    *reinterpret_cast<uint2*>(row + 2 * n) = f16;
    char* p8 = row + h2_off8(N, n);
    *reinterpret_cast<unsigned*>(p8) = hi8;
    *reinterpret_cast<unsigned*>(p8 + 16) = lo8;
    return;
  }
  uint2 hi;
  hi.x = pack2bf(v.x, v.y);
  hi.y = pack2bf(v.z, v.w);
  bf16_t* o = reinterpret_cast<bf16_t*>(e.ln_op) + m * (size_t)e.ln_op_ld + n;
  *reinterpret_cast<uint2*>(o) = hi;
  if (e.ln_op_kind == LNOP_PAIR) {
    uint2 lo;
    lo.x = pack2bf(v.x - __uint_as_float(hi.x << 16), v.y - __uint_as_float(hi.x & 0xffff0000u));
    lo.y = pack2bf(v.z - __uint_as_float(hi.y << 16), v.w - __uint_as_float(hi.y & 0xffff0000u));
    *reinterpret_cast<uint2*>(o + (e.ln_op_ld >> 1)) = lo;
  }
}

// LNF: the folded-LayerNorm producer (GemmEpi::ln_part): every lane stays in the loop (columns n >= N contribute nothing) because the row
// statistics are reduced across the 32 lanes of a 128-column group.  One pass: (sum, sum of squares) of v - k, k = the row's PREVIOUS mean
// (GemmEpi::ln_shift: the residual update moves a row's mean by a fraction of its spread, so the shifted sums carry no cancellation);
// ln_finalize_kernel turns the groups into (mean, rstd).
template <int RB, int STEP, int ITER, int C4, bool LNF, typename RowMap>
__device__ __forceinline__ void drain_resid(const char* sm, int pitch, const GemmEpi& e, const ColParams& cp, int M, int N, int n, int c4, int tid, RowMap rowmap) {
  const int rb = tid / C4;
  const bool nv = !LNF || n < N;
  // lanes 16..31 of a group hold its complete sums (half32_sum_hi): lane 16 + j writes row j of a batch, so the RB rows' statistics leave in ONE
  // store instruction (the epilogue's time goes with its store instructions, not their bytes)
  const int wj = (c4 & 31) - 16;
  const bool writer = LNF && wj >= 0 && wj < RB && (n & ~127) < N;
#pragma unroll 1
  for (int it = 0; it < ITER; it += RB) {
    float4 r[RB];
    int mm[RB];
    float kk[RB];
#pragma unroll
    for (int j = 0; j < RB; ++j) {
      mm[j] = rowmap(rb + (it + j) * STEP);
      r[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (mm[j] < M && nv) r[j] = *reinterpret_cast<const float4*>(e.resid + (size_t)mm[j] * e.ldr + n);
      if (LNF) kk[j] = reinterpret_cast<const float2*>(sm + (rb + (it + j) * STEP) * pitch + C4 * 16)->x;      // stage_row_stats
    }
    float2 mine = make_float2(0.f, 0.f);      // writer lanes: the sums of row wj of this batch
#pragma unroll
    for (int j = 0; j < RB; ++j) {
      if (!LNF && mm[j] >= M) continue;
      const bool ok = nv && mm[j] < M;      // (LNF: predicated instead of skipped -- the group reductions below are wave-level operations)
      float4 v = *reinterpret_cast<const float4*>(sm + (rb + (it + j) * STEP) * pitch + c4 * 16);
      v.x += cp.bias.x; v.y += cp.bias.y; v.z += cp.bias.z; v.w += cp.bias.w;
      if (e.act == ACT_GELU) { gelu_fast4(v); }
      else if (e.act == ACT_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      v.x = fmaf(v.x, cp.scale.x, r[j].x); v.y = fmaf(v.y, cp.scale.y, r[j].y);
      v.z = fmaf(v.z, cp.scale.z, r[j].z); v.w = fmaf(v.w, cp.scale.w, r[j].w);
      if (ok) *reinterpret_cast<float4*>(e.out_f32 + (size_t)mm[j] * e.ldc + n) = v;
      if (LNF) {
        if (ok && e.ln_op) ln_store_op(e, (size_t)mm[j], n, N, v);
        const float a = v.x - kk[j], b = v.y - kk[j], c = v.z - kk[j], d = v.w - kk[j];
        const float sum = half32_sum_hi(nv ? (a + b) + (c + d) : 0.f);
        const float sq = half32_sum_hi(nv ? (a * a + b * b) + (c * c + d * d) : 0.f);
        if (wj == j) mine = make_float2(sum, sq);
      }
    }
    if (LNF && writer) {
      int mw = mm[0];
#pragma unroll
      for (int j = 1; j < RB; ++j) mw = wj == j ? mm[j] : mw;
      if (mw < M) e.ln_part[(size_t)mw * e.ln_npart + (n >> 7)] = mine;
    }
  }
}

// rows of the LDS tile map to global rows through `rowmap` (row_l -> m)
template <int ROWS, int COLS, int NT, int LN = LN_NONE, typename RowMap>
__device__ __forceinline__ void drain_tile(const char* sm, int pitch, const GemmEpi& e, const ColParams& cp, int M, int N,
                                           int n0, int tid, RowMap rowmap) {
  constexpr int C4 = COLS / 4;
  static_assert(NT % C4 == 0, "a thread must keep its column group");
  static_assert(C4 % 32 == 0, "a 128-column statistics group is 32 lanes of one wave");
  const int c4 = tid % C4;
  const int n = n0 + 4 * c4;
  if (n >= N && LN != LN_PROD) return;
  if (e.resid && e.out_f32 && e.rows_per_img == 0 && !e.a_scale && (!e.a_bs || e.w_bs)) {      // (fp8 with BOTH operands block-scaled: nothing left to dequantise here)
    // The in-place fp32 residual epilogue (out-proj, fc2): a thread visits ROWS / (NT / C4) rows, and with one residual load in flight
    // per thread the pass is a chain of memory latencies (tools/pp_timeline.py: 63 k cycles per 256x256 tile, 15 GB/s per CU).
    // Rows go in batches of RB: RB residual loads are issued before the first of them is consumed.  (Unrolling the whole generic loop
    // instead makes the 256x256 kernels spill: QKV 1345 vs 808 us.)  RB = 4 by default; 8 in flight slow the 1024-thread kernels down
    // (128 VGPRs: out-proj 235 vs 204 us); the 512-thread kernels (256 VGPRs) take a deeper batch through e.rb (round 3).
    constexpr int STEP = NT / C4, ITER = ROWS / STEP;
    static_assert(ROWS % STEP == 0, "whole passes over the tile rows");
    constexpr int RBD = ITER % 4 == 0 ? 4 : (ITER % 2 == 0 ? 2 : 1);
    drain_resid<RBD, STEP, ITER, C4, LN == LN_PROD>(sm, pitch, e, cp, M, N, n, c4, tid, rowmap);
    return;
  }
  if (n >= N) return;
#pragma unroll
  for (int row_l = tid / C4; row_l < ROWS; row_l += NT / C4) {
    const int m = rowmap(row_l);
    if (m >= M) continue;
    float4 v = *reinterpret_cast<const float4*>(sm + row_l * pitch + c4 * 16);
    if (e.a_scale || e.a_bs) {   // fp8 operands: (A_q W_q^T)[m][n] * a_scale[m] * w_scale[n]  (block-scaled A: its scales went into the MFMAs)
      const float sa = e.a_scale ? e.a_scale[m] : 1.0f;
      v.x *= sa * cp.wscale.x; v.y *= sa * cp.wscale.y; v.z *= sa * cp.wscale.z; v.w *= sa * cp.wscale.w;
    }
    if (LN == LN_CONS) {         // folded LayerNorm: (x W'^T - mean c) rstd   (cp.scale holds c; GemmEpi::ln_c)
      const float2 st = *reinterpret_cast<const float2*>(sm + row_l * pitch + C4 * 16);      // stage_row_stats
      v.x = fmaf(-st.x, cp.scale.x, v.x) * st.y; v.y = fmaf(-st.x, cp.scale.y, v.y) * st.y;
      v.z = fmaf(-st.x, cp.scale.z, v.z) * st.y; v.w = fmaf(-st.x, cp.scale.w, v.w) * st.y;
    }
    v.x += cp.bias.x; v.y += cp.bias.y; v.z += cp.bias.z; v.w += cp.bias.w;
    if (e.act == ACT_GELU) { gelu_fast4(v); }
    else if (e.act == ACT_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    if (LN != LN_CONS) { v.x *= cp.scale.x; v.y *= cp.scale.y; v.z *= cp.scale.z; v.w *= cp.scale.w; }
    size_t orow = (size_t)m;
    if (e.rows_per_img > 0) {
      const int b = m / e.rows_per_img, p = m - b * e.rows_per_img;
      orow = (size_t)b * e.out_rows_per_img + 1 + p;
      const float4 p4 = *reinterpret_cast<const float4*>(e.pos + (size_t)(1 + p) * N + n);
      v.x += p4.x; v.y += p4.y; v.z += p4.z; v.w += p4.w;
    }
    if (e.resid) {
      const float4 r4 = *reinterpret_cast<const float4*>(e.resid + orow * e.ldr + n);
      v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w;
    }
    if (e.out_f32) {
      *reinterpret_cast<float4*>(e.out_f32 + orow * e.ldc + n) = v;
    } else if (e.glu) {             // interleaved SwiGLU pairs: columns n .. n+3 = (x1, x2, x1, x2) -> two gated outputs at column n / 2
      const float g0 = silu_mul(v.x, v.y), g1 = silu_mul(v.z, v.w);         // (H2 rows need whole quads: launchers route them to drain_tile_bf16x8)
      const unsigned hi = pack2bf(g0, g1);
      bf16_t* og = e.out_bf16 + orow * e.ldc + (n >> 1);
      *reinterpret_cast<unsigned*>(og) = hi;
      if (e.out_split < 0) *reinterpret_cast<unsigned*>(og - e.out_split) = pack2bf(g0 - __uint_as_float(hi << 16), g1 - __uint_as_float(hi & 0xffff0000u));
    } else if (e.out_h2) {          // H2 operand row of the next GEMM (dod_common.h)
      uint2 f16; unsigned hi8, lo8;
      h2_quad(v, 1.0f, f16, hi8, lo8);
      char* row = reinterpret_cast<char*>(e.out_bf16) + orow * (size_t)e.ldc * 2;
      *reinterpret_cast<uint2*>(row + 2 * n) = f16;
      char* p8 = row + h2_off8(N, n);
      *reinterpret_cast<unsigned*>(p8) = hi8;
      *reinterpret_cast<unsigned*>(p8 + 16) = lo8;
    } else if (e.out_split > 0) {   // bf16x3 activation operand of the next GEMM: [hi | hi | lo]
      uint2 hi, lo;
      hi.x = pack2bf(v.x, v.y);
      hi.y = pack2bf(v.z, v.w);
      lo.x = pack2bf(v.x - __uint_as_float(hi.x << 16), v.y - __uint_as_float(hi.x & 0xffff0000u));
      lo.y = pack2bf(v.z - __uint_as_float(hi.y << 16), v.w - __uint_as_float(hi.y & 0xffff0000u));
      bf16_t* o = e.out_bf16 + orow * e.ldc + n;
      *reinterpret_cast<uint2*>(o) = hi;
      *reinterpret_cast<uint2*>(o + e.out_split) = hi;
      *reinterpret_cast<uint2*>(o + 2 * (size_t)e.out_split) = lo;
    } else if (e.out_split < 0) {   // pair layout [hi | lo] (attention input of the bf16x3 mode)
      uint2 hi, lo;
      hi.x = pack2bf(v.x, v.y);
      hi.y = pack2bf(v.z, v.w);
      lo.x = pack2bf(v.x - __uint_as_float(hi.x << 16), v.y - __uint_as_float(hi.x & 0xffff0000u));
      lo.y = pack2bf(v.z - __uint_as_float(hi.y << 16), v.w - __uint_as_float(hi.y & 0xffff0000u));
      bf16_t* o = e.out_bf16 + orow * e.ldc + n;
      *reinterpret_cast<uint2*>(o) = hi;
      *reinterpret_cast<uint2*>(o - e.out_split) = lo;
    } else {
      uint2 o;
      o.x = pack2bf(v.x, v.y);
      o.y = pack2bf(v.z, v.w);
      *reinterpret_cast<uint2*>(e.out_bf16 + orow * e.ldc + n) = o;
    }
  }
}


// bf16 outputs without residual / row remap (QKV, fc1; plain or pair layout): EIGHT columns per lane -- one 16-byte store per lane
// and row instead of two 8-byte ones: the store tail of a tile is bound by the number of store instructions, not by their bytes
// (guide T21), and a workgroup per CU has nothing else to overlap it with.
__device__ __forceinline__ bool drain8_ok(const GemmEpi& e, int N) {
  return e.out_bf16 && e.out_split <= 0 && !e.resid && e.rows_per_img == 0 && !e.a_scale && (!e.a_bs || e.w_bs) && !e.out_bs && (N & 7) == 0 && (e.ldc & 7) == 0 &&
         (e.out_split == 0 || ((-e.out_split) & 7) == 0);
}
template <int ROWS, int COLS, int NT, int LN = LN_NONE, typename RowMap>
__device__ __forceinline__ void drain_tile_bf16x8(const char* sm, int pitch, const GemmEpi& e, int M, int N, int n0, int tid, RowMap rowmap) {
  constexpr int C8 = COLS / 8;
  static_assert(NT % C8 == 0, "a thread must keep its column group");
  const int c8 = tid % C8;
  const int n = n0 + 8 * c8;
  if (n >= N) return;
  float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0, s0 = make_float4(1.f, 1.f, 1.f, 1.f), s1 = s0;
  if (e.bias) { b0 = *reinterpret_cast<const float4*>(e.bias + n); b1 = *reinterpret_cast<const float4*>(e.bias + n + 4); }
  constexpr bool ln = LN == LN_CONS;          // folded LayerNorm: s0 / s1 hold the column sums c[n] instead of a LayerScale
  const float* sc = ln ? e.ln_c : e.scale;
  if (sc) { s0 = *reinterpret_cast<const float4*>(sc + n); s1 = *reinterpret_cast<const float4*>(sc + n + 4); }
#pragma unroll
  for (int row_l = tid / C8; row_l < ROWS; row_l += NT / C8) {
    const int m = rowmap(row_l);
    if (m >= M) continue;
    float4 v = *reinterpret_cast<const float4*>(sm + row_l * pitch + c8 * 32);
    float4 u = *reinterpret_cast<const float4*>(sm + row_l * pitch + c8 * 32 + 16);
    if (ln) {                     // rstd acc + (b' - mean rstd c): two packed FMAs per pair where the plain epilogue has one packed add
      const float2 st = *reinterpret_cast<const float2*>(sm + row_l * pitch + COLS * 4);      // stage_row_stats
      const f32x2_t t = (f32x2_t)(-st.x * st.y), rs = (f32x2_t)(st.y);
#define LN_PAIR_(va, vb, ca, cb, ba, bb)                                                                         \
      { const f32x2_t c_ = {ca, cb}, b_ = {ba, bb}, a_ = {va, vb};                                                \
        const f32x2_t o_ = __builtin_elementwise_fma(a_, rs, __builtin_elementwise_fma(t, c_, b_));               \
        va = o_.x; vb = o_.y; }
      LN_PAIR_(v.x, v.y, s0.x, s0.y, b0.x, b0.y) LN_PAIR_(v.z, v.w, s0.z, s0.w, b0.z, b0.w)
      LN_PAIR_(u.x, u.y, s1.x, s1.y, b1.x, b1.y) LN_PAIR_(u.z, u.w, s1.z, s1.w, b1.z, b1.w)
#undef LN_PAIR_
    } else {
      v.x += b0.x; v.y += b0.y; v.z += b0.z; v.w += b0.w;
      u.x += b1.x; u.y += b1.y; u.z += b1.z; u.w += b1.w;
    }
    if (e.act == ACT_GELU) {
      gelu_fast4(v); gelu_fast4(u);
    } else if (e.act == ACT_RELU) {
      v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
      u.x = fmaxf(u.x, 0.f); u.y = fmaxf(u.y, 0.f); u.z = fmaxf(u.z, 0.f); u.w = fmaxf(u.w, 0.f);
    }
    if (!ln) {
      v.x *= s0.x; v.y *= s0.y; v.z *= s0.z; v.w *= s0.w;
      u.x *= s1.x; u.y *= s1.y; u.z *= s1.z; u.w *= s1.w;
    }
    if (e.glu) {                    // interleaved SwiGLU pairs: eight columns -> four gated outputs at column n / 2 of an N / 2-column row
      const float4 gq = make_float4(silu_mul(v.x, v.y), silu_mul(v.z, v.w), silu_mul(u.x, u.y), silu_mul(u.z, u.w));
      const int nh = n >> 1;
      if (e.out_h2) {               // H2 operand row (fp16x2 mode)
        uint2 f16; unsigned hi8, lo8;
        h2_quad(gq, 1.0f, f16, hi8, lo8);
        char* row = reinterpret_cast<char*>(e.out_bf16) + (size_t)m * e.ldc * 2;
        *reinterpret_cast<uint2*>(row + 2 * nh) = f16;
        char* p8 = row + h2_off8(N >> 1, nh);
        *reinterpret_cast<unsigned*>(p8) = hi8;
        *reinterpret_cast<unsigned*>(p8 + 16) = lo8;
        continue;
      }
      uint2 g;
      g.x = pack2bf(gq.x, gq.y);
      g.y = pack2bf(gq.z, gq.w);
      bf16_t* og = e.out_bf16 + (size_t)m * e.ldc + nh;
      *reinterpret_cast<uint2*>(og) = g;
      if (e.out_split < 0) {        // pair layout [hi | lo] (bf16x3 mode), each half -out_split columns wide
        uint2 lo;
        lo.x = pack2bf(gq.x - __uint_as_float(g.x << 16), gq.y - __uint_as_float(g.x & 0xffff0000u));
        lo.y = pack2bf(gq.z - __uint_as_float(g.y << 16), gq.w - __uint_as_float(g.y & 0xffff0000u));
        *reinterpret_cast<uint2*>(og - e.out_split) = lo;
      }
      continue;
    }
    if (e.out_h2) {                 // H2 operand row of the next GEMM (dod_common.h): 16 B of fp16, 8 + 8 B of e4m3
      uint2 fa, fb; unsigned h0, l0, h1, l1;
      h2_quad(v, 1.0f, fa, h0, l0);
      h2_quad(u, 1.0f, fb, h1, l1);
      char* row = reinterpret_cast<char*>(e.out_bf16) + (size_t)m * e.ldc * 2;
      *reinterpret_cast<uint4*>(row + 2 * n) = make_uint4(fa.x, fa.y, fb.x, fb.y);
      char* p8 = row + h2_off8(N, n);
      *reinterpret_cast<uint2*>(p8) = make_uint2(h0, h1);
      *reinterpret_cast<uint2*>(p8 + 16) = make_uint2(l0, l1);
      continue;
    }
    uint4 hi;
    hi.x = pack2bf(v.x, v.y); hi.y = pack2bf(v.z, v.w); hi.z = pack2bf(u.x, u.y); hi.w = pack2bf(u.z, u.w);
    bf16_t* o = e.out_bf16 + (size_t)m * e.ldc + n;
    *reinterpret_cast<uint4*>(o) = hi;
    if (e.out_split < 0) {       // pair layout [hi | lo]
      uint4 lo;
      lo.x = pack2bf(v.x - __uint_as_float(hi.x << 16), v.y - __uint_as_float(hi.x & 0xffff0000u));
      lo.y = pack2bf(v.z - __uint_as_float(hi.y << 16), v.w - __uint_as_float(hi.y & 0xffff0000u));
      lo.z = pack2bf(u.x - __uint_as_float(hi.z << 16), u.y - __uint_as_float(hi.z & 0xffff0000u));
      lo.w = pack2bf(u.z - __uint_as_float(hi.w << 16), u.w - __uint_as_float(hi.w & 0xffff0000u));
      *reinterpret_cast<uint4*>(o - e.out_split) = lo;
    }
  }
}
