// Backbone self-attention (K5), flash-style, bf16 MFMA, head_dim 64 (every DINOv2 variant).
// Replaces Dinov2SelfAttention.forward's softmax(Q K^T * dh^-0.5) V
// (site-packages/transformers/models/dinov2/modeling_dinov2.py:153-178, 215-229): no mask, non-causal.
//
// Layout: qkv is the fused-QKV GEMM output [B*N, 3*D] bf16 (q | k | v column blocks, head h at
// columns h*64..h*64+63 of each block); ctx is [B*N, D] bf16.
// One workgroup = 4 waves = 256 query rows of one (batch, head); each wave owns 64 query rows (two 32-row MFMA blocks).
// K/V tiles of 64 keys stream by LDS-DMA into a 3-slot ring (48 KiB, 3 workgroups/CU), two tiles in flight behind
// a counted s_waitcnt vmcnt(4) and one raw s_barrier per tile.
// "Swapped" products keep the query on the MFMA lane so the softmax row state (m, l) is per-lane:
//   S^T = K Q^T   (A = K tile rows from LDS via ds_read_b128, B = Q fragments held in registers)
//   O^T += V^T P^T (A = V^T via ds_read_b64_tr_b16 from the row-major V tile, B = the S^T accumulator
//                  registers themselves, converted to bf16 - no LDS round trip for P)
// N = 1370 is not a multiple of 64: the last tile's keys >= N get -inf before the row max; loads of
// rows >= N are clamped to row N-1 (finite values times p = 0).
#include "dod_common.h"
#include <cstdlib>

#define AT_WAVES 4
// AT_NQ (32-row query blocks per wave) is a template parameter of the kernel: 2 for large launches, 1 for small ones
#define AT_KV 64

__device__ __forceinline__ int kswz(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }
// V tile: rows r and r+2 of a 4-row tr-read block are 256 B apart (same banks): flip the 64-B half on row bit 1
__device__ __forceinline__ int vswz(int row, int chunk) { return chunk ^ (((row >> 1) & 1) << 2); }

typedef bf16x4 __attribute__((address_space(3))) * lds_bf16x4_ptr;

// One K/V tile (64 keys) for one wave and NQ blocks of 32 query rows (query on the lane).  TAIL masks keys >= N
// (last tile only, so the full tiles carry no compare/select).  Softmax in the exp2 domain with the scale folded
// into one FMA: p = exp2(s*c - m*c), m = running max of the raw scores (c > 0).
// NQ = 2 gives the scheduler two independent MFMA -> VALU -> MFMA chains to interleave (PMC on NQ = 1: 42 % of wave
// cycles stalled on instruction dependencies, MFMA pipe 34 % busy) and halves the K/V fragment reads per MFMA.
template <bool TAIL, int NQ>
__device__ __forceinline__ void attn_tile(const char* sK, const char* sV, const bf16x8 (&qf)[NQ][4], f32x16 (&o)[NQ][2],
                                          float (&m_run)[NQ], float (&l_run)[NQ], float c, int kbase, int N,
                                          int lr, int lh, int g16, int tq, int tp) {
  f32x16 s[NQ][2];
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) {
    const int row = kb * 32 + lr;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sK + row * 128 + kswz(row, 2 * t + lh) * 16);
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        s[q][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[q][t], t == 0 ? zero : s[q][kb], 0, 0, 0);
    }
  }
  bf16x8 pf[NQ][4];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    if (TAIL) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kbase + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (key >= N) s[q][kb][r] = -INFINITY;
        }
    }
    float mx = s[q][0][0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[q][0][r]);
#pragma unroll
    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[q][1][r]);
    {   // other half's max by v_permlane32_swap (VALU; __shfl_xor goes through the LDS crossbar and an lgkmcnt wait)
      const unsigned mb = __float_as_uint(mx);
      const auto sw = __builtin_amdgcn_permlane32_swap(mb, mb, false, false);
      mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    }
    const float m_new = fmaxf(m_run[q], mx * c);            // scaled (exp2-domain) running max
    const float alpha = __builtin_amdgcn_exp2f(m_run[q] - m_new);
    m_run[q] = m_new;
    const float nm = -m_new;
    float lsum;
    {      // Plain fp32 VALU only (one value per lane), and the file is built with -fno-slp-vectorize (_build.py) so that hipcc does not re-pack it:
           // `tools/probes/mfma_valu_overlap.hip` (profiles/r04_mfma_valu_overlap_probe.txt) shows that v_pk_*_f32 does NOT overlap an executing MFMA
           // on this part -- a wave of v_pk_fma_f32 beside a wave of MFMAs on the same SIMD takes the SUM of their times, and with two waves per
           // SIMD the packed form has no throughput advantage either (4 cycles per instruction against 2 for the plain one) -- while plain VALU,
           // v_exp_f32, v_max3_f32 and v_cvt_pk_bf16_f32 do overlap it.  Rounds 2-3 had forced v_pk_fma / v_pk_add / v_pk_mul here (fewer
           // instructions: 376 against 420 per tile); this form is 4 % faster (6.06 -> 5.83 ms of attention per batch-64 step).
      float l0 = 0.f, l1 = 0.f, l2 = 0.f, l3 = 0.f;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; r += 4) {
          const float e0 = __builtin_amdgcn_exp2f(fmaf(s[q][kb][r], c, nm)), e1 = __builtin_amdgcn_exp2f(fmaf(s[q][kb][r + 1], c, nm));
          const float e2 = __builtin_amdgcn_exp2f(fmaf(s[q][kb][r + 2], c, nm)), e3 = __builtin_amdgcn_exp2f(fmaf(s[q][kb][r + 3], c, nm));
          l0 += e0; l1 += e1; l2 += e2; l3 += e3;
          s[q][kb][r] = e0; s[q][kb][r + 1] = e1; s[q][kb][r + 2] = e2; s[q][kb][r + 3] = e3;
        }
      lsum = (l0 + l1) + (l2 + l3);
    }
    l_run[q] = fmaf(l_run[q], alpha, lsum);   // per-half partial; the halves are combined once at the end
    // (a thresholded "lazy" rescale behind a wave-uniform branch -- the reference maximum moves only when a row's maximum exceeds it by 2^8 -- was
    //  measured twice: 4 % slower on the packed form of round 3, 5 % slower on this one (attention 6.05 -> 6.38 ms, tools/experiments/r4_exp30.sh;
    //  8 spilled registers): the branch splits the scheduling region, and the 64 multiplies it saves run beside the MFMAs anyway)
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[q][0][r] *= alpha; o[q][1][r] *= alpha; }
    // P^T fragments: accumulator registers 8u..8u+7 of key block kb are the B operand of k-step 2kb+u
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const f32x16& pp = s[q][s4 >> 1];
      const int u = (s4 & 1) * 8;
      uint4 pk;
      pk.x = pack2bf(pp[u + 0], pp[u + 1]);
      pk.y = pack2bf(pp[u + 2], pp[u + 3]);
      pk.z = pack2bf(pp[u + 4], pp[u + 5]);
      pk.w = pack2bf(pp[u + 6], pp[u + 7]);
      pf[q][s4] = __builtin_bit_cast(bf16x8, pk);
    }
  }
  // V^T fragments by ds_read_b64_tr_b16 in inline asm: the builtin form makes hipcc wait vmcnt(0) (it treats the
  // read as aliasing the pending LDS-DMA writes), which would drain the K/V ring every tile.  Eight reads + their
  // lgkmcnt wait form ONE statement (cdna guide 5.7 item 1, form i).  Rows 16*s4 (+8) of a lane's address differ
  // only by immediates because the V swizzle depends on row bit 1 alone.
  {
    const int rowb = 4 * lh + tq;
    const int col0 = 16 * (g16 & 1) + 4 * tp;
    const int chunk0 = col0 >> 3, inb = (col0 & 7) * 2;
    const unsigned vaddr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)sV;
    const unsigned a0 = vaddr + rowb * 128 + vswz(rowb, chunk0) * 16 + inb;        // d-block 0
    const unsigned a1 = vaddr + rowb * 128 + vswz(rowb, chunk0 + 4) * 16 + inb;    // d-block 1
#pragma unroll
    for (int db = 0; db < 2; ++db) {
      bf16x4 v0, v1, v2, v3, v4, v5, v6, v7;
      asm volatile(
          "ds_read_b64_tr_b16 %0, %8\n\t"
          "ds_read_b64_tr_b16 %1, %8 offset:1024\n\t"
          "ds_read_b64_tr_b16 %2, %8 offset:2048\n\t"
          "ds_read_b64_tr_b16 %3, %8 offset:3072\n\t"
          "ds_read_b64_tr_b16 %4, %8 offset:4096\n\t"
          "ds_read_b64_tr_b16 %5, %8 offset:5120\n\t"
          "ds_read_b64_tr_b16 %6, %8 offset:6144\n\t"
          "ds_read_b64_tr_b16 %7, %8 offset:7168\n\t"
          "s_waitcnt lgkmcnt(0)"
          : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7)
          : "v"(db ? a1 : a0)
          : "memory");
      const bf16x8 vf0 = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
      const bf16x8 vf1 = __builtin_shufflevector(v2, v3, 0, 1, 2, 3, 4, 5, 6, 7);
      const bf16x8 vf2 = __builtin_shufflevector(v4, v5, 0, 1, 2, 3, 4, 5, 6, 7);
      const bf16x8 vf3 = __builtin_shufflevector(v6, v7, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        o[q][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf0, pf[q][0], o[q][db], 0, 0, 0);
        o[q][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf1, pf[q][1], o[q][db], 0, 0, 0);
        o[q][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf2, pf[q][2], o[q][db], 0, 0, 0);
        o[q][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf3, pf[q][3], o[q][db], 0, 0, 0);
      }
    }
  }
}

// tuning only (tools/attn_timeline.py): per workgroup {tile-loop cycles, cycles waiting for DMA + barrier, tiles, active,
// s_memrealtime (100 MHz) at kernel entry, at the end of the tile loop}; the exit time is the next workgroup's entry
#ifdef DINODET_TUNING
__device__ unsigned long long* g_attn_stamps = nullptr;
#define ATTN_STAMPS g_attn_stamps
#else
#define ATTN_STAMPS (static_cast<unsigned long long*>(nullptr))      // release build: the stamp code folds away
#endif

// K/V ring, [slot][K|V][64 rows][128 B] = 48 KiB: ONE file-scope array, so that both bodies of the fused kernel address the same
// compile-time-constant LDS locations (a pointer parameter cost the 64-rows-per-wave body three registers and a spill)
__shared__ __attribute__((aligned(16))) char g_attn_smem[3 * 2 * AT_KV * 128];
// the kernel body for workgroup index `blk_` of a launch over query rows [q_lo, q_hi)
// MX (fp8 mode, round 4): the context leaves as block-scaled e4m3 -- ctx is then a BYTE buffer [B*N, D], bs its e8m0 bytes [B*N][2][D / 64] --
// the out-proj GEMM's operand, written here instead of by a quantisation pass over bf16 rows (a head's 64 columns are two blocks of 32)
template <int AT_NQ, bool MX = false>
__device__ __forceinline__ void attn_bf16_body(const int blk_, const bf16_t* __restrict__ qkv, bf16_t* __restrict__ ctx,
                                               int N, int heads, int npairs, float scale_log2e, int q_lo, int q_hi, unsigned char* __restrict__ bs = nullptr) {
  char* const smem = g_attn_smem;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int D = heads * 64, ld = 3 * D;
  const unsigned long long rt_entry = ATTN_STAMPS ? __builtin_amdgcn_s_memrealtime() : 0;
  // XCD-aware block order: workgroups are dealt round-robin over the 8 XCDs, so the q-blocks of one (image, head)
  // -- which all stream the same 350 KB of K/V -- are placed on ONE XCD (pair p -> XCD p % 8) and run back to back
  // there; with the plain (q-block, head, image) grid each XCD saw ~70 different pairs at once (24 MB of K/V
  // against a 4 MB L2: FETCH_SIZE showed K/V fetched 5.5x).  Speed only; any placement is correct.
  constexpr int AT_QW = 32 * AT_NQ;
  const int nqb = (q_hi - q_lo + AT_WAVES * AT_QW - 1) / (AT_WAVES * AT_QW);      // query rows [q_lo, q_hi) of every (image, head); keys 0..N-1
  int b, h, qb;
  {
    const int L = blk_, xcd = L & 7, s = L >> 3;
    const int pair = (s / nqb) * 8 + xcd;
    qb = s - (s / nqb) * nqb;
    if (pair >= npairs) return;            // grid is padded to a multiple of 8 pairs
    b = pair / heads;
    h = pair - b * heads;
  }
  const int q0 = q_lo + qb * (AT_WAVES * AT_QW) + wid * AT_QW;
  const bf16_t* base = qkv + (size_t)b * N * ld;

  // Q fragments: B operand of S^T = K Q^T: lane holds Q[q = lr][d = 16 t + 8 lh + 0..7]
  bf16x8 qf[AT_NQ][4];
#pragma unroll
  for (int q = 0; q < AT_NQ; ++q) {
    int qr = q0 + q * 32 + lr; qr = qr < N ? qr : N - 1;
    const bf16_t* qp = base + (size_t)qr * ld + h * 64 + lh * 8;
#pragma unroll
    for (int t = 0; t < 4; ++t) qf[q][t] = *reinterpret_cast<const bf16x8*>(qp + 16 * t);
  }
  // Retire the Q loads HERE, before any LDS-DMA is in flight: beside a pending global_load_lds hipcc waits
  // vmcnt(0) at the first use of an ordinary load's result, which would land inside the tile loop and drain
  // the DMA ring every iteration.  The empty asm makes the fragments "produced" at this point.
#pragma unroll
  for (int q = 0; q < AT_NQ; ++q)
#pragma unroll
    for (int t = 0; t < 4; ++t) asm volatile("" : "+v"(qf[q][t]));

  // staging: LDS-DMA (global_load_lds_dwordx4) into a 3-slot ring of K/V tiles, two tiles in flight.
  // One piece = 8 key rows x 128 B; per tile a wave issues 2 pieces of K and 2 of V.  lane -> (row = lane>>3,
  // LDS chunk slot = lane&7); the swizzle goes on the SOURCE chunk (K: ^(row>>1)&7, V: ^((row>>1)&1)<<2).
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const int wu = __builtin_amdgcn_readfirstlane(wid);
  const int prow0 = wu * 16 + (lane >> 3), prow1 = prow0 + 8;          // tile rows of this lane's two pieces
  const int kc0 = ((lane & 7) ^ ((prow0 >> 1) & 7)) * 8, kc1 = ((lane & 7) ^ ((prow1 >> 1) & 7)) * 8;
  const int vc0 = ((lane & 7) ^ (((prow0 >> 1) & 1) << 2)) * 8, vc1 = ((lane & 7) ^ (((prow1 >> 1) & 1) << 2)) * 8;
  const bf16_t* kbase = base + D + h * 64;
#define AT_STAGE(slot_, kt_)                                                                                   \
  {                                                                                                            \
    int key0 = (kt_) * AT_KV + prow0, key1 = (kt_) * AT_KV + prow1;                                            \
    key0 = key0 < N ? key0 : N - 1; key1 = key1 < N ? key1 : N - 1;                                            \
    const bf16_t* r0 = kbase + (size_t)key0 * ld;                                                              \
    const bf16_t* r1 = kbase + (size_t)key1 * ld;                                                              \
    char* sK_ = smem + (slot_) * (2 * AT_KV * 128) + wu * 2048;                                                \
    char* sV_ = sK_ + AT_KV * 128;                                                                             \
    __builtin_amdgcn_global_load_lds((gptr_t)(r0 + kc0), (lptr_t)(sK_), 16, 0, 0);                             \
    __builtin_amdgcn_global_load_lds((gptr_t)(r1 + kc1), (lptr_t)(sK_ + 1024), 16, 0, 0);                      \
    __builtin_amdgcn_global_load_lds((gptr_t)(r0 + D + vc0), (lptr_t)(sV_), 16, 0, 0);                         \
    __builtin_amdgcn_global_load_lds((gptr_t)(r1 + D + vc1), (lptr_t)(sV_ + 1024), 16, 0, 0);                  \
  }

  f32x16 o[AT_NQ][2];
  float m_run[AT_NQ], l_run[AT_NQ];
#pragma unroll
  for (int q = 0; q < AT_NQ; ++q) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[q][0][r] = 0.f; o[q][1][r] = 0.f; }
    m_run[q] = -INFINITY; l_run[q] = 0.f;
  }
  const bool active = __builtin_amdgcn_readfirstlane(q0) < q_hi;     // waves past the last row only stage and sync

  // tr-read lane geometry (ds_read_b64_tr_b16: 16-lane groups, lane 4q+p supplies row q, cols 4p..4p+3)
  const int g16 = lane >> 4, i16 = lane & 15, tq = i16 >> 2, tp = i16 & 3;

  const int nkt = (N + AT_KV - 1) / AT_KV;
  const int nfull = N / AT_KV;                 // tiles without a masked key
  AT_STAGE(0, 0)
  if (nkt > 1) AT_STAGE(1, 1)
  int slot = 0;
  unsigned long long* stamps = ATTN_STAMPS;
  unsigned long long t_begin = 0, t_wait = 0;
  if (stamps) t_begin = __builtin_amdgcn_s_memtime();
  for (int kt = 0; kt < nkt; ++kt) {
    unsigned long long tw0 = 0;
    if (stamps) tw0 = __builtin_amdgcn_s_memtime();
    if (kt + 1 < nkt) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");     // tile kt landed, tile kt+1 may fly
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (stamps) t_wait += __builtin_amdgcn_s_memtime() - tw0;
    if (kt + 2 < nkt) {
      const int ns = slot >= 1 ? slot - 1 : 2;      // (slot + 2) % 3: the slot tile kt-1 just vacated
      AT_STAGE(ns, kt + 2)
    }
    const char* sK = smem + slot * (2 * AT_KV * 128);
    const char* sV = sK + AT_KV * 128;
    if (active) {
      if (kt < nfull) attn_tile<false, AT_NQ>(sK, sV, qf, o, m_run, l_run, scale_log2e, kt * AT_KV, N, lr, lh, g16, tq, tp);
      else attn_tile<true, AT_NQ>(sK, sV, qf, o, m_run, l_run, scale_log2e, kt * AT_KV, N, lr, lh, g16, tq, tp);
    }
    slot = slot == 2 ? 0 : slot + 1;
  }

  if (stamps && tid == 0) {
    unsigned long long* o_ = stamps + (size_t)blockIdx.x * 6;
    o_[0] = __builtin_amdgcn_s_memtime() - t_begin; o_[1] = t_wait; o_[2] = nkt; o_[3] = active ? 1 : 0;
    o_[4] = rt_entry; o_[5] = __builtin_amdgcn_s_memrealtime();
  }
#pragma unroll
  for (int qi = 0; qi < AT_NQ; ++qi) {
    const float l_tot = l_run[qi] + __shfl_xor(l_run[qi], 32, 64);
    const float inv = 1.0f / l_tot;
    const int q = q0 + qi * 32 + lr;
    if (MX) {
      // block db of this head = columns db*32 .. +31 of query q: 16 of them in this lane, 16 in lane ^ 32 (every lane takes part in the shuffle)
      const bool okq = active && q < q_hi;
      unsigned char* op8 = reinterpret_cast<unsigned char*>(ctx) + ((size_t)b * N + q) * D + h * 64;
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        float amax = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) amax = fmaxf(amax, fabsf(o[qi][db][r] * inv));
        amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
        const unsigned eb = mx_ebyte(amax);
        const float sc = inv * mx_inv_scale(eb);
        if (okq) {
#pragma unroll
          for (int g = 0; g < 4; ++g)
            *reinterpret_cast<unsigned*>(op8 + db * 32 + 8 * g + 4 * lh) = pack4_fp8(o[qi][db][4 * g] * sc, o[qi][db][4 * g + 1] * sc, o[qi][db][4 * g + 2] * sc, o[qi][db][4 * g + 3] * sc);
          if (lh == 0) bs[((size_t)b * N + q) * (D >> 5) + mx_scale_off(D, h * 2 + db)] = (unsigned char)eb;
        }
      }
    } else if (active && q < q_hi) {
      bf16_t* op = ctx + ((size_t)b * N + q) * D + h * 64;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          uint2 pk;
          pk.x = pack2bf(o[qi][db][4 * g] * inv, o[qi][db][4 * g + 1] * inv);
          pk.y = pack2bf(o[qi][db][4 * g + 2] * inv, o[qi][db][4 * g + 3] * inv);
          *reinterpret_cast<uint2*>(op + db * 32 + 8 * g + 4 * lh) = pk;
        }
    }
  }
}

template <int AT_NQ, bool MX>
__global__ __launch_bounds__(256, 2) void attn_bf16_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ ctx,
                                                        int N, int heads, int npairs, float scale_log2e, int q_lo, int q_hi, unsigned char* __restrict__ bs) {
  attn_bf16_body<AT_NQ, MX>(blockIdx.x, qkv, ctx, N, heads, npairs, scale_log2e, q_lo, q_hi, bs);
}
// ONE launch for a sequence whose last 256-row block is short (N = 1370: 5 blocks + 90 rows): workgroups [0, main_blocks) run the
// 64-rows-per-wave body over rows [0, q_main), the workgroups behind them the 32-rows-per-wave body over the remainder -- dispatched
// last, they fill the main part's end-of-kernel bubble instead of costing a launch of their own (round 3: the separate tail launch
// took 62.6 us per layer for 6.6 % of the rows, profiles/r03_bench_bf16_kernel_stats.csv).  main_blocks % 8 == 0 keeps both XCD maps.
template <bool MX>
__global__ __launch_bounds__(256, 2) void attn_bf16_fused_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ ctx,
                                                              int N, int heads, int npairs, float scale_log2e, int main_blocks, int q_main, unsigned char* __restrict__ bs) {
  if ((int)blockIdx.x < main_blocks) attn_bf16_body<2, MX>(blockIdx.x, qkv, ctx, N, heads, npairs, scale_log2e, 0, q_main, bs);
  else attn_bf16_body<1, MX>((int)blockIdx.x - main_blocks, qkv, ctx, N, heads, npairs, scale_log2e, q_main, N, bs);
}

// (Round 4 built a 512-thread PING-PONG form of this kernel -- both waves of every SIMD in one workgroup, phased against each other with the
// workgroup barrier so that one group's MFMA phase (P V of the previous tile, K Q^T of the next) meets the other group's softmax; the groups split
// the keys and merge (m, l, O) through LDS at the end.  The probe says such phases overlap (tools/probes/mfma_valu_overlap.hip mode 16: 1 088 /
// 1 974 cycles side by side); the kernel was parity-green and 26 % SLOWER (attention 6.52 -> 8.25 ms per batch-64 step, tools/experiments/r4_exp19.sh):
// one workgroup per CU (2 x 48 KiB of rings) exposes every workgroup's prologue and merge, the MFMA phase carries six LDS round trips that two
// free-running workgroups hide from each other, and every phase lasts as long as its slower half.  Not kept.)
#ifdef DINODET_TUNING
extern "C" int dod_debug_attn_stamps(void* dev_buf) {
  unsigned long long* p = (unsigned long long*)dev_buf;
  return hipMemcpyToSymbol(HIP_SYMBOL(g_attn_stamps), &p, sizeof(p)) == hipSuccess ? 0 : 4;
}
#endif

int launch_attn_bf16(const bf16_t* qkv, bf16_t* ctx, int B, int N, int heads, float scale, hipStream_t s, unsigned char* ctx_bs) {
  if (B <= 0 || N <= 0 || heads <= 0) return 1;
  if (ctx_bs && (heads * 64) % 64 != 0) return 2;
  const bool mx = ctx_bs != nullptr;
  const int npairs = B * heads, pairs8 = (npairs + 7) / 8 * 8;
  const float c = scale * 1.44269504088896340736f;
  // 64 query rows per wave once the grid still fills the chip several times over (measured: +5 % at B*heads = 768,
  // -12 % at 96), else 32
  static const char* nqe = DOD_TUNE_ENV("DINODET_ATTN_NQ");   // tuning override
  if (nqe ? nqe[0] == '2' : (long)npairs * ((N + 255) / 256) >= 4 * 256) {
    // A workgroup lasts as long as its busiest wave: the last 256-row block of N = 1370 (90 rows) keeps one wave fully busy and costs a
    // whole block time for a third of the rows.  When the remainder fits 128 rows it goes to a second launch of the 32-rows-per-wave
    // kernel (three waves with one q-block each: about half a block time): 5 + 0.55 instead of 6 block times per (image, head).
    const int rem = N % (AT_WAVES * 64);
    static const char* tse = DOD_TUNE_ENV("DINODET_ATTN_TAILSPLIT");
    const bool split = (tse ? tse[0] != '0' : true) && rem > 0 && rem <= AT_WAVES * 32 && N > AT_WAVES * 64;
    const int q_main = split ? N - rem : N;
    const int nqb = (q_main + AT_WAVES * 64 - 1) / (AT_WAVES * 64);
    static const char* fse = DOD_TUNE_ENV("DINODET_ATTN_FUSED_TAIL");      // "0": the two-launch form (A/B)
    if (split && !(fse && fse[0] == '0')) {
#ifdef DINODET_TUNING
      // tuning builds: DINODET_ATTN_LDS_PAD = bytes of unused dynamic LDS per workgroup (64 KiB leaves ONE workgroup per CU: one wave per SIMD)
      static const int pad = [] { const char* v = getenv("DINODET_ATTN_LDS_PAD"); return v ? atoi(v) : 0; }();
      if (pad > 0 && !mx) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bf16_fused_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, pad);
        hipLaunchKernelGGL(attn_bf16_fused_kernel<false>, dim3(pairs8 * nqb + pairs8), dim3(256), pad, s, qkv, ctx, N, heads, npairs, c, pairs8 * nqb, q_main, ctx_bs);
        return hipGetLastError() == hipSuccess ? 0 : 3;
      }
#endif
      if (mx) hipLaunchKernelGGL(attn_bf16_fused_kernel<true>, dim3(pairs8 * nqb + pairs8), dim3(256), 0, s, qkv, ctx, N, heads, npairs, c, pairs8 * nqb, q_main, ctx_bs);
      else hipLaunchKernelGGL(attn_bf16_fused_kernel<false>, dim3(pairs8 * nqb + pairs8), dim3(256), 0, s, qkv, ctx, N, heads, npairs, c, pairs8 * nqb, q_main, ctx_bs);
    } else {
      if (mx) {
        hipLaunchKernelGGL((attn_bf16_kernel<2, true>), dim3(pairs8 * nqb), dim3(256), 0, s, qkv, ctx, N, heads, npairs, c, 0, q_main, ctx_bs);
        if (split) hipLaunchKernelGGL((attn_bf16_kernel<1, true>), dim3(pairs8), dim3(256), 0, s, qkv, ctx, N, heads, npairs, c, q_main, N, ctx_bs);
      } else {
        hipLaunchKernelGGL((attn_bf16_kernel<2, false>), dim3(pairs8 * nqb), dim3(256), 0, s, qkv, ctx, N, heads, npairs, c, 0, q_main, ctx_bs);
        if (split) hipLaunchKernelGGL((attn_bf16_kernel<1, false>), dim3(pairs8), dim3(256), 0, s, qkv, ctx, N, heads, npairs, c, q_main, N, ctx_bs);
      }
    }
  } else {
    const int nqb = (N + AT_WAVES * 32 - 1) / (AT_WAVES * 32);
    if (mx) hipLaunchKernelGGL((attn_bf16_kernel<1, true>), dim3(pairs8 * nqb), dim3(256), 0, s, qkv, ctx, N, heads, npairs, c, 0, N, ctx_bs);
    else hipLaunchKernelGGL((attn_bf16_kernel<1, false>), dim3(pairs8 * nqb), dim3(256), 0, s, qkv, ctx, N, heads, npairs, c, 0, N, ctx_bs);
  }
  return hipGetLastError() == hipSuccess ? 0 : 3;
}
