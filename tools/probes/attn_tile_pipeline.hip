// Probe (GPU box only; no torch): how many cycles does ONE wave need per 64-query x 64-key attention tile (head_dim 64, bf16 MFMA, the shipped kernel's
// arithmetic) when it runs alone on its SIMD with the whole 512-register file and the three stages of CONSECUTIVE tiles are independent inside
// the loop body -- K Q^T of tile t + 1, the softmax of tile t, P V of tile t - 1 -- so that the compiler can put the MFMAs under the VALU?
// The shipped kernel (attn_bf16.hip) runs two waves per SIMD at 245 registers each, tile order QK -> softmax -> PV inside one wave: 5 000 cycles per
// tile and wave = 2 500 cycles of SIMD time per wave-tile (profiles/r04_attn_timeline.txt).  K / V tiles here sit in LDS for the whole run (no DMA).
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -o attn_tile_pipeline tools/probes/attn_tile_pipeline.hip && ./attn_tile_pipeline
// modes: 0 = serial (one tile after the other, as shipped), 1 = pipelined (three stages of three tiles per iteration); each with one and two
// workgroups (of four waves) per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;

__device__ __forceinline__ unsigned pack2bf(float lo, float hi) { f32x2 v = {lo, hi}; bf16x2 h = __builtin_convertvector(v, bf16x2); return __builtin_bit_cast(unsigned, h); }
__device__ __forceinline__ int kswz(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }
__device__ __forceinline__ int vswz(int row, int chunk) { return chunk ^ (((row >> 1) & 1) << 2); }

__device__ __forceinline__ void qk(const char* sK, const bf16x8 (&qf)[2][4], f32x16 (&s)[2][2], int lr, int lh) {
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) {
    const int row = kb * 32 + lr;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sK + row * 128 + kswz(row, 2 * t + lh) * 16);
#pragma unroll
      for (int q = 0; q < 2; ++q) s[q][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[q][t], t == 0 ? zero : s[q][kb], 0, 0, 0);
    }
  }
}
__device__ __forceinline__ void softmax(f32x16 (&s)[2][2], f32x16 (&o)[2][2], float (&m_run)[2], float (&l_run)[2], bf16x8 (&pf)[2][4], float c) {
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    float mx = s[q][0][0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[q][0][r]);
#pragma unroll
    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[q][1][r]);
    {
      const unsigned mb = __float_as_uint(mx);
      const auto sw = __builtin_amdgcn_permlane32_swap(mb, mb, false, false);
      mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    }
    const float m_new = fmaxf(m_run[q], mx * c);
    const float alpha = __builtin_amdgcn_exp2f(m_run[q] - m_new);
    m_run[q] = m_new;
    const float nm = -m_new;
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
    l_run[q] = fmaf(l_run[q], alpha, (l0 + l1) + (l2 + l3));
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[q][0][r] *= alpha; o[q][1][r] *= alpha; }
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const f32x16& pp = s[q][s4 >> 1];
      const int u = (s4 & 1) * 8;
      uint4 pk;
      pk.x = pack2bf(pp[u + 0], pp[u + 1]); pk.y = pack2bf(pp[u + 2], pp[u + 3]);
      pk.z = pack2bf(pp[u + 4], pp[u + 5]); pk.w = pack2bf(pp[u + 6], pp[u + 7]);
      pf[q][s4] = __builtin_bit_cast(bf16x8, pk);
    }
  }
}
__device__ __forceinline__ void pv(const char* sV, const bf16x8 (&pf)[2][4], f32x16 (&o)[2][2], int lh, int g16, int tq, int tp) {
  const int rowb = 4 * lh + tq;
  const int col0 = 16 * (g16 & 1) + 4 * tp;
  const int chunk0 = col0 >> 3, inb = (col0 & 7) * 2;
  const unsigned vaddr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)sV;
  const unsigned a0 = vaddr + rowb * 128 + vswz(rowb, chunk0) * 16 + inb;
  const unsigned a1 = vaddr + rowb * 128 + vswz(rowb, chunk0 + 4) * 16 + inb;
#pragma unroll
  for (int db = 0; db < 2; ++db) {
    bf16x4 v0, v1, v2, v3, v4, v5, v6, v7;
    // the transposing reads WITHOUT an lgkmcnt wait inside the statement (the shipped kernel waits there: hipcc cannot see these reads complete);
    // here the probe waits once, after issuing both d-blocks' reads, so that they can fly under whatever the scheduler puts between
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
    for (int q = 0; q < 2; ++q) {
      o[q][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf0, pf[q][0], o[q][db], 0, 0, 0);
      o[q][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf1, pf[q][1], o[q][db], 0, 0, 0);
      o[q][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf2, pf[q][2], o[q][db], 0, 0, 0);
      o[q][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf3, pf[q][3], o[q][db], 0, 0, 0);
    }
  }
}

template <int MODE, int WGS>
__global__ __launch_bounds__(256, WGS) void probe(int iters, unsigned long long* out, float* sink) {
  __shared__ __attribute__((aligned(16))) char smem[2 * 64 * 128];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5, g16 = lane >> 4, i16 = lane & 15, tq = i16 >> 2, tp = i16 & 3;
  for (int i = tid; i < 2 * 64 * 128 / 2; i += 256) {
    const unsigned r = (i * 2654435761u) >> 9;
    reinterpret_cast<unsigned short*>(smem)[i] = (unsigned short)(0x3c00u | (r & 0x1ffu) | ((r >> 10 & 1u) << 15));      // bf16 of +-0.5..1 magnitude
  }
  bf16x8 qf[2][4];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int i = 0; i < 8; ++i) qf[q][t][i] = (__bf16)(0.05f * ((lane * 7 + q * 3 + t * 5 + i) % 17 - 8));
  f32x16 o[2][2], s0[2][2], s1[2][2];
  bf16x8 p0[2][4], p1[2][4];
  float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};
#pragma unroll
  for (int q = 0; q < 2; ++q) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[q][0][r] = 0.f; o[q][1][r] = 0.f; s0[q][0][r] = 0.f; s0[q][1][r] = 0.f; s1[q][0][r] = 0.f; s1[q][1][r] = 0.f; }
#pragma unroll
    for (int t = 0; t < 4; ++t) { p0[q][t] = bf16x8{}; p1[q][t] = bf16x8{}; }
  }
  __syncthreads();
  const char* sK = smem; const char* sV = smem + 64 * 128;
  const float c = 0.125f * 1.44269504f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (MODE == 0) {
    for (int it = 0; it < iters; ++it) {
      qk(sK, qf, s0, lr, lh);
      softmax(s0, o, m_run, l_run, p0, c);
      pv(sV, p0, o, lh, g16, tq, tp);
    }
  } else {
    // two tiles per loop trip so that the buffers rotate by name: stage A = K Q^T of the NEXT tile, B = softmax of the CURRENT one, C = P V of the PREVIOUS one
    qk(sK, qf, s0, lr, lh);
    for (int it = 0; it < iters; it += 2) {
      qk(sK, qf, s1, lr, lh);  softmax(s0, o, m_run, l_run, p0, c);  pv(sV, p1, o, lh, g16, tq, tp);
      qk(sK, qf, s0, lr, lh);  softmax(s1, o, m_run, l_run, p1, c);  pv(sV, p0, o, lh, g16, tq, tp);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float acc = 0.f;
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc += o[q][0][r] + o[q][1][r] + s0[q][0][r] + s1[q][1][r];
  acc += l_run[0] + l_run[1];
  if (acc == 1234.5678f) sink[0] = acc;
  if (lane == 0) out[blockIdx.x * 4 + wid] = t1 - t0;
}

template <int MODE, int WGS>
static void run(const char* what, int iters, int cus, unsigned long long* d_out, float* d_sink) {
  const int grid = cus * WGS;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<MODE, WGS>), dim3(grid), dim3(256), 0, 0, 8, d_out, d_sink);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((probe<MODE, WGS>), dim3(grid), dim3(256), 0, 0, iters, d_out, d_sink);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(grid * 4);
  (void)hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
  double cyc = 0; for (auto v : h) cyc += (double)v; cyc /= h.size();
  const double tiles_per_simd = (double)iters * WGS;      // wave-tiles a SIMD completed
  printf("%-58s  %7.0f cycles per tile and wave   %7.1f ns of SIMD time per wave-tile   (%s)\n", what, cyc / iters, ms * 1e6 / tiles_per_simd,
         hipGetErrorString(hipGetLastError()));
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 2000;
  hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr, 0);
  const int cus = pr.multiProcessorCount;
  unsigned long long* d_out; float* d_sink;
  (void)hipMalloc((void**)&d_out, (size_t)cus * 2 * 4 * 8); (void)hipMalloc((void**)&d_sink, 64);
  printf("%s, %d CUs, %d tiles per wave; 64 queries x 64 keys x head_dim 64 per wave-tile, K / V resident in LDS\n", pr.name, cus, iters);
  run<0, 1>("serial stages, one wave per SIMD", iters, cus, d_out, d_sink);
  run<0, 2>("serial stages, two waves per SIMD (the shipped shape)", iters, cus, d_out, d_sink);
  run<1, 1>("three stages of three tiles per trip, one wave per SIMD", iters, cus, d_out, d_sink);
  run<1, 2>("three stages of three tiles per trip, two waves per SIMD", iters, cus, d_out, d_sink);
  return 0;
}
