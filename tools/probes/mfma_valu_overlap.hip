// Probe (GPU box only; no torch): can a gfx950 SIMD run VALU / transcendental work while its matrix pipe executes an MFMA -- (a) from ANOTHER wave on
// the same SIMD, (b) from the SAME wave (independent instructions in program order)?  The flash attention kernels' tile time equals VALU + MFMA
// cycles added up; this says whether a better schedule could hide one under the other.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap tools/probes/mfma_valu_overlap.hip && ./mfma_valu_overlap
// One workgroup of 512 threads per CU (8 waves: waves w and w + 4 share SIMD w), modes:
//   0  every wave: MFMA only (32 x v_mfma_f32_32x32x16_bf16 per iteration, four accumulator chains)
//   1  every wave: VALU only (64 x v_exp_f32 + 192 x v_pk_fma_f32 per iteration: the softmax's mix)
//   2  every wave: both, clustered (32 MFMAs, then the VALU block)
//   3  every wave: both, interleaved in program order (1 MFMA, 2 v_exp, 6 v_pk_fma, repeated)
//   4  waves 0-3 MFMA only, waves 4-7 VALU only (same instruction counts per wave as modes 0 / 1)
//   5  every wave: MFMA + plain VALU only (no transcendentals; 256 x v_pk_fma_f32), interleaved
//   6  every wave: MFMA + v_exp only (64 per iteration), interleaved
//   7  every wave: MFMA + 256 x v_fma_f32 (PLAIN fp32, one value per lane), interleaved
//   8  waves 0-3 MFMA only, waves 4-7 256 x v_fma_f32 only
//   9  waves 0-3 MFMA only, waves 4-7 256 x v_pk_fma_f32 only
//  10  every wave: MFMA + 128 x v_max3_f32 + 128 x v_cvt_pk_bf16_f32, interleaved
//  11  every wave: 256 x v_fma_f32 only;   12  every wave: 256 x v_pk_fma_f32 only
//  13  every wave: 256 x v_exp_f32 only;   14  waves 0-3 256 x v_exp_f32 only, waves 4-7 256 x v_fma_f32 only
//  15  every wave: 64 x v_exp_f32 + 256 x v_fma_f32 interleaved (1 : 4), no MFMA
//  16  waves 0-3 MFMA only, waves 4-7 the interleaved 64 x v_exp_f32 + 256 x v_fma_f32 of mode 15 (a softmax beside another wave's MFMAs)
//  17  waves 0-3 MFMA only, waves 4-7 256 x v_exp_f32 only
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;

#define MFMA(acc_) acc_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc_, 0, 0, 0)
#define EXP2(i_) e[i_] = __builtin_amdgcn_exp2f(e[i_])
#define PKF(i_) p[i_] = __builtin_elementwise_fma(p[i_], c2, d2)
#define SF(i_) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i_]) : "v"(c2.x), "v"(d2.x))
#define PF(i_) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i_]) : "v"(c2), "v"(d2))
#define MX3(i_) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(f[i_]) : "v"(c2.x), "v"(d2.x))
#define CVT(i_) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[i_]) : "v"(f[i_]), "v"(c2.y))

template <int MODE>
__global__ __launch_bounds__(512) void probe(int iters, unsigned long long* out, float* sink) {
  const int wid = threadIdx.x >> 6;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.5f + 0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.25f + 0.002f * (threadIdx.x * 3 + i)); }
  f32x16 acc0 = {}, acc1 = {}, acc2 = {}, acc3 = {};
  float e[8];
  f32x2 p[8];
  const f32x2 c2 = {0.999f, 1.001f}, d2 = {1e-3f, -1e-3f};
  float f[8]; unsigned u[8];
  for (int i = 0; i < 8; ++i) { e[i] = -0.01f * (threadIdx.x + i); p[i] = f32x2{0.5f + i, 0.25f - i}; f[i] = 0.125f * i + threadIdx.x; u[i] = 0; }
  const bool do_m = MODE == 0 || MODE == 2 || MODE == 3 || MODE == 5 || MODE == 6 || (MODE == 4 && wid < 4);
  const bool do_v = MODE == 1 || MODE == 2 || MODE == 3 || (MODE == 4 && wid >= 4);
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 7 || MODE == 10) {
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        MFMA(acc0); if (MODE == 7) { SF(0); SF(1); SF(2); SF(3); SF(4); SF(5); SF(6); SF(7); } else { MX3(0); CVT(0); MX3(1); CVT(1); MX3(2); CVT(2); MX3(3); CVT(3); }
        MFMA(acc1); if (MODE == 7) { SF(0); SF(1); SF(2); SF(3); SF(4); SF(5); SF(6); SF(7); } else { MX3(4); CVT(4); MX3(5); CVT(5); MX3(6); CVT(6); MX3(7); CVT(7); }
        MFMA(acc2); if (MODE == 7) { SF(0); SF(1); SF(2); SF(3); SF(4); SF(5); SF(6); SF(7); } else { MX3(0); CVT(0); MX3(1); CVT(1); MX3(2); CVT(2); MX3(3); CVT(3); }
        MFMA(acc3); if (MODE == 7) { SF(0); SF(1); SF(2); SF(3); SF(4); SF(5); SF(6); SF(7); } else { MX3(4); CVT(4); MX3(5); CVT(5); MX3(6); CVT(6); MX3(7); CVT(7); }
      }
    } else if ((MODE == 16 || MODE == 17) && wid < 4) {
#pragma unroll
      for (int g = 0; g < 8; ++g) { MFMA(acc0); MFMA(acc1); MFMA(acc2); MFMA(acc3); }
      asm volatile("" ::: "memory");
    } else if (MODE == 13 || MODE == 14 || MODE == 15 || MODE == 16 || MODE == 17) {
#pragma unroll
      for (int g = 0; g < 32; ++g) {
        if (MODE == 13 || MODE == 17 || (MODE == 14 && wid < 4)) { EXP2(0); EXP2(1); EXP2(2); EXP2(3); EXP2(4); EXP2(5); EXP2(6); EXP2(7); }
        else if (MODE == 14) { SF(0); SF(1); SF(2); SF(3); SF(4); SF(5); SF(6); SF(7); }
        else { EXP2(g & 7); SF(0); SF(1); SF(2); SF(3); EXP2((g + 4) & 7); SF(4); SF(5); SF(6); SF(7); }
      }
      asm volatile("" ::: "memory");
    } else if (MODE == 8 || MODE == 9 || MODE == 11 || MODE == 12) {
      if ((MODE == 8 || MODE == 9) && wid < 4) {
#pragma unroll
        for (int g = 0; g < 8; ++g) { MFMA(acc0); MFMA(acc1); MFMA(acc2); MFMA(acc3); }
      } else {
#pragma unroll
        for (int g = 0; g < 32; ++g) {
          if (MODE == 8 || MODE == 11) { SF(0); SF(1); SF(2); SF(3); SF(4); SF(5); SF(6); SF(7); }
          else { PF(0); PF(1); PF(2); PF(3); PF(4); PF(5); PF(6); PF(7); }
        }
      }
      asm volatile("" ::: "memory");
    } else if (MODE == 3 || MODE == 5 || MODE == 6) {
#pragma unroll
      for (int g = 0; g < 8; ++g) {      // 8 groups x 4 MFMAs = 32; per MFMA: 2 exp + 6 pk_fma (mode 3), 8 pk_fma (5), 2 exp (6)
        MFMA(acc0); if (MODE != 5) { EXP2(0); EXP2(1); } if (MODE != 6) { PKF(0); PKF(1); PKF(2); PKF(3); PKF(4); PKF(5); if (MODE == 5) { PKF(6); PKF(7); } }
        MFMA(acc1); if (MODE != 5) { EXP2(2); EXP2(3); } if (MODE != 6) { PKF(6); PKF(7); PKF(0); PKF(1); PKF(2); PKF(3); if (MODE == 5) { PKF(4); PKF(5); } }
        MFMA(acc2); if (MODE != 5) { EXP2(4); EXP2(5); } if (MODE != 6) { PKF(4); PKF(5); PKF(6); PKF(7); PKF(0); PKF(1); if (MODE == 5) { PKF(2); PKF(3); } }
        MFMA(acc3); if (MODE != 5) { EXP2(6); EXP2(7); } if (MODE != 6) { PKF(2); PKF(3); PKF(4); PKF(5); PKF(6); PKF(7); if (MODE == 5) { PKF(0); PKF(1); } }
        asm volatile("" ::: "memory");
      }
    } else {
      if (do_m) {
#pragma unroll
        for (int g = 0; g < 8; ++g) { MFMA(acc0); MFMA(acc1); MFMA(acc2); MFMA(acc3); }
      }
      asm volatile("" ::: "memory");
      if (do_v) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
          EXP2(0); EXP2(1); EXP2(2); EXP2(3); EXP2(4); EXP2(5); EXP2(6); EXP2(7);
#pragma unroll
          for (int r = 0; r < 3; ++r) { PKF(0); PKF(1); PKF(2); PKF(3); PKF(4); PKF(5); PKF(6); PKF(7); }
        }
      }
      asm volatile("" ::: "memory");
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i] + acc2[i] + acc3[i];
  for (int i = 0; i < 8; ++i) s += e[i] + p[i].x + p[i].y + f[i] + (float)u[i];
  if (s == 1234.5678f) sink[0] = s;
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wid] = t1 - t0;
}

template <int MODE>
static void run(const char* what, int iters, int cus, unsigned long long* d_out, float* d_sink) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<MODE>, dim3(cus), dim3(512), 0, 0, 10, d_out, d_sink);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(probe<MODE>, dim3(cus), dim3(512), 0, 0, iters, d_out, d_sink);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(cus * 8);
  (void)hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
  double cm = 0, cv = 0;
  for (int i = 0; i < cus; ++i) for (int w = 0; w < 8; ++w) (w < 4 ? cm : cv) += (double)h[i * 8 + w];
  cm /= cus * 4.0; cv /= cus * 4.0;
  // s_memtime counts at 100 MHz on this part: report wall time per iteration instead, and the counter ratio
  printf("mode %d  %-62s  %8.1f ns / iteration  (waves 0-3: %7.2f ticks / it, waves 4-7: %7.2f)\n", MODE, what, ms * 1e6 / iters, cm / iters, cv / iters);
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 4000;
  hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr, 0);
  const int cus = pr.multiProcessorCount;
  unsigned long long* d_out; float* d_sink;
  (void)hipMalloc((void**)&d_out, (size_t)cus * 8 * 8); (void)hipMalloc((void**)&d_sink, 64);
  printf("%s, %d CUs, %d iterations; per iteration and wave: 32 x v_mfma_f32_32x32x16_bf16 and / or 64 x v_exp_f32 + 192 x v_pk_fma_f32; 2 waves per SIMD\n", pr.name, cus, iters);
  run<0>("MFMA only", iters, cus, d_out, d_sink);
  run<1>("VALU only (64 exp + 192 pk_fma)", iters, cus, d_out, d_sink);
  run<2>("both, clustered (32 MFMA, then the VALU block)", iters, cus, d_out, d_sink);
  run<3>("both, interleaved in program order", iters, cus, d_out, d_sink);
  run<4>("waves 0-3 MFMA only, waves 4-7 VALU only (share SIMDs)", iters, cus, d_out, d_sink);
  run<5>("MFMA + 256 pk_fma, interleaved (no transcendentals)", iters, cus, d_out, d_sink);
  run<6>("MFMA + 64 exp, interleaved", iters, cus, d_out, d_sink);
  run<7>("MFMA + 256 plain v_fma_f32, interleaved", iters, cus, d_out, d_sink);
  run<8>("waves 0-3 MFMA only, waves 4-7 256 plain v_fma_f32 only", iters, cus, d_out, d_sink);
  run<9>("waves 0-3 MFMA only, waves 4-7 256 v_pk_fma_f32 only", iters, cus, d_out, d_sink);
  run<10>("MFMA + 128 v_max3_f32 + 128 v_cvt_pk_bf16_f32, interleaved", iters, cus, d_out, d_sink);
  run<11>("256 plain v_fma_f32 only", iters, cus, d_out, d_sink);
  run<12>("256 v_pk_fma_f32 only", iters, cus, d_out, d_sink);
  run<13>("256 v_exp_f32 only", iters, cus, d_out, d_sink);
  run<14>("waves 0-3 256 v_exp_f32 only, waves 4-7 256 plain v_fma_f32 only", iters, cus, d_out, d_sink);
  run<15>("64 v_exp_f32 + 256 plain v_fma_f32 interleaved, no MFMA", iters, cus, d_out, d_sink);
  run<16>("waves 0-3 MFMA only, waves 4-7 64 exp + 256 plain fma interleaved", iters, cus, d_out, d_sink);
  run<17>("waves 0-3 MFMA only, waves 4-7 256 v_exp_f32 only", iters, cus, d_out, d_sink);
  return 0;
}
