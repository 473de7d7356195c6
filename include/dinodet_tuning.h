/* Entry points of -DDINODET_TUNING builds of libdinodet.so (`python -m dinov2_od_amd._build --tuning` -> lib/libdinodet_tuning.so, loaded by
 * tools/ through DINODET_LIB).  NOT part of the drop-in boundary: the release library exports none of these, and no environment variable
 * of the tuning rounds (DINODET_GEMM_TILE, _GM, _ORDER, _STAGGER, _WRES, _KSPLIT0, _TAILSPLIT, DINODET_X3_TILE, DINODET_ATTN_NQ, ...) is read by it. */
#ifndef DINODET_TUNING_H
#define DINODET_TUNING_H
#ifdef __cplusplus
extern "C" {
#endif
/* When dev_buf is non-NULL the 256x128 bf16 GEMM kernel stores 4 x uint64 per workgroup {realtime at start, after the K loop, at exit,
 * blockIdx} (100 MHz s_memrealtime); NULL switches it off (tools/gemm_timeline.py). */
int dod_debug_gemm_stamps(void* dev_buf);
/* same for the ping-pong kernels (gemm_pp.hip): 8 x uint64 per workgroup, shader cycles (tools/pp_timeline.py) */
int dod_debug_pp_stamps(void* dev_buf);
/* same for the bf16 attention kernel: {shader cycles in the tile loop, cycles waiting for DMA + barrier, tiles, active} */
int dod_debug_attn_stamps(void* dev_buf);
/* register-only MFMA loop (shape 16: v_mfma_f32_16x16x32_bf16 x 8 chains, 32: 32x32x16 x 4 chains, 2: v_mfma_f32_32x32x2_f32 x 4 chains,
 * 1: the same as one dependent chain; iters < 0: random operands), `blocks` workgroups of 4 waves, `iters` rounds;
 * dev_out[block*4 + {0,1}] = {shader cycles, 100-MHz ticks}.  Measures the SUSTAINED matrix rate and clock of the part under MFMA load
 * (tools/mfma_peak.py); not used by the forward. */
int dod_debug_mfma_peak(int shape, int iters, int blocks, void* dev_out, void* stream);
/* MFMA / VALU co-issue probe: per iteration 4 independent 32x32x16 MFMAs (mode & 1) and nvalu (16 | 28 | 56) independent
 * v_fma_f32 (mode & 2), interleaved; dev_out[block*4] = shader cycles of wave 0 (tools/mfma_peak.py). */
int dod_debug_mfma_valu_probe(int nvalu, int mode, int iters, int blocks, void* dev_out, void* stream);
#ifdef __cplusplus
}
#endif
#endif
