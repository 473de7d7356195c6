#!/usr/bin/env python3
"""Sustained bf16 MFMA rate and clock of this MI355X under a register-only MFMA loop (no memory traffic):
context for the roofline fractions in DESIGN.md (the 2.5 PFLOP/s denominator assumes the 2.4 GHz peak clock)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools._tuning_lib  # noqa: F401,E402  (the -DDINODET_TUNING build: this tool uses tuning hooks)
import torch
from dinov2_od_amd import _native as nat

L = nat.lib()
dev = torch.device("cuda:0")
for shape, chains, flop in ((16, 8, 2 * 16 * 16 * 32), (32, 4, 2 * 32 * 32 * 16), (2, 4, 2 * 32 * 32 * 2), (1, 4, 2 * 32 * 32 * 2)):   # shape 2 = v_mfma_f32_32x32x2_f32 (4 chains), 1 = the same, one dependent chain
    for wg_per_cu, rnd in ((1, 0), (2, 0), (4, 0), (4, 1)):
        blocks, iters = 256 * wg_per_cu, (-20000 if rnd else 20000)
        out = torch.zeros(blocks * 4, dtype=torch.int64, device=dev)
        for _ in range(2):
            nat.check(L.dod_debug_mfma_peak(shape, iters, blocks, nat.ptr(out), nat.stream_ptr()))
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        nat.check(L.dod_debug_mfma_peak(shape, iters, blocks, nat.ptr(out), nat.stream_ptr()))
        b.record()
        torch.cuda.synchronize()
        t = a.elapsed_time(b) * 1e-3
        o = out.view(blocks, 4).cpu().double()
        cyc, ticks = o[:, 0].median().item(), o[:, 1].median().item()
        iters = abs(iters)
        total = float(blocks) * 4 * iters * chains * flop
        print(f"mfma {shape}{' random operands' if rnd else ''}: {wg_per_cu} WG/CU ({wg_per_cu} waves/SIMD): {total / t / 1e12:7.1f} TFLOP/s wall; "
              f"s_memtime {cyc / (ticks / 100e6) / 1e9:5.2f} G counts/s; {cyc / (iters * chains):6.2f} counts per MFMA per wave")

print("MFMA / VALU co-issue probe (4 x 32x32x16 MFMA = 128 MFMA cycles per iteration; v_fma_f32 = 4 issue cycles each):")
for wg_per_cu in (1, 2):
    for nv in (16, 28, 56):
        row = []
        for mode in (1, 2, 3):
            blocks, iters = 256 * wg_per_cu, 20000
            out = torch.zeros(blocks * 4, dtype=torch.int64, device=dev)
            nat.check(L.dod_debug_mfma_valu_probe(nv, mode, iters, blocks, nat.ptr(out), nat.stream_ptr()))
            torch.cuda.synchronize()
            row.append(out.view(blocks, 4)[:, 0].double().median().item() / iters)
        print(f"  {wg_per_cu} wave(s)/SIMD, {nv} VALU per iteration: MFMA only {row[0]:6.1f}, VALU only {row[1]:6.1f}, both {row[2]:6.1f} cycles/iteration "
              f"(sum {row[0] + row[1]:6.1f}, max {max(row[0], row[1]):6.1f})")
