#!/usr/bin/env python3
"""Matcher cost matrices on device (dod_match_cost) vs the torch-CPU restatement of matching.py:79-98, bench workload's
shapes (B=64, Q=100, C=91, 0..30 targets per image)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from dinov2_od_amd import matching as mt, synth
from oracle import matching_oracle as mo

B, Q, C = 64, 100, 91
counts = (synth.uniform01(3, "bm.counts", (B,)) * 31).astype(int)
G = int(counts.sum())
det = np.concatenate([synth.normal(1, "bm.l", (B, Q, C), 2.0), 0.15 + 0.7 * synth.uniform01(1, "bm.c", (B, Q, 2)),
                      0.05 + 0.4 * synth.uniform01(1, "bm.w", (B, Q, 2))], -1).astype(np.float32)
labels = (synth.uniform01(1, "bm.lab", (G,)) * C).astype(np.int64).clip(0, C - 1)
gt = np.concatenate([0.15 + 0.7 * synth.uniform01(1, "bm.g1", (G, 2)), 0.05 + 0.4 * synth.uniform01(1, "bm.g2", (G, 2))], -1).astype(np.float32)
offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
d, l, g, o = [torch.from_numpy(x).cuda() for x in (det, labels, gt, offs)]
run = lambda: mt.match_cost(d, C, l, g, o, rows_from=-1)
for _ in range(5): run()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(50): run()
b.record(); torch.cuda.synchronize()
t_k = a.elapsed_time(b) / 50 * 1e-3
t0 = time.perf_counter()
for _ in range(10): c = run().cpu()
t_h = (time.perf_counter() - t0) / 10
t0 = time.perf_counter(); want = mo.cost_matrices(det, C, labels, gt, offs, rows_from=-1); t_cpu = time.perf_counter() - t0
alg = det.nbytes + G * 24 + G * Q * 4
print(f"B={B} Q={Q} C={C} G={G}: kernel+alloc {t_k*1e6:.1f} us ({alg/1e6:.2f} MB algorithmic), with D2H {t_h*1e6:.0f} us, "
      f"torch-CPU restatement (as the reference: one [B*Q, n] matrix per image) {t_cpu*1e3:.1f} ms -> {t_cpu/t_h:.0f}x")
