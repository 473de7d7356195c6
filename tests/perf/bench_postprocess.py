#!/usr/bin/env python3
"""Device post-processing (dod_postprocess) vs the CPU loop it replaces, on the bench workload's detections
(B=64, Q=100, C=91).  Prints kernel time (HIP events, detections resident in HBM), end-to-end time incl. the record
copy to the host, and the numpy oracle's time for the same input."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from dinov2_od_amd import _native as nat, postprocess as pp, synth
from oracle import postprocess_oracle as ppo

B, Q, C = int(os.environ.get("PP_B", 64)), int(os.environ.get("PP_Q", 100)), 91
for mean in (-6.0, -3.0):
    logits = (synth.normal(1, "bench.pp", (B, Q, C), 1.5) + mean).astype(np.float32)
    det_np = np.concatenate([logits, synth.uniform01(2, "bench.ppb", (B, Q, 4)).astype(np.float32)], -1)
    det = torch.from_numpy(det_np).cuda()
    L = nat.lib()
    ws = torch.empty(L.dod_postprocess_workspace_bytes(B, Q, C), dtype=torch.uint8, device="cuda")
    cap = B * Q * (C - 1)
    out = torch.empty(cap * 40, dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    run = lambda: nat.check(L.dod_postprocess(nat.ptr(det), B, Q, C, None, 0.05, nat.ptr(out), cap, nat.ptr(cnt), nat.ptr(ws), ws.numel(), nat.stream_ptr()))
    for _ in range(5): run()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50): run()
    b.record(); torch.cuda.synchronize()
    t_k = a.elapsed_time(b) / 50 * 1e-3
    n = int(cnt.item())
    t0 = time.perf_counter()
    for _ in range(10): rec = pp.postprocess_packed(det, C)
    t_e2e = (time.perf_counter() - t0) / 10
    t0 = time.perf_counter(); want = ppo.postprocess(det_np, C); t_cpu = time.perf_counter() - t0
    alg = det_np.nbytes + n * 40
    print(f"B={B} Q={Q} C={C} kept {n} ({100.0*n/cap:.1f}%): kernels {t_k*1e6:.1f} us ({alg/t_k/1e9:.1f} GB/s of {alg/1e6:.2f} MB algorithmic), "
          f"to host records {t_e2e*1e6:.0f} us, numpy oracle {t_cpu*1e3:.1f} ms -> {t_cpu/t_e2e:.0f}x")
