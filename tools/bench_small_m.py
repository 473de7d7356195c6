#!/usr/bin/env python3
"""Plain-bf16 GEMM at the decoder's query-side shapes (few thousand rows, K' = 3K of the split product): tile choice A/B through
DINODET_GEMM_TILE (read per call).  Usage: python tools/bench_small_m.py"""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools._tuning_lib  # noqa: F401,E402  (the -DDINODET_TUNING build: this tool uses tuning hooks)
import torch
from dinov2_od_amd import _native as nat
from tools.bench_ops import timeit
L = nat.lib(); dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
tiles = [t for t in os.environ.get("TILES", "default,128,8").split(",")]
for M in [int(v) for v in os.environ.get("MS", "800,2400,3200,4800,6400,9600").split(",")]:
    for N, K in [tuple(int(u) for u in v.split("x")) for v in os.environ.get("NK", "768x2304,2304x2304,1024x2304,768x3072,768x768").split(",")]:
        A = (torch.randn(M, K, generator=g) * 0.5).to(dev).to(torch.bfloat16)
        W = (torch.randn(N, K, generator=g) * 0.05).to(dev).to(torch.bfloat16)
        bias = torch.randn(N, generator=g).to(dev)
        out = torch.empty(M, N, device=dev, dtype=torch.float32)
        def setv(t):
            os.environ.pop("DINODET_GEMM_TILE", None)
            if t != "default": os.environ["DINODET_GEMM_TILE"] = t
        f = lambda: L.dod_op_linear(1, nat.ptr(A), K, nat.ptr(W), K, M, N, K, nat.ptr(bias), None, None, 0, nat.ptr(out), 0, N, 0, nat.stream_ptr())
        ref, devs, times, bad = None, {}, {t: [] for t in tiles}, {}
        for t in tiles:                       # correctness: every variant against the first
            setv(t)
            out.zero_()
            rc = f(); torch.cuda.synchronize()
            if rc: bad[t] = rc; continue
            o = out.clone()
            if ref is None: ref = o
            devs[t] = (o - ref).abs().max().item()
        for _ in range(int(os.environ.get("ROUNDS", "5"))):      # interleaved rounds, median per variant (the clock drifts over a run)
            for t in tiles:
                if t in bad: continue
                setv(t)
                times[t].append(timeit(f, iters=20))
        row = [f"{t}: rc={bad[t]}" if t in bad else f"{t}: {statistics.median(times[t])*1e6:6.1f} us {2.0*M*N*K/statistics.median(times[t])/1e12:6.1f} TF d={devs[t]:.1e}" for t in tiles]
        os.environ.pop("DINODET_GEMM_TILE", None)
        print(f"M={M:5d} N={N:4d} K={K:4d} | " + " | ".join(row), flush=True)
