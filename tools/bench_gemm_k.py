#!/usr/bin/env python3
"""GEMM time vs K at fixed M,N: separates the per-K-tile cost from the fixed prologue+epilogue cost."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dinov2_od_amd import _native as nat
from tools.bench_ops import timeit
L = nat.lib(); dev = torch.device("cuda:0")
M = int(os.environ.get("GM", "87680"))
g = torch.Generator().manual_seed(0)
for N, obf in ((2304, True), (768, False)):
    for K in (64, 128, 256, 768, 1536, 3072):
        A = (torch.randn(M, K, generator=g) * 0.5).to(dev).to(torch.bfloat16)
        W = (torch.randn(N, K, generator=g) * 0.05).to(dev).to(torch.bfloat16)
        bias = torch.randn(N, generator=g).to(dev); scale = torch.randn(N, generator=g).to(dev)
        x = torch.randn(M, N, generator=g).to(dev)
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        if obf:
            f = lambda: L.dod_op_linear(1, nat.ptr(A), K, nat.ptr(W), K, M, N, K, nat.ptr(bias), None, None, 0, nat.ptr(out), 1, N, 0, nat.stream_ptr())
        else:
            f = lambda: L.dod_op_linear(1, nat.ptr(A), K, nat.ptr(W), K, M, N, K, nat.ptr(bias), nat.ptr(scale), nat.ptr(x), N, nat.ptr(x), 0, N, 0, nat.stream_ptr())
        t = timeit(f, iters=10)
        print(f"M={M} N={N} K={K:5d} {'bf16out' if obf else 'resid  '}: {t*1e6:8.1f} us  {2.0*M*N*K/t/1e12:7.1f} TF")
