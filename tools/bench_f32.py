#!/usr/bin/env python3
"""fp32 MFMA GEMM micro-benchmark on the decoder shapes (M = B*Q)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dinov2_od_amd import _native as nat
from tools.bench_ops import timeit
L = nat.lib(); dev = torch.device("cuda:0")
M = 6400
g = torch.Generator().manual_seed(0)
for N, K, act in ((2304, 768, 0), (768, 768, 0), (1024, 768, 1), (768, 1024, 0), (384, 768, 1), (91, 768, 0), (50, 768, 0)):
    A = torch.randn(M, K, generator=g).to(dev); W = (torch.randn(N, K, generator=g) * 0.05).to(dev)
    bias = torch.randn(N, generator=g).to(dev); x = torch.randn(M, N, generator=g).to(dev)
    out = torch.empty(M, N, device=dev)
    f = lambda: L.dod_op_linear(0, nat.ptr(A), K, nat.ptr(W), K, M, N, K, nat.ptr(bias), None, nat.ptr(x), N, nat.ptr(out), 0, N, act, nat.stream_ptr())
    t = timeit(f, iters=10)
    print(f"f32 M={M} N={N:5d} K={K:5d}: {t*1e6:8.1f} us  {2.0*M*N*K/t/1e12:6.1f} TF")
