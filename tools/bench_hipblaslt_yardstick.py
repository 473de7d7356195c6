#!/usr/bin/env python3
"""Yardstick only (nothing in the product calls it): torch.matmul (hipBLASLt / rocBLAS) on the bench workload's GEMM shapes,
bf16 in / bf16 out, no epilogue, next to the hand-written kernels through dod_op_linear (bias epilogue, bf16 out)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dinov2_od_amd import _native as nat
from tools.bench_ops import timeit
L = nat.lib(); dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
M = 64 * 1370
for name, n, k in (("qkv", 2304, 768), ("proj", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072)):
    A = (torch.randn(M, k, generator=g) * 0.5).to(dev).to(torch.bfloat16); W = (torch.randn(n, k, generator=g) * 0.05).to(dev).to(torch.bfloat16)
    bias = torch.randn(n, generator=g).to(dev); out = torch.empty(M, n, dtype=torch.bfloat16, device=dev)
    Wt = W.t()
    t_lib = timeit(lambda: torch.matmul(A, Wt, out=out))
    t_own = timeit(lambda: L.dod_op_linear(1, nat.ptr(A), k, nat.ptr(W), k, M, n, k, nat.ptr(bias), None, None, 0, nat.ptr(out), 1, n, 0, nat.stream_ptr()))
    fl = 2.0 * M * n * k
    print(f"{name:5s} M={M} N={n} K={k}: torch.matmul {t_lib*1e6:7.1f} us {fl/t_lib/1e12:6.1f} TFLOP/s | this library (bias, bf16 out) {t_own*1e6:7.1f} us {fl/t_own/1e12:6.1f} TFLOP/s")
