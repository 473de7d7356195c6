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

# attention yardstick: torch's scaled_dot_product_attention (the ROCm flash kernel) vs attn_bf16.hip, same problem
import torch.nn.functional as F
B, N, H, D = 64, 1370, 12, 768
qkv = (torch.randn(B, N, 3 * D, generator=g) * 0.5).to(dev).to(torch.bfloat16)
ctx = torch.empty(B, N, D, device=dev, dtype=torch.bfloat16)
q, k, v = [t.view(B, N, H, 64).transpose(1, 2).contiguous() for t in qkv.split(D, dim=-1)]
fl = 4.0 * B * N * N * D
try:
    t_lib = timeit(lambda: F.scaled_dot_product_attention(q, k, v))
    print(f"attention B={B} N={N} heads={H}: torch SDPA {t_lib*1e6:7.1f} us {fl/t_lib/1e12:6.1f} TFLOP/s (head-major contiguous q/k/v, layout copy not counted)")
except Exception as e:
    print("torch SDPA failed:", type(e).__name__, e)
t_own = timeit(lambda: L.dod_op_attention_bf16(nat.ptr(qkv), nat.ptr(ctx), B, N, H, 0.125, nat.stream_ptr()))
print(f"attention B={B} N={N} heads={H}: attn_bf16.hip {t_own*1e6:7.1f} us {fl/t_own/1e12:6.1f} TFLOP/s (reads the fused-QKV GEMM output in place)")
