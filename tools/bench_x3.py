#!/usr/bin/env python3
"""Split-product (bf16x3) GEMM and attention on the ViT-B shapes of the bench workload (batch 64)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dinov2_od_amd import _native as nat
from tools.bench_ops import timeit
L = nat.lib(); dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
B = int(os.environ.get("X3_B", "64")); N = 1370; D = 768; M = B * N
def pair(x):
    out = torch.empty(x.shape[0], 2 * x.shape[1], dtype=torch.bfloat16, device=dev)
    nat.check(L.dod_op_split_pair(nat.ptr(x), x.stride(0), x.shape[0], x.shape[1], nat.ptr(out), nat.stream_ptr())); return out
for name, n, k, layout, act, resid in (("qkv", 3 * D, D, 2, 0, False), ("proj", D, D, 0, 0, True), ("fc1", 4 * D, D, 2, 2, False), ("fc2", D, 4 * D, 0, 0, True)):
    A2 = pair((torch.randn(M, k, generator=g) * 0.5).to(dev)); W2 = pair((torch.randn(n, k, generator=g) * 0.05).to(dev))
    bias = torch.randn(n, generator=g).to(dev); x = torch.randn(M, n, generator=g).to(dev) if resid else None
    out = x if resid else torch.empty(M, 2 * n if layout == 2 else n, dtype=torch.bfloat16 if layout else torch.float32, device=dev)
    t = timeit(lambda: nat.check(L.dod_op_linear_x3(nat.ptr(A2), nat.ptr(W2), M, n, k, nat.ptr(bias), None, nat.ptr(x), n if resid else 0, nat.ptr(out), layout, out.shape[1], act, nat.stream_ptr())))
    print(f"gemm_x3 {name:5s} M={M} N={n} K={k}: {t*1e6:8.1f} us  {2.0*M*n*k/t/1e12:6.1f} TFLOP/s algorithmic ({6.0*M*n*k/t/1e12:6.1f} executed)")
qkv2 = pair((torch.randn(M, 3 * D, generator=g) * 0.5).to(dev)); ctx2 = torch.empty(M, 2 * D, dtype=torch.bfloat16, device=dev)
t = timeit(lambda: nat.check(L.dod_op_attention_x3(nat.ptr(qkv2), nat.ptr(ctx2), B, N, D // 64, 0.125, nat.stream_ptr())))
print(f"attn_x3 B={B}: {t*1e6:8.1f} us  {4.0*B*N*N*D/t/1e12:6.1f} TFLOP/s algorithmic")
