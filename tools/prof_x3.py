#!/usr/bin/env python3
"""three launches each of the split-product attention and (PROF_GEMM=1) the x3 QKV GEMM at the bench shapes, for rocprofv3 --pmc"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dinov2_od_amd import _native as nat
L = nat.lib(); dev = torch.device("cuda:0")
B, N, D = int(os.environ.get("PROF_B", "64")), 1370, 768
M = B * N
g = torch.Generator().manual_seed(0)
def pair(x):
    out = torch.empty(x.shape[0], 2 * x.shape[1], dtype=torch.bfloat16, device=dev)
    nat.check(L.dod_op_split_pair(nat.ptr(x), x.stride(0), x.shape[0], x.shape[1], nat.ptr(out), nat.stream_ptr())); return out
qkv2 = pair((torch.randn(M, 3 * D, generator=g) * 0.5).to(dev)); ctx2 = torch.empty(M, 2 * D, dtype=torch.bfloat16, device=dev)
for _ in range(3):
    nat.check(L.dod_op_attention_x3(nat.ptr(qkv2), nat.ptr(ctx2), B, N, D // 64, 0.125, nat.stream_ptr()))
if os.environ.get("PROF_GEMM"):
    A2 = pair((torch.randn(M, D, generator=g) * 0.5).to(dev)); W2 = pair((torch.randn(3 * D, D, generator=g) * 0.05).to(dev))
    bias = torch.randn(3 * D, generator=g).to(dev); out = torch.empty(M, 6 * D, dtype=torch.bfloat16, device=dev)
    for _ in range(3):
        nat.check(L.dod_op_linear_x3(nat.ptr(A2), nat.ptr(W2), M, 3 * D, D, nat.ptr(bias), None, None, 0, nat.ptr(out), 2, 6 * D, 0, nat.stream_ptr()))
torch.cuda.synchronize()
