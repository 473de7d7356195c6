#!/usr/bin/env python3
"""One launch set of the four ViT-B block GEMM shapes + attention (for rocprofv3 --pmc passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dinov2_od_amd import _native as nat
L = nat.lib()
dev = torch.device("cuda:0")
import os as _o
B, N, D = int(_o.environ.get("PROF_B", "8")), 1370, 768
M = B * N
g = torch.Generator().manual_seed(0)
rnd = lambda *s: (torch.randn(*s, generator=g) * 0.5).to(dev)
for name, m, n, k, epi, obf in [("qkv", M, 3 * D, D, "none", True), ("proj", M, D, D, "resid", False), ("fc1", M, 4 * D, D, "gelu", True), ("fc2", M, D, 4 * D, "resid", False)]:
    A = rnd(m, k).to(torch.bfloat16); W = (rnd(n, k) * 0.1).to(torch.bfloat16); bias = rnd(n); scale = rnd(n); x = rnd(m, n)
    out = torch.empty(m, n, device=dev, dtype=torch.bfloat16 if obf else torch.float32)
    for _ in range(3):
        if epi == "resid":
            L.dod_op_linear(1, nat.ptr(A), k, nat.ptr(W), k, m, n, k, nat.ptr(bias), nat.ptr(scale), nat.ptr(x), n, nat.ptr(x), 0, n, 0, nat.stream_ptr())
        else:
            L.dod_op_linear(1, nat.ptr(A), k, nat.ptr(W), k, m, n, k, nat.ptr(bias), None, None, 0, nat.ptr(out), 1, n, 2 if epi == "gelu" else 0, nat.stream_ptr())
    torch.cuda.synchronize()
qkv = rnd(B, N, 3 * D).to(torch.bfloat16); ctx = torch.empty(B, N, D, device=dev, dtype=torch.bfloat16)
for _ in range(3):
    L.dod_op_attention_bf16(nat.ptr(qkv), nat.ptr(ctx), B, N, D // 64, 0.125, nat.stream_ptr())
torch.cuda.synchronize()
