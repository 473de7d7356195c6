#!/usr/bin/env python3
"""three launches each of the H2 QKV and fc2 GEMMs, the split-product QKV GEMM and (PROF_FP8=1) the fp8 ping-pong GEMM at the bench
shapes, for rocprofv3 --pmc (LDS bank conflicts, MFMA busy cycles, VALU / LDS instruction counts)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dinov2_od_amd import _native as nat
L = nat.lib(); dev = torch.device("cuda:0")
M, D = int(os.environ.get("PROF_B", "64")) * 1370, 768
g = torch.Generator().manual_seed(0)
def h2(x, weight=False):
    out = torch.empty(x.shape[0], (3 if weight else 4) * x.shape[1], dtype=torch.uint8, device=dev)
    we = torch.empty(x.shape[0], dtype=torch.uint8, device=dev) if weight else None
    nat.check(L.dod_op_split_h2(nat.ptr(x), x.stride(0), x.shape[0], x.shape[1], nat.ptr(out), nat.ptr(we), nat.stream_ptr())); return out, we
def pair(x):
    out = torch.empty(x.shape[0], 2 * x.shape[1], dtype=torch.bfloat16, device=dev)
    nat.check(L.dod_op_split_pair(nat.ptr(x), x.stride(0), x.shape[0], x.shape[1], nat.ptr(out), nat.stream_ptr())); return out
for n, k in ((3 * D, D), (D, 4 * D)):
    a = (torch.randn(M, k, generator=g) * 0.5).to(dev); w = (torch.randn(n, k, generator=g) * 0.05).to(dev)
    Ah, _ = h2(a); Wh, we = h2(w, True); bias = torch.randn(n, generator=g).to(dev)
    out = torch.empty(M, 2 * n, dtype=torch.bfloat16, device=dev)
    for _ in range(3):
        nat.check(L.dod_op_linear_h2(nat.ptr(Ah), nat.ptr(Wh), nat.ptr(we), M, n, k, nat.ptr(bias), None, None, 0, nat.ptr(out), 2, 2 * n, 0, nat.stream_ptr()))
    if n == 3 * D:
        A2, W2 = pair(a), pair(w)
        for _ in range(3):
            nat.check(L.dod_op_linear_x3(nat.ptr(A2), nat.ptr(W2), M, n, k, nat.ptr(bias), None, None, 0, nat.ptr(out), 2, 2 * n, 0, nat.stream_ptr()))
    del Ah, Wh, a, w
torch.cuda.synchronize()
