#!/usr/bin/env python3
"""fp8 vs bf16 GEMM on the ViT-g/14 (cfg5: 32 images of 518^2 per GPU) and ViT-B linear shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dinov2_od_amd import _native as nat
from tools.bench_ops import timeit

L = nat.lib(); dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
def shapes(M, D, F, swiglu):
    return [("qkv", M, 3 * D, D), ("mlp_in", M, 2 * F if swiglu else F, D), ("mlp_out", M, D, F)]
for name, M, D, F, sw in (("vitg b32", 32 * 1370, 1536, 4096, True), ("vitb b64", 64 * 1370, 768, 3072, False)):
    for nm, m, n, k in shapes(M, D, F, sw):
        A = (torch.randn(m, k, generator=g) * 0.5).to(dev); W = (torch.randn(n, k, generator=g) * 0.05).to(dev)
        bias = torch.randn(n, generator=g).to(dev)
        Ab, Wb = A.to(torch.bfloat16), W.to(torch.bfloat16)
        qa = torch.empty(m, k, dtype=torch.uint8, device=dev); sa = torch.empty(m, device=dev)
        qw = torch.empty(n, k, dtype=torch.uint8, device=dev); sw_ = torch.empty(n, device=dev)
        nat.check(L.dod_op_quant_rows_fp8(nat.ptr(A), 0, k, m, k, nat.ptr(qa), k, nat.ptr(sa), nat.stream_ptr()))
        nat.check(L.dod_op_quant_rows_fp8(nat.ptr(W), 0, k, n, k, nat.ptr(qw), k, nat.ptr(sw_), nat.stream_ptr()))
        out = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
        t8 = timeit(lambda: L.dod_op_linear_fp8(nat.ptr(qa), k, nat.ptr(sa), nat.ptr(qw), k, nat.ptr(sw_), m, n, k, nat.ptr(bias), None, None, 0, nat.ptr(out), 1, n, 0, nat.stream_ptr()))
        tb = timeit(lambda: L.dod_op_linear(1, nat.ptr(Ab), k, nat.ptr(Wb), k, m, n, k, nat.ptr(bias), None, None, 0, nat.ptr(out), 1, n, 0, nat.stream_ptr()))
        tq = timeit(lambda: L.dod_op_quant_rows_fp8(nat.ptr(Ab), 1, k, m, k, nat.ptr(qa), k, nat.ptr(sa), nat.stream_ptr()))
        tmx = tqm = float("nan")
        if k % 256 == 0:       # block-scaled activations (one e8m0 byte per 32 k): the same GEMM and its operator-level quantiser
            bs = torch.empty(m, k // 32, dtype=torch.uint8, device=dev)
            nat.check(L.dod_op_quant_mx_fp8(nat.ptr(A), 0, k, m, k, nat.ptr(qa), k, nat.ptr(bs), nat.stream_ptr()))
            qm = torch.empty_like(qa)
            nat.check(L.dod_op_quant_mx_fp8(nat.ptr(A), 0, k, m, k, nat.ptr(qm), k, nat.ptr(bs), nat.stream_ptr()))
            nat.check(L.dod_op_quant_rows_fp8(nat.ptr(A), 0, k, m, k, nat.ptr(qa), k, nat.ptr(sa), nat.stream_ptr()))
            f_mx = lambda: L.dod_op_linear_fp8_mx(nat.ptr(qm), k, nat.ptr(bs), nat.ptr(qw), k, nat.ptr(sw_), m, n, k, nat.ptr(bias), None, None, 0, nat.ptr(out), 1, n, 0, nat.stream_ptr())
            f_row = lambda: L.dod_op_linear_fp8(nat.ptr(qa), k, nat.ptr(sa), nat.ptr(qw), k, nat.ptr(sw_), m, n, k, nat.ptr(bias), None, None, 0, nat.ptr(out), 1, n, 0, nat.stream_ptr())
            import statistics
            a_, b_ = [], []
            for _ in range(5):                     # interleaved: the clock drifts over a run
                a_.append(timeit(f_row)); b_.append(timeit(f_mx))
            t8, tmx = statistics.median(a_), statistics.median(b_)
            tqm = timeit(lambda: L.dod_op_quant_mx_fp8(nat.ptr(Ab), 1, k, m, k, nat.ptr(qm), k, nat.ptr(bs), nat.stream_ptr()))
        fl = 2.0 * m * n * k
        print(f"   block-scaled A: fp8 {tmx*1e6:7.1f} us {fl/tmx/1e12:7.1f} TF, block quantiser {tqm*1e6:.1f} us")
        print(f"{name} {nm:7s} M={m} N={n} K={k}: fp8 {t8*1e6:7.1f} us {fl/t8/1e12:7.1f} TF | bf16 {tb*1e6:7.1f} us {fl/tb/1e12:7.1f} TF | x{tb/t8:.2f} | row-quant of A (bf16 in) {tq*1e6:.1f} us")
