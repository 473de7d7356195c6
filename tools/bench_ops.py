#!/usr/bin/env python3
"""Micro-benchmark of the hot kernels through the C ABI operator entry points (GPU box only).
    python tools/bench_ops.py [--batch 8] [--res 518]
Times each ViT-B GEMM shape of one encoder block and the attention kernel with HIP events."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dinov2_od_amd import _native as nat


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--res", type=int, default=518)
    ap.add_argument("--hidden", type=int, default=768)
    ap.add_argument("--layerscale", action="store_true", help="LayerScale in the residual epilogues (fp32-mode form; the bf16 path folds it into W)")
    ap.add_argument("--rows", type=int, default=0, help="override M of the GEMMs (tail / wave-quantisation studies)")
    a = ap.parse_args()
    L = nat.lib()
    dev = torch.device("cuda:0")
    N = (a.res // 14) ** 2 + 1
    M, D = a.rows or a.batch * N, a.hidden
    heads = D // 64
    g = torch.Generator(device="cpu").manual_seed(0)
    rnd = lambda *s: (torch.randn(*s, generator=g) * 0.5).to(dev)
    shapes = [("qkv", M, 3 * D, D, "none", True), ("proj", M, D, D, "resid", False), ("fc1", M, 4 * D, D, "gelu", True),
              ("fc2", M, D, 4 * D, "resid", False)]
    for name, m, n, k, epi, obf in shapes:
        A = rnd(m, k).to(torch.bfloat16)
        W = (rnd(n, k) * 0.1).to(torch.bfloat16)
        bias = rnd(n)
        scale = rnd(n)
        x = rnd(m, n)
        out = torch.empty(m, n, device=dev, dtype=torch.bfloat16 if obf else torch.float32)

        def run():
            if epi == "resid":
                rc = L.dod_op_linear(1, nat.ptr(A), k, nat.ptr(W), k, m, n, k, nat.ptr(bias), nat.ptr(scale) if a.layerscale else None, nat.ptr(x), n, nat.ptr(x), 0, n, 0, nat.stream_ptr())
            else:
                rc = L.dod_op_linear(1, nat.ptr(A), k, nat.ptr(W), k, m, n, k, nat.ptr(bias), None, None, 0, nat.ptr(out), 1, n, 2 if epi == "gelu" else 0, nat.stream_ptr())
            assert rc == 0
        t = timeit(run)
        print(f"gemm_bf16 {name:5s} M={m} N={n} K={k}: {t*1e6:8.1f} us  {2.0*m*n*k/t/1e12:7.1f} TFLOP/s")
    if a.rows:
        return
    qkv = rnd(a.batch, N, 3 * D).to(torch.bfloat16)
    ctx = torch.empty(a.batch, N, D, device=dev, dtype=torch.bfloat16)
    t = timeit(lambda: L.dod_op_attention_bf16(nat.ptr(qkv), nat.ptr(ctx), a.batch, N, heads, 0.125, nat.stream_ptr()))
    print(f"attn_bf16 B={a.batch} N={N} heads={heads}: {t*1e6:8.1f} us  {4.0*a.batch*N*N*D/t/1e12:7.1f} TFLOP/s")
    x = rnd(M, D)
    gam, bet = rnd(D), rnd(D)
    y = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
    t = timeit(lambda: L.dod_op_layernorm(nat.ptr(x), None, nat.ptr(gam), nat.ptr(bet), 1e-6, M, D, nat.ptr(y), 1, nat.stream_ptr()))
    print(f"layernorm rows={M} D={D}: {t*1e6:8.1f} us  {(M*D*6)/t/1e9:7.1f} GB/s")


if __name__ == "__main__":
    main()
