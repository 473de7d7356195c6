#!/usr/bin/env python3
"""Experiment: the bench batch (64 images) as TWO half-batches on two HIP streams (two engines = two workspaces), captured in one
hipGraph, against the single 64-image forward -- does the other half's work fill one half's LayerNorm / tail / epilogue bubbles?
    DINODET_GEMM_TAILSPLIT=0 python tools/bench_two_streams.py [precision]      (the tail-split scratch is one buffer per device)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools._tuning_lib  # noqa: F401,E402  (the -DDINODET_TUNING build: this tool uses tuning hooks)
os.environ.setdefault("DINODET_GEMM_TAILSPLIT", "0")
import torch
from bench import build, make_images
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
dev = torch.device("cuda:0")
B, R, Q = 64, 518, 100
x = make_images(B, R, 0, dev)
m0, bb, dc = build("facebook/dinov2-base", Q, prec, dev)
m1, _, _ = build("facebook/dinov2-base", Q, prec, dev)

def timed(fn, steps=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(steps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / steps

with torch.no_grad():
    ref = m0.forward_packed(x).clone()
    # --- single stream, one 64-image forward, captured
    g1 = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        m0.forward_packed(x); torch.cuda.synchronize()
        with torch.cuda.graph(g1, stream=s):
            o1 = m0.forward_packed(x)
    t1 = timed(g1.replay)
    # --- two streams, two 32-image forwards, captured in one graph (fork / join)
    xa, xb = x[:32].contiguous(), x[32:].contiguous()
    m0.forward_packed(xa); m1.forward_packed(xb); torch.cuda.synchronize()
    g2 = torch.cuda.CUDAGraph(); sa = torch.cuda.Stream(); sb = torch.cuda.Stream()
    with torch.cuda.stream(sa):
        with torch.cuda.graph(g2, stream=sa):
            sb.wait_stream(sa)
            oa = m0.forward_packed(xa)
            with torch.cuda.stream(sb):
                ob = m1.forward_packed(xb)
            sa.wait_stream(sb)
    t2 = timed(g2.replay)
    g2.replay(); torch.cuda.synchronize()
    err = max(float((oa - ref[:32]).abs().max()), float((ob - ref[32:]).abs().max()))
print(f"{prec}: one 64-image forward {t1*1e3:.2f} ms = {B/t1:.0f} images/s | two concurrent 32-image forwards {t2*1e3:.2f} ms = {B/t2:.0f} images/s | max abs diff vs the single forward {err:.2e}")
