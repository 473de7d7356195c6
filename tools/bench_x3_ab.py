#!/usr/bin/env python3
"""A/B of the split-product GEMM kernels (DINODET_X3_TILE: default = 16-wave gemm_x3_256x256, p = 8-wave ping-pong) on the ViT-B
block shapes, interleaved rounds in one process; checks the variants agree (GPU box only)."""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools._tuning_lib  # noqa: F401,E402  (the -DDINODET_TUNING build: this tool uses tuning hooks)
import torch
from dinov2_od_amd import _native as nat
L = nat.lib(); L.dod_reserve_gemm_scratch(64 << 20); dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
B = int(os.environ.get("X3_B", "64")); N = 1370; D = 768; M = int(os.environ.get("X3_ROWS", B * N))
variants = os.environ.get("X3_VARIANTS", "w;p").replace(",", ";").split(";") if "@" not in os.environ.get("X3_VARIANTS", "") else os.environ["X3_VARIANTS"].split(";")
def setv(v):
    os.environ.pop("DINODET_GEMM_STAGGER", None)
    if "@" in v:
        v, st = v.split("@"); os.environ["DINODET_GEMM_STAGGER"] = st
    if v == "default": os.environ.pop("DINODET_X3_TILE", None)      # the shape heuristic
    else: os.environ["DINODET_X3_TILE"] = v
def pair(x):
    out = torch.empty(x.shape[0], 2 * x.shape[1], dtype=torch.bfloat16, device=dev)
    nat.check(L.dod_op_split_pair(nat.ptr(x), x.stride(0), x.shape[0], x.shape[1], nat.ptr(out), nat.stream_ptr())); return out
for name, n, k, layout, act, resid in (("qkv", 3 * D, D, 2, 0, False), ("proj", D, D, 0, 0, True), ("fc1", 4 * D, D, 2, 2, False), ("fc2", D, 4 * D, 0, 0, True)):
    A2 = pair((torch.randn(M, k, generator=g) * 0.5).to(dev)); W2 = pair((torch.randn(n, k, generator=g) * 0.05).to(dev))
    if os.environ.get("X3_ZERO"): A2.zero_(); W2.zero_()
    bias = torch.randn(n, generator=g).to(dev); x0 = torch.randn(M, n, generator=g).to(dev) if resid else None
    x = x0.clone() if resid else None
    out = x if resid else torch.empty(M, 2 * n if layout == 2 else n, dtype=torch.bfloat16 if layout else torch.float32, device=dev)
    run = lambda: nat.check(L.dod_op_linear_x3(nat.ptr(A2), nat.ptr(W2), M, n, k, nat.ptr(bias), None, nat.ptr(x), n if resid else 0, nat.ptr(out), layout, out.shape[1], act, nat.stream_ptr()))
    ref, devs = None, {}
    for v in variants:
        setv(v)
        if resid: x.copy_(x0)
        run(); torch.cuda.synchronize()
        got = out.float().clone()
        ref = got if ref is None else ref
        devs[v] = float((got - ref).abs().max() / ref.abs().max())
    times = {v: [] for v in variants}
    for _ in range(4):
        for v in variants:
            setv(v); run(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8): run()
            e1.record(); torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1) / 8 * 1e-3)
    fl = 2.0 * M * n * k
    print(f"gemm_x3 {name:5s} M={M} N={n} K={k}: " + "  ".join(f"[{v}] {statistics.median(times[v])*1e6:7.1f} us {fl/statistics.median(times[v])/1e12:6.1f} TF alg ({3*fl/statistics.median(times[v])/1e12:6.1f} exec) dev {devs[v]:.1e}" for v in variants), flush=True)
setv("default")
