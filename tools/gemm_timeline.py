#!/usr/bin/env python3
"""Per-workgroup timeline of the 256x128 GEMM (s_memrealtime stamps, 100 MHz): prologue+loop vs epilogue durations."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools._tuning_lib  # noqa: F401,E402  (the -DDINODET_TUNING build: this tool uses tuning hooks)
import torch, numpy as np
from dinov2_od_amd import _native as nat
L = nat.lib(); dev = torch.device("cuda:0")

M = 87680
g = torch.Generator().manual_seed(0)
for N, K, obf in ((2304, 768, True), (768, 768, False), (768, 3072, False)):
    A = (torch.randn(M, K, generator=g) * 0.5).to(dev).to(torch.bfloat16); W = (torch.randn(N, K, generator=g) * 0.05).to(dev).to(torch.bfloat16)
    bias = torch.randn(N, generator=g).to(dev); scale = torch.randn(N, generator=g).to(dev); x = torch.randn(M, N, generator=g).to(dev)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    tiles = ((M + 255) // 256) * (N // 128)
    buf = torch.zeros(tiles * 4, dtype=torch.int64, device=dev)
    def run():
        if obf: L.dod_op_linear(1, nat.ptr(A), K, nat.ptr(W), K, M, N, K, nat.ptr(bias), None, None, 0, nat.ptr(out), 1, N, 0, nat.stream_ptr())
        else: L.dod_op_linear(1, nat.ptr(A), K, nat.ptr(W), K, M, N, K, nat.ptr(bias), nat.ptr(scale) if os.environ.get('TL_LAYERSCALE') else None, nat.ptr(x), N, nat.ptr(x), 0, N, 0, nat.stream_ptr())
    for _ in range(3): run()
    torch.cuda.synchronize()
    L.dod_debug_gemm_stamps(C.c_void_p(buf.data_ptr()))
    run(); torch.cuda.synchronize()
    L.dod_debug_gemm_stamps(C.c_void_p(0))
    t = buf.cpu().numpy().reshape(tiles, 4).astype(np.float64)
    t0 = t[:, 0].min()
    start, loop, end = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0, (t[:, 2] - t0) / 100.0   # us
    print(f"N={N} K={K}: tiles={tiles} kernel span {end.max():.1f} us; per-WG: loop {np.median(loop-start):.2f} us (p10 {np.percentile(loop-start,10):.2f} p90 {np.percentile(loop-start,90):.2f}), "
          f"epilogue {np.median(end-loop):.2f} us (p10 {np.percentile(end-loop,10):.2f} p90 {np.percentile(end-loop,90):.2f})")
    # concurrency over time: how many WGs are in epilogue / in loop at sampled instants
    ts = np.linspace(0, end.max(), 41)[1:-1]
    inl = [(np.sum((start <= x_) & (loop > x_)), np.sum((loop <= x_) & (end > x_))) for x_ in ts]
    print("   in-loop/in-epilogue WGs at 39 instants:", " ".join(f"{a}/{b}" for a, b in inl))
    first = np.argsort(start)[:512]
    print(f"   first-wave start spread {start[first].max()-start[first].min():.2f} us; WG start order vs blockIdx monotone: {np.all(np.diff(start[np.argsort(t[:,3])][:512])>=-0.5)}")
