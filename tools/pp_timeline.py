#!/usr/bin/env python3
"""Where a ping-pong GEMM workgroup spends its cycles (s_memtime stamps of wave 0: prologue / K loop / epilogue; GPU box only).
    X3=1 python tools/pp_timeline.py        # split-product kernel (DINODET_X3_TILE=p) ; X3=0: plain bf16 (DINODET_GEMM_TILE=q)"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools._tuning_lib  # noqa: F401,E402  (the -DDINODET_TUNING build: this tool uses tuning hooks)
import torch, numpy as np
from dinov2_od_amd import _native as nat
L = nat.lib(); L.dod_reserve_gemm_scratch(64 << 20); dev = torch.device("cuda:0")
X3 = os.environ.get("X3", "1") == "1"
if X3: os.environ["DINODET_X3_TILE"] = os.environ.get("PP_VARIANT", "p")
else: os.environ["DINODET_GEMM_TILE"] = os.environ.get("PP_VARIANT", "q")
M = int(os.environ.get("PP_ROWS", 87680)); D = 768
g = torch.Generator().manual_seed(0)
def pair(x):
    out = torch.empty(x.shape[0], 2 * x.shape[1], dtype=torch.bfloat16, device=dev)
    nat.check(L.dod_op_split_pair(nat.ptr(x), x.stride(0), x.shape[0], x.shape[1], nat.ptr(out), nat.stream_ptr())); return out
for name, n, k, layout, act, resid in (("qkv", 3 * D, D, 2, 0, False), ("proj", D, D, 0, 0, True), ("fc1", 4 * D, D, 2, 2, False), ("fc2", D, 4 * D, 0, 0, True)):
    a = (torch.randn(M, k, generator=g) * 0.5).to(dev); w = (torch.randn(n, k, generator=g) * 0.05).to(dev)
    bias = torch.randn(n, generator=g).to(dev); x = torch.randn(M, n, generator=g).to(dev) if resid else None
    if X3:
        A2, W2 = pair(a), pair(w)
        out = x if resid else torch.empty(M, 2 * n if layout == 2 else n, dtype=torch.bfloat16 if layout else torch.float32, device=dev)
        run = lambda: nat.check(L.dod_op_linear_x3(nat.ptr(A2), nat.ptr(W2), M, n, k, nat.ptr(bias), None, nat.ptr(x), n if resid else 0, nat.ptr(out), layout, out.shape[1], act, nat.stream_ptr()))
    else:
        A, W = a.to(torch.bfloat16), w.to(torch.bfloat16)
        out = x if resid else torch.empty(M, n, dtype=torch.bfloat16, device=dev)
        if resid: run = lambda: nat.check(L.dod_op_linear(1, nat.ptr(A), k, nat.ptr(W), k, M, n, k, nat.ptr(bias), None, nat.ptr(x), n, nat.ptr(x), 0, n, 0, nat.stream_ptr()))
        else: run = lambda: nat.check(L.dod_op_linear(1, nat.ptr(A), k, nat.ptr(W), k, M, n, k, nat.ptr(bias), None, None, 0, nat.ptr(out), 1, n, 2 if act else 0, nat.stream_ptr()))
    tiles = ((M + 255) // 256) * ((n + 255) // 256)
    buf = torch.zeros(tiles * 8, dtype=torch.int64, device=dev)
    for _ in range(20): run()
    torch.cuda.synchronize()
    L.dod_debug_pp_stamps(C.c_void_p(buf.data_ptr()))
    run(); torch.cuda.synchronize()
    L.dod_debug_pp_stamps(C.c_void_p(0))
    t = buf.cpu().numpy().reshape(tiles, 8).astype(np.float64)
    t = t[t[:, 3] > 0]
    pro, loop, epi, tot = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 3] - t[:, 0]
    clk = tot / ((t[:, 5] - t[:, 4]) * 10.0)         # cycles per ns -> GHz
    span_us = (t[:, 5].max() - t[:, 4].min()) / 100.0
    kt = k // (32 if X3 else 64)
    mf = kt * (96 if X3 else 64) * 16 * 2            # MFMA-pipe cycles per SIMD over the K loop (two waves per SIMD)
    med = lambda v: float(np.median(v))
    print(f"{name:4s} N={n} K={k}: {len(t)} WGs, kernel span {span_us:.0f} us, clock {med(clk):.2f} GHz; per WG cycles: prologue {med(pro):.0f}, K loop {med(loop):.0f} "
          f"(MFMA-ideal {mf}, {mf / med(loop) * 100:.0f} %; {med(loop) / kt:.0f} per K-tile), epilogue {med(epi):.0f}, total {med(tot):.0f} = {med(tot) / med(clk) / 1e3:.1f} us", flush=True)
