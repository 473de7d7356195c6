#!/usr/bin/env python3
"""fp16x2 (H2) GEMM against the bf16 split-product GEMM on the ViT-B block shapes, interleaved rounds in one process (GPU box only).
TF figures are ALGORITHMIC (2MNK / time)."""
import os, statistics, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools._tuning_lib  # noqa: F401,E402  (the -DDINODET_TUNING build: this tool uses tuning hooks)
import torch, numpy as np
from dinov2_od_amd import _native as nat
L = nat.lib(); L.dod_reserve_gemm_scratch(64 << 20); dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
B = int(os.environ.get("X3_B", "64")); D = 768; M = int(os.environ.get("X3_ROWS", B * 1370))
def pair(x):
    out = torch.empty(x.shape[0], 2 * x.shape[1], dtype=torch.bfloat16, device=dev)
    nat.check(L.dod_op_split_pair(nat.ptr(x), x.stride(0), x.shape[0], x.shape[1], nat.ptr(out), nat.stream_ptr())); return out
def h2(x, weight=False):
    out = torch.empty(x.shape[0], (3 if weight else 4) * x.shape[1], dtype=torch.uint8, device=dev)
    we = torch.empty(x.shape[0], dtype=torch.uint8, device=dev) if weight else None
    nat.check(L.dod_op_split_h2(nat.ptr(x), x.stride(0), x.shape[0], x.shape[1], nat.ptr(out), nat.ptr(we), nat.stream_ptr())); return out, we
variants = os.environ.get("H2_VARIANTS", "x3,h2,h2d").split(",")
for name, n, k, act, resid in (("qkv", 3 * D, D, 0, False), ("proj", D, D, 0, True), ("fc1", 4 * D, D, 2, False), ("fc2", D, 4 * D, 0, True)):
    a = (torch.randn(M, k, generator=g) * 0.5).to(dev); w = (torch.randn(n, k, generator=g) * 0.05).to(dev)
    A2, W2 = pair(a), pair(w); Ah, _ = h2(a); Wh, we = h2(w, True)
    bias = torch.randn(n, generator=g).to(dev); x0 = torch.randn(M, n, generator=g).to(dev) if resid else None
    x = x0.clone() if resid else None
    lay_x3, lay_h2 = (0, 0) if resid else (2, 3 if name == "fc1" else 2)        # QKV feeds the split attention (pair layout) in both modes
    out = x if resid else torch.empty(M, 2 * n, dtype=torch.bfloat16, device=dev)
    def run(v):
        if v == "x3":
            nat.check(L.dod_op_linear_x3(nat.ptr(A2), nat.ptr(W2), M, n, k, nat.ptr(bias), None, nat.ptr(x), n if resid else 0, nat.ptr(out), lay_x3, n if resid else 2 * n, act, nat.stream_ptr()))
        else:
            if v == "h2d": os.environ["DINODET_H2_TILE"] = "d"
            else: os.environ.pop("DINODET_H2_TILE", None)
            nat.check(L.dod_op_linear_h2(nat.ptr(Ah), nat.ptr(Wh), nat.ptr(we), M, n, k, nat.ptr(bias), None, nat.ptr(x), n if resid else 0, nat.ptr(out), lay_h2, n if resid else 2 * n, act, nat.stream_ptr()))
    times = {v: [] for v in variants}
    for _ in range(4):
        for v in variants:
            run(v); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8): run(v)
            e1.record(); torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1) / 8 * 1e-3)
    fl = 2.0 * M * n * k
    print(f"{name:5s} M={M} N={n} K={k}: " + "  ".join(f"[{v}] {statistics.median(times[v])*1e6:7.1f} us {fl/statistics.median(times[v])/1e12:6.1f} TF" for v in variants), flush=True)
if os.environ.get("H2_TIMELINE", "1") == "1":      # cycle anatomy of the H2 kernel (wave 0 stamps)
    for name, n, k, act, resid in (("qkv", 3 * D, D, 0, False), ("fc2", D, 4 * D, 0, True)):
        a = (torch.randn(M, k, generator=g) * 0.5).to(dev); w = (torch.randn(n, k, generator=g) * 0.05).to(dev)
        Ah, _ = h2(a); Wh, we = h2(w, True); bias = torch.randn(n, generator=g).to(dev)
        x = torch.randn(M, n, generator=g).to(dev) if resid else None
        out = x if resid else torch.empty(M, 2 * n, dtype=torch.bfloat16, device=dev)
        os.environ.pop("DINODET_H2_TILE", None)
        run = lambda: nat.check(L.dod_op_linear_h2(nat.ptr(Ah), nat.ptr(Wh), nat.ptr(we), M, n, k, nat.ptr(bias), None, nat.ptr(x), n if resid else 0, nat.ptr(out), 0 if resid else 2, n if resid else 2 * n, act, nat.stream_ptr()))
        tiles = ((M + 255) // 256) * ((n + 255) // 256)
        buf = torch.zeros(tiles * 8, dtype=torch.int64, device=dev)
        for _ in range(20): run()
        torch.cuda.synchronize()
        L.dod_debug_pp_stamps(C.c_void_p(buf.data_ptr())); run(); torch.cuda.synchronize(); L.dod_debug_pp_stamps(C.c_void_p(0))
        t = buf.cpu().numpy().reshape(tiles, 8).astype(np.float64); t = t[t[:, 3] > 0]
        pro, loop, epi, tot = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 3] - t[:, 0]
        clk = tot / ((t[:, 5] - t[:, 4]) * 10.0); kt = k // 32; mf = kt * 8 * 256
        med = lambda v: float(np.median(v))
        print(f"h2 {name}: clock {med(clk):.2f} GHz; cycles: prologue {med(pro):.0f}, K loop {med(loop):.0f} (MFMA-ideal {mf}, {mf/med(loop)*100:.0f} %; {med(loop)/kt:.0f} per K-tile), epilogue {med(epi):.0f}, total {med(tot):.0f}", flush=True)
