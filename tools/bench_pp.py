#!/usr/bin/env python3
"""A/B of bf16 GEMM kernel variants on the four ViT-B block shapes, interleaved rounds in ONE process (GPU box only).
    python tools/bench_pp.py [--batch 64] [--variants default,p] [--rounds 5]
A variant is a value of DINODET_GEMM_TILE ("default" = the shape heuristic).  Prints the median time per variant and the
max deviation of each variant's output from the first one's."""
import argparse
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools._tuning_lib  # noqa: F401,E402  (the -DDINODET_TUNING build: this tool uses tuning hooks)
import torch
from dinov2_od_amd import _native as nat


def set_variant(v):
    os.environ.pop("DINODET_GEMM_STAGGER", None)
    os.environ.pop("DINODET_EPI_RB", None)
    if "#" in v:                               # "<tile>#<rb>": residual loads in flight per thread (512-thread kernels)
        v, rb = v.split("#")
        os.environ["DINODET_EPI_RB"] = rb
    if "@" in v:                               # "<tile>@<groups>,<step_us>": start stagger
        v, st = v.split("@")
        os.environ["DINODET_GEMM_STAGGER"] = st
    os.environ.pop("DINODET_GEMM_WRES", None)
    os.environ.pop("DINODET_GEMM_GM", None)
    if v.startswith("w"):                      # "w": weight-resident tile map of the ping-pong kernels
        os.environ["DINODET_GEMM_WRES"] = "1"
        return
    if v.startswith("g"):                      # "g<N>": m-tiles per group of the time-ordered map
        os.environ["DINODET_GEMM_GM"] = v[1:]
        return
    if v.startswith("o"):                      # tile-order A/B: "o0".."o3"
        os.environ["DINODET_GEMM_ORDER"] = v[1:]
        return
    if v == "default":
        os.environ.pop("DINODET_GEMM_TILE", None)
    else:
        os.environ["DINODET_GEMM_TILE"] = v


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--rows", type=int, default=0)
    ap.add_argument("--hidden", type=int, default=768)
    ap.add_argument("--variants", default="default,p")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--producer", action="store_true", help="re-write A (rows 0 -> M, as a producer kernel would) right before every timed GEMM, which is timed alone: the in-forward cache state instead of a warm one")
    ap.add_argument("--orders", default="", help="comma list of DINODET_GEMM_ORDER values to A/B (variants then must be one entry)")
    ap.add_argument("--lda0", action="store_true", help="every A row aliases row 0 (A traffic becomes cache hits): isolates the A-panel miss cost")
    ap.add_argument("--zeros", action="store_true", help="all-zero operands (clock stays high: the DVFS bound)")
    ap.add_argument("--noout", action="store_true", help="ldc = 0: every output row aliases row 0 (output / residual traffic becomes cache hits)")
    a = ap.parse_args()
    variants = a.variants.split(";") if ";" in a.variants else a.variants.split(",")
    if a.orders:
        variants = ["o" + o for o in a.orders.split(",")]
    L = nat.lib(); L.dod_reserve_gemm_scratch(int(os.environ.get("DINODET_GEMM_SCRATCH_MB", "64")) << 20)
    dev = torch.device("cuda:0")
    M, D = a.rows or a.batch * 1370, a.hidden
    g = torch.Generator(device="cpu").manual_seed(0)
    rnd = lambda *s: (torch.randn(*s, generator=g) * 0.5).to(dev)
    shapes = [("qkv", M, 3 * D, D, "none", True), ("proj", M, D, D, "resid", False), ("fc1", M, 4 * D, D, "gelu", True),
              ("fc2", M, D, 4 * D, "resid", False)]
    for name, m, n, k, epi, obf in shapes:
        A = rnd(m, k).to(torch.bfloat16)
        W = (rnd(n, k) * 0.1).to(torch.bfloat16)
        bias, x0 = rnd(n), rnd(m, n)
        if a.zeros:
            A.zero_(); W.zero_()
        la = 0 if a.lda0 else k
        lc = 0 if a.noout else n
        out = torch.empty(m, n, device=dev, dtype=torch.bfloat16 if obf else torch.float32)
        x = x0.clone()

        def run():
            if epi == "resid":
                rc = L.dod_op_linear(1, nat.ptr(A), la, nat.ptr(W), k, m, n, k, nat.ptr(bias), None, nat.ptr(x), lc, nat.ptr(x), 0, lc, 0, nat.stream_ptr())
            else:
                rc = L.dod_op_linear(1, nat.ptr(A), la, nat.ptr(W), k, m, n, k, nat.ptr(bias), None, None, 0, nat.ptr(out), 1, lc, 2 if epi == "gelu" else 0, nat.stream_ptr())
            assert rc == 0, rc
        # correctness: every variant against the first
        ref = None
        devs = {}
        for v in variants:
            set_variant(v)
            x.copy_(x0)
            run()
            torch.cuda.synchronize()
            got = (x if epi == "resid" else out).float().clone()
            if ref is None:
                ref = got
            devs[v] = float((got - ref).abs().max() / ref.abs().max())
        times = {v: [] for v in variants}
        A_src = A.clone() if a.producer else None
        for _ in range(a.rounds):
            for v in variants:
                set_variant(v)
                run()
                torch.cuda.synchronize()
                if a.producer:
                    tot = 0.0
                    for _ in range(a.iters):
                        A.copy_(A_src)                         # the producer: writes A front to back
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        run()
                        e1.record()
                        torch.cuda.synchronize()
                        tot += e0.elapsed_time(e1)
                    times[v].append(tot / a.iters * 1e-3)
                    continue
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.iters):
                    run()
                e1.record()
                torch.cuda.synchronize()
                times[v].append(e0.elapsed_time(e1) / a.iters * 1e-3)
        fl = 2.0 * m * n * k
        print(f"{name:5s} M={m} N={n} K={k}: " + "  ".join(
            f"[{v}] {statistics.median(times[v]) * 1e6:7.1f} us {fl / statistics.median(times[v]) / 1e12:6.1f} TF (best {fl / min(times[v]) / 1e12:6.1f}) dev {devs[v]:.1e}"
            for v in variants), flush=True)
    set_variant("default")


if __name__ == "__main__":
    main()
