#!/usr/bin/env python3
"""Headline benchmark: images/sec of DINOv2ObjectDetector.forward on MI355X.

    python bench.py --gpus 1 --steps K --warmup W            (single GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W   (one rank per GPU, RCCL)

A "step" is one forward of the hot path over one batch of synthetic images already resident in HBM:
pixel_values [B,3,518,518] fp32 -> packed detections [B,100,95] fp32 (multi-GPU: after the single
RCCL all-gather of the packed detections).  Workload at every N: the configuration BASELINE.json's
metric is quoted on -- configs[2]: ViT-B/14 518x518, 100 queries, batch 64 -- which fits ONE MI355X, so
each GPU runs a full batch of 64 (weak scaling: global batch 64 N; `--batch 8` gives configs[2]'s
8-per-GPU sharding instead).
Prints ONE JSON line on rank 0 (contract in the task statement), with
  roofline     : the dominant kernel (bf16 MFMA GEMM) -- algorithmic FLOPs / its summed launch time,
                 measured with HIP events on the launch stream by the library's profile mode over
                 extra forwards of the same step, right after the timed region;
  cpu_baseline : the CPU oracle (kind "port": oracle/dinodet_oracle.py, the parity-checked CPU
                 restatement of the reference) timed on this box's host cores on a bounded sample; its first
                 batch also checks the GPU detections of the same two images (`gpu_vs_oracle`);
  parity_gated_mode / fp32_mode : the same workload in the two modes that meet the 1e-3 gate (bf16x3: split
                 products on the bf16 MFMA cores; fp32), each with its own `gpu_vs_oracle` (N = 1 only).
Other workloads / modes: --workload {vits224,vitb224,vitl518,vitg518}, --precision {bf16,bf16x3,fp32,fp8}.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch

WORKLOADS = {
    # name: (model name, R, queries, per-GPU batch, description)
    "vitb518": ("facebook/dinov2-base", 518, 100, 64, "ViT-B/14 518x518 bf16, 100 queries, batch 64 per GPU (BASELINE configs[2]'s batch of 64; it fits one GPU, so every GPU runs the whole configuration: weak scaling)"),
    "vitb224": ("facebook/dinov2-base", 224, 100, 32, "ViT-B/14 224x224 bf16, 100 queries, batch 32 per GPU (BASELINE configs[1])"),
    "vitl518": ("facebook/dinov2-large", 518, 300, 16, "ViT-L/14 518x518 bf16, 300 queries, batch 16 per GPU (BASELINE configs[3])"),
    "vitg518": ("facebook/dinov2-giant", 518, 300, 32, "ViT-g/14 518x518, 300 queries, batch 32 per GPU (BASELINE configs[4]; --precision fp8 for its fp8 MFMA form)"),
    "vits224": ("facebook/dinov2-small", 224, 100, 2, "--lightweight ViT-S/14 224x224, batch 2 (BASELINE configs[0])"),
}
PEAK_BF16 = 2.5e15      # dense bf16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_F32 = 157.3e12
PEAK_FP8 = 5.0e15       # dense fp8 MFMA (v_mfma_f32_32x32x64_f8f6f4), same guide


def build(name, queries, precision, device):
    from dinov2_od_amd.models import DINOv2ObjectDetector
    from dinov2_od_amd.config import BackboneConfig, DecoderConfig
    from dinov2_od_amd import synth
    kw = dict(num_queries=queries)
    if "small" in name:   # train.py:607-640 lightweight preset
        kw.update(hidden_dim=256, num_decoder_layers=2, dim_feedforward=512, lora_r=1, nheads=4)
    m = DINOv2ObjectDetector(dino_model_name=name, pretrained=False, precision=precision, **kw)
    bb, dc = m._bb_cfg, m._dc_cfg
    sd = synth.detector_state_dict(bb, dc, seed=1)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    del sd
    return m.to(device).eval(), bb, dc


def host_cores():
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota (the GPU box
    exposes all 256 hardware threads but gives a 1-GPU job a 16-core share)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(p) + 0.5)))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("DINODET_CPU_THREADS", "16"))))


def cpu_baseline(bb, dc, R, seconds_budget=20.0, gpu_det=None):
    """oracle on the host cores: bounded sample (>= 1 batch of 2 images, up to the time budget).  The first batch's
    outputs also check the GPU detections of the same two images (gpu_det: packed [>=2, Q, C+4])."""
    from oracle import dinodet_oracle as orc
    from dinov2_od_amd import synth
    cores = host_cores()
    torch.set_num_threads(cores)
    sd = {k: torch.from_numpy(v) for k, v in synth.detector_state_dict(bb, dc, seed=1).items()}
    x = torch.from_numpy(synth.make_pixels(2, R, R, seed=0))
    orc.detector_forward(sd, bb, dc, x[:1])          # warm-up (thread pools, first-touch)
    n, t0 = 0, time.perf_counter()
    first = None
    while True:
        o = orc.detector_forward(sd, bb, dc, x)
        if first is None:
            first = o
        n += 2
        if time.perf_counter() - t0 > seconds_budget or n >= 32:
            break
    dt = time.perf_counter() - t0
    res = {"value": n / dt, "unit": "images/s", "cores": cores, "kind": "port",
           "sample": f"{n} images (batches of 2) of the same workload, fp32, torch {torch.__version__} CPU, {dt:.1f} s"}
    if gpu_det is not None:
        res["gpu_vs_oracle"] = oracle_error(gpu_det, first, dc.num_classes)
    return res, first


def oracle_error(gpu_det, oracle_out, C):
    """max |gpu - oracle| / max |oracle| on the images both computed (the metric of tests/cases.py::rel_err)."""
    g = gpu_det.float().cpu()
    n = min(g.shape[0], oracle_out["pred_logits"].shape[0])
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
    return {"images": n, "pred_logits_max_rel": rel(g[:n, :, :C], oracle_out["pred_logits"][:n]),
            "pred_boxes_max_rel": rel(g[:n, :, C:], oracle_out["pred_boxes"][:n])}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="vitb518", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch override")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32", "fp8", "bf16x3"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of a captured hipGraph")
    a = ap.parse_args()

    from dinov2_od_amd import dist as ddist
    from dinov2_od_amd.config import flops_per_image
    # RCCL ("nccl") is the backend of record; DINODET_DIST_BACKEND=gloo only exists to rehearse the N > 1 control flow on a
    # one-GPU box (all ranks on cuda:0)
    rank, world, local = ddist.init_from_env(os.environ.get("DINODET_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else None))
    if os.environ.get("DINODET_DIST_BACKEND") == "gloo":
        local = 0
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the hot path has no CPU fallback)")
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)

    name, R, Q, B_local, desc = WORKLOADS[a.workload]
    if a.batch:
        B_local = a.batch
        desc += f" [per-GPU batch overridden to {B_local}]"
    from dinov2_od_amd import synth
    model, bb, dc = build(name, Q, a.precision, device)
    lo, hi = ddist.shard_bounds(B_local * world, rank, world)
    x = torch.empty(B_local, 3, R, R, device=device)
    for i, g in enumerate(range(lo, hi)):          # image g of the global batch, resident in HBM
        x[i] = torch.from_numpy(synth.uniform01(0, f"pixel_values.{R}x{R}.{g % 16}", (3, R, R))).to(device)
    gathered = torch.empty(B_local * world, Q, dc.num_classes + 4, device=device) if world > 1 else None

    def step():
        det = model.forward_packed(x)
        return ddist.gather_detections_equal(det, gathered) if world > 1 else det

    with torch.no_grad():
        out = step()                                # packs weights, sizes workspace, sets func attributes
        torch.cuda.synchronize()
        use_graph = not a.no_graph
        graph = None
        if use_graph:
            # the forward (all kernels of the hot path) is captured in a hipGraph; at N > 1 the RCCL all-gather
            # stays an eager call on the same stream right after the replay
            try:
                s = torch.cuda.Stream()
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    model.forward_packed(x)
                torch.cuda.current_stream().wait_stream(s)
                graph = torch.cuda.CUDAGraph()
                # thread_local: an RCCL watchdog thread polling events must not invalidate the capture
                with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                    static_det = model.forward_packed(x)
                if world > 1:
                    def run():
                        graph.replay()
                        return ddist.gather_detections_equal(static_det, gathered)
                else:
                    run = graph.replay
            except Exception as e:                  # capture unsupported -> eager, say so
                print(f"[bench] hipGraph capture failed ({type(e).__name__}: {e}); running eager", file=sys.stderr)
                graph, use_graph = None, False
                run = step
        else:
            run = step
        for _ in range(a.warmup):
            run()
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]     # per-step spread (median, p10/p90)
        t0 = time.perf_counter()
        evs[0].record()
        for i in range(a.steps):
            run()
            evs[i + 1].record()
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        step_ms = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(a.steps))
        pct = lambda q: step_ms[min(len(step_ms) - 1, int(q * len(step_ms)))]
        if world > 1:
            t = torch.tensor([dt], device=device, dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dt = float(t.item())

        # ---- roofline leg: same step, library event timing per kernel class (eager, outside the timed region)
        eng = model._get_engine()
        nprof = max(3, min(10, a.steps))
        eng.profile(True)
        for _ in range(nprof):
            model.forward_packed(x)
        torch.cuda.synchronize()
        prof = eng.profile_read()
        eng.profile(False)

    global_batch = B_local * world
    ips = global_batch * a.steps / dt
    fpi = flops_per_image(bb, dc, R, R)
    dom = {"bf16": "gemm_bf16", "fp32": "gemm_f32", "fp8": "gemm_fp8", "bf16x3": "gemm_bf16"}[a.precision]
    peak = {"bf16": PEAK_BF16, "fp32": PEAK_F32, "fp8": PEAK_FP8, "bf16x3": PEAK_BF16}[a.precision]
    d = prof[dom]
    ach = d["flops"] / (d["ms"] * 1e-3) if d["ms"] > 0 else 0.0
    traffic, traffic_src = None, None
    tj = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if a.workload == "vitb518" and a.precision == "bf16" and B_local == 64 and os.path.exists(tj):
        # HBM bytes per launch from the committed rocprofv3 --pmc passes of this same command (not measurable live)
        tdat = json.load(open(tj))
        traffic, traffic_src = tdat["gemm_bf16_avg_bytes_per_launch"], "profiles/r01_pmc_traffic.json"
    roof = {"bound": "mfma", "kernel": {"bf16": "gemm_bf16_256x128/256x256_kernel (all bf16 MFMA GEMM launches of the step)", "fp32": "gemm_f32_kernel",
                                        "fp8": "gemm_fp8_256x128_kernel (the e4m3 MFMA GEMM launches: QKV, MLP-in, SwiGLU MLP-out)",
                                        "bf16x3": "bf16 MFMA GEMM launches on split operands (K' = 3K; achieved = ALGORITHMIC 2MNK FLOPs / time)"}[a.precision],
            "achieved": ach / 1e12, "peak": peak / 1e12, "unit": "TFLOP/s",
            "frac": ach / peak, "traffic": traffic, "traffic_source": traffic_src,
            "launches_per_step": d["launches"] // nprof, "avg_launch_us": 1e3 * d["ms"] / max(1, d["launches"]),
            "flops_per_launch_avg": d["flops"] / max(1, d["launches"]),
            "other_kernels": {k: {"ms_per_step": v["ms"] / nprof, "tflops": (v["flops"] / (v["ms"] * 1e-3) / 1e12) if v["ms"] > 0 and v["flops"] > 0 else None}
                              for k, v in prof.items() if v["launches"]}}
    res = {"metric": "images/sec forward, DINOv2 ViT-B/14 518x518 + 100-query head" if a.workload == "vitb518" else f"images/sec forward, {a.workload}",
           "value": ips, "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
           "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": a.precision, "data": "synthetic",
           "config": {"workload": desc, "global_batch": global_batch, "per_gpu_batch": B_local, "image": R, "queries": Q,
                      "parallelism": f"dp{world}", "hipgraph": bool(use_graph),
                      "gflop_per_image": fpi / 1e9},
           "step_ms_p10_p50_p90": [pct(0.10), pct(0.50), pct(0.90)],
           "mfma_roofline_frac_end_to_end": ips * fpi / (peak * world),
           "roofline": roof}
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        with torch.no_grad():
            det_now = model.forward_packed(x[:2]).clone()
            torch.cuda.synchronize()
        res["cpu_baseline"], oracle_first = cpu_baseline(bb, dc, R, gpu_det=det_now)
        res["precision_note"] = ("single-pass bf16 MFMA operands (the configuration BASELINE names) cannot meet the 1e-3 gate on logits "
                                 "(8-bit mantissa; DESIGN.md section 2): see gpu_vs_oracle.  The parity-gated modes -- bf16x3 "
                                 "(split products on the same bf16 MFMA kernels) and fp32 -- do, and are measured below on the same workload")
        if a.precision == "bf16":
            def gated(prec, nb, steps):
                m2, _, _ = build(name, Q, prec, device)
                xs = x[:nb].contiguous()
                with torch.no_grad():
                    d2 = m2.forward_packed(xs).clone()
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    for _ in range(steps):
                        m2.forward_packed(xs)
                    torch.cuda.synchronize()
                    t2 = (time.perf_counter() - t1) / steps
                out = {"precision": prec, "value": nb / t2, "unit": "images/s", "batch": nb, "ms_per_step": 1e3 * t2,
                       "gpu_vs_oracle": oracle_error(d2, oracle_first, dc.num_classes),
                       "mfma_roofline_frac_end_to_end_algorithmic": nb / t2 * fpi / PEAK_BF16}
                del m2
                torch.cuda.empty_cache()
                return out
            del model
            torch.cuda.empty_cache()
            for key, prec, nb, steps in (("parity_gated_mode", "bf16x3", B_local, 5), ("fp32_mode", "fp32", min(8, B_local), 2)):
                try:
                    res[key] = gated(prec, nb, steps)
                except Exception as e:
                    res[key] = {"error": f"{type(e).__name__}: {e}"}
            if "value" in res.get("parity_gated_mode", {}):
                res["parity_gated_mode"]["note"] = ("every backbone product as bf16 split products (Ah Wh + Ah Wl + Al Wh) on the bf16 MFMA "
                                                    "kernels incl. the flash attention, fp32 elsewhere: within 1e-3 of the reference "
                                                    "(tests/test_gpu_forward.py GATED); 3x the MFMA work of the single-pass bf16 mode")
    if rank == 0:
        print(json.dumps(res))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
