#!/usr/bin/env python3
"""Headline benchmark: images/sec of DINOv2ObjectDetector.forward on MI355X.

    python bench.py --gpus N --steps K --warmup W

N = 1 runs in this process.  N > 1 without a torchrun-style environment makes THIS process a launcher: it starts N fresh
child processes of this script (one rank per GPU; RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT), relays
rank 0's JSON line and exits with the job's code -- the launch the reference does with mp.spawn (train.py:1501-1506); the
launcher never touches the GPU.  Started under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` the
environment is already there and each process is a rank.

A "step" is one forward of the hot path over one batch of synthetic images already resident in HBM:
pixel_values [B,3,518,518] fp32 -> packed detections [B,100,95] fp32; at N > 1 followed by the path's one collective, the RCCL
all-gather of the packed detections, issued on a side stream so that it overlaps the next step's forward (every gather has
completed when the timed region ends).  Workload at every N: the configuration BASELINE.json's metric is quoted on --
configs[2]: ViT-B/14 518x518, 100 queries, batch 64 -- which fits ONE MI355X, so each GPU runs a full batch of 64 (weak
scaling: global batch 64 N).  At N > 1 the same line also carries `global64_sharded`: configs[2] split literally, 64 / N
images per GPU.
Prints ONE JSON line on rank 0 (contract in the task statement), with
  roofline     : the dominant kernel class (bf16 MFMA GEMM) -- ALGORITHMIC FLOPs (2 M N K per launch) / its summed launch
                 time, measured with HIP events on the launch stream by the library's profile mode over extra forwards of the
                 same step, right after the timed region;
  cpu_baseline : the CPU oracle (kind "port": oracle/dinodet_oracle.py, the parity-checked CPU restatement of the reference)
                 timed on this box's host cores on a bounded sample; its first batch also checks images 0 and 1 of the TIMED
                 batch's own output (`gpu_vs_oracle`, taken from the last timed step's buffer -- the same kernels, dispatch,
                 micro-batch split and graph as the measurement; `batch2_launch` = a separate forward of just those two);
  parity_gated_mode : the same workload in the DEFAULT precision of the drop-in modules, bf16x3 (split products on the bf16
                 MFMA cores, within the 1e-3 gate), with its own value / roofline / gpu_vs_oracle -- measured exactly like the
                 headline (N = 1 only);
  also         : BASELINE configs[1] (ViT-B/14 224x224, batch 32) in both modes, configs[3] (ViT-L/14 518x518 bf16, 16 images) and
                 configs[4] (ViT-g/14 518x518 fp8 MFMA, 32 images) on their per-GPU shards, and `train_step`: the native train()-mode step
                 (section 8 row f1) against the autograd composite on the same frozen prefix (N = 1 only).
Other workloads / modes: --workload {vits224,vitb224,vitl518,vitg518}, --precision {bf16,bf16x3,fp16x2,fp32,fp8}.
`--rehearse-cpu` swaps the model for a stub on the CPU with gloo: it exercises launcher, sharding, barriers, timing and the
overlapped gather without a GPU (tests/test_dist_cpu.py) and labels its line as a rehearsal -- never a measurement.
"""
import argparse
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

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
PEAK = {"bf16": PEAK_BF16, "fp32": PEAK_F32, "fp8": PEAK_FP8, "bf16x3": PEAK_BF16, "fp16x2": PEAK_BF16}
DOMINANT = {"bf16": "gemm_bf16", "fp32": "gemm_f32", "fp8": "gemm_fp8", "bf16x3": "gemm_bf16", "fp16x2": "gemm_bf16"}
KERNEL_NAMES = {"bf16": "bf16 MFMA GEMM launches of the step (gemm_bf16_* / gemm_x3_256x256_kernel<PLAIN>)", "fp32": "gemm_f32_kernel",
                "fp8": "gemm_fp8_256x128_kernel (the e4m3 MFMA GEMM launches: QKV, out-proj, MLP-in, SwiGLU MLP-out)",
                "bf16x3": "gemm_x3_256x256_kernel: bf16 MFMA GEMM on split operands (3 products per K step; achieved = ALGORITHMIC 2MNK FLOPs / time)",
                "fp16x2": "gemm_h2_256x256_kernel: fp16 MFMA main product + both cross terms on one block-scaled e4m3 MFMA (2 bf16-equivalents per K step; achieved = ALGORITHMIC 2MNK FLOPs / time against the bf16 peak)"}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="vitb518", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch override")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32", "fp8", "bf16x3", "fp16x2"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip parity_gated_mode / also / global64_sharded")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of a captured hipGraph")
    ap.add_argument("--no-overlap", action="store_true", help="N > 1: all-gather on the compute stream instead of the side stream")
    ap.add_argument("--rehearse-cpu", action="store_true", help="CPU/gloo rehearsal of the control flow with a stub model (not a measurement)")
    return ap.parse_args(argv)


def build(name, queries, precision, device):
    import torch
    from dinov2_od_amd.models import DINOv2ObjectDetector
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


def train_step_ms(device, steps=10, warmup=3):
    """SURVEY section 8 row f1 beside the forward numbers: one train()-mode step (forward + backward of the trainable subset:
    decoder, heads, LoRA A/B of the last two blocks; train.py:1079-1109) on ViT-B/14 224x224, batch 16 -- the native HIP step and,
    for comparison, the PyTorch autograd composite on the same native frozen prefix (DINODET_NATIVE_TRAIN=0)."""
    import torch
    m, _, _ = build("facebook/dinov2-base", 100, "bf16", device)
    m.train()
    x = torch.rand(16, 3, 224, 224, device=device)

    def step():
        m.zero_grad(set_to_none=True)
        o = m(x)
        (o["pred_logits"].square().mean() + o["pred_boxes"].mean()).backward()

    out = {"workload": "ViT-B/14 224x224, 100 queries, batch 16, forward + backward (dropout 0.1), fp32 trainable path"}
    prev = os.environ.get("DINODET_NATIVE_TRAIN")
    try:
        for key, env in (("native_ms", "1"), ("composite_on_native_prefix_ms", "0")):
            os.environ["DINODET_NATIVE_TRAIN"] = env
            for _ in range(warmup):
                step()
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(steps):
                step()
            torch.cuda.synchronize()
            out[key] = round((time.perf_counter() - t) / steps * 1e3, 3)
    finally:
        if prev is None:
            os.environ.pop("DINODET_NATIVE_TRAIN", None)
        else:
            os.environ["DINODET_NATIVE_TRAIN"] = prev
    return out


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
    import torch
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


def timed_vs_oracle(st, model, x, oracle_first, C):
    """images 0 and 1 of the TIMED batch's own output (the last timed step: same kernels, same dispatch, same micro-batch split and
    graph as the measurement) against the oracle's evaluation of those two images; `batch2_launch` = the round-1/2 figure, a separate
    forward of just those two images (smaller-M kernels), kept for comparison"""
    import torch
    det = st.last_det[:2].clone()
    with torch.no_grad():
        d2 = model.forward_packed(x[:2]).clone()
    torch.cuda.synchronize()
    e = oracle_error(det, oracle_first, C)
    e["taken_from"] = f"rows 0-1 of the timed batch-{x.shape[0]} output (last timed step" + (", hipGraph replay)" if st.graph is not None else ", eager)")
    e["batch2_launch"] = oracle_error(d2, oracle_first, C)
    return e


def oracle_error(gpu_det, oracle_out, C):
    """max |gpu - oracle| / max |oracle| on the images both computed (the metric of tests/cases.py::rel_err)."""
    g = gpu_det.float().cpu()
    n = min(g.shape[0], oracle_out["pred_logits"].shape[0])
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
    l2 = lambda a, b: float((a - b).double().norm() / b.double().norm())
    return {"images": n, "pred_logits_max_rel": rel(g[:n, :, :C], oracle_out["pred_logits"][:n]),
            "pred_boxes_max_rel": rel(g[:n, :, C:], oracle_out["pred_boxes"][:n]),
            "pred_logits_rel_l2": l2(g[:n, :, :C], oracle_out["pred_logits"][:n]),
            "pred_boxes_rel_l2": l2(g[:n, :, C:], oracle_out["pred_boxes"][:n])}


# gate of a rank's detections against the CPU oracle (max-relative, tests/cases.py::rel_err): the parity gate for the gated modes; the
# throughput modes' own bound (bf16 logits < 8e-2, boxes < 3e-2; fp8 looser) -- a broken rank is off by O(1)
RANK_GATE = {"fp32": (1e-3, 1e-3), "bf16x3": (1e-3, 1e-3), "fp16x2": (1e-3, 1e-3), "bf16": (8e-2, 3e-2), "fp8": (5e-1, 1e-1)}


def rank_oracle_check(st, lo, bb, dc, R, precision, world, device, cpu, stub=None):
    """N > 1: EVERY rank checks image 0 of its own shard (global image `lo`) of the timed output against the CPU oracle evaluated on its
    host cores; the errors are all-gathered, rank 0 reports them and the job exits non-zero when any rank leaves its gate."""
    import torch
    dist = torch.distributed
    det = st.last_det[:1].float().cpu() if st.last_det is not None else None
    if cpu:
        want = _StubModel(stub.Q, stub.C).forward_packed(st.x[:1]).float()      # a clean evaluation, not the rank's own
        C = want.shape[-1] - 4
        e = [float((det[..., :C] - want[..., :C]).abs().max() / want[..., :C].abs().max()), float((det[..., C:] - want[..., C:]).abs().max() / want[..., C:].abs().max())]
        gate = (1e-6, 1e-6)
    else:
        from oracle import dinodet_oracle as orc
        from dinov2_od_amd import synth
        torch.set_num_threads(max(1, host_cores() // max(1, world)))
        sd = {k: torch.from_numpy(v) for k, v in synth.detector_state_dict(bb, dc, seed=1).items()}
        img = torch.from_numpy(synth.uniform01(0, f"pixel_values.{R}x{R}.{lo}", (3, R, R)))[None]
        o = orc.detector_forward(sd, bb, dc, img)
        oe = oracle_error(det.to(device), o, dc.num_classes)
        e = [oe["pred_logits_max_rel"], oe["pred_boxes_max_rel"]]
        gate = RANK_GATE[precision]
    mine = torch.tensor(e, device=device, dtype=torch.float64)
    allr = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allr, mine)
    rows = [{"rank": r, "global_image": None, "pred_logits_max_rel": float(v[0]), "pred_boxes_max_rel": float(v[1]),
             "in_gate": bool(float(v[0]) < gate[0] and float(v[1]) < gate[1])} for r, v in enumerate(allr)]
    return {"gate_logits_boxes": list(gate), "ranks": rows, "all_in_gate": all(r["in_gate"] for r in rows)}


def make_images(B_local, R, lo, device):
    """image g of the global batch (every image distinct: data-dependent clocks), resident on `device`"""
    import torch
    from dinov2_od_amd import synth
    x = torch.empty(B_local, 3, R, R, device=device)
    for i in range(B_local):
        x[i] = torch.from_numpy(synth.uniform01(0, f"pixel_values.{R}x{R}.{lo + i}", (3, R, R))).to(device)
    return x


class _StubModel:
    """--rehearse-cpu: stands for the detector; cheap, deterministic per image (so the gather can be checked)"""

    def __init__(self, Q, C, offset=0.0):
        self.Q, self.C, self.offset = Q, C, offset      # offset != 0: a rank whose detections are wrong (tests the per-rank gate)

    def forward_packed(self, x):
        import torch
        s = x.flatten(1).mean(dim=1) + self.offset
        return s[:, None, None] + torch.arange(self.Q * (self.C + 4), dtype=torch.float32).view(1, self.Q, self.C + 4)


class Stepper:
    """one forward (+ the gather at N > 1) per call, from a captured hipGraph when possible"""

    def __init__(self, model, x, world, Q, C, use_graph, overlap, cpu=False):
        import torch
        self.torch, self.model, self.x, self.world, self.cpu = torch, model, x, world, cpu
        self.graph, self.static_det, self.i = None, None, 0
        self.overlap = overlap and world > 1 and not cpu
        B = x.shape[0]
        dev = x.device
        if world > 1:
            nbuf = 2 if self.overlap else 1
            self.gathered = [torch.empty(B * world, Q, C + 4, device=dev) for _ in range(nbuf)]
            if self.overlap:
                self.stage = [torch.empty(B, Q, C + 4, device=dev) for _ in range(2)]
                self.comm = torch.cuda.Stream(device=dev)
                self.ready = [torch.cuda.Event() for _ in range(2)]
                self.done = [torch.cuda.Event() for _ in range(2)]
        with torch.no_grad():
            self.out = self._eager()                      # packs weights, sizes workspace, sets func attributes
            self.sync()
            if use_graph and not cpu:
                # the forward (all kernels of the hot path) is captured in a hipGraph; the RCCL all-gather stays an eager call
                try:
                    s = torch.cuda.Stream()
                    s.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(s):
                        model.forward_packed(x)
                    torch.cuda.current_stream().wait_stream(s)
                    g = torch.cuda.CUDAGraph()
                    # thread_local: an RCCL watchdog thread polling events must not invalidate the capture
                    with torch.cuda.graph(g, capture_error_mode="thread_local"):
                        self.static_det = model.forward_packed(x)
                    self.graph = g
                except Exception as e:                    # capture unsupported -> eager, say so
                    print(f"[bench] hipGraph capture failed ({type(e).__name__}: {e}); running eager", file=sys.stderr)
                    self.graph = None
            if world > 1:
                # one untimed step incl. the collective: RCCL builds its communicator lazily on the first call (seconds), which must
                # not land in the timed region even with --warmup 0
                self()
                self.sync()

    def sync(self):
        if not self.cpu:
            self.torch.cuda.synchronize()

    def _eager(self):
        return self.model.forward_packed(self.x)

    def _forward(self):
        if self.graph is not None:
            self.graph.replay()
            return self.static_det
        return self._eager()

    def __call__(self):
        from dinov2_od_amd import dist as ddist
        torch = self.torch
        det = self._forward()
        self.last_det = det                               # the timed steps' own output (graph replays: the static buffer)
        if self.world == 1:
            return det
        if not self.overlap:
            return ddist.gather_detections_equal(det, self.gathered[0])
        # side-stream gather: step i's detections are copied to a staging buffer (two of them: the forward of step i + 1
        # overwrites the graph's static output) and gathered on `comm` while the compute stream runs step i + 1
        k = self.i & 1
        self.i += 1
        main = torch.cuda.current_stream()
        main.wait_event(self.done[k])                     # the gather of step i - 2 has read stage[k] / written gathered[k]
        self.stage[k].copy_(det, non_blocking=True)
        self.ready[k].record(main)
        with torch.cuda.stream(self.comm):
            self.comm.wait_event(self.ready[k])
            ddist.gather_detections_equal(self.stage[k], self.gathered[k])
            self.done[k].record(self.comm)
        return self.gathered[k]

    def last_gathered(self):
        """the most recent gathered buffer, complete (host-synchronised)"""
        self.sync()
        if self.world == 1:
            return None
        return self.gathered[(self.i - 1) & 1] if self.overlap else self.gathered[0]


def timed(stepper, steps, warmup, world, device, cpu=False):
    """W untimed warm-up steps, then EXACTLY K steps bracketed by barrier + synchronize on both sides; MAX over ranks"""
    import torch
    dist = torch.distributed
    with torch.no_grad():
        for _ in range(warmup):
            stepper()
        stepper.sync()
        if world > 1:
            dist.barrier()
        stepper.sync()
        evs = None if cpu else [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]   # per-step spread
        t0 = time.perf_counter()
        if evs:
            evs[0].record()
        for i in range(steps):
            stepper()
            if evs:
                evs[i + 1].record()
        stepper.sync()
        if world > 1:
            dist.barrier()
        stepper.sync()
        dt = time.perf_counter() - t0
    step_ms = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(steps)) if evs else [1e3 * dt / steps] * steps
    pct = lambda q: step_ms[min(len(step_ms) - 1, int(q * len(step_ms)))]
    if world > 1:
        # every rank's own clock around the same K steps + its median step from the event pairs (the spread says whether one GPU or one xGMI
        # link drags the job); `dt` of the job = MAX over ranks
        mine = torch.tensor([dt, pct(0.50)], device=device, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        timed.per_rank = [{"rank": r, "ms_per_step": 1e3 * float(v[0]) / steps, "step_ms_p50": float(v[1])} for r, v in enumerate(allr)]
        dt = max(float(v[0]) for v in allr)
    else:
        timed.per_rank = None
    return dt, [pct(0.10), pct(0.50), pct(0.90)]


def roofline_leg(model, x, precision, steps, traffic=None, traffic_src=None):
    """same step, library event timing per kernel class (eager, outside the timed region)"""
    import torch
    eng = model._get_engine()
    nprof = max(3, min(10, steps))
    # kernel durations are taken with the batch on ONE stream: the timed steps run it as concurrent micro-batches (engine.py
    # micro_streams), under which two kernels share the chip and an event pair around one of them measures both
    ms_saved = eng.micro_streams
    eng.micro_streams = 1
    with torch.no_grad():
        eng.profile(True)
        for _ in range(nprof):
            model.forward_packed(x)
        torch.cuda.synchronize()
        prof = eng.profile_read()
        eng.profile(False)
    eng.micro_streams = ms_saved
    d = prof[DOMINANT[precision]]
    peak = PEAK[precision]
    ach = d["flops"] / (d["ms"] * 1e-3) if d["ms"] > 0 else 0.0
    return {"bound": "mfma", "kernel": KERNEL_NAMES[precision],
            "achieved": ach / 1e12, "peak": peak / 1e12, "unit": "TFLOP/s",
            "frac": ach / peak, "traffic": traffic, "traffic_source": traffic_src,
            "measured": "HIP event pairs per launch over extra forwards right after the timed region, the batch on one stream (DINODET_MICRO_STREAMS=1: the timed steps overlap two micro-batches, under which per-kernel durations are not separable)",
            "launches_per_step": d["launches"] // nprof, "avg_launch_us": 1e3 * d["ms"] / max(1, d["launches"]),
            "flops_per_launch_avg": d["flops"] / max(1, d["launches"]),
            "class_ms_per_step": d["ms"] / nprof,
            "other_kernels": {k: {"ms_per_step": v["ms"] / nprof, "tflops": (v["flops"] / (v["ms"] * 1e-3) / 1e12) if v["ms"] > 0 and v["flops"] > 0 else None}
                              for k, v in prof.items() if v["launches"]}}


def pmc_traffic(workload, precision, B_local):
    """HBM-side bytes per launch of the dominant kernel class from the committed rocprofv3 --pmc passes of this same command
    (separate runs: not measurable live); latest round's file"""
    for tj in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic*.json")), reverse=True):
        try:
            tdat = json.load(open(tj))
        except Exception:
            continue
        if tdat.get("workload", "vitb518") == workload and tdat.get("precision", "bf16") == precision and tdat.get("batch", 64) == B_local:
            return tdat.get("gemm_bf16_avg_bytes_per_launch"), os.path.relpath(tj, ROOT)
    return None, None


def measure_mode(name, Q, R, precision, x, steps, warmup, device, use_graph, oracle_first=None):
    """build `precision`, time it like the headline (graph replay, barrier-free at N = 1), roofline leg, optional oracle check"""
    import torch
    from dinov2_od_amd.config import flops_per_image
    model, bb, dc = build(name, Q, precision, device)
    st = Stepper(model, x, 1, Q, dc.num_classes, use_graph, False)
    dt, pcts = timed(st, steps, warmup, 1, device)
    B = x.shape[0]
    ips = B * steps / dt
    fpi = flops_per_image(bb, dc, R, R)
    out = {"precision": precision, "value": ips, "unit": "images/s", "batch": B, "steps": steps, "ms_per_step": 1e3 * dt / steps,
           "step_ms_p10_p50_p90": pcts, "hipgraph": st.graph is not None,
           "mfma_roofline_frac_end_to_end": ips * fpi / PEAK[precision],
           "roofline": roofline_leg(model, x, precision, steps)}
    if oracle_first is not None:
        out["gpu_vs_oracle"] = timed_vs_oracle(st, model, x, oracle_first, dc.num_classes)
    del st, model
    torch.cuda.empty_cache()
    return out


def worker(a):
    import torch
    from dinov2_od_amd import dist as ddist
    from dinov2_od_amd.config import flops_per_image
    cpu = a.rehearse_cpu
    # RCCL ("nccl") is the backend of record; DINODET_DIST_BACKEND=gloo only exists to rehearse the N > 1 control flow on a
    # one-GPU box (all ranks on cuda:0)
    backend = "gloo" if cpu else (os.environ.get("DINODET_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else None))
    rank, world, local = ddist.init_from_env(backend)
    if os.environ.get("DINODET_DIST_BACKEND") == "gloo":
        local = 0
    if not cpu and not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the hot path has no CPU fallback)")
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if cpu:
        device = torch.device("cpu")
    else:
        torch.cuda.set_device(local)
        device = torch.device("cuda", local)
    n_ranks_seen = torch.distributed.get_world_size() if world > 1 else 1

    name, R, Q, B_local, desc = WORKLOADS[a.workload]
    if a.batch:
        B_local = a.batch
        desc += f" [per-GPU batch overridden to {B_local}]"
    if cpu:
        R, C = 28, 91
        model, bb, dc = _StubModel(Q, C, 1.0 if os.environ.get("BENCH_REHEARSE_BREAK_RANK") == str(rank) else 0.0), None, None
    else:
        model, bb, dc = build(name, Q, a.precision, device)
        C = dc.num_classes
    lo, hi = ddist.shard_bounds(B_local * world, rank, world)
    x = make_images(B_local, R, lo, device)
    use_graph = not a.no_graph and not cpu

    st = Stepper(model, x, world, Q, C, use_graph, not a.no_overlap, cpu)
    dt, pcts = timed(st, a.steps, a.warmup, world, device, cpu)
    global_batch = B_local * world
    ips = global_batch * a.steps / dt
    per_rank = timed.per_rank
    gather_ok = None
    rank_check = None
    if world > 1:       # the gathered buffer holds every rank's detections in rank order (checked on the last step's buffer)
        g = st.last_gathered()
        with torch.no_grad():
            mine = model.forward_packed(x)
        ok_here = bool(torch.allclose(g[lo:hi].cpu(), mine.cpu(), rtol=0, atol=0)) and g.shape[0] == global_batch
        flag = torch.tensor([1.0 if ok_here else 0.0], device=device)
        torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)      # every rank's view of the gathered buffer, not only rank 0's
        gather_ok = bool(flag.item() > 0.5)
        rank_check = rank_oracle_check(st, lo, bb, dc, R, a.precision, world, device, cpu, stub=model if cpu else None)
        for r_ in rank_check["ranks"]:
            r_["global_image"] = ddist.shard_bounds(global_batch, r_["rank"], world)[0]

    res = {"metric": "images/sec forward, DINOv2 ViT-B/14 518x518 + 100-query head" if a.workload == "vitb518" else f"images/sec forward, {a.workload}",
           "value": ips, "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
           "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": a.precision, "data": "synthetic",
           "config": {"workload": desc, "global_batch": global_batch, "per_gpu_batch": B_local, "image": R, "queries": Q,
                      "parallelism": f"dp{world}", "hipgraph": st.graph is not None, "n_ranks_seen": n_ranks_seen,
                      "collective": None if world == 1 else ("all_gather_into_tensor of packed [B_local,Q,C+4] fp32 per step, "
                                                             + ("side stream, overlapped with the next forward" if st.overlap else "compute stream")),
                      "gather_checked": gather_ok,
                      "micro_batches": None if cpu else (lambda e: e.micro_streams if e._micro_ok(B_local, R, R) else 1)(model._get_engine())},
           "step_ms_p10_p50_p90": pcts}
    if world > 1:
        ms = [r_["ms_per_step"] for r_ in per_rank]
        res["per_rank"] = {"ms_per_step": per_rank, "spread_ms": max(ms) - min(ms), "slowest_rank": ms.index(max(ms)),
                           "gpu_vs_oracle": rank_check}
    if cpu:
        res["metric"] = "REHEARSAL (CPU stub model, gloo): control flow only, not a measurement"
        res["data"] = "rehearsal-cpu"
    else:
        fpi = flops_per_image(bb, dc, R, R)
        res["config"]["gflop_per_image"] = fpi / 1e9
        res["mfma_roofline_frac_end_to_end"] = ips * fpi / (PEAK[a.precision] * world)
        traffic, traffic_src = pmc_traffic(a.workload, a.precision, B_local)
        res["roofline"] = roofline_leg(model, x, a.precision, a.steps, traffic, traffic_src)

    # ---- N > 1: BASELINE configs[2] split literally -- global batch 64, 64 / N images per GPU (strong-scaling view)
    if world > 1 and not a.no_extras and a.workload == "vitb518" and not a.batch and 64 % world == 0:
        bs = 64 // world
        lo2, _ = ddist.shard_bounds(64, rank, world)
        xs = x[:bs].contiguous() if cpu else make_images(bs, R, lo2, device)
        st2 = Stepper(model, xs, world, Q, C, use_graph, not a.no_overlap, cpu)
        dt2, pcts2 = timed(st2, a.steps, a.warmup, world, device, cpu)
        res["global64_sharded"] = {"value": 64 * a.steps / dt2, "unit": "images/s", "global_batch": 64, "per_gpu_batch": bs,
                                   "ms_per_step": 1e3 * dt2 / a.steps, "step_ms_p10_p50_p90": pcts2, "scaling": "strong",
                                   "note": "BASELINE configs[2] as written: batch 64 sharded over the N GPUs"}
        if not cpu:
            res["global64_sharded"]["mfma_roofline_frac_end_to_end"] = 64 * a.steps / dt2 * fpi / (PEAK[a.precision] * world)
        del st2

    if rank == 0 and world == 1 and not cpu and not a.no_cpu_baseline:
        # the oracle's two images are make_pixels(2, R, R, seed=0) = images 0 and 1 of the batch
        res["cpu_baseline"], oracle_first = cpu_baseline(bb, dc, R)
        res["cpu_baseline"]["gpu_vs_oracle"] = timed_vs_oracle(st, model, x, oracle_first, C)
        if a.precision == "bf16":
            res["precision_note"] = ("`value` is the configuration BASELINE names: single-pass bf16 MFMA operands, which cannot meet the 1e-3 gate "
                                     "on logits (8-bit mantissa; DESIGN.md section 2): see cpu_baseline.gpu_vs_oracle.  The drop-in modules' DEFAULT "
                                     "precision is the parity-gated bf16x3 mode, measured the same way in parity_gated_mode")
        if a.precision == "bf16" and not a.no_extras:
            del st, model
            torch.cuda.empty_cache()
            try:
                res["parity_gated_mode"] = measure_mode(name, Q, R, "bf16x3", x, max(5, a.steps // 4), max(2, a.warmup // 4), device, use_graph, oracle_first)
                res["parity_gated_mode"]["note"] = ("every backbone product as bf16 split products (Ah Wh + Ah Wl + Al Wh) on the bf16 MFMA "
                                                    "kernels incl. the flash attention, fp32 elsewhere: within 1e-3 of the reference "
                                                    "(tests/test_gpu_forward.py GATED); 3x the MFMA work of the single-pass bf16 mode; "
                                                    "roofline.achieved counts ALGORITHMIC FLOPs")
            except Exception as e:
                res["parity_gated_mode"] = {"error": f"{type(e).__name__}: {e}"}
            try:
                res["fp16x2_mode"] = measure_mode(name, Q, R, "fp16x2", x, max(5, a.steps // 4), max(2, a.warmup // 4), device, use_graph, oracle_first)
                res["fp16x2_mode"]["note"] = ("the second parity-gated mode (opt-in): the block linears as fp16 main product + both cross terms on one "
                                              "block-scaled e4m3 MFMA (2 bf16-MFMA-equivalents per product instead of 3), everything else as bf16x3; "
                                              "roofline.achieved counts ALGORITHMIC FLOPs against the bf16 peak")
            except Exception as e:
                res["fp16x2_mode"] = {"error": f"{type(e).__name__}: {e}"}
            if a.workload == "vitb518":
                res["also"] = {}
                n2, R2, Q2, B2, d2 = WORKLOADS["vitb224"]
                x2 = make_images(B2, R2, 0, device)
                for prec in ("bf16", "bf16x3"):
                    try:
                        r2 = measure_mode(n2, Q2, R2, prec, x2, max(10, a.steps // 2), max(3, a.warmup // 2), device, use_graph)
                        r2["workload"] = d2
                        res["also"][f"vitb224_{prec}"] = r2
                    except Exception as e:
                        res["also"][f"vitb224_{prec}"] = {"error": f"{type(e).__name__}: {e}"}
                del x2
                torch.cuda.empty_cache()
                # BASELINE configs[2] as written shards its 64 images over 8 GPUs: the 8-image shard on this GPU (the headline runs the whole
                # batch of 64 per GPU -- weak scaling; an 8-GPU run of the literal configuration moves at 8 x this figure less the all-gather)
                for prec in ("bf16", "bf16x3"):
                    try:
                        r8 = measure_mode(name, Q, R, prec, x[:8].contiguous(), max(10, a.steps // 2), max(3, a.warmup // 2), device, use_graph)
                        r8["workload"] = "ViT-B/14 518x518, 100 queries, batch 8 (the per-GPU shard of BASELINE configs[2] as written)"
                        res["also"][f"vitb518_shard8_{prec}"] = r8
                    except Exception as e:
                        res["also"][f"vitb518_shard8_{prec}"] = {"error": f"{type(e).__name__}: {e}"}
                torch.cuda.empty_cache()
                # BASELINE configs[3] and configs[4] on their per-GPU shards (16 / 32 images), a few timed steps each: the driver-visible
                # number for the ViT-L bf16 and the ViT-g fp8-MFMA configurations (parity: tests/test_gpu_timed_shapes.py runs these
                # very batches against the reference's G7 / G8 goldens)
                for wl, prec in (("vitl518", "bf16"), ("vitg518", "fp8")):
                    try:
                        n3, R3, Q3, B3, d3 = WORKLOADS[wl]
                        x3 = make_images(B3, R3, 0, device)
                        r3 = measure_mode(n3, Q3, R3, prec, x3, 5, 2, device, use_graph)
                        r3["workload"] = d3
                        res["also"][f"{wl}_{prec}"] = r3
                        del x3
                    except Exception as e:
                        res["also"][f"{wl}_{prec}"] = {"error": f"{type(e).__name__}: {e}"}
                    torch.cuda.empty_cache()
                try:
                    res["also"]["train_step"] = train_step_ms(device)
                except Exception as e:
                    res["also"]["train_step"] = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0:
        print(json.dumps(res), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
        # a rank whose detections left the gate, or a gathered buffer that does not hold every rank's rows, fails the whole job (every
        # rank exits non-zero: the launcher propagates the first non-zero code)
        if not gather_ok or not rank_check["all_in_gate"]:
            if rank == 0:
                print(f"[bench] FAILED: gather_checked={gather_ok}, per-rank oracle check {rank_check}", file=sys.stderr)
            sys.exit(3)


def main():
    a = parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        # launcher: fresh child processes, one per GPU; this process never touches the GPU (no torch import up to here)
        from dinov2_od_amd.launch import spawn_ranks
        sys.exit(spawn_ranks(os.path.abspath(__file__), sys.argv[1:], a.gpus))
    worker(a)


if __name__ == "__main__":
    main()
