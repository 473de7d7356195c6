#!/usr/bin/env python3
"""Yardstick: the same forward as plain PyTorch-ROCm ops (the train()-mode composite with dropout 0, every block in torch:
hipBLASLt linears, torch SDPA, a vectorised deformable gather instead of the reference's Python loop) under bf16 autocast,
next to the native path.  ViT-B/14 518x518, 100 queries, batch 64 (bench workload).  Nothing in the product uses this."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["DINODET_COMPOSITE_FULL"] = "1"
import torch
from bench import build
B = int(os.environ.get("EAGER_B", "64"))
m, bb, dc = build("facebook/dinov2-base", 100, "bf16", torch.device("cuda"))
x = torch.rand(B, 3, 518, 518, device="cuda")
def timeit(fn, n=5):
    fn(); fn(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n
m.eval()
with torch.no_grad():
    t_nat = timeit(lambda: m.forward_packed(x), 10)
m.train(); m._dropout_p = m.decoder._dropout_p = 0.0
for mod in m.modules():
    if isinstance(mod, torch.nn.MultiheadAttention): mod.dropout = 0.0
with torch.no_grad():
    t_f32 = timeit(lambda: m(x))
    with torch.autocast("cuda", dtype=torch.bfloat16):
        t_bf = timeit(lambda: m(x))
print(f"batch {B}: native bf16 {B/t_nat:7.1f} images/s ({t_nat*1e3:.1f} ms) | PyTorch-ROCm eager, bf16 autocast {B/t_bf:7.1f} images/s ({t_bf*1e3:.1f} ms) "
      f"| PyTorch-ROCm eager fp32 {B/t_f32:7.1f} images/s ({t_f32*1e3:.1f} ms)")
