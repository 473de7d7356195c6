#!/usr/bin/env python3
"""train()-mode step (forward + backward) on the drop-in detector, ViT-B/14 224x224, batch 16: the native step (HIP forward with a
tape + HIP backward; the default) or, with DINODET_NATIVE_TRAIN=0, the native frozen prefix + the PyTorch autograd composite --
against the all-composite evaluation (selected by an input that requires grad).
    python tools/bench_train_step.py [resolution] [batch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build
m, bb, dc = build("facebook/dinov2-base", 100, os.environ.get("DINODET_PRECISION", "bf16"), torch.device("cuda"))
m.train()
R = int(sys.argv[1]) if len(sys.argv) > 1 else 224
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
x = torch.rand(B, 3, R, R, device="cuda")
def step(inp):
    m.zero_grad(set_to_none=True)
    o = m(inp)
    (o["pred_logits"].square().mean() + o["pred_boxes"].mean()).backward()
first = "native prefix + composite" if os.environ.get("DINODET_NATIVE_TRAIN", "1") == "0" else "native step"
for name, mk in ((first, lambda: x), ("all-composite", lambda: x.clone().requires_grad_(True))):
    for _ in range(2): step(mk())
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5): step(mk())
    torch.cuda.synchronize()
    print(f"{name:24s}: {(time.perf_counter() - t) / 5 * 1e3:7.1f} ms per forward+backward (batch {B}, {R}x{R})")
