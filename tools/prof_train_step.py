#!/usr/bin/env python3
"""The native train()-mode step alone (forward + backward, ViT-B/14 224x224, batch 16) for a rocprofv3 kernel trace:
    rocprofv3 --kernel-trace -d out -- python3 tools/prof_train_step.py [steps] [resolution] [batch]
    python3 tools/rocprof_by_grid.py out/.../*_results.db"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
m, bb, dc = build("facebook/dinov2-base", 100, os.environ.get("DINODET_PRECISION", "bf16"), torch.device("cuda"))
m.train()
R = int(sys.argv[2]) if len(sys.argv) > 2 else 224
B = int(sys.argv[3]) if len(sys.argv) > 3 else 16
x = torch.rand(B, 3, R, R, device="cuda")
def step():
    m.zero_grad(set_to_none=True)
    o = m(x)
    (o["pred_logits"].square().mean() + o["pred_boxes"].mean()).backward()
for _ in range(2): step()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(steps): step()
torch.cuda.synchronize()
print(f"native train step: {(time.perf_counter() - t) / steps * 1e3:7.2f} ms per forward+backward (batch {B}, {R}x{R}; {steps} steps)")
