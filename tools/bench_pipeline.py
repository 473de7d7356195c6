#!/usr/bin/env python3
"""The widened path end to end on one GPU: decoded uint8 images (64 x 480x640, host memory) -> device resize + ToTensor
(dod_preprocess) -> forward (bf16) -> device post-processing (dod_postprocess) -> COCO records on the host."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import build
from dinov2_od_amd import preprocess as pre, postprocess as post

R = int(os.environ.get("PIPE_R", "518"))
m, bb, dc = build("facebook/dinov2-base", 100, "bf16", torch.device("cuda"))
rng = np.random.default_rng(0)
imgs = [rng.integers(0, 256, size=(480, 640, 3), dtype=np.uint8) for _ in range(64)]
ids = list(range(1000, 1064))
def once():
    t = [time.perf_counter()]
    x = pre.preprocess_batch(imgs, (R, R)); torch.cuda.synchronize(); t.append(time.perf_counter())
    with torch.no_grad():
        det = m.forward_packed(x)
    torch.cuda.synchronize(); t.append(time.perf_counter())
    rec = post.postprocess_packed(det, dc.num_classes, ids, 0.05); t.append(time.perf_counter())
    return [b - a for a, b in zip(t, t[1:])], len(rec)
once(); once()
acc = np.zeros(3); n = 0
for _ in range(5):
    d, nrec = once(); acc += d; n += 1
acc /= n
print(f"64 images 480x640 -> {R}^2: preprocess incl. H2D {acc[0]*1e3:.1f} ms, forward {acc[1]*1e3:.1f} ms, postprocess incl. D2H {acc[2]*1e3:.2f} ms "
      f"({nrec} records) -> {64 / acc.sum():.0f} images/s end to end ({64 / acc[1]:.0f} forward only)")
