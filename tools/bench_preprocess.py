#!/usr/bin/env python3
"""Device input pipeline (dod_preprocess) vs Pillow on the host: 64 COCO-sized images (480x640) -> 224^2 and 518^2."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from PIL import Image
from dinov2_od_amd import _native as nat, preprocess as pre

rng = np.random.default_rng(0)
imgs = [rng.integers(0, 256, size=(480, 640, 3), dtype=np.uint8) for _ in range(64)]
for R in (224, 518):
    L = nat.lib(); B = len(imgs)
    hs = np.array([a.shape[0] for a in imgs], np.int32); ws = np.array([a.shape[1] for a in imgs], np.int32)
    so = np.concatenate([[0], np.cumsum(hs.astype(np.int64) * ws * 3)[:-1]]).astype(np.int64)
    to = np.concatenate([[0], np.cumsum(hs.astype(np.int64) * R * 3)[:-1]]).astype(np.int64)
    src = torch.from_numpy(np.concatenate([a.reshape(-1) for a in imgs])).cuda()
    m = [torch.from_numpy(x).cuda() for x in (so, hs, ws, to)]
    tmp = torch.empty(int((hs.astype(np.int64) * R * 3).sum()), dtype=torch.uint8, device="cuda")
    out = torch.empty(B, 3, R, R, device="cuda")
    run = lambda: nat.check(L.dod_preprocess(nat.ptr(src), nat.ptr(m[0]), nat.ptr(m[1]), nat.ptr(m[2]), B, 480, 640, R, R, nat.ptr(tmp), nat.ptr(m[3]), nat.ptr(out), nat.stream_ptr()))
    for _ in range(3): run()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): run()
    b.record(); torch.cuda.synchronize()
    t_k = a.elapsed_time(b) / 20 * 1e-3
    t0 = time.perf_counter(); o = pre.preprocess_batch(imgs, (R, R)); torch.cuda.synchronize(); t_e2e = time.perf_counter() - t0
    t0 = time.perf_counter()
    ref = [np.transpose(np.array(Image.fromarray(im, "RGB").resize((R, R), Image.BILINEAR)), (2, 0, 1)).astype(np.float32) / 255 for im in imgs]
    t_cpu = time.perf_counter() - t0
    alg = src.numel() + tmp.numel() * 2 + out.numel() * 4
    print(f"64 x 480x640 -> {R}^2: kernels {t_k*1e6:.0f} us ({alg/t_k/1e9:.0f} GB/s of {alg/1e6:.0f} MB algorithmic: source + temp write/read + fp32 out), "
          f"incl. H2D of the raw bytes {t_e2e*1e3:.1f} ms, Pillow on one host core {t_cpu*1e3:.0f} ms")
