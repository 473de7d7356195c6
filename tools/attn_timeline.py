#!/usr/bin/env python3
"""Per-workgroup cycle split of the bf16 attention kernel: tile loop total vs time waiting for K/V DMA + barrier."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools._tuning_lib  # noqa: F401,E402  (the -DDINODET_TUNING build: this tool uses tuning hooks)
import torch, numpy as np
from dinov2_od_amd import _native as nat
L = nat.lib(); dev = torch.device("cuda:0")
B, N, D = int(os.environ.get("PROF_B", "64")), 1370, 768
g = torch.Generator().manual_seed(0)
qkv = (torch.randn(B, N, 3 * D, generator=g) * 0.5).to(dev).to(torch.bfloat16); ctx = torch.empty(B, N, D, device=dev, dtype=torch.bfloat16)
run = lambda: L.dod_op_attention_bf16(nat.ptr(qkv), nat.ptr(ctx), B, N, D // 64, 0.125, nat.stream_ptr())
for _ in range(3): run()
torch.cuda.synchronize()
nblk = ((B * 12 + 7) // 8 * 8) * ((N + 255) // 256)
buf = torch.zeros(nblk * 6 + 64, dtype=torch.int64, device=dev)
L.dod_debug_attn_stamps(C.c_void_p(buf.data_ptr())); run(); torch.cuda.synchronize(); L.dod_debug_attn_stamps(C.c_void_p(0))
t = buf.cpu().numpy()[: nblk * 6].reshape(nblk, 6).astype(np.float64)
t = t[t[:, 2] > 0]
act = t[t[:, 3] > 0]
print(f"workgroups {len(t)}; tile loop cycles/tile median {np.median(act[:,0]/act[:,2]):.0f}; waiting (DMA+barrier) {100*np.median(act[:,1]/act[:,0]):.1f}% of it "
      f"(p10 {100*np.percentile(act[:,1]/act[:,0],10):.1f}% p90 {100*np.percentile(act[:,1]/act[:,0],90):.1f}%)")
t0 = act[:, 4].min()
entry, loop_end = (act[:, 4] - t0) / 100.0, (act[:, 5] - t0) / 100.0
print(f"kernel span (first entry -> last loop end) {loop_end.max():.1f} us; per WG entry -> loop end median {np.median(loop_end - entry):.2f} us "
      f"(p10 {np.percentile(loop_end - entry, 10):.2f}, p90 {np.percentile(loop_end - entry, 90):.2f}); tiles/WG {act[0,2]:.0f}")
slots = 512
print(f"sum of WG (entry -> loop end) / {slots} slots = {np.sum(loop_end - entry) / slots:.1f} us")
order = np.argsort(entry)
e = entry[order]
print("entries per 50-us window:", np.histogram(e, bins=np.arange(0, loop_end.max() + 50, 50))[0])
