#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dinov2_od_amd import _native as nat
L = nat.lib(); dev = torch.device("cuda:0")
B, N, D = int(os.environ.get("PROF_B", "64")), 1370, 768
g = torch.Generator().manual_seed(0)
qkv = (torch.randn(B, N, 3 * D, generator=g) * 0.5).to(dev).to(torch.bfloat16); ctx = torch.empty(B, N, D, device=dev, dtype=torch.bfloat16)
for _ in range(3):
    L.dod_op_attention_bf16(nat.ptr(qkv), nat.ptr(ctx), B, N, D // 64, 0.125, nat.stream_ptr())
torch.cuda.synchronize()
