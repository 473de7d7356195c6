#!/usr/bin/env python3
"""Which scale byte does v_mfma_scale_f32_32x32x64_f8f6f4 apply where?  All-ones e4m3 operands through dod_op_linear_fp8_mx with
scale patterns that switch single (row, half, K-tile) blocks on (byte 127 = 2^0) and everything else off (byte 1 = 2^-126)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dinov2_od_amd import _native as nat
L = nat.lib()
M, N, K = 256, 128, 512
A = torch.full((M, K), 0x38, dtype=torch.uint8, device="cuda")
W = torch.full((N, K), 0x38, dtype=torch.uint8, device="cuda")
sw = torch.ones(N, device="cuda")
def run(bs):
    out = torch.empty(M, N, device="cuda")
    nat.check(L.dod_op_linear_fp8_mx(nat.ptr(A), K, nat.ptr(bs.contiguous()), nat.ptr(W), K, nat.ptr(sw), M, N, K, None, None, None, 0, nat.ptr(out), nat.DOD_F32, N, 0, nat.stream_ptr()))
    torch.cuda.synchronize()
    return out.cpu()
def pat(fn):
    bs = torch.ones(M, 2, K // 64, dtype=torch.uint8)
    for r in range(M):
        for h in range(2):
            for kt in range(K // 64):
                if fn(r, h, kt): bs[r, h, kt] = 127
    return bs.cuda()
o = run(pat(lambda r, h, kt: True)); print("all on: expect", K, "->", o[0, 0].item(), o[100, 77].item(), o.min().item(), o.max().item())
o = run(pat(lambda r, h, kt: h == 0)); print("half 0 on: expect", K // 2, "->", o[0, 0].item(), o[37, 5].item(), o.min().item(), o.max().item())
o = run(pat(lambda r, h, kt: h == 1)); print("half 1 on: expect", K // 2, "->", o[0, 0].item(), o[37, 5].item(), o.min().item(), o.max().item())
for k in range(K // 64):
    o = run(pat(lambda r, h, kt: kt == k)); print(f"K-tile {k} on: expect 64 ->", o[0, 0].item(), o[200, 100].item(), o.min().item(), o.max().item())
o = run(pat(lambda r, h, kt: r % 2 == 0)); print("even rows on: rows 0..5 ->", [o[r, 0].item() for r in range(6)], "rows 32..35", [o[r, 3].item() for r in range(32, 36)])
o = run(pat(lambda r, h, kt: r == 5 and h == 1 and kt == 2)); nz = (o > 1e-3).nonzero(); print("single (5, 1, 2): nonzero rows", sorted(set(nz[:, 0].tolist()))[:8], "value", o[5, 0].item())
# random block scales on all-ones data, several tiles and groups
for (M, N, K) in ((515, 384, 1536), (256, 128, 1536), (515, 384, 512), (300, 128, 256)):
    A = torch.full((M, K), 0x38, dtype=torch.uint8, device="cuda")
    W = torch.full((N, K), 0x38, dtype=torch.uint8, device="cuda")
    sw = torch.ones(N, device="cuda")
    g = torch.Generator().manual_seed(1)
    eb = torch.randint(120, 135, (M, K // 32), generator=g)                       # block order
    lay = eb.reshape(M, K // 64, 2).permute(0, 2, 1).reshape(M, K // 32).to(torch.uint8).cuda()
    want = (32.0 * torch.pow(torch.tensor(2.0, dtype=torch.float64), (eb - 127).double())).sum(1)
    out = torch.empty(M, N, device="cuda")
    nat.check(L.dod_op_linear_fp8_mx(nat.ptr(A), K, nat.ptr(lay), nat.ptr(W), K, nat.ptr(sw), M, N, K, None, None, None, 0, nat.ptr(out), nat.DOD_F32, N, 0, nat.stream_ptr()))
    torch.cuda.synchronize()
    o = out.cpu().double()
    rel = ((o - want[:, None]).abs() / want[:, None])
    bad = (rel > 1e-5).nonzero()
    print(f"random scales M={M} N={N} K={K}: max rel {rel.max().item():.3e}, bad elements {len(bad)}", "first bad rows", sorted(set(bad[:, 0].tolist()))[:10], "cols", sorted(set(bad[:, 1].tolist()))[:6])
# which ELEMENTS does a lane's scale apply to?  one element of row 7 doubled, one block (half h of K-tile 0) on
M, N, K = 256, 128, 256
W = torch.full((N, K), 0x38, dtype=torch.uint8, device="cuda"); sw = torch.ones(N, device="cuda")
for p in (0, 15, 16, 31, 32, 47, 48, 63, 64, 100):
    A = torch.full((M, K), 0x38, dtype=torch.uint8); A[7, p] = 0x40; A = A.cuda()
    res = []
    for h in (0, 1):
        for kt in (0, 1):
            bs = torch.ones(M, 2, K // 64, dtype=torch.uint8); bs[:, h, kt] = 127; bs = bs.cuda()
            out = torch.empty(M, N, device="cuda")
            nat.check(L.dod_op_linear_fp8_mx(nat.ptr(A), K, nat.ptr(bs), nat.ptr(W), K, nat.ptr(sw), M, N, K, None, None, None, 0, nat.ptr(out), nat.DOD_F32, N, 0, nat.stream_ptr()))
            torch.cuda.synchronize()
            res.append((h, kt, out[7, 0].item()))
    print(f"element {p:3d} doubled: (half, K-tile, out[7]) =", res)
