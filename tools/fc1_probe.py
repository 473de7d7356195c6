import os, sys
sys.path.insert(0, "/root/repo")
import torch
from dinov2_od_amd import _native as nat
from tools.bench_ops import timeit
L = nat.lib(); dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
m, n, k = 87680, 3072, 768
A = (torch.randn(m, k, generator=g) * 0.5).to(dev).to(torch.bfloat16); W = (torch.randn(n, k, generator=g) * 0.05).to(dev).to(torch.bfloat16)
bias = torch.randn(n, generator=g).to(dev); out = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
for act, name in ((0, "none"), (1, "relu"), (2, "gelu")):
    t = timeit(lambda: L.dod_op_linear(1, nat.ptr(A), k, nat.ptr(W), k, m, n, k, nat.ptr(bias), None, None, 0, nat.ptr(out), 1, n, act, nat.stream_ptr()))
    print(f"fc1 shape act={name}: {t*1e6:.1f} us {2.0*m*n*k/t/1e12:.1f} TF  (DINODET_GEMM_TILE={os.environ.get('DINODET_GEMM_TILE')})")
