#!/usr/bin/env python3
"""Three launches of the exact-fp32 MFMA GEMM (gemm_f32_kernel) at the bench shape (87 680 x 768 x 768) and of the register-only
v_mfma_f32_32x32x2_f32 probe (one dependent chain per wave, as gemm_f32_kernel issues), for `rocprofv3 --pmc`:
    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS -d out -- python3 tools/prof_f32.py
    rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d out -- python3 tools/prof_f32.py
(why every fp32-MFMA kernel levels off at 93-100 TFLOP/s against 155 for the probe.)  Without rocprofv3 it
prints the two rates."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools._tuning_lib  # noqa: F401,E402  (the -DDINODET_TUNING build: this tool uses tuning hooks)
import torch
from dinov2_od_amd import _native as nat
L = nat.lib(); dev = torch.device("cuda:0")
M, N, K = int(os.environ.get("PROF_B", "64")) * 1370, 768, 768
g = torch.Generator().manual_seed(0)
A = (torch.randn(M, K, generator=g) * 0.5).to(dev); W = (torch.randn(N, K, generator=g) * 0.05).to(dev)
bias = torch.randn(N, generator=g).to(dev); out = torch.empty(M, N, device=dev)
def gemm():
    nat.check(L.dod_op_linear(0, nat.ptr(A), K, nat.ptr(W), K, M, N, K, nat.ptr(bias), None, None, 0, nat.ptr(out), 0, N, 0, nat.stream_ptr()))
probe_out = torch.zeros(1024 * 4, dtype=torch.int64, device=dev)
def probe():
    nat.check(L.dod_debug_mfma_peak(1, 20000, 1024, nat.ptr(probe_out), nat.stream_ptr()))      # 4 waves / SIMD, one dependent chain each
for f in (gemm, probe):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3):
        f()
    b.record(); torch.cuda.synchronize()
    t = a.elapsed_time(b) / 3 * 1e-3
    fl = 2.0 * M * N * K if f is gemm else 1024 * 4 * 20000 * 2.0 * 32 * 32 * 2
    print(f"{f.__name__}: {t * 1e6:9.1f} us  {fl / t / 1e12:6.1f} TFLOP/s")
