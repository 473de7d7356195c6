#!/bin/bash
# round 3, batch 13: fp8 SwiGLU epilogue that quantises (block scales) + block-scaled weights_out GEMM, wired into the forward
set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_forward.py tests/test_gpu_timed_shapes.py tests/test_gpu_fp8.py -m gpu -x -q -k "fp8 or giant" > gpurun_out/r3e13_tests.log 2>&1 || { tail -30 gpurun_out/r3e13_tests.log; exit 1; }
tail -2 gpurun_out/r3e13_tests.log
for mx in 1 0 1 0; do
  DINODET_FP8_MX_GATE=$mx timeout -k 10 300 python bench.py --workload vitg518 --precision fp8 --steps 5 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('MX_GATE=$mx vitg518 fp8', round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms')"
done
