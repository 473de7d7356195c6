#!/bin/bash
# round 3, batch 7: packed GELU epilogue
set -o pipefail
mkdir -p gpurun_out/r3e7
timeout -k 10 700 python -m pytest tests/test_gpu_ops.py tests/test_gpu_forward.py tests/test_gpu_timed_shapes.py -m gpu -x -q > gpurun_out/r3e7/tests.log 2>&1 || { tail -30 gpurun_out/r3e7/tests.log; exit 1; }
tail -2 gpurun_out/r3e7/tests.log
for b in 64 32; do timeout -k 10 200 python tools/bench_pp.py --batch $b --variants "default" --rounds 5 || exit 1; done
for p in bf16 bf16x3 fp16x2 bf16; do
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --precision $p > gpurun_out/r3e7/b.json 2> gpurun_out/r3e7/b.err || { tail -5 gpurun_out/r3e7/b.err; exit 1; }
  python - "$p" <<'P'
import json, sys
d = json.loads(open("gpurun_out/r3e7/b.json").read().strip().splitlines()[-1])
print(f"{sys.argv[1]:8s} {d['value']:8.1f} img/s {d['ms_per_step']:.3f} ms  gemm class {d['roofline'].get('class_ms_per_step')}", flush=True)
P
done
