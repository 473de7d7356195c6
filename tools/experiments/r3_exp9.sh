#!/bin/bash
# round 3, batch 9: decoder operand split fused into the producers
set -o pipefail
mkdir -p gpurun_out/r3e9
timeout -k 10 900 python -m pytest tests/test_gpu_forward.py tests/test_gpu_timed_shapes.py tests/test_gpu_ops.py -m gpu -x -q > gpurun_out/r3e9/tests.log 2>&1 || { tail -40 gpurun_out/r3e9/tests.log; exit 1; }
tail -2 gpurun_out/r3e9/tests.log
for f in 1 0 1 0; do
  for wl in "vitb224" "vitb518 --batch 8" "vitb518"; do
    DINODET_DEC_FUSED_SPLIT=$f timeout -k 10 300 python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/r3e9/b.json 2> gpurun_out/r3e9/b.err || { tail -5 gpurun_out/r3e9/b.err; exit 1; }
    python - "$f" "$wl" <<'P'
import json, sys
d = json.loads(open("gpurun_out/r3e9/b.json").read().strip().splitlines()[-1])
print(f"FUSED_SPLIT={sys.argv[1]} {sys.argv[2]:18s} {d['value']:8.1f} img/s {d['ms_per_step']:.3f} ms", flush=True)
P
  done
done
