#!/bin/bash
# round 3, batch 4: small-grid tile rule (128x128 kernel for grids of a few dozen 256x128 tiles)
set -o pipefail
mkdir -p gpurun_out/r3e4
MS=4800,6400 timeout -k 10 200 python tools/bench_small_m.py > gpurun_out/r3e4/smallm.log 2>&1 || exit 1
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_forward.py -m gpu -x -q > gpurun_out/r3e4/tests.log 2>&1 || { tail -30 gpurun_out/r3e4/tests.log; exit 1; }
for sg in 1 0 1 0; do
  for wl in "vitb224" "vitb518 --batch 8" "vitb518"; do
    DINODET_GEMM_SMALLGRID=$sg timeout -k 10 300 python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/r3e4/b.json 2> gpurun_out/r3e4/b.err || { tail -5 gpurun_out/r3e4/b.err; exit 1; }
    python - "$sg" "$wl" <<'P'
import json, sys
d = json.loads(open("gpurun_out/r3e4/b.json").read().strip().splitlines()[-1])
print(f"SMALLGRID={sys.argv[1]} {sys.argv[2]:18s} {d['value']:8.1f} img/s {d['ms_per_step']:.3f} ms", flush=True)
P
  done
done
