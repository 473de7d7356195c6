#!/bin/bash
# round 3, batch 10: K-split of the half-filled N = 768 grids at the 8-image shard (10 960 rows: 129 tiles of 256x256)
set -o pipefail
export DINODET_GEMM_SCRATCH_MB=160
for k in 0 7; do
  echo "== DINODET_GEMM_KSPLIT0=$k"
  DINODET_GEMM_KSPLIT0=$k timeout -k 10 200 python tools/bench_pp.py --rows 10960 --variants default --rounds 5 || exit 1
done
for k in 0 7 0 7; do
  DINODET_GEMM_KSPLIT0=$k timeout -k 10 300 python bench.py --workload vitb518 --batch 8 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/r3e10_b.json 2> gpurun_out/r3e10_b.err || { tail -5 gpurun_out/r3e10_b.err; exit 1; }
  python - "$k" <<'P'
import json, sys
d = json.loads(open("gpurun_out/r3e10_b.json").read().strip().splitlines()[-1])
print(f"KSPLIT0={sys.argv[1]} vitb518 --batch 8 {d['value']:8.1f} img/s {d['ms_per_step']:.3f} ms", flush=True)
P
done
