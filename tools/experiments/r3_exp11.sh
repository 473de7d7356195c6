#!/bin/bash
# round 3, batch 11: tail split of a short third round (fc1 at the 8-image shard: 516 tiles = 2 rounds + 4)
set -o pipefail
for k in 0 1; do
  echo "== DINODET_GEMM_TAILSPLIT_SHORT=$k"
  DINODET_GEMM_TAILSPLIT_SHORT=$k timeout -k 10 200 python tools/bench_pp.py --rows 10960 --variants default --rounds 5 || exit 1
done
for k in 0 1 0 1; do
  DINODET_GEMM_TAILSPLIT_SHORT=$k timeout -k 10 300 python bench.py --workload vitb518 --batch 8 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/r3e11_b.json 2> gpurun_out/r3e11_b.err || { tail -5 gpurun_out/r3e11_b.err; exit 1; }
  python - "$k" <<'P'
import json, sys
d = json.loads(open("gpurun_out/r3e11_b.json").read().strip().splitlines()[-1])
print(f"TAILSPLIT_SHORT={sys.argv[1]} vitb518 --batch 8 {d['value']:8.1f} img/s {d['ms_per_step']:.3f} ms", flush=True)
P
done
