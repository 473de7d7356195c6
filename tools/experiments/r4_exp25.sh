#!/bin/bash
# round 4, batch 25: the fp8 256x256 tile with one / two / three micro-batch streams (is its gain the interleaving of one-workgroup-per-CU grids?)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4e25
mkdir -p $O
cd $R
export DINODET_LIB=$R/dinov2_od_amd/lib/libdinodet_tuning.so
for st in 1 2 3; do
  for v in 0 1; do
    DINODET_MICRO_STREAMS=$st DINODET_FP8_TILE=$v timeout -k 10 300 python bench.py --workload vitg518 --steps 8 --warmup 3 --no-cpu-baseline --no-extras --precision fp8 > $O/b_${st}_$v.json 2> $O/b_${st}_$v.err || { echo "bench failed"; tail -5 $O/b_${st}_$v.err; exit 1; }
    python - <<PY
import json
d = json.loads(open("$O/b_${st}_$v.json").read().strip().splitlines()[-1])
print("vitg518 fp8 streams=$st tile=$v: %.1f img/s  %.3f ms/step" % (d["value"], d["ms_per_step"]))
PY
  done
done
