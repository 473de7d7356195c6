#!/bin/bash
# round 4, batch 28: the fp8 256x256 tile for every block-scaled GEMM (1) or for all but the gated weights_in (2), two-stream step
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4e28
mkdir -p $O
cd $R
export DINODET_LIB=$R/dinov2_od_amd/lib/libdinodet_tuning.so
for v in 1 2 0 1 2; do
  DINODET_FP8_TILE=$v timeout -k 10 300 python bench.py --workload vitg518 --steps 8 --warmup 3 --no-cpu-baseline --no-extras --precision fp8 > $O/b_$v.json 2> $O/b_$v.err || { echo "bench failed"; tail -5 $O/b_$v.err; exit 1; }
  python - <<PY
import json
d = json.loads(open("$O/b_$v.json").read().strip().splitlines()[-1])
print("vitg518 fp8 tile=$v: %.1f img/s  %.3f ms/step" % (d["value"], d["ms_per_step"]))
PY
done
