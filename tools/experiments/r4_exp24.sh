#!/bin/bash
# round 4, batch 24: does the one-workgroup-per-CU interleaving that paid for fp8 also pay for the bf16 out-proj (256x128, two per CU, in the
# two-stream headline)?  DINODET_GEMM_M16_BIAS=100 sends every M >= 4096 GEMM to a 256x256 tile
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4e24
mkdir -p $O
cd $R
export DINODET_LIB=$R/dinov2_od_amd/lib/libdinodet_tuning.so
for wl in "vitb518 0" "vitb224 0" "vitb518 8" "vitl518 0"; do
  set -- $wl
  b=""; [ "$2" != "0" ] && b="--batch $2"
  for v in 1.02 100 1.02 100; do
    DINODET_GEMM_M16_BIAS=$v timeout -k 10 300 python bench.py --workload $1 $b --steps 15 --warmup 4 --no-cpu-baseline --no-extras --precision bf16 > $O/b_$1_$2_$v.json 2> $O/b_$1_$2_$v.err || { echo "bench failed"; tail -5 $O/b_$1_$2_$v.err; exit 1; }
    python - <<PY
import json
d = json.loads(open("$O/b_$1_$2_$v.json").read().strip().splitlines()[-1])
print("$1 batch $2 m16 bias $v: %.1f img/s  %.3f ms/step" % (d["value"], d["ms_per_step"]))
PY
  done
done
