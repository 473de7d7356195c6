#!/bin/bash
# round 4, batch 13: slice target of the fp32 K split (workgroups aimed at for 16 m-tiles), tuning library
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4e13
mkdir -p $O
cd $R
export DINODET_LIB=$R/dinov2_od_amd/lib/libdinodet_tuning.so
for wl in "vitb518 8" "vitb224 0"; do
  set -- $wl
  b=""; [ "$2" != "0" ] && b="--batch $2"
  for v in 512 768 1024 1536 512 1024; do
    DINODET_F32_KSPLIT=$v timeout -k 10 200 python bench.py --workload $1 $b --steps 30 --warmup 5 --no-cpu-baseline --no-extras --precision bf16 > $O/b_$1_$2_$v.json 2> $O/b_$1_$2_$v.err || { echo "bench failed"; tail -5 $O/b_$1_$2_$v.err; exit 1; }
    python - <<PY
import json
d = json.loads(open("$O/b_$1_$2_$v.json").read().strip().splitlines()[-1])
print("$1 batch $2 f32 ksplit target $v: %.1f img/s  %.3f ms/step" % (d["value"], d["ms_per_step"]))
PY
  done
done
