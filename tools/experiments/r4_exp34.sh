#!/bin/bash
# round 4, batch 34: every bf16 block GEMM on the 16-wave kernel (fc2 leaves the 8-wave ping-pong kernel + tail split) in the two-stream headline
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4e34
mkdir -p $O
cd $R
export DINODET_LIB=$R/dinov2_od_amd/lib/libdinodet_tuning.so
for v in default x default x q; do
  if [ $v = default ]; then unset DINODET_GEMM_TILE; else export DINODET_GEMM_TILE=$v; fi
  timeout -k 10 300 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-extras --precision bf16 > $O/b_$v.json 2> $O/b_$v.err || { echo "bench failed"; tail -5 $O/b_$v.err; exit 1; }
  python - <<PY
import json
d = json.loads(open("$O/b_$v.json").read().strip().splitlines()[-1])
print("bf16 tile=$v: %.1f img/s  %.3f ms/step  gemm class %.2f" % (d["value"], d["ms_per_step"], d["roofline"]["class_ms_per_step"]))
PY
done
