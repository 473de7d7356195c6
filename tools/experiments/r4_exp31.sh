#!/bin/bash
# round 4, batch 31: uneven micro-batches (the two streams then run kernels of different lengths and drift apart instead of meeting in the same phase)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4e31
mkdir -p $O
cd $R
for sp in 0.5 0.53125 0.5625 0.625 0.5 0.484375; do
  DINODET_MICRO_SPLIT=$sp timeout -k 10 300 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-extras --precision bf16 > $O/b_$sp.json 2> $O/b_$sp.err || { echo "bench failed"; tail -5 $O/b_$sp.err; exit 1; }
  python - <<PY
import json
d = json.loads(open("$O/b_$sp.json").read().strip().splitlines()[-1])
print("bf16 split $sp: %.1f img/s  %.3f ms/step" % (d["value"], d["ms_per_step"]))
PY
done
