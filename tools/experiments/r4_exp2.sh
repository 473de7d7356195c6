#!/bin/bash
# round 4, batch 2: folded LayerNorm on the other configurations (small grids: 2 workgroups per CU overlap an epilogue with a K loop) and the
# per-kernel evidence for the headline (rocprofv3 kernel trace, one stream, fold on / off)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4e2
mkdir -p $O
cd $R
for wl in "vitb224 0" "vitb518 8" "vitl518 0"; do
  set -- $wl
  for f in 0 1; do
    DINODET_LN_FOLD=$f timeout -k 10 200 python bench.py --workload $1 --batch $2 --steps 30 --warmup 8 --no-cpu-baseline --no-extras --precision bf16 > $O/b_$1_$2_$f.json 2> $O/b_$1_$2_$f.err || { echo "bench $wl $f failed"; tail -5 $O/b_$1_$2_$f.err; exit 1; }
    python - <<PY
import json
d = json.loads(open("$O/b_$1_$2_$f.json").read().strip().splitlines()[-1])
r = d.get("roofline", {})
ok = {k: round(v["ms_per_step"], 3) for k, v in r.get("other_kernels", {}).items()}
print("$1 batch $2 fold=$f: %.1f img/s  %.3f ms/step  classes %s" % (d["value"], d["ms_per_step"], ok))
PY
  done
done
cd /tmp && export TMPDIR=/tmp
for f in 0 1; do
  DINODET_MICRO_STREAMS=1 DINODET_LN_FOLD=$f timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/trace$f -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --precision bf16 > $O/trace$f.json 2> $O/trace$f.err || { tail -5 $O/trace$f.err; exit 1; }
  db=$(find $O/trace$f -name "*.db" | head -1)
  python3 $R/tools/rocprof_by_grid.py $db > $O/headline_bf16_fold${f}_by_grid.txt 2>&1 || true
  head -14 $O/headline_bf16_fold${f}_by_grid.txt | cut -c1-175
  rm -rf $O/trace$f
done
