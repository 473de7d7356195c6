#!/bin/bash
# round 4, batch 15: the default bench line and the small-configuration by-grid views of the round's final binary
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4e15
mkdir -p $O
cd $R
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > $O/r04_bench_1gpu.json 2> $O/r04_bench_1gpu.err || { tail -5 $O/r04_bench_1gpu.err; exit 1; }
python - <<PY
import json
d = json.loads(open("$O/r04_bench_1gpu.json").read().strip().splitlines()[-1])
print("bf16 %.1f  frac %.3f  x3 %.1f  h2 %.1f" % (d["value"], d["roofline"]["frac"], d["parity_gated_mode"]["value"], d["fp16x2_mode"]["value"]))
for k, v in d.get("also", {}).items():
    if isinstance(v, dict) and "value" in v: print(" ", k, round(v["value"], 1))
    elif isinstance(v, dict): print(" ", k, {a: (round(b, 2) if isinstance(b, float) else b) for a, b in v.items() if not isinstance(b, (dict, list))})
PY
cd /tmp && export TMPDIR=/tmp
export DINODET_MICRO_STREAMS=1
for wl in "vitb224 0" "vitb518 8"; do
  set -- $wl
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/trace -- python3 $R/bench.py --workload $1 --batch $2 --steps 20 --warmup 3 --no-cpu-baseline --no-extras --precision bf16 > $O/b_$1_$2.json 2> $O/b_$1_$2.err || { tail -5 $O/b_$1_$2.err; exit 1; }
  db=$(find $O/trace -name "*.db" | head -1)
  python3 $R/tools/rocprof_by_grid.py $db > $O/r04_$1_b$2_by_grid_final.txt 2>&1 || true
  echo "== $1 batch $2"; head -24 $O/r04_$1_b$2_by_grid_final.txt | cut -c1-150
  rm -rf $O/trace
done
