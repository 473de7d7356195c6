#!/bin/bash
# round 4, batch 6: where configs[1] (ViT-B 224x224 x 32) and the 8-image shard of configs[2] spend their step on the round-4 binary
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4e6
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export DINODET_MICRO_STREAMS=1
for wl in "vitb224 0" "vitb518 8"; do
  set -- $wl
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/trace -- python3 $R/bench.py --workload $1 --batch $2 --steps 20 --warmup 3 --no-cpu-baseline --no-extras --precision bf16 > $O/b_$1_$2.json 2> $O/b_$1_$2.err || { tail -5 $O/b_$1_$2.err; exit 1; }
  db=$(find $O/trace -name "*.db" | head -1)
  python3 $R/tools/rocprof_by_grid.py $db > $O/r04_$1_b$2_by_grid.txt 2>&1 || true
  echo "== $1 batch $2"; head -32 $O/r04_$1_b$2_by_grid.txt | cut -c1-160
  rm -rf $O/trace
done
