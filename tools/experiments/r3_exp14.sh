#!/bin/bash
# round 3, batch 14: where the ViT-L bf16 step goes (rocprofv3 kernel trace, single stream)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3e14
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export DINODET_MICRO_STREAMS=1
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $O/trace -- python3 $R/bench.py --workload vitl518 --precision bf16 --steps 6 --warmup 2 --no-cpu-baseline --no-extras > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
db=$(find $O/trace -name "*.db" | head -1)
python3 $R/tools/rocprof_by_grid.py $db > $O/vitl_bf16_by_grid.txt 2>&1 || true
head -16 $O/vitl_bf16_by_grid.txt | cut -c1-150
