#!/bin/bash
# round 3, batch 12: where the ViT-g fp8 step goes (rocprofv3 kernel trace, single stream)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3e12
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export DINODET_MICRO_STREAMS=1
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $O/trace -- python3 $R/bench.py --workload vitg518 --precision fp8 --steps 4 --warmup 2 --no-cpu-baseline --no-extras > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
db=$(find $O/trace -name "*.db" | head -1)
python3 $R/tools/rocprof_stats.py $db $O/vitg_fp8_kernel_stats.csv || exit 1
python3 $R/tools/rocprof_by_grid.py $db > $O/vitg_fp8_by_grid.txt 2>&1 || true
head -30 $O/vitg_fp8_by_grid.txt | cut -c1-170
