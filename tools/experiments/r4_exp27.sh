#!/bin/bash
# round 4, batch 27: by-grid view of the ViT-g fp8 step on the final binary (256x256 fp8 tile), one stream
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4e27
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export DINODET_MICRO_STREAMS=1
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/trace -- python3 $R/bench.py --workload vitg518 --precision fp8 --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/vitg_fp8.json 2> $O/vitg_fp8.err || { tail -5 $O/vitg_fp8.err; exit 1; }
db=$(find $O/trace -name "*.db" | head -1)
python3 $R/tools/rocprof_by_grid.py $db > $O/r04_vitg518_fp8_by_grid.txt 2>&1 || true
head -16 $O/r04_vitg518_fp8_by_grid.txt | cut -c1-150
rm -rf $O/trace
