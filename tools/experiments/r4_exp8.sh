#!/bin/bash
# round 4, batch 8: the round's profile artefacts on one box -- default bench line, rocprof/PMC passes, by-grid views of the small configs
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_1gpu.json 2> gpurun_out/r04_bench_1gpu.err || { tail -5 gpurun_out/r04_bench_1gpu.err; exit 1; }
tail -c 600 gpurun_out/r04_bench_1gpu.json; echo
bash tools/profile_round.sh r04 || exit 1
bash tools/experiments/r4_exp6.sh || exit 1
