#!/bin/bash
# round 3, batch 6: does the weight-resident tile map lower the L2-miss traffic of fc1 (and does the time follow)?
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3e6
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export DINODET_MICRO_STREAMS=1
for w in 0 1; do
  export DINODET_GEMM_WRES=$w
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    timeout -k 10 400 rocprofv3 --pmc $set -d $O/pmc_w$w --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-extras --precision bf16 > /dev/null 2>> $O/pmc_w$w.err || exit 1
  done
  python3 $R/tools/pmc_traffic.py $O/pmc_w$w $O/traffic_w$w.json "DINODET_GEMM_WRES=$w bench.py --steps 2 --warmup 1 --no-graph" vitb518 bf16 64 || exit 1
  python3 - $O/traffic_w$w.json <<'P'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d["per_kernel"].items():
    if "gemm_ppm" in k: print(k, round(v["hbm_bytes_per_launch"] / 1e6, 1), "MB  L2 hit", round(v.get("l2_hit_rate", 0), 3), "launches", v["launches_profiled"])
P
done
