#!/bin/bash
# round 4, batch 14: the headline's micro-batches put every block GEMM just over a round boundary (172 m-tiles: fc2 516 tiles = 2 rounds + 4):
# forced tail split on / off in the two-stream step, and three micro-batches
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4e14
mkdir -p $O
cd $R
export DINODET_LIB=$R/dinov2_od_amd/lib/libdinodet_tuning.so
run() {
  local tag=$1 envs=$2; shift 2
  env $envs timeout -k 10 300 python bench.py "$@" --no-cpu-baseline --no-extras > $O/$tag.json 2> $O/$tag.err || { echo "bench $tag failed"; tail -5 $O/$tag.err; exit 1; }
  python - <<PY
import json
d = json.loads(open("$O/$tag.json").read().strip().splitlines()[-1])
r = d.get("roofline", {})
print("$tag: %.1f img/s  %.3f ms/step" % (d["value"], d["ms_per_step"]))
PY
}
for i in 1 2; do
  run bf16_default_$i "X=1" --steps 20 --warmup 5 --precision bf16
  run bf16_force_$i "DINODET_GEMM_TAILSPLIT=2" --steps 20 --warmup 5 --precision bf16
done
run bf16_3streams "DINODET_MICRO_STREAMS=3" --steps 20 --warmup 5 --precision bf16
run bf16_1stream "DINODET_MICRO_STREAMS=1" --steps 20 --warmup 5 --precision bf16
run bf16_1stream_force "DINODET_MICRO_STREAMS=1 DINODET_GEMM_TAILSPLIT=2" --steps 20 --warmup 5 --precision bf16
run fp16x2_default "X=1" --steps 10 --warmup 3 --precision fp16x2
run fp16x2_force "DINODET_GEMM_TAILSPLIT=2" --steps 10 --warmup 3 --precision fp16x2
run vitl_default "X=1" --workload vitl518 --steps 10 --warmup 3 --precision bf16
run vitl_force "DINODET_GEMM_TAILSPLIT=2" --workload vitl518 --steps 10 --warmup 3 --precision bf16
