#!/bin/bash
# round 4, batch 10: (1) the narrowed cut-off rule on the three bf16 workloads; (2) decoder split-3 threshold at the 8-image shard;
# (3) forced tail split (K-split of the short last round) for the compensated QKV at configs[1]; (4) where the fp8 ViT-g step goes now
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4e10
mkdir -p $O
cd $R
export DINODET_LIB=$R/dinov2_od_amd/lib/libdinodet_tuning.so
run() {   # tag, env string, bench args
  local tag=$1 envs=$2; shift 2
  env $envs timeout -k 10 300 python bench.py "$@" --no-cpu-baseline --no-extras > $O/$tag.json 2> $O/$tag.err || { echo "bench $tag failed"; tail -5 $O/$tag.err; exit 1; }
  python - <<PY
import json
d = json.loads(open("$O/$tag.json").read().strip().splitlines()[-1])
print("$tag: %.1f img/s  %.3f ms/step" % (d["value"], d["ms_per_step"]))
PY
}
run b224_default "X=1" --workload vitb224 --steps 30 --warmup 5 --precision bf16
run b224_off "DINODET_GEMM_REMCUT=0 DINODET_GEMM_RING=0" --workload vitb224 --steps 30 --warmup 5 --precision bf16
run s8_default "X=1" --workload vitb518 --batch 8 --steps 30 --warmup 5 --precision bf16
run s8_off "DINODET_GEMM_REMCUT=0 DINODET_GEMM_RING=0" --workload vitb518 --batch 8 --steps 30 --warmup 5 --precision bf16
run s8_q256 "DINODET_QSPLIT_ROWS=256" --workload vitb518 --batch 8 --steps 30 --warmup 5 --precision bf16
run s8_q256_1s "DINODET_QSPLIT_ROWS=256 DINODET_MICRO_STREAMS=1" --workload vitb518 --batch 8 --steps 30 --warmup 5 --precision bf16
run s8_1s "DINODET_MICRO_STREAMS=1" --workload vitb518 --batch 8 --steps 30 --warmup 5 --precision bf16
for p in bf16x3 fp16x2; do
  run b224_${p}_default "X=1" --workload vitb224 --steps 20 --warmup 5 --precision $p
  run b224_${p}_force "DINODET_GEMM_TAILSPLIT=2" --workload vitb224 --steps 20 --warmup 5 --precision $p
  run s8_${p}_default "X=1" --workload vitb518 --batch 8 --steps 20 --warmup 5 --precision $p
  run s8_${p}_force "DINODET_GEMM_TAILSPLIT=2" --workload vitb518 --batch 8 --steps 20 --warmup 5 --precision $p
done
unset DINODET_LIB
cd /tmp && export TMPDIR=/tmp
export DINODET_MICRO_STREAMS=1
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/trace -- python3 $R/bench.py --workload vitg518 --precision fp8 --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/vitg_fp8.json 2> $O/vitg_fp8.err || { tail -5 $O/vitg_fp8.err; exit 1; }
db=$(find $O/trace -name "*.db" | head -1)
python3 $R/tools/rocprof_by_grid.py $db > $O/r04_vitg518_fp8_by_grid.txt 2>&1 || true
head -24 $O/r04_vitg518_fp8_by_grid.txt | cut -c1-150
rm -rf $O/trace
