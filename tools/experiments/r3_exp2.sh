#!/bin/bash
# round-3 experiment batch 2 (GPU box): where configs[1] and the 8-image shard spend their step (kernels grouped by grid), PMC
# instruction mix of the split-product attention
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3e2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export DINODET_MICRO_STREAMS=1
for wl in "vitb224 0" "vitb518 8"; do
  set -- $wl
  extra=""; [ "$2" != "0" ] && extra="--batch $2"
  timeout -k 10 300 rocprofv3 --kernel-trace -d $O/trace_$1_$2 -- python3 $R/bench.py --workload $1 $extra --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/trace_$1_$2.json 2> $O/trace_$1_$2.err || exit 1
  db=$(find $O/trace_$1_$2 -name "*.db" | head -1)
  python3 $R/tools/rocprof_by_grid.py $db 45 > $O/by_grid_$1_$2.txt || exit 1
  cat $O/by_grid_$1_$2.txt
done
unset DINODET_MICRO_STREAMS
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_LDS_IDX_ACTIVE"; do
  timeout -k 10 200 rocprofv3 --pmc $set -d $O/pmc_attn_x3 --output-format csv -- python3 $R/tools/prof_x3.py > /dev/null 2>> $O/pmc_attn_x3.err || exit 1
  timeout -k 10 200 rocprofv3 --pmc $set -d $O/pmc_attn_bf16 --output-format csv -- python3 $R/tools/prof_attn.py > /dev/null 2>> $O/pmc_attn_bf16.err || exit 1
done
python3 $R/tools/pmc_summary.py $O/pmc_attn_x3 > $O/pmc_attn_x3_summary.txt
python3 $R/tools/pmc_summary.py $O/pmc_attn_bf16 > $O/pmc_attn_bf16_summary.txt
grep -A 18 "attn_" $O/pmc_attn_x3_summary.txt $O/pmc_attn_bf16_summary.txt
cd $R
python -m pytest tests/test_gpu_train_native.py tests/test_gpu_train_loop.py -q -s -k "tail_backward or reference_backward or train" > $O/tests_flash.log 2>&1 || tail -40 $O/tests_flash.log
grep -E "passed|failed|backbone tail" $O/tests_flash.log | cut -c1-220
for f in 1 0; do
  echo "DINODET_ATTN_BWD_FLASH=$f"; DINODET_ATTN_BWD_FLASH=$f python tools/bench_train_step.py 224 16 2>&1 | grep "native step"
  DINODET_ATTN_BWD_FLASH=$f python tools/bench_train_step.py 518 8 2>&1 | grep "native step"
done
