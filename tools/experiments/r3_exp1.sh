#!/bin/bash
# round-3 experiment batch 1 (GPU box): new tests, residual-prefetch A/B, patch-embed timing, fp32-MFMA PMC passes
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3e1
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_train_native.py -k "reference_backward" -q -s > $O/tests.log 2>&1 || tail -30 $O/tests.log
tail -3 $O/tests.log
python tools/bench_pp.py --batch 64 --variants "default;q;q#8" --rounds 3 > $O/pp_rb.log 2>&1 || exit 1
cat $O/pp_rb.log
for rb in 0 8; do
  for prec in fp16x2 bf16x3; do
    DINODET_EPI_RB=$rb python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --precision $prec > $O/bench_${prec}_rb$rb.json 2> $O/bench_${prec}_rb$rb.err || exit 1
    python - <<P
import json
d=json.load(open("$O/bench_${prec}_rb$rb.json")); r=d["roofline"]
print("$prec rb=$rb", round(d["value"],1), "img/s; class", round(r["class_ms_per_step"],2), {k:round(v["ms_per_step"],2) for k,v in r["other_kernels"].items()})
P
  done
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/trace_bf16 -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/trace_bf16.json 2> $O/trace_bf16.err || exit 1
db=$(find $O/trace_bf16 -name "*.db" | head -1)
python3 $R/tools/rocprof_stats.py $db $O/trace_bf16_kernel_stats.csv || exit 1
grep -i "patch_embed\|layernorm" $O/trace_bf16_kernel_stats.csv
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU"; do
  timeout -k 10 200 rocprofv3 --pmc $set -d $O/pmc_f32 --output-format csv -- python3 $R/tools/prof_f32.py > $O/pmc_f32.log 2>> $O/pmc_f32.err || exit 1
done
python3 $R/tools/pmc_summary.py $O/pmc_f32 > $O/pmc_f32_summary.txt
cat $O/pmc_f32.log; cat $O/pmc_f32_summary.txt
