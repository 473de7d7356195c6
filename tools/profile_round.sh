#!/bin/bash
# rocprofv3 artefacts of one round for profiles/ (run on the GPU box through gpurun): kernel-trace statistics of the bench command in
# both headline modes, and the separate PMC passes (FETCH_SIZE / WRITE_SIZE / L2 hit) the roofline's `traffic` comes from.
#   bash tools/profile_round.sh r02
set -o pipefail
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# (1) the batch on ONE stream (DINODET_MICRO_STREAMS=1): per-kernel durations are separable -- this is what bench.py's roofline leg
#     measures and what its avg_launch_us / class_ms_per_step must agree with;  (2) the default command, two concurrent micro-batches:
#     the step the headline `value` times (kernels of the two streams overlap: their durations include each other)
for prec in bf16 bf16x3 fp16x2; do
  export DINODET_MICRO_STREAMS=1
  timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/trace_$prec -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --precision $prec > $O/trace_$prec.json 2> $O/trace_$prec.err || exit 1
  db=$(find $O/trace_$prec -name "*.db" | head -1)
  python3 $R/tools/rocprof_stats.py $db $O/${TAG}_bench_${prec}_kernel_stats.csv || exit 1
  unset DINODET_MICRO_STREAMS
  timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/tracem_$prec -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --precision $prec > $O/tracem_$prec.json 2> $O/tracem_$prec.err || exit 1
  db=$(find $O/tracem_$prec -name "*.db" | head -1)
  python3 $R/tools/rocprof_stats.py $db $O/${TAG}_bench_${prec}_micro2_kernel_stats.csv || exit 1
done
export DINODET_MICRO_STREAMS=1
for prec in bf16 bf16x3 fp16x2; do
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    timeout -k 10 400 rocprofv3 --pmc $set -d $O/pmc_$prec --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-extras --precision $prec > /dev/null 2>> $O/pmc_$prec.err || exit 1
  done
  python3 $R/tools/pmc_traffic.py $O/pmc_$prec $O/${TAG}_pmc_traffic_$prec.json "bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-extras --precision $prec" vitb518 $prec 64 || exit 1
  python3 $R/tools/pmc_summary.py $O/pmc_$prec > $O/${TAG}_pmc_bench_${prec}_summary.txt
done
ls -la $O/*.csv $O/*.json $O/*.txt
