#!/bin/bash
# round 4, batch 12: fp32 GEMM K split across workgroups (decoder small linears) -- parity, then A/B (tuning library, DINODET_F32_KSPLIT=0/1)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4e12
mkdir -p $O
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu > $O/tests_ops.log 2>&1
rc=$?
tail -5 $O/tests_ops.log
if [ $rc -ne 0 ]; then echo "op tests rc $rc: no further GPU step"; exit $rc; fi
timeout -k 10 700 python -m pytest tests/test_gpu_forward.py tests/test_gpu_timed_shapes.py -x -q -m gpu -k "not giant" > $O/tests_fwd.log 2>&1
rc=$?
tail -5 $O/tests_fwd.log
if [ $rc -ne 0 ]; then echo "forward tests rc $rc: no further GPU step"; exit $rc; fi
export DINODET_LIB=$R/dinov2_od_amd/lib/libdinodet_tuning.so
for wl in "vitb518 8" "vitb224 0" "vitb518 0"; do
  set -- $wl
  b=""; [ "$2" != "0" ] && b="--batch $2"
  for v in 0 1 0 1; do
    DINODET_F32_KSPLIT=$v timeout -k 10 200 python bench.py --workload $1 $b --steps 30 --warmup 5 --no-cpu-baseline --no-extras --precision bf16 > $O/b_$1_$2_$v.json 2> $O/b_$1_$2_$v.err || { echo "bench failed"; tail -5 $O/b_$1_$2_$v.err; exit 1; }
    python - <<PY
import json
d = json.loads(open("$O/b_$1_$2_$v.json").read().strip().splitlines()[-1])
print("$1 batch $2 f32 ksplit $v: %.1f img/s  %.3f ms/step" % (d["value"], d["ms_per_step"]))
PY
  done
done
