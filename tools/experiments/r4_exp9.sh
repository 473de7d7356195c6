#!/bin/bash
# round 4, batch 9: the cut-off last round (DINODET_GEMM_REMCUT) and the 3-slot ring of the 128x128 kernel (DINODET_GEMM_RING) --
# op parity, then A/B on the small configurations and the headline (tuning library so that both switches can be turned off)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4e9
mkdir -p $O
cd $R
timeout -k 10 500 python -m pytest tests/test_gpu_lnfold.py tests/test_gpu_ops.py tests/test_gpu_timed_shapes.py -x -q -m gpu > $O/tests.log 2>&1
rc=$?
tail -5 $O/tests.log
if [ $rc -ne 0 ]; then echo "tests rc $rc: no further GPU step"; exit $rc; fi
export DINODET_LIB=$R/dinov2_od_amd/lib/libdinodet_tuning.so
for wl in "vitb224 0" "vitb518 8" "vitb518 0"; do
  set -- $wl
  for v in "0 0" "6 0" "0 1" "6 1"; do
    set -- $wl $v
    b=""; [ "$2" != "0" ] && b="--batch $2"
    DINODET_GEMM_REMCUT=$3 DINODET_GEMM_RING=$4 timeout -k 10 200 python bench.py --workload $1 $b --steps 30 --warmup 5 --no-cpu-baseline --no-extras --precision bf16 > $O/b_$1_$2_$3_$4.json 2> $O/b_$1_$2_$3_$4.err || { echo "bench $wl $v failed"; tail -5 $O/b_$1_$2_$3_$4.err; exit 1; }
    python - <<PY
import json
d = json.loads(open("$O/b_$1_$2_$3_$4.json").read().strip().splitlines()[-1])
print("$1 batch $2 remcut $3 ring $4: %.1f img/s  %.3f ms/step" % (d["value"], d["ms_per_step"]))
PY
  done
done
