#!/bin/bash
# round 4, batch 32: fp8 256x256 tile with 128-byte LDS rows (two K-tiles per barrier, two 64-KiB slots) against the 64-byte-row / 3-slot form
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4e32
mkdir -p $O
cd $R
export DINODET_LIB=$R/dinov2_od_amd/lib/libdinodet_tuning.so
DINODET_FP8_TILE=3 timeout -k 10 400 python -m pytest tests/test_gpu_fp8.py -x -q -m gpu > $O/tests_ops.log 2>&1
rc=$?
tail -4 $O/tests_ops.log
if [ $rc -ne 0 ]; then echo "op tests rc $rc: no further GPU step"; exit $rc; fi
for v in 1 3 1 3; do
  DINODET_FP8_TILE=$v timeout -k 10 300 python bench.py --workload vitg518 --steps 8 --warmup 3 --no-cpu-baseline --no-extras --precision fp8 > $O/b_$v.json 2> $O/b_$v.err || { echo "bench failed"; tail -5 $O/b_$v.err; exit 1; }
  python - <<PY
import json
d = json.loads(open("$O/b_$v.json").read().strip().splitlines()[-1])
r = d.get("roofline", {})
print("vitg518 fp8 tile=$v: %.1f img/s  %.3f ms/step  fp8 gemm %.2f ms (%.0f TF)" % (d["value"], d["ms_per_step"], r.get("class_ms_per_step", 0), r.get("achieved", 0)))
PY
done
DINODET_FP8_TILE=3 timeout -k 10 500 python -m pytest tests/test_gpu_timed_shapes.py tests/test_gpu_forward.py -x -q -m gpu -k "fp8" > $O/tests_fwd.log 2>&1
tail -3 $O/tests_fwd.log
