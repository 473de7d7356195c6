#!/bin/bash
# round 4, batch 23: 256x256 / 16-wave fp8 kernel with both operands block-scaled: op parity, the ViT-g forward parity, A/B (DINODET_FP8_TILE=0/1)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4e23
mkdir -p $O
cd $R
timeout -k 10 400 python -m pytest tests/test_gpu_fp8.py -x -q -m gpu > $O/tests_ops.log 2>&1
rc=$?
tail -6 $O/tests_ops.log
if [ $rc -ne 0 ]; then echo "op tests rc $rc: no further GPU step"; exit $rc; fi
export DINODET_LIB=$R/dinov2_od_amd/lib/libdinodet_tuning.so
for v in 0 1 0 1; do
  DINODET_FP8_TILE=$v timeout -k 10 300 python bench.py --workload vitg518 --steps 8 --warmup 3 --no-cpu-baseline --no-extras --precision fp8 > $O/b_$v.json 2> $O/b_$v.err || { echo "bench failed"; tail -5 $O/b_$v.err; exit 1; }
  python - <<PY
import json
d = json.loads(open("$O/b_$v.json").read().strip().splitlines()[-1])
r = d.get("roofline", {})
print("vitg518 fp8 tile=$v: %.1f img/s  %.3f ms/step  fp8 gemm %.2f ms (%.0f TF)" % (d["value"], d["ms_per_step"], r.get("class_ms_per_step", 0), r.get("achieved", 0)))
PY
done
unset DINODET_LIB
timeout -k 10 600 python -m pytest tests/test_gpu_timed_shapes.py tests/test_gpu_forward.py -x -q -m gpu -k "fp8 or giant" > $O/tests_fwd.log 2>&1
tail -5 $O/tests_fwd.log
