#!/bin/bash
# round 4, batch 19: ping-pong bf16 attention (8 waves, the two waves of a SIMD phased against each other): op parity, forward parity, A/B
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4e19
mkdir -p $O
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "attention_bf16" > $O/tests_ops.log 2>&1
rc=$?
tail -6 $O/tests_ops.log
if [ $rc -ne 0 ]; then echo "op tests rc $rc: no further GPU step"; exit $rc; fi
export DINODET_LIB=$R/dinov2_od_amd/lib/libdinodet_tuning.so
for wl in "vitb518 bf16" "vitg518 fp8"; do
  set -- $wl
  for v in 0 1 0 1; do
    DINODET_ATTN_PP=$v timeout -k 10 300 python bench.py --workload $1 --steps 15 --warmup 4 --no-cpu-baseline --no-extras --precision $2 > $O/b_$1_$v.json 2> $O/b_$1_$v.err || { echo "bench failed"; tail -5 $O/b_$1_$v.err; exit 1; }
    python - <<PY
import json
d = json.loads(open("$O/b_$1_$v.json").read().strip().splitlines()[-1])
r = d.get("roofline", {}).get("other_kernels", {})
print("$1 $2 pp=$v: %.1f img/s  %.3f ms/step  attention %.2f ms/step" % (d["value"], d["ms_per_step"], r.get("attn_bf16", {}).get("ms_per_step", 0)))
PY
  done
done
unset DINODET_LIB
timeout -k 10 700 python -m pytest tests/test_gpu_timed_shapes.py tests/test_gpu_forward.py -x -q -m gpu -k "not giant" > $O/tests_fwd.log 2>&1
tail -5 $O/tests_fwd.log
