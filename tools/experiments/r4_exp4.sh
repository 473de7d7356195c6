#!/bin/bash
# round 4, batch 4: fp8 mode block-scaled on both operands -- operator / forward tests, then the ViT-g fp8 bench against the round-3 library
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4e4
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_fp8.py tests/test_gpu_train_native.py tests/test_gpu_forward.py tests/test_gpu_timed_shapes.py -q -m gpu -k "fp8 or g9_grad or deterministic or detector_train_step or giant" > $O/tests.log 2>&1
rc=$?
tail -25 $O/tests.log | cut -c1-220
grep -E "fp8 vs the reference|median distance|deterministic vs fast" $O/tests.log | cut -c1-250
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "tests ended with rc $rc: no further GPU step"; exit $rc; fi
for v in r3lib new; do
  env=""; [ $v = r3lib ] && env="DINODET_LIB=$R/build/head/libdinodet_r3.so"
  env $env timeout -k 10 300 python bench.py --workload vitg518 --precision fp8 --steps 8 --warmup 3 --no-cpu-baseline --no-extras > $O/vitg_$v.json 2> $O/vitg_$v.err || { echo "bench $v failed"; tail -5 $O/vitg_$v.err; exit 1; }
  python - <<PY
import json
d = json.loads(open("$O/vitg_$v.json").read().strip().splitlines()[-1])
r = d.get("roofline", {})
ok = {k: round(v["ms_per_step"], 2) for k, v in r.get("other_kernels", {}).items()}
print("vitg518 fp8 $v: %.1f img/s  %.2f ms/step  classes %s" % (d["value"], d["ms_per_step"], ok))
PY
done
exit $rc
