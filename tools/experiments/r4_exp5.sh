#!/bin/bash
# round 4, batch 5: epilogue passes written out (16-wave GEMM, fp8 mx GEMM: no spills), fp8 block-scaled schedule -- parity, then the bench
# legs against the round-3 library on the same box
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4e5
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_lnfold.py tests/test_gpu_fp8.py tests/test_gpu_x3.py tests/test_gpu_ops.py tests/test_gpu_timed_shapes.py tests/test_gpu_forward.py -q -m gpu -rP -k "not fp32" > $O/tests.log 2>&1
rc=$?
tail -4 $O/tests.log | cut -c1-220
grep -E "fp8 vs the reference|timed config B=64|giant B=32" $O/tests.log | cut -c1-250
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "tests ended with rc $rc: no further GPU step"; exit $rc; fi
for prec in bf16 bf16x3 fp16x2; do
  for v in r3lib new; do
    env=""; [ $v = r3lib ] && env="DINODET_LIB=$R/build/head/libdinodet_r3.so"
    env $env timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --precision $prec > $O/bench_${prec}_$v.json 2> $O/bench_${prec}_$v.err || { echo "bench $prec $v failed"; tail -5 $O/bench_${prec}_$v.err; exit 1; }
    python - <<PY
import json
d = json.loads(open("$O/bench_${prec}_$v.json").read().strip().splitlines()[-1])
r = d.get("roofline", {})
ok = {k: round(v["ms_per_step"], 2) for k, v in r.get("other_kernels", {}).items()}
print("$prec $v: %.1f img/s  %.2f ms/step  classes %s" % (d["value"], d["ms_per_step"], ok))
PY
  done
done
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
