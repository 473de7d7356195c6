#!/bin/bash
# round 4, batch 1: folded LayerNorm -- op-level parity, the forward suites, then the A/B of the bench workload
# (round-3 library / this library with DINODET_LN_FOLD=0 / this library) in the three MFMA modes, same box.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4e1
mkdir -p $O
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_lnfold.py -x -q -m gpu > $O/tests_ops.log 2>&1
rc=$?
tail -6 $O/tests_ops.log
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "op tests ended with rc $rc: no further GPU step"; exit $rc; fi
DINODET_LN_FOLD=1 timeout -k 10 700 python -m pytest tests/test_gpu_forward.py tests/test_gpu_timed_shapes.py -x -q -m gpu -k "not giant and not fp32" > $O/tests_fwd.log 2>&1
rc2=$?
tail -12 $O/tests_fwd.log
if [ $rc2 -ne 0 ] && [ $rc2 -ne 1 ]; then echo "forward tests ended with rc $rc2: no further GPU step"; exit $rc2; fi
[ $rc -eq 0 ] && rc=$rc2
for prec in bf16 bf16x3 fp16x2; do
  for v in r3lib nofold fold; do
    case $v in
      r3lib) env="DINODET_LIB=$R/build/head/libdinodet_r3.so";;
      nofold) env="DINODET_LN_FOLD=0";;
      fold) env="DINODET_LN_FOLD=1";;
    esac
    env $env timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --precision $prec > $O/bench_${prec}_$v.json 2> $O/bench_${prec}_$v.err || { echo "bench $prec $v failed"; tail -5 $O/bench_${prec}_$v.err; exit 1; }
    python - <<PY
import json
d = json.loads(open("$O/bench_${prec}_$v.json").read().strip().splitlines()[-1])
r = d.get("roofline", {})
ok = {k: round(v["ms_per_step"], 2) for k, v in r.get("other_kernels", {}).items()}
print("$prec $v: %.1f img/s  %.2f ms/step  gemm-class %.2f ms  classes %s" % (d["value"], d["ms_per_step"], r.get("class_ms_per_step", -1), ok))
PY
  done
done
exit $rc
