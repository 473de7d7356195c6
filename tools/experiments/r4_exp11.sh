#!/bin/bash
# round 4, batch 11: operand rows of the folded producer as lane-pair 16-byte stores -- op parity, then A/B against a build without it
# (build/head/libdinodet_nopair.so: -DDINODET_AB_NO_OP_PAIR), three MFMA modes, headline + configs[1]
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4e11
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_lnfold.py tests/test_gpu_tailsplit.py tests/test_gpu_timed_shapes.py -x -q -m gpu > $O/tests.log 2>&1
rc=$?
tail -5 $O/tests.log
if [ $rc -ne 0 ]; then echo "tests rc $rc: no further GPU step"; exit $rc; fi
for prec in bf16 bf16x3 fp16x2; do
  for wl in vitb518 vitb224; do
    for v in nopair pair nopair pair; do
      lib=$R/dinov2_od_amd/lib/libdinodet.so; [ $v = nopair ] && lib=$R/build/head/libdinodet_nopair.so
      DINODET_LIB=$lib timeout -k 10 200 python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --no-extras --precision $prec > $O/b_${wl}_${prec}_$v.json 2> $O/b_${wl}_${prec}_$v.err || { echo "bench failed"; tail -5 $O/b_${wl}_${prec}_$v.err; exit 1; }
      python - <<PY
import json
d = json.loads(open("$O/b_${wl}_${prec}_$v.json").read().strip().splitlines()[-1])
r = d.get("roofline", {})
print("$wl $prec $v: %.1f img/s  %.3f ms/step  gemm class %.2f ms" % (d["value"], d["ms_per_step"], r.get("class_ms_per_step", 0)))
PY
    done
  done
done
