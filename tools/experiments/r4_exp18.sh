#!/bin/bash
# round 4, batch 18: both flash attentions without packed fp32 (plain softmax arithmetic + -fno-slp-vectorize): parity, then A/B against the
# library before the change (build/head/libdinodet_base.so), three MFMA modes
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4e18
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_x3.py tests/test_gpu_forward.py tests/test_gpu_timed_shapes.py -x -q -m gpu -k "not giant" > $O/tests.log 2>&1
rc=$?
tail -4 $O/tests.log
if [ $rc -ne 0 ]; then echo "tests rc $rc: no further GPU step"; exit $rc; fi
for prec in bf16x3 fp16x2 bf16; do
  for v in base new base new; do
    lib=$R/dinov2_od_amd/lib/libdinodet.so; [ $v = base ] && lib=$R/build/head/libdinodet_base.so
    DINODET_LIB=$lib timeout -k 10 300 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-extras --precision $prec > $O/b_${prec}_$v.json 2> $O/b_${prec}_$v.err || { echo "bench failed"; tail -5 $O/b_${prec}_$v.err; exit 1; }
    python - <<PY
import json
d = json.loads(open("$O/b_${prec}_$v.json").read().strip().splitlines()[-1])
r = d.get("roofline", {}).get("other_kernels", {})
print("$prec $v: %.1f img/s  %.3f ms/step  attention %.2f ms/step" % (d["value"], d["ms_per_step"], r.get("attn_bf16", {}).get("ms_per_step", 0)))
PY
  done
done
