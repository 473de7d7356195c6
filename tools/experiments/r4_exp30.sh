#!/bin/bash
# round 4, batch 30: bf16 attention with a thresholded lazy rescale (plain-VALU form: 64 accumulator multiplies per tile = 13 % of the VALU pipe's cycles)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4e30
mkdir -p $O
cd $R
DINODET_LIB=$R/build/head/libdinodet_lazy.so timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "attention_bf16" > $O/tests.log 2>&1
rc=$?
tail -4 $O/tests.log
if [ $rc -ne 0 ]; then echo "tests rc $rc: no further GPU step"; exit $rc; fi
for v in base lazy base lazy; do
  DINODET_LIB=$R/build/head/libdinodet_$v.so timeout -k 10 300 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-extras --precision bf16 > $O/b_$v.json 2> $O/b_$v.err || { echo "bench failed"; tail -5 $O/b_$v.err; exit 1; }
  python - <<PY
import json
d = json.loads(open("$O/b_$v.json").read().strip().splitlines()[-1])
r = d.get("roofline", {}).get("other_kernels", {})
g = d["cpu_baseline"]["gpu_vs_oracle"] if "cpu_baseline" in d and d["cpu_baseline"] else {}
print("bf16 $v: %.1f img/s  %.3f ms/step  attention %.2f ms/step" % (d["value"], d["ms_per_step"], r.get("attn_bf16", {}).get("ms_per_step", 0)))
PY
done
