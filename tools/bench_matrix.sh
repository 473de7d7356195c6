#!/bin/bash
# every BASELINE workload x precision mode on the current binary (one line each: images/s, ms/step, dominant-class TFLOP/s)
OUT=${1:-gpurun_out/matrix.log}
: > $OUT
for w in vitb518 vitb224 vits224 vitl518 vitg518; do
  for p in bf16 bf16x3 fp8; do
    timeout -k 10 400 python bench.py --workload $w --precision $p --no-cpu-baseline --no-extras --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']
        print('$w $p', round(j['value'],1), 'img/s', round(j['ms_per_step'],2), 'ms', 'e2e', round(j['mfma_roofline_frac_end_to_end'],4), 'dominant', round(r['achieved'],1), 'TF', {k:round(v['ms_per_step'],2) for k,v in r['other_kernels'].items()})" >> $OUT
    echo "$w $p done"
  done
done
cat $OUT
