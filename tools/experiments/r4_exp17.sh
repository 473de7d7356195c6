#!/bin/bash
# round 4, batch 17: bf16 flash attention with plain fp32 VALU only (no v_pk_*_f32: the probe of batch 16 shows packed fp32 serialises with an
# executing MFMA while plain VALU, v_exp, v_max3, v_cvt_pk overlap it) -- parity, then A/B against the shipped kernel
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4e17
mkdir -p $O
cd $R
DINODET_LIB=$R/build/head/libdinodet_attnplain.so timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "attention" > $O/tests.log 2>&1
rc=$?
tail -4 $O/tests.log
if [ $rc -ne 0 ]; then echo "tests rc $rc: no further GPU step"; exit $rc; fi
for wl in "vitb518 bf16" "vitb224 bf16" "vitg518 fp8"; do
  set -- $wl
  for v in base plain base plain; do
    lib=$R/dinov2_od_amd/lib/libdinodet.so; [ $v = plain ] && lib=$R/build/head/libdinodet_attnplain.so
    DINODET_LIB=$lib timeout -k 10 300 python bench.py --workload $1 --steps 15 --warmup 4 --no-cpu-baseline --no-extras --precision $2 > $O/b_$1_$v.json 2> $O/b_$1_$v.err || { echo "bench failed"; tail -5 $O/b_$1_$v.err; exit 1; }
    python - <<PY
import json
d = json.loads(open("$O/b_$1_$v.json").read().strip().splitlines()[-1])
r = d.get("roofline", {}).get("other_kernels", {})
print("$1 $2 $v: %.1f img/s  %.3f ms/step  attention %.2f ms/step" % (d["value"], d["ms_per_step"], r.get("attn_bf16", {}).get("ms_per_step", 0)))
PY
  done
done
