#!/bin/bash
# round 4, batch 21: the two-workgroups-per-CU GEMM kernels (bf16 256x128, fp8 256x128) built without SLP vectorisation (their epilogues run beside
# the co-resident workgroup's MFMAs; packed fp32 does not overlap an MFMA) against the shipped build
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4e21
mkdir -p $O
cd $R
for wl in "vitg518 fp8" "vitb224 bf16" "vitb518 bf16"; do
  set -- $wl
  for v in base noslp base noslp; do
    lib=$R/dinov2_od_amd/lib/libdinodet.so; [ $v = noslp ] && lib=$R/build/head/libdinodet_noslp.so
    DINODET_LIB=$lib timeout -k 10 300 python bench.py --workload $1 --steps 15 --warmup 4 --no-cpu-baseline --no-extras --precision $2 > $O/b_$1_$v.json 2> $O/b_$1_$v.err || { echo "bench failed"; tail -5 $O/b_$1_$v.err; exit 1; }
    python - <<PY
import json
d = json.loads(open("$O/b_$1_$v.json").read().strip().splitlines()[-1])
r = d.get("roofline", {})
print("$1 $2 $v: %.1f img/s  %.3f ms/step  dominant class %.2f ms" % (d["value"], d["ms_per_step"], r.get("class_ms_per_step", 0)))
PY
  done
done
