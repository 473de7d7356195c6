#!/bin/bash
# round-3 experiment batch 3 (GPU box): flash adjoint accuracy (row max / sum kept separate), K-split of the decoder's tiny grids
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3e3
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_train_native.py tests/test_gpu_train_loop.py -q -s -k "tail_backward or reference_backward or train" > $O/tests_flash.log 2>&1 || tail -40 $O/tests_flash.log
grep -E "passed|failed|backbone tail" $O/tests_flash.log | cut -c1-220
python -m pytest tests/test_gpu_timed_shapes.py tests/test_gpu_tailsplit.py tests/test_gpu_forward.py -q -k "timed_vitb518 or tail or decoder or split3 or cfg1 or strict_vitb" > $O/tests_ksplit.log 2>&1 || tail -40 $O/tests_ksplit.log
tail -3 $O/tests_ksplit.log
for k0 in 0 4; do
  for wl in vitb224 vitb518; do
    DINODET_GEMM_KSPLIT0=$k0 python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_${wl}_k$k0.json 2> $O/bench_${wl}_k$k0.err || exit 1
    python - <<P
import json
d=json.load(open("$O/bench_${wl}_k$k0.json")); r=d["roofline"]
print("$wl KSPLIT0=$k0", round(d["value"],1), "img/s", round(d["ms_per_step"],3), "ms; class", round(r["class_ms_per_step"],3), "frac", round(r["frac"],4))
P
  done
done
for f in 1 0; do
  echo "DINODET_ATTN_BWD_FLASH=$f"; DINODET_ATTN_BWD_FLASH=$f python tools/bench_train_step.py 518 8 2>&1 | grep "native step"
done
