#!/bin/bash
# round 4, batch 33: release library after the last kernel change (fp8 tile with 128-byte rows): fp8 tests, the ViT-g forward tests, the default bench line
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_fp8.py tests/test_gpu_timed_shapes.py tests/test_gpu_forward.py -x -q -m gpu -k "fp8 or giant" > gpurun_out/r4_fp8_final.log 2>&1
rc=$?
tail -3 gpurun_out/r4_fp8_final.log
if [ $rc -ne 0 ]; then echo "tests rc $rc: no further GPU step"; exit $rc; fi
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_1gpu.json 2> gpurun_out/r04_bench_1gpu.err || { tail -5 gpurun_out/r04_bench_1gpu.err; exit 1; }
python - <<PY
import json
d = json.loads(open("gpurun_out/r04_bench_1gpu.json").read().strip().splitlines()[-1])
print("bf16 %.1f  frac %.3f  x3 %.1f  h2 %.1f" % (d["value"], d["roofline"]["frac"], d["parity_gated_mode"]["value"], d["fp16x2_mode"]["value"]))
for k, v in d.get("also", {}).items():
    if isinstance(v, dict) and "value" in v: print(" ", k, round(v["value"], 1))
PY
