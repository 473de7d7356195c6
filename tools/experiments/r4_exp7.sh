#!/bin/bash
# round 4, batch 7: decoder layer-0 constants at pack time + statistics finished in the consumer's epilogue -- parity, then the default bench line
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4e7
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_lnfold.py tests/test_gpu_timed_shapes.py tests/test_gpu_forward.py tests/test_gpu_train_loop.py -q -m gpu -k "not fp32 or strict" > $O/tests.log 2>&1
rc=$?
tail -6 $O/tests.log | cut -c1-220
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "tests ended with rc $rc: no further GPU step"; exit $rc; fi
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python - <<PY
import json
d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("value", round(d["value"], 1), "ms", round(d["ms_per_step"], 3), "frac", round(d["roofline"]["frac"], 4), "e2e", round(d.get("mfma_roofline_frac_end_to_end"), 4))
print("gated", round(d["parity_gated_mode"]["value"], 1), d["parity_gated_mode"].get("gpu_vs_oracle", {}).get("pred_logits_max_rel"), "fp16x2", round(d["fp16x2_mode"]["value"], 1), d["fp16x2_mode"].get("gpu_vs_oracle", {}).get("pred_logits_max_rel"))
for k, v in d["also"].items():
    print(k, round(v["value"], 1) if "value" in v else v)
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["gpu_vs_oracle"]["pred_logits_max_rel"])
PY
exit $rc
