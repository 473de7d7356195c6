#!/bin/bash
# round 3, batch 8: packed row-sum / split in the attention kernels
set -o pipefail
mkdir -p gpurun_out/r3e8
timeout -k 10 700 python -m pytest tests -m gpu -x -q -k "attn or attention or forward or timed" > gpurun_out/r3e8/tests.log 2>&1 || { tail -30 gpurun_out/r3e8/tests.log; exit 1; }
tail -2 gpurun_out/r3e8/tests.log
for p in bf16 bf16x3 fp16x2 bf16 bf16x3; do
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --precision $p > gpurun_out/r3e8/b.json 2> gpurun_out/r3e8/b.err || { tail -5 gpurun_out/r3e8/b.err; exit 1; }
  python - "$p" <<'P'
import json, sys
d = json.loads(open("gpurun_out/r3e8/b.json").read().strip().splitlines()[-1])
o = d['roofline'].get('other_kernels', {})
print(f"{sys.argv[1]:8s} {d['value']:8.1f} img/s {d['ms_per_step']:.3f} ms  gemm class {d['roofline'].get('class_ms_per_step'):.2f}  attn", {k: round(v['ms_per_step'], 2) for k, v in o.items() if 'attn' in k}, flush=True)
P
done
