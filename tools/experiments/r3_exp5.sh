#!/bin/bash
# round 3, batch 5: weight-resident tile map / group depth A/B on the block GEMM shapes
set -o pipefail
mkdir -p gpurun_out/r3e5
for b in 64 32; do
  echo "== batch $b warm"; timeout -k 10 200 python tools/bench_pp.py --batch $b --variants "default,w,g1,g2,g8" --rounds 5 || exit 1
  echo "== batch $b producer"; timeout -k 10 200 python tools/bench_pp.py --batch $b --variants "default,w,g1,g8" --rounds 3 --producer || exit 1
done
