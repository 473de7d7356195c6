#!/bin/bash
# round 4, batch 22: repeat the tests that depend on the cross-workgroup K split of the fp32 GEMM (agent-scope atomics, self-resetting counters, slabs
# shared by launches of one stream, two concurrent micro-batch streams) -- one pytest process at a time, five times over
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4e22
mkdir -p $O
cd $R
for i in 1 2 3 4 5; do
  timeout -k 10 400 python -m pytest tests/test_gpu_ops.py tests/test_gpu_forward.py -q -m gpu -x -k "k_split or micro_batches or fused_operand_split or decoder or hipgraph or strict_batch" > $O/run$i.log 2>&1
  rc=$?
  tail -2 $O/run$i.log
  if [ $rc -ne 0 ]; then echo "run $i rc $rc: stop"; exit $rc; fi
done
