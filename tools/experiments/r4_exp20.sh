#!/bin/bash
# round 4, batch 20: full GPU suite on the round's final sources, then the round's profile artefacts again (attention kernels changed)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -q -m gpu > gpurun_out/r4_full5.log 2>&1
rc=$?
tail -6 gpurun_out/r4_full5.log
if [ $rc -ne 0 ]; then echo "suite rc $rc: no further GPU step"; exit $rc; fi
bash tools/experiments/r4_exp8.sh
