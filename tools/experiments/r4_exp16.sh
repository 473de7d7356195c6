#!/bin/bash
# round 4, batch 16: does a gfx950 SIMD overlap VALU / transcendental work with an executing MFMA (other wave / same wave)?  tools/probes/mfma_valu_overlap.hip
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4e16
mkdir -p $O
cd $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $O/mfma_valu_overlap $R/tools/probes/mfma_valu_overlap.hip > $O/build.log 2>&1 || { tail -5 $O/build.log; exit 1; }
timeout -k 10 120 $O/mfma_valu_overlap 4000 > $O/probe.txt 2>&1 || { tail -5 $O/probe.txt; exit 1; }
cat $O/probe.txt
timeout -k 10 120 $O/mfma_valu_overlap 4000 >> $O/probe.txt 2>&1
tail -8 $O/probe.txt
rm -f $O/mfma_valu_overlap
