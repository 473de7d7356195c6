#!/bin/bash
# round 4, batch 35: tools/probes/attn_tile_pipeline.hip -- cycles per attention tile for one wave alone / two per SIMD, stages serial or pipelined across tiles
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4e35
mkdir -p $O
cd $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -o $O/attn_tile_pipeline $R/tools/probes/attn_tile_pipeline.hip > $O/build.log 2>&1 || { tail -5 $O/build.log; exit 1; }
timeout -k 10 120 $O/attn_tile_pipeline 2000 > $O/probe.txt 2>&1 || { tail -5 $O/probe.txt; exit 1; }
timeout -k 10 120 $O/attn_tile_pipeline 2000 >> $O/probe.txt 2>&1
cat $O/probe.txt
rm -f $O/attn_tile_pipeline
