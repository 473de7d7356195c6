#!/bin/bash
# round 4, batch 3: the folded-LayerNorm producer on the 8-wave ping-pong kernel instead of the 16-wave k64 kernel (tuning build, DINODET_GEMM_TILE=q)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4e3
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export DINODET_LIB=$R/dinov2_od_amd/lib/libdinodet_tuning.so
for t in default q; do
for f in 0 1; do
  tile=""; [ $t = q ] && tile="DINODET_GEMM_TILE=q"
  env $tile DINODET_MICRO_STREAMS=1 DINODET_LN_FOLD=$f timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/trace -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --precision bf16 > $O/t_${t}_$f.json 2> $O/t_${t}_$f.err || { tail -5 $O/t_${t}_$f.err; exit 1; }
  db=$(find $O/trace -name "*.db" | head -1)
  python3 $R/tools/rocprof_by_grid.py $db > $O/bf16_tile_${t}_fold${f}_by_grid.txt 2>&1 || true
  echo "== tile $t fold $f"; head -8 $O/bf16_tile_${t}_fold${f}_by_grid.txt | cut -c1-150
  rm -rf $O/trace
done
done
