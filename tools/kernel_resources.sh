#!/bin/bash
# Per-kernel register / scratch usage of one csrc file (hipcc -Rpass-analysis=kernel-resource-usage), one line per kernel.
# usage: tools/kernel_resources.sh gemm_pp.hip [srcdir]
cd "$(dirname "$0")/.."
d=${2:-dinov2_od_amd/csrc}
# per-source flags of the real build (dinov2_od_amd/_build.py EXTRA: the attention kernels are built without SLP vectorisation)
extra=$(python3 -c "import sys; sys.path.insert(0, '.'); from dinov2_od_amd import _build as b; print(' '.join(b.EXTRA.get('$1', [])))" 2>/dev/null)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 $extra -Iinclude -I$d -c "$d/$1" -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 | \
  awk '/remark: Function Name:/ {name=$(NF-1)} /remark:     VGPRs:/ {v=$(NF-1)} /AGPRs:/ {a=$(NF-1)} /ScratchSize/ {sc=$(NF-1)} /SGPRs Spill/ {ss=$(NF-1)} /VGPRs Spill/ {vs=$(NF-1)} /Occupancy/ {oc=$(NF-1)} /LDS Size/ {printf "vgpr %3d agpr %3d scratch %4d sgpr_spill %3d vgpr_spill %3d occ %d  %s\n", v, a, sc, ss, vs, oc, name}' | sort -u -k13 | c++filt | sed 's/(.*//'
