#!/bin/bash
# 32x32x16 vs 16x16x32 in the ring8 step structure, constant vs random operands, long runs (DVFS settled)
cd $(dirname $0)
out=${1:-/tmp}
run() {
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $1 -o $out/mfma_loop_bin mfma_loop.hip 2>/dev/null || { echo "build failed: $1"; return; }
  printf "%-44s " "[$1]"
  $out/mfma_loop_bin 512 ${2:-8000}
}
run ""
run "-DRANDOM"
run "-DRANDOM -DSHAPE16"
run "-DSHAPE16"
run "-DRANDOM -DSETPRIO"
run "-DRANDOM -DSHAPE16 -DSETPRIO"
run "-DRANDOM -DPIPE"
run "-DRANDOM -DNO_DMA"
run "-DRANDOM -DSHAPE16 -DNO_DMA"
run "-DRANDOM"
run "-DRANDOM -DSHAPE16"
