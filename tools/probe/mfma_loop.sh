#!/bin/bash
# builds tools/probe/mfma_loop.hip in several variants on the GPU box and prints the TFLOP/s of each
cd $(dirname $0)
out=${1:-/tmp}
run() {
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $1 -o $out/mfma_loop_bin mfma_loop.hip 2>/dev/null || { echo "build failed: $1"; return; }
  printf "%-44s " "[$1]"
  $out/mfma_loop_bin ${2:-512}
}
run ""
run "-DNO_DMA"
run "-DNO_WAIT"
run "-DNO_BARRIER"
run "-DNO_BARRIER -DNO_DMA"
run "-DSETPRIO"
run "-DPIPE"
run "-DPIPE -DSETPRIO"
run "-DPIPE -DNO_DMA"
run "-DPIPE -DNO_BARRIER -DNO_DMA"
run "-DKB=128 -DRING=3"
run "-DKB=128 -DRING=3 -DNO_DMA"
run "-DKB=128 -DRING=4 -DPIPE"
run "-DWAVES=4"
run "-DWAVES=4 -DPIPE"
run "-DWAVES=4 -DKB=128 -DRING=3" 
run "-DKB=32 -DNSTEPS=392" 1024
run "-DKB=32 -DNSTEPS=392" 512
run "-DKB=32 -DNSTEPS=392 -DPIPE" 1024
run "-DKB=32 -DNSTEPS=392 -DRING=6" 1024
run "" 256
run "-DPIPE" 256
run "-DPIPE" 1024
