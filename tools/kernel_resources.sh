#!/bin/bash
# Print VGPR/SGPR/scratch/occupancy of every sweep kernel of one app translation unit.
# usage: tools/kernel_resources.sh stencilstream_amd/csrc/app_jacobi.hip
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
SRC="$1"
hipcc -O3 -std=c++20 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=iterative-ilp -I"$ROOT/include" -I"$ROOT/include/compat" \
  -I"$ROOT/stencilstream_amd/csrc" -c "$SRC" -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 |
  grep -E "remark:" | sed -e 's/.*remark: *//' -e 's/ \[-Rpass.*//' |
  awk '/Function Name/ {name=$3} /^VGPRs:/ {v=$2} /TotalSGPRs/ {s=$2} /ScratchSize/ {sc=$3} /Occupancy/ {o=$3} /LDS Size/ {print name, "vgpr="v, "sgpr="s, "scratch="sc, "occ="o}' |
  while read -r name rest; do echo "$(echo "$name" | c++filt | sed -e 's/void stencil::hip::internal::sweep_kernel<stencil::hip::internal::Sweep<//' -e 's/> >(.*//' ) $rest"; done
