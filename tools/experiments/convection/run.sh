#!/bin/bash
# Build (in the build container) or run (on the GPU box) the convection shape sweep.
# usage: run.sh build | run
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/build/experiments/convection"; mkdir -p "$OUT"
FLAGS="-x hip -std=c++20 -O3 --hipstdpar --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=iterative-ilp -w -I$REPO/include -I$REPO/include/compat -I/root/reference/examples/convection -DSTENCILSTREAM_BACKEND_CUDA -DCONVECTION_SPIT_CELL_STRUCT=true"
LINK="-L$REPO/stencilstream_amd -lststhip -Wl,-rpath,$REPO/stencilstream_amd"
SHAPES="1:1:2:false 1:1:4:false 1:3:2:false 1:3:2:true 2:3:2:false 2:2:2:false 2:6:2:false 1:1:2:true"
if [ "$1" = build ]; then
  hipcc $FLAGS -DSHAPE_DEFAULT "$REPO/tools/experiments/convection/convection_shapes.cpp" -o "$OUT/pt_default" $LINK &
  for s in $SHAPES; do IFS=: read t w p i <<< "$s"
    hipcc $FLAGS -DSHAPE_T=$t -DSHAPE_W=$w -DSHAPE_P=$p -DSHAPE_INTERIOR=$i "$REPO/tools/experiments/convection/convection_shapes.cpp" -o "$OUT/pt_t${t}_w${w}_p${p}_$i" $LINK &
  done; wait; ls "$OUT"
else
  for b in "$OUT"/pt_*; do "$b" 2>&1 | tail -1; done
fi
