#!/bin/bash
# Convection end to end at res = 1024 (tools/data/convection_bench.json: 3071 x 1023 cells of 88 bytes, 2 time steps
# of 1000 pseudo-transient iterations): the reference's unchanged example against the driver with the device-side
# convergence check (examples/convection_device_reduce.cpp).  Run on the GPU box.
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
EX="$REPO/build/examples"
W=/tmp/stst_convection; rm -rf $W; mkdir -p $W/a $W/b
for run in 1 2 3; do
  echo "== run $run: convection_hip (host-side scan of convection.cpp:412-438)"
  $EX/convection_hip $REPO/tools/data/convection_bench.json $W/a | tail -4
  echo "== run $run: convection_reduce_hip (stencil::hip::max_abs on the device)"
  $EX/convection_reduce_hip $REPO/tools/data/convection_bench.json $W/b | tail -4
done
