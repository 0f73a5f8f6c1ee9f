#!/bin/bash
# The upload in row blocks against one copy at grid sizes between "not split" and the benchmark's (jacobi example, ms).
C="0.2 0.2 0.2 0.2 0.2"
for n in 6144 8192 10240 12288 24576; do for rep in 1 2; do
  a=$(STSTHIP_STREAM_UPLOAD=0 build/examples/jacobi_Jacobi5General_hip $n $n 1000 /dev/null $C | grep Wall | awk '{print $2*1000}')
  b=$(STSTHIP_STREAM_UPLOAD=1 build/examples/jacobi_Jacobi5General_hip $n $n 1000 /dev/null $C | grep Wall | awk '{print $2*1000}')
  echo "jacobi ${n}^2 x 1000: one copy $a ms, row blocks $b ms"
done; done
