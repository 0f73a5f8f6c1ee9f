#!/bin/bash
# Does the depth probe of a one-shot run pick the same depth every time?  (bench.py's template_api_fma leg has been seen
# at 0.080 s instead of 0.058 s for a whole bench invocation.)  The FMA build of the unchanged jacobi example, N runs.
J="16384 16384 1000 /dev/null 0.2 0.2 0.2 0.2 0.2"
for i in $(seq 1 ${1:-25}); do
  STSTHIP_TRACE_STREAM=1 build/examples/jacobi_Jacobi5General_hip_fma $J 2>&1 | grep "depth probe\|Walltime" | tr '\n' ' '; echo
done
