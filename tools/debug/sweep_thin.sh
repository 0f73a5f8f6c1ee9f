#!/bin/bash
# thin-strip sweeps of the staged kernel's launch parameters (one process per setting; prints bench_strip's lines)
set -e
mkdir -p gpurun_out
out=gpurun_out/sweep_thin.txt
: > $out
run() {
  echo "## $*" >> $out
  env "$@" python3 tools/bench_strip.py --rows 2048 4096 --exchange-every 4 --reps 3 >> $out 2>&1
}
run A=0
run STSTHIP_TAPER=
for c in 64 96 160 200 256; do
  run STSTHIP_CHUNK_ROWS=$c
  run STSTHIP_CHUNK_ROWS=$c STSTHIP_TAPER=
done
run STSTHIP_MAX_GENERATIONS=8
