#!/bin/bash
# Thin strips (the per-rank share of the 16384^2 strong-scaling series): chunk length, taper and depth knobs for
# 2048 / 4096 rows of 16384 columns, band launches included (STSTHIP_STRIP_DEBUG_BANDS=1), no exchange.
# usage: tools/tune_thin_strip.sh  > profiles/rNN_tune_thin_strip.txt   (on the GPU box)
run() {
  env STSTHIP_STRIP_DEBUG_BANDS=1 "$@" python bench.py --strip-domain --rows-per-gpu $ROWS --steps 3 --warmup 1 --no-cpu-baseline --no-verify 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('rows $ROWS', '$*', '| Gcell/s', round(d['value'], 1), 'ms_per_step', round(d['ms_per_step'], 3))"
}
for ROWS in 2048 4096; do
  run A=default
  for c in 24 32 48 64 96 128 192 256 342 512 1024; do run STSTHIP_CHUNK_ROWS=$c; done
  for c in 64 128 256; do run STSTHIP_CHUNK_ROWS=$c STSTHIP_TAPER=; done
  for t in 100 200 350 700 1000 2000; do run STSTHIP_TAIL_PERMILLE=$t; done
  run STSTHIP_MAX_GENERATIONS=6
  run STSTHIP_MAX_GENERATIONS=6 STSTHIP_CHUNK_ROWS=128
done
