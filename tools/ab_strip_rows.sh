#!/bin/bash
# One strip of a multi-GPU run on one MI355X (tools/ab_strip_rows.sh "ENV=..." ...): rows per GPU of the strong-scaling
# series, with the band launches of a strip that has neighbours (STSTHIP_STRIP_DEBUG_BANDS=1: no exchange, timing only).
for combo in "$@"; do
  for rows in 2048 4096 8192; do
    env $combo python bench.py --strip-domain --rows-per-gpu $rows --steps 3 --warmup 1 --no-cpu-baseline --no-verify 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$combo rows $rows', '| value', round(d['value'], 1), 'ms_per_step', round(d['ms_per_step'], 2))"
  done
done
