#!/bin/bash
# HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); streams that share a queue serialise.
run() {
  env STSTHIP_STRIP_DEBUG_BANDS=1 "$@" python bench.py --strip-domain --rows-per-gpu $ROWS --steps 4 --warmup 1 --no-cpu-baseline --no-verify 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('rows $ROWS', '$*', '| Gcell/s', round(d['value'], 1), 'ms_per_step', round(d['ms_per_step'], 3))"
}
for ROWS in 2048 8192; do
 for q in 2 4 8 16; do
  run GPU_MAX_HW_QUEUES=$q STSTHIP_BANDS_APART=0 STSTHIP_BAND_WAVE_PRIORITY=0
  run GPU_MAX_HW_QUEUES=$q STSTHIP_BANDS_APART=1 STSTHIP_BAND_WAVE_PRIORITY=0
  run GPU_MAX_HW_QUEUES=$q STSTHIP_BANDS_APART=1 STSTHIP_BAND_WAVE_PRIORITY=1
 done
done
for q in 2 4 8 16; do
  env GPU_MAX_HW_QUEUES=$q STSTHIP_BAND_WAVE_PRIORITY=0 python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('bench queues $q', d['value'], d['ms_per_step'], d.get('verified'), d.get('general_coefficients', {}).get('value'))"
done
