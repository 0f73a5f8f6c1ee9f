#!/bin/bash
# Band scheduling of a strip with neighbours on both sides (one rank of a multi-GPU run, on one GPU, no exchange):
# top / bottom band on one stream or two (STSTHIP_BANDS_APART), raised wave priority of the band launches
# (STSTHIP_BAND_WAVE_PRIORITY), priority of the exchange stream.
run() {
  env STSTHIP_STRIP_DEBUG_BANDS=1 "$@" python bench.py --strip-domain --rows-per-gpu $ROWS --steps 4 --warmup 1 --no-cpu-baseline --no-verify 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('rows $ROWS', '$*', '| Gcell/s', round(d['value'], 1), 'ms_per_step', round(d['ms_per_step'], 3))"
}
for ROWS in 2048 4096 8192; do
  run STSTHIP_BANDS_APART=0 STSTHIP_BAND_WAVE_PRIORITY=0
  run STSTHIP_BANDS_APART=1 STSTHIP_BAND_WAVE_PRIORITY=0
  run STSTHIP_BANDS_APART=0 STSTHIP_BAND_WAVE_PRIORITY=1
  run STSTHIP_BANDS_APART=1 STSTHIP_BAND_WAVE_PRIORITY=1
  run STSTHIP_BANDS_APART=1 STSTHIP_BAND_WAVE_PRIORITY=1 STSTHIP_COMM_STREAM_PRIORITY=0
  run STSTHIP_BANDS_APART=1 STSTHIP_BAND_WAVE_PRIORITY=1 STSTHIP_VIRTUAL_STRIPS=1
  run STSTHIP_BANDS_APART=1 STSTHIP_BAND_WAVE_PRIORITY=1 STSTHIP_VIRTUAL_STRIPS=2
done
for ROWS in 8192; do
  env STSTHIP_STRIP_DEBUG_BANDS=0 python bench.py --strip-domain --rows-per-gpu $ROWS --steps 4 --warmup 1 --no-cpu-baseline --no-verify 2>/dev/null | tail -1 | cut -c 1-140
  env STSTHIP_STRIP_DEBUG_BANDS=0 STSTHIP_COMM_STREAM_PRIORITY=0 python bench.py --strip-domain --rows-per-gpu $ROWS --steps 4 --warmup 1 --no-cpu-baseline --no-verify 2>/dev/null | tail -1 | cut -c 1-140
done
for w in "" "STSTHIP_BAND_WAVE_PRIORITY=0"; do
  env $w python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('bench $w', d['value'], d['ms_per_step'], d.get('verified'), d.get('general_coefficients', {}).get('value'))"
done
