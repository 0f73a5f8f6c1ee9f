#!/bin/bash
# boundary bands swept with one cell per lane (the narrow form) instead of the default shape: STSTHIP_NARROW_BAND_ROWS
run() {
  env STSTHIP_STRIP_DEBUG_BANDS=1 "$@" python bench.py --strip-domain --rows-per-gpu $ROWS --steps 4 --warmup 1 --no-cpu-baseline --no-verify 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('strip rows $ROWS', '$*', '| Gcell/s', round(d['value'], 1), 'ms_per_step', round(d['ms_per_step'], 3))"
}
for ROWS in 2048 4096 8192; do
  for n in 0 32 0 32; do run STSTHIP_NARROW_BAND_ROWS=$n; done
done
for n in 0 32 0 32; do
  env STSTHIP_NARROW_BAND_ROWS=$n python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('bench narrow band rows $n:', round(d['value'], 1), round(d['ms_per_step'], 3), d.get('verified'), round(d.get('general_coefficients', {}).get('value', 0), 1))"
done
for n in 0 32; do
  env STSTHIP_NARROW_BAND_ROWS=$n python tools/bench_apps.py hotspot conway 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('narrow band rows $n:', d['app'], d['Gcell_updates_per_s'])"
done
