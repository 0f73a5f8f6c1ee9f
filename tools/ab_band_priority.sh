#!/bin/bash
# boundary bands on highest-priority streams (default) against normal-priority streams of their own
for p in 1 0 1 0; do
  env STSTHIP_BAND_STREAM_PRIORITY=$p python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-verify 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('bench 16384^2, band stream priority $p:', round(d['value'], 1), round(d['ms_per_step'], 3), round(d.get('general_coefficients', {}).get('value', 0), 1))"
done
for p in 1 0; do
  env STSTHIP_BAND_STREAM_PRIORITY=$p python bench.py --rows-per-gpu 4096 --steps 8 --warmup 2 --no-cpu-baseline --no-verify 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('pass driver 4096 x 16384, band stream priority $p:', round(d['value'], 1))"
  env STSTHIP_BAND_STREAM_PRIORITY=$p python tools/bench_apps.py hotspot hotspot_f64 fdtd_grouped conway 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('band stream priority $p:', d['app'], d['Gcell_updates_per_s'])"
  env STSTHIP_BAND_STREAM_PRIORITY=$p STSTHIP_STRIP_DEBUG_BANDS=1 python bench.py --strip-domain --rows-per-gpu 8192 --steps 4 --warmup 1 --no-cpu-baseline --no-verify 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('strip 8192 rows with two bands, band stream priority $p:', round(d['value'], 1))"
  env STSTHIP_BAND_STREAM_PRIORITY=$p STSTHIP_STRIP_DEBUG_BANDS=1 python bench.py --strip-domain --rows-per-gpu 2048 --steps 4 --warmup 1 --no-cpu-baseline --no-verify 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('strip 2048 rows with two bands, band stream priority $p:', round(d['value'], 1))"
done
