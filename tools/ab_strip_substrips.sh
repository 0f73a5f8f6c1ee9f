run() {
  env "$@" python bench.py --strip-domain --rows-per-gpu $ROWS --steps 4 --warmup 1 --no-cpu-baseline --no-verify 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('rows $ROWS', '$*', '| Gcell/s', round(d['value'], 1), 'ms_per_step', round(d['ms_per_step'], 3))"
}
for ROWS in 2048 4096 8192 16384; do
  run STSTHIP_STRIP_DEBUG_BANDS=1
  run STSTHIP_STRIP_DEBUG_BANDS=1 STSTHIP_VIRTUAL_STRIPS=1
  run STSTHIP_STRIP_DEBUG_BANDS=1 STSTHIP_VIRTUAL_STRIPS=2
  run STSTHIP_STRIP_DEBUG_BANDS=0
  run STSTHIP_STRIP_DEBUG_BANDS=0 STSTHIP_VIRTUAL_STRIPS=1
  run STSTHIP_STRIP_DEBUG_BANDS=0 STSTHIP_VIRTUAL_STRIPS=2
done
ROWS=8192; run STSTHIP_STRIP_DEBUG_BANDS=1 STSTHIP_BANDS_ONE_LAUNCH=0
ROWS=2048; run STSTHIP_STRIP_DEBUG_BANDS=1 STSTHIP_BANDS_ONE_LAUNCH=0
ROWS=2048; run STSTHIP_STRIP_DEBUG_BANDS=1 STSTHIP_BANDS_APART=1 STSTHIP_BANDS_ONE_LAUNCH=0
