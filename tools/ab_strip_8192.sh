run() {
  env STSTHIP_STRIP_DEBUG_BANDS=1 "$@" python bench.py --strip-domain --rows-per-gpu $ROWS --steps 4 --warmup 1 --no-cpu-baseline --no-verify 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('rows $ROWS', '$*', '| Gcell/s', round(d['value'], 1), 'ms_per_step', round(d['ms_per_step'], 3))"
}
for ROWS in 8192 16384; do
run A=default
run A=default
run STSTHIP_VIRTUAL_STRIPS=1
run STSTHIP_VIRTUAL_STRIPS=2
run STSTHIP_VIRTUAL_STRIPS=2 STSTHIP_TAIL_PERMILLE=500
run STSTHIP_VIRTUAL_STRIPS=2 STSTHIP_TAIL_PERMILLE=1000
run STSTHIP_VIRTUAL_STRIPS=2 STSTHIP_BANDS_BESIDE_INTERIOR=0
run STSTHIP_VIRTUAL_STRIPS=3
done
ROWS=8192
env STSTHIP_STRIP_DEBUG_BANDS=0 python bench.py --strip-domain --rows-per-gpu 8192 --steps 4 --warmup 1 --no-cpu-baseline --no-verify 2>/dev/null | tail -1 | cut -c 1-200
