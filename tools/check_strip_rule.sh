#!/bin/bash
# after a change of suggest_row_strips: the configurations of tools/bench_apps.py, the bench line, strips with bands
python tools/bench_apps.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['app'], d['grid'], d['Gcell_updates_per_s'], d['ms_per_launch'])"
for ROWS in 2048 4096 8192; do
env STSTHIP_STRIP_DEBUG_BANDS=1 python bench.py --strip-domain --rows-per-gpu $ROWS --steps 4 --warmup 1 --no-cpu-baseline --no-verify 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('strip with two bands, rows $ROWS', round(d['value'],1), round(d['ms_per_step'],3))"
done
for ROWS in 4096 8192; do
python bench.py --rows-per-gpu $ROWS --steps 5 --warmup 1 --no-cpu-baseline --no-verify 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('pass driver $ROWS x 16384', round(d['value'],1), round(d['ms_per_step'],3))"
done
python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('bench', d['value'], d['ms_per_step'], d.get('verified'), d.get('general_coefficients', {}).get('value'))"
