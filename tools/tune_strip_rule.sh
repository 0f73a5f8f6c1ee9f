#!/bin/bash
# Where two row strips (two launches side by side, boundary bands beside them) beat one launch per pass: the data
# behind suggest_row_strips (runtime.hip).  usage (GPU box): tools/tune_strip_rule.sh > profiles/rNN_tune_strip_rule.txt
one() { # app rows cols
  for v in 0 1 2 3; do
    env STSTHIP_VIRTUAL_STRIPS=$v BENCH_APPS_ROWS=$2 BENCH_APPS_COLS=$3 python tools/bench_apps.py $1 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1 $2 x $3 strips', '$v' if '$v' != '0' else 'rule', '|', d['Gcell_updates_per_s'], 'Gcell/s', d['ms_per_launch'], 'ms per launch')"
  done
}
one jacobi 1024 16384
one jacobi 2048 16384
one jacobi 4096 16384
one jacobi 8192 16384
one jacobi 4096 4096
one jacobi 6144 6144
one jacobi 8192 8192
one jacobi_general 4096 16384
one jacobi_general 8192 8192
one hotspot 4096 4096
one hotspot 6144 6144
one hotspot 2048 8192
one hotspot_f64 4096 4096
one hotspot_f64 6144 6144
one fdtd_grouped 2304 2304
one fdtd_grouped 4608 4608
one fdtd_grouped 9216 4608
one fdtd_aos 4608 4608
one conway 8192 8192
one conway 4096 16384
