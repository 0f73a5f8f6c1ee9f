for r in 1024 2048 4096; do
for a in jacobi_general jacobi_general_persistent; do
for extra in "A=1" "STSTHIP_FINE_ROWS=16" "STSTHIP_PERSISTENT_PHASES=8"; do
 env $extra STSTHIP_VIRTUAL_STRIPS=1 BENCH_APPS_ROWS=$r python tools/bench_apps.py $a 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$a rows $r $extra |', d['Gcell_updates_per_s'], 'Gcell/s', d['ms_per_launch'], 'ms per launch')"
done
done
done
