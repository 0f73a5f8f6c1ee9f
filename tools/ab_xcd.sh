for i in 1 2; do for X in 0 1; do echo -n "XCD_REMAP=$X: "; STSTHIP_XCD_REMAP=$X python tools/bench_apps.py jacobi hotspot_aos fdtd_aos 2>&1 | grep "^{" | python -c "
import sys,json
print(' '.join(f\"{json.loads(l)['app']}={json.loads(l)['Gcell_updates_per_s']}\" for l in sys.stdin))"; done; done
