#!/bin/bash
# Thin strips, second sweep: launches of less than one residency round -- chunk lengths that give every SIMD the same
# whole number m of equal waves (98 column strips x chunks = m x 1024), taper off.
run() {
  env STSTHIP_STRIP_DEBUG_BANDS=1 "$@" python bench.py --strip-domain --rows-per-gpu $ROWS --steps 3 --warmup 1 --no-cpu-baseline --no-verify 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('rows $ROWS', '$*', '| Gcell/s', round(d['value'], 1), 'ms_per_step', round(d['ms_per_step'], 3))"
}
ROWS=2048
run A=default
for c in 25 29 34 40 45 50 56 67 103; do run STSTHIP_CHUNK_ROWS=$c STSTHIP_TAPER=; done
run STSTHIP_CHUNK_ROWS=50 STSTHIP_TAPER=120:2
ROWS=4096
run A=default
for c in 50 57 67 79 90 100 112 133; do run STSTHIP_CHUNK_ROWS=$c STSTHIP_TAPER=; done
run STSTHIP_CHUNK_ROWS=100 STSTHIP_TAPER=120:2
ROWS=8192
run A=default
for c in 100 114 133 158 200 ; do run STSTHIP_CHUNK_ROWS=$c STSTHIP_TAPER=; done
