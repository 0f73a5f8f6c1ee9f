#!/bin/bash
# launch parameters of the fdtd_max_grid and hotspot_8192 legs of bench.py (environment knobs act on every launch shape)
mkdir -p gpurun_out
out=gpurun_out/sweep_fdtd.txt
: > $out
run() {
  echo "## $*" >> $out
  env "$@" python3 bench.py --no-cpu-baseline --no-verify --steps 2 --warmup 1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); l=d['legs']; print(round(l['fdtd_max_grid']['value'],1), round(l['hotspot_8192']['value'],1), round(l['general_coefficients']['value'],1))" >> $out
}
run A=0
run STSTHIP_VIRTUAL_STRIPS=1
run STSTHIP_VIRTUAL_STRIPS=2
run STSTHIP_VIRTUAL_STRIPS=3
for t in 150 250 500 700 1000; do run STSTHIP_TAIL_PERMILLE=$t; done
run STSTHIP_TAPER=
run STSTHIP_TAPER=120:4
run STSTHIP_TAPER=200:2
run STSTHIP_TAPER=250:4
for c in 48 64 96 128 192 256; do run STSTHIP_CHUNK_ROWS=$c; done
run A=0
