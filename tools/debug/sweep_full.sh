#!/bin/bash
# full-grid sweeps of the chunk model for the staged headline kernel (bench.py's timed path: two strips side by side)
mkdir -p gpurun_out
out=gpurun_out/sweep_full.txt
: > $out
run() {
  echo "## $*" >> $out
  env "$@" python3 bench.py --no-legs --no-cpu-baseline --no-verify --steps 8 --warmup 3 2>>$out | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['value'],1), d['ms_per_step'], d.get('launches_per_step'), d['roofline'].get('kernel_ms'))" >> $out
}
run A=0
for c in 164 205 256 410 512 683 1024 2048; do
  run STSTHIP_CHUNK_ROWS=$c
done
for t in 100 200 500 700 1000; do
  run STSTHIP_TAIL_PERMILLE=$t
done
run STSTHIP_TAPER=
run STSTHIP_TAPER=150:4
run STSTHIP_TAPER=300:2
run STSTHIP_TAPER=100:2
run STSTHIP_VIRTUAL_STRIPS=1
run STSTHIP_VIRTUAL_STRIPS=3
run STSTHIP_VIRTUAL_STRIPS=4
run A=0
