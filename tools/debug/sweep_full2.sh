#!/bin/bash
mkdir -p gpurun_out
out=gpurun_out/sweep_full2.txt
: > $out
run() {
  echo "## $*" >> $out
  env "$@" python3 bench.py --no-legs --no-cpu-baseline --no-verify --steps 8 --warmup 3 2>>$out | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['value'],1), d['ms_per_step'], d.get('launches_per_step'), d['roofline'].get('kernel_ms'))" >> $out
}
run A=0
for c in 0 328 410 455 512 585; do
  run STSTHIP_TAPER= STSTHIP_CHUNK_ROWS=$c
  run STSTHIP_TAPER=100:2 STSTHIP_CHUNK_ROWS=$c
  run STSTHIP_TAPER=60:2 STSTHIP_CHUNK_ROWS=$c
done
run A=0
for t in 250 350 450; do
  run STSTHIP_TAPER= STSTHIP_TAIL_PERMILLE=$t
done
echo "## apps default" >> $out
python3 tools/bench_apps.py hotspot fdtd_grouped fdtd_aos conway 2>/dev/null | grep -o '"app": "[a-z_0-9]*"\|"Gcell_updates_per_s": [0-9.]*' | paste - - >> $out
echo "## apps TAPER=" >> $out
STSTHIP_TAPER= python3 tools/bench_apps.py hotspot fdtd_grouped fdtd_aos conway 2>/dev/null | grep -o '"app": "[a-z_0-9]*"\|"Gcell_updates_per_s": [0-9.]*' | paste - - >> $out
echo "## apps default" >> $out
python3 tools/bench_apps.py hotspot fdtd_grouped fdtd_aos conway 2>/dev/null | grep -o '"app": "[a-z_0-9]*"\|"Gcell_updates_per_s": [0-9.]*' | paste - - >> $out
echo "## general default / TAPER=" >> $out
python3 bench.py --no-cpu-baseline --no-verify --steps 5 --warmup 2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], {k: (v.get('value') if isinstance(v, dict) else v) for k, v in d.get('legs', {}).items()})" >> $out
STSTHIP_TAPER= python3 bench.py --no-cpu-baseline --no-verify --steps 5 --warmup 2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], {k: (v.get('value') if isinstance(v, dict) else v) for k, v in d.get('legs', {}).items()})" >> $out
