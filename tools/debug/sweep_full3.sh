#!/bin/bash
mkdir -p gpurun_out
out=gpurun_out/sweep_full3.txt
: > $out
legs() {
  echo "## $*" >> $out
  env "$@" python3 bench.py --no-cpu-baseline --no-verify --steps 8 --warmup 3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['value'],1), {k: (round(v.get('value'),1) if isinstance(v, dict) and v.get('value') else v) for k, v in d.get('legs', {}).items()})" >> $out
  env "$@" python3 tools/bench_apps.py hotspot hotspot_aos fdtd_grouped fdtd_aos conway 2>/dev/null | grep -o '"app": "[a-z_0-9]*"\|"Gcell_updates_per_s": [0-9.]*' | paste - - | tr '\n' ' ' >> $out
  echo >> $out
  env "$@" python3 tools/bench_strip.py --rows 2048 4096 8192 --exchange-every 4 --reps 3 2>/dev/null | grep -o '"rows_per_gpu": [0-9]*\|"Gcell_updates_per_s_per_gpu": [0-9.]*' | paste - - | tr '\n' ' ' >> $out
  echo >> $out
}
legs A=0
legs STSTHIP_TAPER= STSTHIP_TAIL_PERMILLE=200
legs STSTHIP_TAPER= STSTHIP_TAIL_PERMILLE=250
legs STSTHIP_TAPER=
legs A=0
legs STSTHIP_TAPER= STSTHIP_TAIL_PERMILLE=200
