#!/bin/bash
# the uniform Jacobi form at other grid sizes: the library's rule against the general rule forced by the environment
mkdir -p gpurun_out
out=gpurun_out/sweep_sizes.txt
: > $out
run() {
  echo "## $*" >> $out
  env "${@:2}" python3 bench.py --size $1 --no-legs --no-cpu-baseline --no-verify --steps 8 --warmup 3 2>>$out | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['value'],1), d['ms_per_step'], d.get('launches_per_step'))" >> $out
}
for n in 4096 6144 8192 12288 14336 16384 20480 24576 32768; do
  run $n A=0
  run $n STSTHIP_TAIL_PERMILLE=350 STSTHIP_TAPER=150:2
done
