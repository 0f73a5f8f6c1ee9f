mkdir -p gpurun_out/harness_depth
for e in fdtd_render_hip fdtd_lut_hip; do
for t in 1 2 4 8; do
  d=gpurun_out/harness_depth/${e}_T$t; mkdir -p $d
  env STSTHIP_MAX_GENERATIONS=$t python tools/benchmark.py max_perf fdtd --exe build/examples/$e --out $d --samples 2 > $d/log.txt 2>&1
  python -c "
import json; d=json.load(open('$d/metrics.$e.json')); print('$e max generations $t:', round(d['measured']/1e9,1), 'Gcell-updates/s')"
done
done
