#!/bin/bash
# A/B of one environment knob on the headline benchmark: tools/ab_bench_env.sh NAME v1 v2 [v1 v2 ...]
NAME="$1"; shift
for v in "$@"; do
  env "$NAME=$v" python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$NAME=$v', 'value', round(d['value'], 1), 'ms_per_step', round(d['ms_per_step'], 2), 'verified', d.get('verified'),
      'general', round(d.get('general_coefficients', {}).get('value', 0), 1))"
done
