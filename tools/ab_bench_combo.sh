#!/bin/bash
# headline benchmark under several environment settings: tools/ab_bench_combo.sh "A=1 B=2" "A=3" ...
for combo in "$@"; do
  env $combo python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$combo', '| value', round(d['value'], 1), 'ms_per_step', round(d['ms_per_step'], 2), 'verified', d.get('verified'),
      'general', round(d.get('general_coefficients', {}).get('value', 0), 1))"
done
