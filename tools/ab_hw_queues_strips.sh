for q in 4 8; do for v in 2 3 4; do
  env GPU_MAX_HW_QUEUES=$q STSTHIP_VIRTUAL_STRIPS=$v python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-verify 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('bench 16384^2 hw queues $q strips $v:', round(d['value'], 1), round(d['ms_per_step'], 3), round(d.get('general_coefficients', {}).get('value', 0), 1))"
done; done
