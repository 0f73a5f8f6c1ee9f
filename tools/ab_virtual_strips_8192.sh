for i in 1 2; do
for v in 1 2; do
  env STSTHIP_VIRTUAL_STRIPS=$v python bench.py --rows-per-gpu 8192 --steps 5 --warmup 1 --no-cpu-baseline --no-verify 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('pass driver 8192x16384 strips $v', round(d['value'],1), round(d['ms_per_step'],3))"
done
done
for v in 1 2; do
  env STSTHIP_VIRTUAL_STRIPS=$v python bench.py --rows-per-gpu 4096 --steps 5 --warmup 1 --no-cpu-baseline --no-verify 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('pass driver 4096x16384 strips $v', round(d['value'],1), round(d['ms_per_step'],3))"
done
for ROWS in 2048 8192; do
env STSTHIP_STRIP_DEBUG_BANDS=1 python bench.py --strip-domain --rows-per-gpu $ROWS --steps 4 --warmup 1 --no-cpu-baseline --no-verify 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('strip $ROWS default', round(d['value'],1), round(d['ms_per_step'],3))"
done
