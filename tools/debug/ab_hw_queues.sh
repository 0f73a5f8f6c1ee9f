#!/bin/bash
# Hardware queues (GPU_MAX_HW_QUEUES, read by the HIP runtime at start) against the pass driver's strips with moving boundaries
# (no band streams any more) and the strip driver's sub-strips: does a third stream pay when it has a queue of its own?
for q in 4 8; do for v in 2 3; do
  GPU_MAX_HW_QUEUES=$q STSTHIP_VIRTUAL_STRIPS=$v python bench.py --gpus 1 --steps 8 --warmup 2 --no-cpu-baseline --no-legs 2>/dev/null > /tmp/hwq.json
  python3 - $q $v <<'PY'
import json, sys
d = json.loads(open("/tmp/hwq.json").read().strip().splitlines()[-1])
print("hw queues", sys.argv[1], "strips", sys.argv[2], round(d["value"], 1), d["verified"], d["launches_per_step"], flush=True)
PY
done; done
for q in 4 8; do
  echo "== strip driver, hw queues $q"
  GPU_MAX_HW_QUEUES=$q python tools/bench_strip.py --rows 2048 4096 --exchange-every 4 2>/dev/null | grep rows_per_gpu | cut -c1-140
done
