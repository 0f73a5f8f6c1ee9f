#!/bin/bash
# Two strips with a moving boundary (STSTHIP_SKEWED_STRIPS=1) against two strips with boundary bands, every bench leg.
for m in 0 1 0 1; do
  STSTHIP_SKEWED_STRIPS=$m python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null > /tmp/ab_skew_$m.json
  python3 - $m <<'PY'
import json, sys
m = sys.argv[1]
d = json.loads(open(f"/tmp/ab_skew_{m}.json").read().strip().splitlines()[-1])
print("skewed", m, round(d["value"], 1), d["verified"], {k: round(v["value"], 1) for k, v in d["legs"].items() if v.get("value")})
PY
done
