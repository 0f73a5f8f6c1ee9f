#!/bin/bash
# bench.py's template_api_fma leg has been seen ~40 % slow for a whole invocation (all four runs of the binary), the legs
# around it normal.  Repeat the bench; when the leg is slow, run the binary by hand right away with the driver's narration.
for i in 1 2 3 4; do
  python bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null > /tmp/catch_$i.json
  python3 - $i <<'PY'
import json, sys
d = json.loads(open(f"/tmp/catch_{sys.argv[1]}.json").read().strip().splitlines()[-1])
l = d["legs"]
print("bench", sys.argv[1], round(d["value"]), "template_api", l["template_api"].get("walltime_s"), "template_api_fma", l["template_api_fma"].get("walltime_s"), flush=True)
PY
  tools/debug/probe_stability.sh 2
done
