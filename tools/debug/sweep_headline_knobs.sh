#!/bin/bash
# Launcher knobs on the bench headline now that the strips have no boundary bands: tail weight of the chunk rule, taper.
run() { # label, env...
  label=$1; shift
  env "$@" python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-legs 2>/dev/null > /tmp/knob.json
  python3 - "$label" <<'PY'
import json, sys
d = json.loads(open("/tmp/knob.json").read().strip().splitlines()[-1])
print(sys.argv[1], round(d["value"], 1), d["verified"], flush=True)
PY
}
for rep in 1 2; do
run default X=1
run tail150 STSTHIP_TAIL_PERMILLE=150
run tail250 STSTHIP_TAIL_PERMILLE=250
run tail350 STSTHIP_TAIL_PERMILLE=350
run taper_off STSTHIP_TAPER=
run taper_120_4 STSTHIP_TAPER=120:4
run taper_250_2 STSTHIP_TAPER=250:2
run skew480 STSTHIP_STRIP_SKEW_PERMILLE=480
run skew520 STSTHIP_STRIP_SKEW_PERMILLE=520
done
