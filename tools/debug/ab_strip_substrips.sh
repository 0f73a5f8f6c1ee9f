#!/bin/bash
# One rank's compute side of a multi-GPU run (tools/bench_strip.py: the middle strip of three, an exchange that moves
# nothing) with the strip as one launch per pass (STSTHIP_STRIP_SUBSTRIPS=1) and as two sub-strips with a moving boundary (2),
# and what the rule picks (-1).
for rep in 1 2; do for sub in 1 2 -1; do
  echo "== substrips=$sub"
  STSTHIP_STRIP_SUBSTRIPS=$sub python tools/bench_strip.py --rows 2048 4096 8192 --exchange-every 4 2>/dev/null | grep -v "^$" | tail -4
done; done
