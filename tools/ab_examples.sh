#!/bin/bash
# A/B of example binaries built with different SweepTuning rules (copies kept under build/ab/); GPU box.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p /tmp/fd /tmp/cv
python3 - <<'PY'
import json
e = json.load(open("tools/data/fdtd_max_grid_nosnap.json"))
e["time"]["t_max"] = e["time"]["t_max"] / 8   # ~23000 steps at 4608^2
json.dump(e, open("/tmp/fd/short.json", "w"))
PY
for b in build/ab/fdtd_lut_hip_t4 build/ab/fdtd_lut_hip_t8 build/examples/fdtd_lut_hip build/examples/fdtd_hip; do
  [ -x "$b" ] || continue
  for i in 1 2; do echo "$b: $($b -c /tmp/fd/short.json -o /tmp/fd 2>&1 < /dev/null | grep -i walltime)"; done
done
for i in 1 2; do build/examples/convection_hip tools/data/convection_bench.json /tmp/cv 2>&1 < /dev/null | tail -2; done
