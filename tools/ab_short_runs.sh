W=/tmp/stst_examples; mkdir -p $W
python3 - <<PY
import numpy as np
n = 8192
np.full((n, n), 30.0, dtype=np.float32).tofile("$W/temp.bin")
p = np.zeros((n, n), dtype=np.float32); p[n//4-1:3*n//4, n//4-1:3*n//4] = 0.5; p.tofile("$W/power.bin")
PY
for i in 1 2; do
for e in "A=1" "STSTHIP_ALLOW_SPILLING_DEPTHS=1" "STSTHIP_BANDS_BESIDE_INTERIOR=0" "STSTHIP_VIRTUAL_STRIPS=1"; do
  echo "$e jacobi: $(env $e build/examples/jacobi_Jacobi5General_hip 16384 16384 1000 /dev/null 0.2 0.2 0.2 0.2 0.2 | grep Walltime)  hotspot: $(env $e build/examples/hotspot_hip 8192 8192 1000 $W/temp.bin $W/power.bin /dev/null | grep Walltime)"
done
done
