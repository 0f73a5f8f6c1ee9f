#!/bin/bash
# The reference's example binaries (unchanged sources, built by examples/Makefile against this backend) on the
# BASELINE.json configurations; prints the applications' own "Walltime:" lines and derived Gcell/s.
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
EX="$REPO/build/examples"
W=/tmp/stst_examples; mkdir -p $W/fd
python3 - <<PY
import numpy as np
n = 8192
np.full((n, n), 30.0, dtype=np.float32).tofile("$W/temp.bin")
p = np.zeros((n, n), dtype=np.float32); p[n//4-1:3*n//4, n//4-1:3*n//4] = 0.5; p.tofile("$W/power.bin")
PY
rate() { python3 -c "import sys; print(f'  => {float(sys.argv[1])*float(sys.argv[2])*float(sys.argv[3])/float(sys.argv[4])/1e9:.1f} Gcell-updates/s (application walltime, includes upload and scatter/gather)')" "$@"; }
echo "== jacobi (Jacobi5General) 16384 x 16384, 1000 generations"
t=$($EX/jacobi_Jacobi5General_hip 16384 16384 1000 /dev/null 0.2 0.2 0.2 0.2 0.2 | grep Walltime | awk '{print $2}'); echo "Walltime: $t s"; rate 16384 16384 1000 $t
echo "== hotspot (split cell structure) 8192 x 8192, 1000 iterations"
t=$($EX/hotspot_hip 8192 8192 1000 $W/temp.bin $W/power.bin /dev/null | grep Walltime | awk '{print $2}'); echo "Walltime: $t s"; rate 8192 8192 1000 $t
echo "== hotspot (AoS) 8192 x 8192, 1000 iterations"
t=$($EX/hotspot_aos_hip 8192 8192 1000 $W/temp.bin $W/power.bin /dev/null | grep Walltime | awk '{print $2}'); echo "Walltime: $t s"; rate 8192 8192 1000 $t
echo "== fdtd max_grid.json without snapshots: 4608 x 4608, 184911 time steps (2 sub-iterations each)"
t=$($EX/fdtd_hip -c $REPO/tools/data/fdtd_max_grid_nosnap.json -o $W/fd | grep Walltime | awk '{print $2}'); echo "Walltime: $t s"; rate 4608 4608 184911 $t
echo "== conway 512 x 512, 100 generations (stencil::cuda build; BASELINE config 0 is the cpu backend)"
python3 -c "
import numpy as np
g = (np.random.default_rng(0xC0FFEE).random((512,512)) < 0.35)
open('$W/conway.txt','w').write(''.join(''.join('X' if v else '.' for v in r)+'\n' for r in g))"
( time $EX/conway_hip 512 512 100 < $W/conway.txt | md5sum ) 2>&1 | grep -E "real|-"
( time $EX/conway_cpu 512 512 100 < $W/conway.txt | md5sum ) 2>&1 | grep -E "real|-"
