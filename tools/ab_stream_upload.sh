#!/bin/bash
# One-shot runs of the unchanged example binaries (their own `Walltime:`, upload included) with the upload in one
# copy in front of the first pass (STSTHIP_STREAM_UPLOAD=0) against the upload in row blocks that the pass driver
# follows (default, ABI 6: ststhip_set_source_arrival), alternating, same box.  -> profiles/r04_stream_upload.txt
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
EX="$REPO/build/examples"
W=/tmp/stst_stream; mkdir -p $W
python3 - <<PY
import numpy as np
n = 8192
np.full((n, n), 30.0, dtype=np.float32).tofile("$W/temp.bin")
p = np.zeros((n, n), dtype=np.float32); p[n//4-1:3*n//4, n//4-1:3*n//4] = 0.5; p.tofile("$W/power.bin")
PY
one() { # name cells binary args...
    name=$1; cells=$2; shift 2
    for rep in 1 2 3; do
        for mode in 0 1; do
            t=$(STSTHIP_STREAM_UPLOAD=$mode "$@" | grep Walltime | awk '{print $2}')
            python3 -c "import sys; print(f'$name stream_upload=$mode Walltime {float(sys.argv[1]):.4f} s  {float(sys.argv[2])/float(sys.argv[1])/1e9:.0f} Gcell-updates/s')" $t $cells
        done
    done
}
J="16384 16384 1000 /dev/null 0.2 0.2 0.2 0.2 0.2"
one jacobi5general 268435456000 $EX/jacobi_Jacobi5General_hip $J
one jacobi5general_fma 268435456000 $EX/jacobi_Jacobi5General_hip_fma $J
one hotspot 67108864000 $EX/hotspot_hip 8192 8192 1000 $W/temp.bin $W/power.bin /dev/null
one hotspot_aos 67108864000 $EX/hotspot_aos_hip 8192 8192 1000 $W/temp.bin $W/power.bin /dev/null
one hotspot_fma 67108864000 $EX/hotspot_hip_fma 8192 8192 1000 $W/temp.bin $W/power.bin /dev/null
