python3 -c "
import numpy as np
n=8192
import os; os.makedirs('/tmp/stst_stream',exist_ok=True)
np.full((n,n),30.0,dtype=np.float32).tofile('/tmp/stst_stream/temp.bin')
p=np.zeros((n,n),dtype=np.float32); p[n//4-1:3*n//4,n//4-1:3*n//4]=0.5; p.tofile('/tmp/stst_stream/power.bin')
"
export STSTHIP_TRACE_STREAM=1
H="8192 8192 1000 /tmp/stst_stream/temp.bin /tmp/stst_stream/power.bin /dev/null"
for b in hotspot_hip hotspot_hip hotspot_aos_hip hotspot_aos_hip; do echo == $b; build/examples/$b $H 2>&1 | grep "ststhip\|Wall"; done
J="16384 16384 1000 /dev/null 0.2 0.2 0.2 0.2 0.2"
for b in jacobi_Jacobi5General_hip jacobi_Jacobi5General_hip jacobi_Jacobi5General_hip_fma jacobi_Jacobi5General_hip_fma; do echo == $b; build/examples/$b $J 2>&1 | grep "ststhip\|Wall"; done
