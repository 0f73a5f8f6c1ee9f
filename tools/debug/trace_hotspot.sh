#!/bin/bash
# GPU timeline of one run of the unchanged hotspot example (planes): upload, scatter, sweeps, gather.
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/r3h"; rm -rf "$OUT"; mkdir -p "$OUT"
W=/tmp/stst_examples; mkdir -p $W
python3 - <<PY
import numpy as np
n = 8192
np.full((n, n), 30.0, dtype=np.float32).tofile("$W/temp.bin")
p = np.zeros((n, n), dtype=np.float32); p[n//4-1:3*n//4, n//4-1:3*n//4] = 0.5; p.tofile("$W/power.bin")
PY
cd /tmp && export TMPDIR=/tmp
"$REPO/build/examples/hotspot_hip" 8192 8192 1000 $W/temp.bin $W/power.bin /dev/null | grep Walltime
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$OUT/trace" -- "$REPO/build/examples/hotspot_hip" 8192 8192 1000 $W/temp.bin $W/power.bin /dev/null > "$OUT/run.log" 2>&1
grep Walltime "$OUT/run.log"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
k = glob.glob(out + "/trace/*/*kernel_trace.csv")[0]; m = glob.glob(out + "/trace/*/*memory_copy_trace.csv")[0]
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(k))]
ms = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction") or r.get("Name"), r.get("Bytes") or "") for r in csv.DictReader(open(m))]
t0 = min(ms[0][0], ks[0][0])
for a, b, d, n in ms:
    print("copy", d, round((a - t0) / 1e6, 2), "->", round((b - t0) / 1e6, 2), "ms")
ks.sort()
print("first kernel", round((ks[0][0] - t0) / 1e6, 2), "last kernel end", round((max(b for a, b, _ in ks) - t0) / 1e6, 2), "n kernels", len(ks))
by = collections.defaultdict(lambda: [0, 0.0, 1e30, 0])
for a, b, n in ks:
    e = by[n[:90]]; e[0] += 1; e[1] += (b - a) / 1e6; e[2] = min(e[2], a); e[3] = max(e[3], b)
for n, (c, t, a, b) in sorted(by.items(), key=lambda kv: kv[1][2]):
    print(f"{c:5d} x {t:8.2f} ms  [{(a - t0) / 1e6:8.2f} .. {(b - t0) / 1e6:8.2f}]  {n}")
PY
