#!/bin/bash
# GPU timeline of one run of the unchanged jacobi example: when the upload runs, when the sweeps run.
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/r3g"; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$OUT/trace" -- "$REPO/build/examples/jacobi_Jacobi5General_hip" 16384 16384 1000 /dev/null 0.2 0.2 0.2 0.2 0.2 > "$OUT/run.log" 2>&1
grep Walltime "$OUT/run.log"
python3 - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
k = glob.glob(out + "/trace/*/*kernel_trace.csv")[0]; m = glob.glob(out + "/trace/*/*memory_copy_trace.csv")[0]
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:50]) for r in csv.DictReader(open(k))]
ms = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction") or r.get("Name"), r.get("Bytes") or "") for r in csv.DictReader(open(m))]
t0 = min(ms[0][0], ks[0][0])
for a, b, d, n in ms:
    print("copy", d, round((a - t0) / 1e6, 2), "->", round((b - t0) / 1e6, 2), "ms")
ks.sort()
print("first kernel", round((ks[0][0] - t0) / 1e6, 2), "last kernel end", round((max(b for a, b, _ in ks) - t0) / 1e6, 2), "n kernels", len(ks))
print("sum kernel ms", round(sum(b - a for a, b, _ in ks) / 1e6, 2))
PY
