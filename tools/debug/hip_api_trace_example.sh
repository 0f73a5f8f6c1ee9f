REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/hiptrace"; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --hip-trace --output-format csv -d "$OUT/trace" -- "$REPO/build/examples/jacobi_Jacobi5General_hip" 16384 16384 1000 /dev/null 0.2 0.2 0.2 0.2 0.2 > "$OUT/run.log" 2>&1
grep Walltime "$OUT/run.log"
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/trace/*/*hip_api_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
print(rows[0].keys())
calls = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"]) for r in rows]
calls.sort()
# find the first big H2D async copy
idx = [i for i, c in enumerate(calls) if c[2] == "hipMemcpyAsync"]
big = idx[2] if len(idx) > 2 else idx[0]
t0 = calls[big][0]
for a, b, fn in calls[max(0, big - 25): big + 60]:
    print(f"{(a - t0) / 1e6:9.3f} ms  {(b - a) / 1e3:9.1f} us  {fn}")
PY
