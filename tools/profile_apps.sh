#!/bin/bash
# rocprofv3 over every BASELINE GPU configuration (tools/bench_apps.py, full-grid launches): kernel-trace
# statistics and, in separate passes, the HBM counters.  Condensed by tools/summarize_apps_profile.py.
# usage (GPU box): tools/profile_apps.sh <tag>
TAG="${1:-r01_apps}"
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/prof_$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export STSTHIP_VIRTUAL_STRIPS=1   # one launch per pass, so a launch is the whole grid
APPS="jacobi jacobi_general hotspot hotspot_aos hotspot_f64 fdtd fdtd_aos conway"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$REPO/tools/bench_apps.py" $APPS > "$OUT/stats.log" 2>&1 < /dev/null
echo "stats pass done" >> "$OUT/progress.txt"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 "$REPO/tools/bench_apps.py" $APPS > "$OUT/fetch.log" 2>&1 < /dev/null
echo "fetch pass done" >> "$OUT/progress.txt"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 "$REPO/tools/bench_apps.py" $APPS > "$OUT/write.log" 2>&1 < /dev/null
echo "write pass done" >> "$OUT/progress.txt"
find "$OUT" -name "*kernel_trace.csv" -delete
python3 "$REPO/tools/summarize_apps_profile.py" "$TAG" < /dev/null
