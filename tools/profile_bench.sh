#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats and, in separate passes, the HBM counters
# for the bench command.  Outputs under gpurun_out/prof_<tag>/.
# usage: tools/profile_bench.sh <tag> [bench args...]
TAG="$1"; shift
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/prof_$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --single-strip $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$REPO/bench.py" $ARGS > "$OUT/stats.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 "$REPO/bench.py" $ARGS > "$OUT/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 "$REPO/bench.py" $ARGS > "$OUT/write.log" 2>&1
find "$OUT" -name "*.csv" | head -20
# keep only what is needed (the traces can be large)
find "$OUT" -name "*kernel_trace.csv" -size +20M -delete
ls -la "$OUT"/*/* 2>/dev/null | head -30
