#!/bin/bash
# GPU timeline of one run of an unchanged example with the upload in row blocks: when each block arrives, which
# tiles (rows = grid size / waves) run when, on how many streams side by side.
# usage: trace_stream_upload.sh <tag> <binary> <args...>
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
TAG=$1; shift
OUT="$REPO/gpurun_out/$TAG"; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$OUT/trace" -- "$@" > "$OUT/run.log" 2>&1
grep Walltime "$OUT/run.log"
python3 "$REPO/tools/debug/timeline_summary.py" "$OUT"
