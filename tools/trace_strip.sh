#!/bin/bash
# kernel trace of one strip of a multi-GPU run on one GPU: tools/trace_strip.sh <rows> [ENV=...]
ROWS=$1; shift
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/trace_strip_$ROWS"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
export STSTHIP_STRIP_DEBUG_BANDS=1
rocprofv3 --kernel-trace --output-format csv -d "$OUT" -- python3 "$REPO/bench.py" --strip-domain --rows-per-gpu $ROWS --steps 2 --warmup 1 --no-cpu-baseline --no-verify > "$OUT/log.txt" 2>&1
python3 "$REPO/tools/trace_gaps.py" "$OUT" 300
