#!/bin/bash
# kernel trace of a small grid through the pass driver: how much of the span is launch gap
SIZE=$1; shift
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/trace_small_$SIZE"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
rocprofv3 --kernel-trace --output-format csv -d "$OUT" -- python3 "$REPO/bench.py" --size $SIZE --steps 5 --warmup 1 --no-cpu-baseline --no-verify > "$OUT/log.txt" 2>&1
tail -1 "$OUT/log.txt" | cut -c 1-160
python3 "$REPO/tools/trace_gaps.py" "$OUT" 100
