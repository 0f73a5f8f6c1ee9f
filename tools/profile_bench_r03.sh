#!/bin/bash
# Run on the GPU box (via gpurun).  Profiles the command the driver times -- `python3 bench.py --gpus 1 --steps 20
# --warmup 5`, the program directly after `--` -- with rocprofv3 --kernel-trace --stats, then collects the counters
# of the same program in separate --pmc passes (FETCH_SIZE; WRITE_SIZE; SQ instruction counts), as
# MI355X_MICROARCH.md prescribes.  tools/summarize_bench_profile.py condenses the outputs into the files that are
# committed under profiles/.
# usage: tools/profile_bench_r03.sh <tag> [extra bench args...]
TAG="${1:-r03_bench}"; shift
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/prof_$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
set -e
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$REPO/bench.py" --gpus 1 --steps 20 --warmup 5 "$@" > "$OUT/stats.log" 2> "$OUT/stats.err"
tail -n 1 "$OUT/stats.log" | cut -c 1-400
export STSTHIP_BENCH_MINIMAL=1
for PASS in "fetch FETCH_SIZE" "write WRITE_SIZE" "sq SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES"; do
    set -- $PASS; NAME=$1; shift
    rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$NAME" -- python3 "$REPO/bench.py" --gpus 1 --steps 2 --warmup 1 --no-verify > "$OUT/$NAME.log" 2> "$OUT/$NAME.err"
    echo "pmc pass $NAME done"
done
python3 "$REPO/tools/summarize_bench_profile.py" "$TAG"
# the raw traces are large; the condensed files are what is kept
find "$OUT" -name "*kernel_trace.csv" -size +8M -delete
find "$OUT" -name "*counter_collection.csv" -size +8M -delete
