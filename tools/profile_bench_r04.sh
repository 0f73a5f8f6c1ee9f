#!/bin/bash
# Run on the GPU box (via gpurun).  Round 4's profile of what bench.py reports:
#  1. the command the driver times -- `python3 bench.py --gpus 1 --steps 20 --warmup 5`, the program directly after
#     `--` -- under rocprofv3 --kernel-trace --stats (kernel averages, the timed steps' span);
#  2. `python3 bench.py --profile-legs` -- every kernel leg of the bench line three times, nothing else -- under
#     --kernel-trace and then under one --pmc pass per counter group (FETCH_SIZE; WRITE_SIZE; SQ instruction counts;
#     the SQ wave-cycle breakdown), as MI355X_MICROARCH.md prescribes (counters in their own runs, never with a trace);
#  3. the FETCH_SIZE / WRITE_SIZE calibration copies (tools/microbench/fetch_calibration.hip) under the same two passes.
# tools/summarize_legs_profile.py condenses everything into the files that are committed under profiles/.
# usage: tools/profile_bench_r04.sh <tag> [skip-driver-command]
TAG="${1:-r04_bench}"
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/prof_$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
set -e
if [ "$2" != "skip-driver-command" ]; then
    rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$REPO/bench.py" --gpus 1 --steps 20 --warmup 5 > "$OUT/stats.log" 2> "$OUT/stats.err"
    tail -n 1 "$OUT/stats.log" | cut -c 1-300
    echo "driver command traced"
fi
rocprofv3 --kernel-trace --output-format csv -d "$OUT/legs_trace" -- python3 "$REPO/bench.py" --profile-legs > "$OUT/legs_trace.log" 2> "$OUT/legs_trace.err"
echo "legs traced"
# the depths the (all but unperturbed) traced run measured for the families that measure theirs: pinned for the counter passes
export STSTHIP_BENCH_LEG_DEPTHS="$(python3 - "$OUT/legs_trace.log" <<'PY'
import json, sys
for line in reversed(open(sys.argv[1]).read().splitlines()):
    if line.startswith("{") and '"profile_legs"' in line:
        print(",".join(f"{s['leg']}={s['depth_by_measurement']}" for s in json.loads(line)["profile_legs"] if s.get("depth_by_measurement")))
        break
PY
)"
echo "depths pinned for the counter passes: $STSTHIP_BENCH_LEG_DEPTHS"
for PASS in "fetch FETCH_SIZE" "write WRITE_SIZE" "sq SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS" \
            "sqwait SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU"; do
    set -- $PASS; NAME=$1; shift
    rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$NAME" -- python3 "$REPO/bench.py" --profile-legs > "$OUT/$NAME.log" 2> "$OUT/$NAME.err"
    echo "pmc pass $NAME done"
done
if [ -x "$REPO/build/fetch_calibration" ]; then
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/cal_fetch" -- "$REPO/build/fetch_calibration" > "$OUT/cal_fetch.log" 2> "$OUT/cal_fetch.err"
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/cal_write" -- "$REPO/build/fetch_calibration" > "$OUT/cal_write.log" 2> "$OUT/cal_write.err"
    echo "calibration passes done"
fi
python3 "$REPO/tools/summarize_legs_profile.py" "$TAG"
# the raw traces are large; the condensed files are what is kept
find "$OUT" -name "*kernel_trace.csv" -size +8M -delete
find "$OUT" -name "*counter_collection.csv" -size +8M -delete
