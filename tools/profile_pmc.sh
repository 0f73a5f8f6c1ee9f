#!/bin/bash
# One rocprofv3 PMC pass (counters in $2...) over a short bench run; output in gpurun_out/pmc_<tag>/.
TAG="$1"; shift
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/pmc_$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && export STSTHIP_BENCH_MINIMAL=1
rocprofv3 --pmc "$@" --output-format csv -d "$OUT" -- python3 "$REPO/bench.py" --steps 1 --warmup 1 --generations 120 --no-cpu-baseline --single-strip > "$OUT/run.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")
if not f:
    print("no counter file"); print(open(sys.argv[1] + "/run.log").read()[-2000:]); sys.exit(0)
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if "sweep_kernel" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        agg["_duration_ns"].append(dur)
for k, v in sorted(agg.items()):
    print(f"{k:28s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
PY
