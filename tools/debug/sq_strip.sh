#!/bin/bash
# SQ counters of the uniform Jacobi form on a thin strip (strip driver, m = 4) and on the full grid (bench.py), one pass each
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/sq_strip"; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES"
timeout -k 10 280 rocprofv3 --pmc $C --output-format csv -d "$OUT/strip" -- python3 "$REPO/tools/bench_strip.py" --rows 2048 --exchange-every 4 --reps 1 > "$OUT/strip.log" 2>&1 || echo "strip pass failed"
STSTHIP_BENCH_MINIMAL=1 timeout -k 10 280 rocprofv3 --pmc $C --output-format csv -d "$OUT/full" -- python3 "$REPO/bench.py" --gpus 1 --steps 2 --warmup 1 --no-verify > "$OUT/full.log" 2>&1 || echo "full pass failed"
python3 - "$OUT" <<'PY'
import collections, csv, glob, json, sys
out = sys.argv[1]
for kind in ("strip", "full"):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{out}/{kind}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "sweep_kernel" not in r["Kernel_Name"]:
                continue
            g = int(r.get("Grid_Size_X") or r.get("Grid_Size") or 0)
            per[g][r["Counter_Name"]].append(float(r["Counter_Value"]))
            per[g]["ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for g, v in sorted(per.items()):
        n = len(v["SQ_WAVE_CYCLES"])
        if n < 4: continue
        m = {k: sum(x) / len(x) for k, x in v.items()}
        w = m["SQ_WAVE_CYCLES"]
        print(f"{kind} grid {g}: launches {n}, {m['ns']/1e3:.1f} us (under the profiler), waves {m['SQ_WAVES']:.0f}, VALU insts {m['SQ_INSTS_VALU']:.4g}, "
              f"wave cycles {w:.4g}: waiting {m['SQ_WAIT_ANY']/w:.3f}, waiting to issue {m['SQ_WAIT_INST_ANY']/w:.3f}, issuing {m['SQ_ACTIVE_INST_ANY']/w:.3f} (VALU {m['SQ_ACTIVE_INST_VALU']/w:.3f}); "
              f"busy cycles {m['SQ_BUSY_CYCLES']:.4g}")
PY
