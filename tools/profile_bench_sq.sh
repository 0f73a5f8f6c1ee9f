#!/bin/bash
# Where the waves of the bench kernel spend their cycles: SQ counters over the full-grid launches of bench.py's
# roofline leg (grid 721408), one --pmc pass.  usage (GPU box): tools/profile_bench_sq.sh
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/prof_bench_sq"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export STSTHIP_BENCH_MINIMAL=1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES \
    --output-format csv -d "$OUT/sq" -- python3 "$REPO/bench.py" --gpus 1 --steps 2 --warmup 1 --no-verify > "$OUT/sq.log" 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM \
    --output-format csv -d "$OUT/sq2" -- python3 "$REPO/bench.py" --gpus 1 --steps 2 --warmup 1 --no-verify > "$OUT/sq2.log" 2>&1
python3 - "$OUT" <<'PY'
import collections, csv, glob, json, sys
out = sys.argv[1]
res = {}
for kind in ("sq", "sq2"):
    for f in glob.glob(f"{out}/{kind}/*/*counter_collection.csv"):
        vals = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            grid = int(r.get("Grid_Size_X") or r.get("Grid_Size") or 0)
            if "Jacobi5Uniform<false, false>, false, 12," in r["Kernel_Name"] and grid == 721408:
                vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
                vals["_ns_" + kind].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        for k, v in vals.items():
            res[k] = sum(v) / len(v)
if "SQ_WAVE_CYCLES" in res:
    w = res["SQ_WAVE_CYCLES"]
    res["frac_wait_any"] = res["SQ_WAIT_ANY"] / w
    res["frac_wait_inst_any"] = res["SQ_WAIT_INST_ANY"] / w
    res["frac_active_inst_any"] = res["SQ_ACTIVE_INST_ANY"] / w
print(json.dumps(res, indent=1))
json.dump(res, open(f"{out}/summary.json", "w"), indent=1)
PY
