#!/bin/bash
# SQ counters of one registered sweep (AB_ONLY) on one grid shape, single launches per pass.
# usage (GPU box): tools/profile_staged_sq.sh <family> <app> <rows>x<cols> <tag>
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
FAMILY="$1"; APP="$2"; SHAPE="$3"; TAG="$4"
OUT="$REPO/gpurun_out/prof_staged_$TAG"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export AB_ONLY="$APP" STSTHIP_VIRTUAL_STRIPS=1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES \
    --output-format csv -d "$OUT/sq" -- python3 "$REPO/tools/ab_staged.py" "$FAMILY" "$SHAPE" > "$OUT/sq.log" 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR \
    --output-format csv -d "$OUT/sq2" -- python3 "$REPO/tools/ab_staged.py" "$FAMILY" "$SHAPE" > "$OUT/sq2.log" 2>&1
rocprofv3 --pmc FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS \
    --output-format csv -d "$OUT/sq3" -- python3 "$REPO/tools/ab_staged.py" "$FAMILY" "$SHAPE" > "$OUT/sq3.log" 2>&1
python3 - "$OUT" <<'PY'
import collections, csv, glob, json, sys
out = sys.argv[1]
res = {}
for kind in ("sq", "sq2", "sq3"):
    for f in glob.glob(f"{out}/{kind}/*/*counter_collection.csv"):
        per = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            if "sweep_kernel" not in r["Kernel_Name"]:
                continue
            key = (r["Kernel_Name"][:0] + str(int(r.get("Grid_Size_X") or r.get("Grid_Size") or 0)))
            per[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            per[key]["_ns_" + kind].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            per[key]["_vgpr"].append(float(r.get("VGPR_Count") or 0))
            per[key]["_lds"].append(float(r.get("LDS_Block_Size") or 0))
        for key, vals in per.items():
            d = res.setdefault(key, {})
            for k, v in vals.items():
                d[k] = sum(v) / len(v)
            d["_launches_" + kind] = len(vals["_ns_" + kind])
for key, d in res.items():
    if "SQ_WAVE_CYCLES" in d:
        w = d["SQ_WAVE_CYCLES"]
        d["frac_wait_any"] = d["SQ_WAIT_ANY"] / w
        d["frac_wait_inst_any"] = d["SQ_WAIT_INST_ANY"] / w
        d["frac_active_inst_any"] = d["SQ_ACTIVE_INST_ANY"] / w
print(json.dumps(res, indent=1))
json.dump(res, open(f"{out}/summary.json", "w"), indent=1)
PY
