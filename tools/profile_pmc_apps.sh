#!/bin/bash
# rocprofv3 PMC passes over tools/ab_coop.py <families>: where the waves of each sweep kernel spend their cycles
# (SQ_WAIT_ANY = parked at s_waitcnt / s_barrier, SQ_WAIT_INST_ANY = issue stalls, SQ_ACTIVE_INST_ANY = issuing),
# VALU instruction counts, and the HBM bytes (FETCH_SIZE, WRITE_SIZE in separate passes).
# usage (GPU box): tools/profile_pmc_apps.sh <tag> <family> [family...]
TAG="$1"; shift
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/pmc_$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export STSTHIP_VIRTUAL_STRIPS=1   # one launch per pass, so a launch is the whole grid
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES \
    --output-format csv -d "$OUT/sq" -- python3 "$REPO/tools/ab_coop.py" "$@" > "$OUT/sq.log" 2>&1 < /dev/null
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR \
    --output-format csv -d "$OUT/sq2" -- python3 "$REPO/tools/ab_coop.py" "$@" > "$OUT/sq2.log" 2>&1 < /dev/null
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 "$REPO/tools/ab_coop.py" "$@" > "$OUT/fetch.log" 2>&1 < /dev/null
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 "$REPO/tools/ab_coop.py" "$@" > "$OUT/write.log" 2>&1 < /dev/null
python3 - "$OUT" <<'PY'
import collections, csv, glob, json, sys
out = sys.argv[1]
table = collections.defaultdict(dict)
for kind in ("sq", "sq2", "fetch", "write"):
    for f in glob.glob(f"{out}/{kind}/*/*counter_collection.csv"):
        vals = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            if "sweep_kernel" not in r["Kernel_Name"]:
                continue
            grid = int(r.get("Grid_Size_X") or r.get("Grid_Size") or 0)
            vals[(r["Kernel_Name"], grid)][r["Counter_Name"]].append(float(r["Counter_Value"]))
            vals[(r["Kernel_Name"], grid)]["_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            vals[(r["Kernel_Name"], grid)]["_vgpr"] = [float(r.get("VGPR_Count") or 0)]
            vals[(r["Kernel_Name"], grid)]["_lds"] = [float(r.get("LDS_Block_Size") or 0)]
        # per kernel name: the launch shape with the largest grid (full-grid launches of the deepest kernel)
        best = {}
        for (name, grid), c in vals.items():
            if name not in best or grid > best[name][0]:
                best[name] = (grid, c)
        for name, (grid, c) in best.items():
            short = name.split("Sweep<")[1][:110] if "Sweep<" in name else name[:110]
            for k, v in c.items():
                if k == "_ns":
                    table[short][f"us_under_{kind}"] = round(sum(v) / len(v) / 1e3, 1)
                else:
                    table[short][k.lstrip("_")] = sum(v) / len(v)
            table[short]["grid"] = grid
for name, c in table.items():
    if "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"]:
        w = c["SQ_WAVE_CYCLES"]
        c["frac_wait_any"] = round(c.get("SQ_WAIT_ANY", 0) / w, 3)
        c["frac_wait_inst"] = round(c.get("SQ_WAIT_INST_ANY", 0) / w, 3)
        c["frac_active"] = round(c.get("SQ_ACTIVE_INST_ANY", 0) / w, 3)
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        c["hbm_read_MB"] = round(2 * c["FETCH_SIZE"] * 1024 / 1e6, 1)
        c["hbm_write_MB"] = round(c["WRITE_SIZE"] * 1024 / 1e6, 1)
json.dump(table, open(f"{out}/summary.json", "w"), indent=1)
for name, c in table.items():
    print(name)
    print("   ", {k: (round(v, 1) if isinstance(v, float) else v) for k, v in sorted(c.items())})
PY
find "$OUT" -name "*counter_collection.csv" -size +8M -delete
