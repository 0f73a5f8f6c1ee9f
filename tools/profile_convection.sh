#!/bin/bash
# rocprofv3 passes over the convection example with the device-side convergence check (res = 1024, 2 x 1000
# pseudo-transient iterations): kernel trace + PMC (SQ instruction counts and wave cycles; FETCH_SIZE; WRITE_SIZE), one
# pass each.  Writes gpurun_out/prof_convection/summary.json (copy to profiles/r03_pmc_convection.json).
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/prof_convection"; rm -rf "$OUT"; mkdir -p "$OUT/w"
BIN="$REPO/build/examples/convection_reduce_hip"; CFG="$REPO/tools/data/convection_bench.json"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- "$BIN" "$CFG" "$OUT/w" > "$OUT/stats.log" 2>&1
for PASS in "sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" "fetch FETCH_SIZE" "write WRITE_SIZE" "lds SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS"; do
  set -- $PASS; NAME=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$NAME" -- "$BIN" "$CFG" "$OUT/w" > "$OUT/$NAME.log" 2>&1
  echo "pass $NAME done"
done
python3 - "$OUT" <<'PY'
import collections, csv, glob, json, sys
out = sys.argv[1]
res = collections.defaultdict(dict)
for kind in ("sq", "fetch", "write", "lds"):
    for f in glob.glob(f"{out}/{kind}/*/*counter_collection.csv"):
        per = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            if "sweep_kernel" not in name:
                continue
            key = ("PseudoTransientKernel" if "PseudoTransient" in name else "ThermalSolverKernel" if "ThermalSolver" in name else "other") + \
                  ":grid" + str(int(r.get("Grid_Size_X") or r.get("Grid_Size") or 0))
            per[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            per[key]["_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size"):
                per[key]["_" + k] = [float(r.get(k) or 0)]
            per[key]["_kernel"] = name[:200]
        for key, vals in per.items():
            for k, v in vals.items():
                if k == "_kernel":
                    res[key]["kernel"] = v
                elif k == "_ns":
                    res[key]["avg_us_" + kind] = sum(v) / len(v) / 1e3
                    res[key]["launches_" + kind] = len(v)
                else:
                    res[key][k.lstrip("_")] = sum(v) / len(v)
for key, d in res.items():
    if "SQ_WAVE_CYCLES" in d:
        w = d["SQ_WAVE_CYCLES"]
        d["frac_parked"] = d["SQ_WAIT_ANY"] / w
        d["frac_waiting_to_issue"] = d["SQ_WAIT_INST_ANY"] / w
        d["frac_issuing"] = d["SQ_ACTIVE_INST_ANY"] / w
        # fp64 vector instructions issue at half the fp32 rate: the peak used here is the fp32 one (1.09 ns), so this
        # fraction under-states how busy the VALU is for an fp64 kernel by up to a factor of two
        d["valu_issue_fraction_fp32_peak"] = d["SQ_INSTS_VALU"] * 1.09e-9 / 1024 / (d["avg_us_sq"] * 1e-6)
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        d["hbm_bytes_per_launch"] = 2 * d["FETCH_SIZE"] * 1024 + d["WRITE_SIZE"] * 1024
        d["hbm_TBps"] = d["hbm_bytes_per_launch"] / (d.get("avg_us_fetch", 1) * 1e-6) / 1e12
json.dump(res, open(f"{out}/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1)[:6000])
PY
grep -h "Total time\|transient" "$OUT/stats.log" | tail -2
