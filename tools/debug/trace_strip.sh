#!/bin/bash
# kernel timeline of one rank's compute side (2048-row strip, m = 4): launch durations and the gaps between them
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/trace_strip"; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT/t" -- python3 "$REPO/tools/bench_strip.py" --rows ${1:-2048} --exchange-every 4 --reps 2 > "$OUT/run.log" 2>&1
grep rows_per_gpu "$OUT/run.log" | cut -c1-160
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/t/*/*kernel_trace.csv")[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r.get("Grid_Size_X") or r.get("Grid_Size")), r["Kernel_Name"][:60]) for r in csv.DictReader(open(f)) if "sweep_kernel" in r["Kernel_Name"]]
rows.sort()
rows = rows[-141 * 2:]         # the last two steps (141 launches each with two sub-strips)
shapes = collections.Counter(g for _, _, g, _ in rows)
big = max(shapes, key=lambda g: (shapes[g], g))
print("launch shapes", dict(shapes))
durs = collections.defaultdict(list)
for s, e, g, _ in rows: durs[g].append((e - s) / 1e3)
for g, d in durs.items(): print(f"grid {g}: n={len(d)} mean {sum(d)/len(d):.1f} us  min {min(d):.1f} max {max(d):.1f}")
# gaps between consecutive launches in time order (whatever the stream)
gaps = []
cur_e = rows[0][1]
for s, e, g, _ in rows[1:]:
    if s > cur_e: gaps.append((s - cur_e) / 1e3)
    cur_e = max(cur_e, e)
span = (rows[-1][1] - rows[0][0]) / 1e3
print(f"span {span:.1f} us for {len(rows)} launches; idle gaps: n={len(gaps)} sum {sum(gaps):.1f} us mean {sum(gaps)/max(len(gaps),1):.2f} us max {max(gaps):.1f}")
PY
python3 - "$OUT" <<'PY'
# one group of launches in time order: start, end, rows (grid), stream
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/t/*/*kernel_trace.csv")[0]
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Grid_Size_X"]), r["Stream_Id"]) for r in csv.DictReader(open(f)) if "sweep_kernel" in r["Kernel_Name"])
rows = rows[-141:]
t0 = rows[40][0]
for s, e, g, st in rows[40:62]:
    print(f"  {(s - t0) / 1e3:8.1f} -> {(e - t0) / 1e3:8.1f} us  {(e - s) / 1e3:6.1f} us  grid {g:7d}  stream {st}")
PY
