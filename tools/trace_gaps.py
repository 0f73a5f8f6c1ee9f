"""Condense a rocprofv3 --kernel-trace CSV: per launch shape the count and mean duration, and how much of the span
between the first and the last launch no kernel was running (launch gaps / dependency latency).
usage: python tools/trace_gaps.py <dir or kernel_trace.csv> [skip_first_n_launches]"""
import csv
import glob
import os
import sys


def main():
    path = sys.argv[1]
    skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    if os.path.isdir(path):
        path = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))[-1]
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"]
            if "sweep_kernel" not in name:
                continue
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r["Grid_Size"]),
                         r.get("Queue_Id", "")))
    rows.sort()
    rows = rows[skip:]
    shapes = {}
    for s, e, g, q in rows:
        c = shapes.setdefault(g, [0, 0])
        c[0] += 1
        c[1] += e - s
    span = rows[-1][1] - rows[0][0]
    # union of busy intervals
    busy, cur_s, cur_e = 0, rows[0][0], rows[0][1]
    for s, e, g, q in rows[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    print(f"{path}: {len(rows)} sweep launches, span {span / 1e6:.3f} ms, some kernel running {busy / 1e6:.3f} ms "
          f"({100.0 * busy / span:.1f} %), idle {100.0 * (span - busy) / span:.1f} %")
    for g, (n, t) in sorted(shapes.items(), key=lambda kv: -kv[1][1]):
        print(f"  grid {g:>9}: {n:>5} launches, mean {t / n / 1e3:8.1f} us, total {t / 1e6:8.3f} ms")


main()
