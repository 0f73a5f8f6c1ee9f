import csv, glob, os, sys
out = sys.argv[1]
k = sorted(glob.glob(out + "/trace/*/*kernel_trace.csv"), key=os.path.getmtime)[-1]
m = sorted(glob.glob(out + "/trace/*/*memory_copy_trace.csv"), key=os.path.getmtime)[-1]
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Grid_Size_X"])) for r in csv.DictReader(open(k)) if "sweep" in r["Kernel_Name"])
ms = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(m))]
big = [x for x in ms if x[1]-x[0] > 200000]
t0 = big[0][0]
print("copies:", " ".join(f"{(b-t0)/1e6:.2f}" for a,b in big))
# union busy
busy=0; cur_a,cur_b=ks[0][0],ks[0][1]; gaps=[]
for a,b,g in ks[1:]:
    if a>cur_b:
        busy+=cur_b-cur_a; gaps.append((cur_b,a)); cur_a,cur_b=a,b
    else: cur_b=max(cur_b,b)
busy+=cur_b-cur_a
print(f"first kernel {(ks[0][0]-t0)/1e6:.2f} ms, last end {(cur_b-t0)/1e6:.2f} ms, busy union {busy/1e6:.2f} ms, n {len(ks)}")
print("gaps > 30us:", " ".join(f"{(a-t0)/1e6:.2f}+{(b-a)/1e3:.0f}us" for a,b in gaps if b-a>30000)[:1500])
# per 2ms window: sum of kernel time (concurrency-weighted) 
import collections
w=collections.Counter()
for a,b,g in ks:
    w[int((a-t0)/2e6)]+= (b-a)
print("kernel-time per 2 ms window:", " ".join(f"{i*2}:{v/2e6:.2f}" for i,v in sorted(w.items())))
