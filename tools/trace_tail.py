"""Per-kernel average duration over the LAST `n` launches of each kernel in a rocprofv3 --kernel-trace CSV (the aged state of a run, not
its average).  python tools/trace_tail.py <kernel_trace.csv> [n]"""
import csv, sys, collections
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    d[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
rows = []
for k, v in d.items():
    v.sort()
    tail = [x[1] for x in v[-n:]]
    rows.append((sum(tail) / len(tail) / 1e3, len(v), k))
for us, cnt, k in sorted(rows, reverse=True)[:24]:
    print("%9.1f us  x%-5d %s" % (us, cnt, k[:110]))
