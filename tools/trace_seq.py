"""The kernel sequence of the last `n` launches of a rocprofv3 --kernel-trace CSV: start offset, duration, gap to the previous kernel's
end, grid, name.  python tools/trace_seq.py <kernel_trace.csv> [n]"""
import csv, sys
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Grid_Size_X", "?"), r.get("Workgroup_Size_X", "?"),
                 r.get("LDS_Block_Size", "?"), r.get("VGPR_Count", "?"), r.get("Scratch_Size", "?")))
rows.sort()
tail = rows[-n:]
t0 = tail[0][0]
prev = None
busy = 0
for s, e, k, g, w, lds, vg, sc in tail:
    gap = (s - prev) / 1e3 if prev is not None else 0.0
    busy += e - s
    print("%9.1f us  dur %8.1f  gap %6.1f  grid %8s wg %5s lds %6s vgpr %4s scr %4s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, g, w, lds, vg, sc, k[:100]))
    prev = e
print("span %.1f us, busy %.1f us" % ((tail[-1][1] - t0) / 1e3, busy / 1e3))
