#!/usr/bin/env python3
"""Last step of a rocprofv3 --kernel-trace csv: start, end, gap to the previous kernel (us), queue, kernel."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = sys.argv[2] if len(sys.argv) > 2 else "k_morton"
idx = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]][-2]
t0 = int(rows[idx - 1]["Start_Timestamp"]) if idx > 0 else int(rows[idx]["Start_Timestamp"])
prev_end = None
for r in rows[idx - 1: idx + int(sys.argv[3]) if len(sys.argv) > 3 else idx + 14]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = "" if prev_end is None else "%6.1f" % ((s - prev_end) / 1e3)
    print("%8.1f %8.1f  gap %6s  q%s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, gap, r.get("Queue_Id", "?"), r["Kernel_Name"][:56]))
    prev_end = e
