#!/usr/bin/env python3
"""Diagnostic: print the kernels of a rocprofv3 kernel trace (csv) in start order with their queue, for a window of
the LAST timed step -- which launches overlap which.

    python tools/trace_timeline.py <dir with *_kernel_trace.csv> [first_launch [count]]
"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last step = after the last init kernel
last_init = max(i for i, r in enumerate(rows) if "init_kernel" in r["Kernel_Name"])
rows = rows[last_init:]
t0 = int(rows[0]["Start_Timestamp"])
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
count = int(sys.argv[3]) if len(sys.argv) > 3 else 80
queues = {}
for r in rows[first:first + count]:
    q = queues.setdefault(r["Queue_Id"], len(queues))
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    name = r["Kernel_Name"].split("(")[0].replace("mi32::", "").replace("void ", "")[:44]
    print("%9.1f %9.1f %8.1f us  q%d %s%-44s grid %s" % (s, e, e - s, q, "    " * q, name, r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", "")))
