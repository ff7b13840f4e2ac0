#!/usr/bin/env python3
"""Diagnostic: per-CU timeline statistics from gpurun_out/rb_stamps.csv (tools/rank_bw_bench.hip -DMI32_RB_STAMPS)."""
import csv
import sys
from collections import defaultdict

import numpy as np

rows = list(csv.DictReader(open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/rb_stamps.csv")))
t = np.array([[int(r["t0"]), int(r["t1"]), int(r["t2"]), int(r["t3"])] for r in rows], dtype=np.int64)
hw = np.array([int(r["hwid"]) for r in rows])
xcc = np.array([int(r["xcc"]) for r in rows])
keep = t[:, 0] > 0
t, hw, xcc = t[keep], hw[keep], xcc[keep]
t0 = t[:, 0].min()
t -= t0
span = t[:, 3].max()
# HW_ID: bits 8-11 cu_id, 12-14 sh_id?, 15-17 se_id (gfx9); use (xcc, hwid & 0x3ff00 >> 8) as the CU key
cu = xcc * 256 + ((hw >> 8) & 0xFF)  # cu_id[11:8], sh_id[12], se_id[15:13]
print(f"{len(rows)} workgroups on {len(set(cu))} CUs, launch span {span} cycles")
print("per-WG medians: prologue %d  k-loop %d  epilogue %d" % (np.median(t[:, 1] - t[:, 0]), np.median(t[:, 2] - t[:, 1]),
                                                             np.median(t[:, 3] - t[:, 2])))
per_cu = defaultdict(list)
for i, c in enumerate(cu):
    per_cu[c].append(t[i])
cnt = np.array([len(v) for v in per_cu.values()])
print("workgroups per CU: min %d median %d max %d" % (cnt.min(), np.median(cnt), cnt.max()))
# fraction of the launch span during which at least one / how many WGs of a CU are inside their k-loop
G = 512
grid = np.linspace(0, span, G)
busy_any, conc = [], []
for v in per_cu.values():
    v = np.array(v)
    k = ((grid[:, None] >= v[None, :, 1]) & (grid[:, None] < v[None, :, 2])).sum(axis=1)
    busy_any.append((k > 0).mean())
    conc.append(k.mean())
print("fraction of span with >=1 WG in its k-loop, per CU: min %.2f median %.2f max %.2f" %
      (min(busy_any), np.median(busy_any), max(busy_any)))
print("mean concurrent k-loops per CU: %.2f" % np.mean(conc))
# chip-wide: how many WGs are in their epilogue at each time
ep = ((grid[:, None] >= t[None, :, 2]) & (grid[:, None] < t[None, :, 3])).sum(axis=1)
kl = ((grid[:, None] >= t[None, :, 1]) & (grid[:, None] < t[None, :, 2])).sum(axis=1)
print("chip-wide WGs in epilogue over time (16 buckets):", [int(x) for x in ep.reshape(16, -1).mean(axis=1)])
print("chip-wide WGs in k-loop  over time (16 buckets):", [int(x) for x in kl.reshape(16, -1).mean(axis=1)])
# the slowest CU
last = max(per_cu.items(), key=lambda kv: max(x[3] for x in kv[1]))
print("last CU to finish runs %d workgroups:" % len(last[1]))
for x in sorted(last[1], key=lambda x: x[0]):
    print("   start %7d  kloop %7d..%7d  end %7d" % tuple(x))
