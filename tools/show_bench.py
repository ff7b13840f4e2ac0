#!/usr/bin/env python3
"""Diagnostic: print the per-kernel breakdown of bench.py JSON lines (files given on the command line)."""
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "ERR", e); continue
    print(f, "ms/step %.3f" % d["ms_per_step"], "value %.1f" % d["value"], "res %.1e" % d["residual_inf"], d["config"]["blocking"])
    for k, v in (d.get("kernel_breakdown") or {}).items():
        if k.startswith("_"): continue
        print("    %-16s %8.3f ms/step  %7.1f launches  %8.2f us avg" % (k, v["ms_per_step"], v["launches_per_step"], v["avg_us"]))
    r = d.get("roofline")
    if r: print("    roofline %.1f %s frac %.3f" % (r["achieved"], r["unit"], r["frac"]))
