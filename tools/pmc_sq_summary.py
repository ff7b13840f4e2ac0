#!/usr/bin/env python3
"""Summarise the SQ counter pass of tools/collect_profiles.sh into profiles/roundN/pmc_sq_summary.json.

    tools/pmc_sq_summary.py gpurun_out/prof profiles/round2/pmc_sq_summary.json

Per configuration and kernel class: per-launch means of the counters and three derived fractions (see `note`)."""
import collections
import csv
import glob
import json
import os
import sys

src, out = sys.argv[1], sys.argv[2]
CLASSES = ["gj_subpanel_kernel", "gj_inblock_update_kernel", "gj_rank_bw2_kernel"]
RUNS = {"c1": "C1_n4096_b1", "c2": "C2_n2048_b64"}
table = {
    "note": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY "
            "GRBM_GUI_ACTIVE --kernel-trace (one pass, tools/collect_profiles.sh; MI32_LOOKAHEAD=0); per-launch means. "
            "SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs (= 64 x the number of v_mfma_f32_32x32x2_f32), "
            "GRBM_GUI_ACTIVE is summed over the 8 XCDs, SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles "
            "summed over waves (MI355X_MICROARCH.md). mfma_pipe_busy_frac = MFMA_BUSY / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs).",
}
for run, key in RUNS.items():
    files = glob.glob(os.path.join(src, f"pmc_SQ_VALU_MFMA_BUSY_CYCLES_{run}", "**", "*counter_collection.csv"), recursive=True)
    if not files:
        continue
    per = collections.defaultdict(lambda: collections.defaultdict(dict))  # class -> dispatch -> counter -> value
    for r in csv.DictReader(open(files[0])):
        for c in CLASSES:
            if c in r["Kernel_Name"] and "persistent" not in r["Kernel_Name"]:
                per[c][r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
    table[key] = {}
    for c in CLASSES:
        disp = list(per[c].values())
        if not disp:
            continue
        m = {k: sum(d.get(k, 0.0) for d in disp) / len(disp) for k in disp[0]}
        m["launches_averaged"] = len(disp)
        wc = m.get("SQ_WAVE_CYCLES", 0.0)
        if m.get("GRBM_GUI_ACTIVE"):
            m["mfma_pipe_busy_frac"] = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (m["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
        if wc:
            m["waves_issuing_frac"] = m.get("SQ_ACTIVE_INST_ANY", 0.0) / wc
            m["waves_parked_frac"] = m.get("SQ_WAIT_ANY", 0.0) / wc
            m["waves_issue_stalled_frac"] = m.get("SQ_WAIT_INST_ANY", 0.0) / wc
        table[key][c] = m
json.dump(table, open(out, "w"), indent=1)
for key in RUNS.values():
    for c, m in table.get(key, {}).items():
        print(key, c, m["launches_averaged"], round(m.get("mfma_pipe_busy_frac", 0), 4))
