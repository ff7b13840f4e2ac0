#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE, one pass each, per configuration) into
profiles/roundN/pmc_traffic.json, the table bench.py reads `roofline.traffic` from -- one entry per exact
configuration (algo, n, batch, block width), never replayed for another one.

    tools/pmc_traffic.py gpurun_out/prof profiles/round2/pmc_traffic.json

Expects <src>/pmc_<CTR>_<name>/pmc_<CTR>_<name>_counter_collection.csv for every <name> in CONFIGS (written by
tools/collect_profiles.sh).  Units and corrections as MI355X_MICROARCH.md prescribes: both counters are KiB; on
gfx950 FETCH_SIZE reports half the bytes of a wide coalesced streaming read (x2); WRITE_SIZE is exact for
streaming stores."""
import collections
import csv
import glob
import json
import os
import sys

src, out = sys.argv[1], sys.argv[2]
# name -> (table key, kernel substring of the roofline kernel, algorithmic bytes per launch)
CONFIGS = {
    "c1": ("blocked_n4096_b1_bw256", "gj_rank_bw2_kernel", 8.0 * 4096 * (4096 - 256) + 8.0 * 4096 * 256),
    "c2": ("blocked_n2048_b64_bw128", "gj_rank_bw2_kernel", 64 * (8.0 * 2048 * (2048 - 128) + 8.0 * 2048 * 128)),
    "sweep": ("sweep_n4096_b1", "gj_sweep_step_kernel", 8.0 * 4096 * 4097),
    "c4": ("blocked_n16384_b1_bw256", "gj_rank_bw2_kernel", 8.0 * 16384 * (16384 - 256) + 8.0 * 16384 * 256),
}
table = {}
for name, (key, kern, alg) in CONFIGS.items():
    vals = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        files = glob.glob(os.path.join(src, f"pmc_{ctr}_{name}", "**", "*counter_collection.csv"), recursive=True)
        if not files:
            break
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(files[0])):
            if r["Counter_Name"] == ctr and kern in r["Kernel_Name"] and "persistent" not in r["Kernel_Name"]:
                agg[kern].append(float(r["Counter_Value"]))
        if not agg[kern]:
            break
        v = agg[kern]
        vals[ctr] = (sum(v) / len(v) * 1024.0, len(v))
    if len(vals) != 2:
        continue
    fetch, nl = vals["FETCH_SIZE"]
    write, _ = vals["WRITE_SIZE"]
    table[key] = {
        "kernel": kern, "launches_averaged": nl,
        "fetch_bytes_raw": fetch, "fetch_bytes_x2": 2 * fetch, "write_bytes": write,
        "hbm_bytes_per_launch_corrected": 2 * fetch + write,
        "algorithmic_bytes_per_launch": alg,
        "traffic_over_algorithmic": (2 * fetch + write) / alg,
        "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes (tools/collect_profiles.sh, run '{name}'), "
                  "MI355X; FETCH_SIZE x2 per MI355X_MICROARCH.md (gfx950 counts half of wide coalesced reads; uncalibrated "
                  "for this kernel's 4-byte-per-lane C reads), WRITE_SIZE exact",
    }
os.makedirs(os.path.dirname(out), exist_ok=True)
json.dump(table, open(out, "w"), indent=1)
print(json.dumps(table, indent=1))
