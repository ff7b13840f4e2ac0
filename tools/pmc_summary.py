#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE, one pass each) into the per-launch HBM traffic of
the rank-bw update kernel, the JSON bench.py reads for roofline.traffic.
    tools/pmc_summary.py gpurun_out/prof profiles/round1/pmc_rank_bw2_n4096.json
Units and corrections as MI355X_MICROARCH.md prescribes: both counters are in KiB; on gfx950 FETCH_SIZE reports
half of the bytes of a wide coalesced streaming read (x2); WRITE_SIZE is exact for streaming stores."""
import csv, json, sys, collections

src, out = sys.argv[1], sys.argv[2]
res = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f"{src}/pmc_{ctr}/pmc_{ctr}_counter_collection.csv")):
        if r["Counter_Name"] != ctr:
            continue
        k = r["Kernel_Name"]
        name = ("rank_bw" if "gj_rank_bw2_kernel" in k else "transpose" if "gj_panel_transpose" in k else
                "subpanel" if "gj_subpanel_kernel" in k else "inblock_update" if "gj_inblock_update" in k else None)
        if name:
            agg[name].append(float(r["Counter_Value"]))
    for name, v in agg.items():
        res[f"{ctr}_{name}_mean_KiB"] = sum(v) / len(v)
        res[f"{ctr}_{name}_launches"] = len(v)
n, bw = 4096, 256
fetch = res["FETCH_SIZE_rank_bw_mean_KiB"] * 1024
write = res["WRITE_SIZE_rank_bw_mean_KiB"] * 1024
res.update({
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/collect_profiles.sh): MI32_LOOKAHEAD=0 "
              "python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile-pass, MI355X, N=4096 batch=1 bw=256",
    "kernel": "gj_rank_bw2_kernel<16,3>",
    "hbm_bytes_per_launch_raw": fetch + write,
    "hbm_bytes_per_launch_corrected": 2 * fetch + write,
    "correction_note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of wide coalesced reads); this "
                       "kernel's C-tile reads are 4-byte-per-lane accesses (128 B per half-wave), for which the guide calls "
                       "the factor uncalibrated; WRITE_SIZE is exact",
    "algorithmic_bytes_per_launch": {"C_read": 4 * n * (n - bw), "C_write": 4 * n * (n - bw),
                                     "A_B_panels_once": 2 * 4 * n * bw, "panel_copy_read_write": 2 * 4 * n * bw},
})
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
