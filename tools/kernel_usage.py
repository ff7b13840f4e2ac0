#!/usr/bin/env python3
"""Print a per-kernel resource table (VGPR/SGPR/scratch/LDS/occupancy) for the HIP sources."""
import re, subprocess, sys, os
here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpu_matrix_inversion_amd", "csrc")
files = sys.argv[1:] or ["mi32_sweep.hip", "mi32_blocked.hip", "mi32_residual.hip"]
for f in files:
    # exactly the flags of csrc/Makefile (a different flag set gives different register allocation)
    mk = open(os.path.join(here, "Makefile")).read()
    flags = re.search(r"^FLAGS\s*\?=\s*(.*?)(?<!\\)$", mk, re.M | re.S).group(1).replace("\\\n", " ").split()
    flags = [fl.replace("$(ARCH)", "gfx950") for fl in flags]
    out = subprocess.run(["/opt/rocm/bin/hipcc"] + flags + ["-c", f, "-o", "/dev/null",
                          "-Rpass-analysis=kernel-resource-usage"], cwd=here, capture_output=True, text=True).stderr
    cur = None
    rows = {}
    for line in out.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            cur = re.sub(r"\(.*", "", cur)
            rows[cur] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+) \[-Rpass", line)
        if m and cur:
            rows[cur][m.group(1).strip()] = int(m.group(2))
    for k, v in rows.items():
        print(f"{k:60s} vgpr={v.get('VGPRs',-1):4d} agpr={v.get('AGPRs',-1):3d} sgpr={v.get('TotalSGPRs',-1):4d} scratch={v.get('ScratchSize',-1):5d} vspill={v.get('VGPRs Spill',-1):4d} lds={v.get('LDS Size',-1):6d} occ={v.get('Occupancy',-1)}")
