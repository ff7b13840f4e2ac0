"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` remarks: VGPRs, spills, scratch, occupancy per kernel.

    hipcc ... -c file.hip -o /dev/null -Rpass-analysis=kernel-resource-usage 2> res.txt
    python tools/kernel_resources.py res.txt [filter]
"""
import re
import subprocess
import sys


def main():
    txt = open(sys.argv[1]).read()
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    blocks = re.split(r"remark: [^\n]*Function Name: ", txt)
    names = [b.split("\n")[0].strip() for b in blocks[1:]]
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True,
                         text=True).stdout.split("\n")
    for b, dn in zip(blocks[1:], dem):
        def g(k):
            m = re.search(k + r": (\d+)", b)
            return int(m.group(1)) if m else -1
        dn = dn.replace("mi32::", "").replace("void ", "")
        dn = re.sub(r"\(.*", "", dn)[:64]
        if flt and flt not in dn:
            continue
        print("%-66s vgpr %3d agpr %3d spill %3d scratch %4d occ %2d lds %6d" % (
            dn, g("VGPRs"), g("AGPRs"), g("VGPR Spill"), g(r"ScratchSize \[bytes/lane\]"),
            g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")))


if __name__ == "__main__":
    main()
