#!/usr/bin/env python3
"""Diagnostic (not part of the product): where a panel launch spends its cycles.

Needs the stamped build (`make -C gpu_matrix_inversion_amd/csrc stamps`): every panel launch of one
inversion records s_memtime at its phase boundaries (wave 0 of the first panel workgroup).  Prints, per
thread geometry, the median cycles of: slab load, prologue, each pivot step, epilogue, and the phase
split of one representative step.
    python tools/panel_stamps.py [n] [batch]
"""
import ctypes
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from gpu_matrix_inversion_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.environ.get("MI32_STAMPS_LIB") or os.path.join(ROOT, "gpu_matrix_inversion_amd", "lib", "libmat_inv_32_stamps.so")
import gpu_matrix_inversion_amd as g  # noqa: E402


def gate(n, seed):
    rng = np.random.default_rng(seed)
    a = rng.uniform(-1.0, 1.0, (n, n)) + np.sqrt(n) * np.eye(n)
    return a[rng.permutation(n)].astype(np.float32)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    lib = _lib.load()
    a = torch.from_numpy(np.stack([gate(n, 1000 + b) for b in range(batch)])).cuda()
    inv = g.Inverter(algo="blocked")
    inv.inv(a)
    torch.cuda.synchronize()
    buf = torch.zeros(1024 * 64, dtype=torch.int64, device="cuda")
    lib.mi32_debug_panel_stamps.restype = ctypes.c_int
    lib.mi32_debug_panel_stamps.argtypes = [ctypes.c_void_p]
    assert lib.mi32_debug_panel_stamps(ctypes.c_void_p(buf.data_ptr())) == 0
    inv.inv(a)
    torch.cuda.synchronize()
    lib.mi32_debug_panel_stamps(None)
    st = buf.cpu().numpy().reshape(1024, 64).astype(np.uint64)
    groups = defaultdict(list)
    for q in st:
        if q[0] == 0 or q[49] == 0:
            continue
        meta = int(q[51])
        nt, rpt, fused, w = meta >> 32, (meta >> 16) & 0xFFFF, (meta >> 8) & 0xFF, meta & 0xFF
        groups[(nt, rpt, w, fused)].append(q.astype(np.int64))
    print(f"N={n} batch={batch}: cycles (s_memtime), medians over the launches of each geometry")
    for key in sorted(groups, reverse=True):
        qs = np.stack(groups[key])
        nt, rpt, w, fused = key
        d = lambda a_, b_: float(np.median(qs[:, b_] - qs[:, a_]))  # noqa: E731
        steps = [d(2 if r == 0 else 3 + r - 1, 3 + r) for r in range(w)]
        total = d(0, 49)
        print(f"\n<{nt},{rpt},{w},{'fused' if fused else 'plain'}> x{len(qs)} launches, rows {int(qs[0, 52])}..{int(qs[-1, 52])}: "
              f"total {total:.0f}")
        print(f"  load {d(0, 1):.0f}  prologue {d(1, 2):.0f}  steps sum {sum(steps):.0f} (per step med {np.median(steps):.0f}, "
              f"min {min(steps):.0f}, max {max(steps):.0f})  post-steps sync {d(3 + w - 1, 48):.0f}  epilogue {d(48, 49):.0f}")
        print("  steps: " + " ".join(f"{v:.0f}" for v in steps))
        r = w // 2
        names = ["search", "dpp-max", "ballot+cand-write", "cand-readback", "divide", "publish", "barrier", "key-read",
                 "prn-read", "fma", "labels+winner"]
        slots = [32, 33, 34, 35, 36, 37, 38, 39, 40, 41, 42, 3 + r]
        parts = [d(slots[i], slots[i + 1]) for i in range(len(names))]
        print(f"  step {r}: " + "  ".join(f"{nm} {v:.0f}" for nm, v in zip(names, parts)) + f"  = {sum(parts):.0f}")
    inv.close()


if __name__ == "__main__":
    main()
