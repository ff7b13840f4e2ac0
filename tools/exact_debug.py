"""Where does the blocked HIP path differ from the reference-order oracle?  (debug aid, GPU box)

    python tools/exact_debug.py [n ...]      env: MI32_FUSED_ROWS=0 (no fused launches), LOOKAHEAD=0
Prints, per size and distribution, the number of differing entries and the bounding box / first rows and
columns of the differences, which localises a bug to a block, a sub-panel or a row class.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import gpu_matrix_inversion_amd as g  # noqa: E402
import oracle as O  # noqa: E402
from conftest import gate_matrix  # noqa: E402


def main():
    sizes = [int(v) for v in sys.argv[1:]] or [100, 300]
    inv = g.Inverter(algo="blocked")
    if os.environ.get("LOOKAHEAD", "1") == "0":
        inv.set_lookahead(False)
    bad_total = 0
    for n in sizes:
        for kind in ("gate", "u100"):
            a = gate_matrix(n, 100 + n) if kind == "gate" else np.random.default_rng(n).uniform(0, 100, (n, n)).astype(np.float32)
            want, info = (O.matrix_inv_32_inplace if n <= 1024 else O.matrix_inv_32_blocked_exact)(a, n, return_info=True)
            x, st = inv.inv(torch.from_numpy(a).cuda())
            torch.cuda.synchronize()
            got = x.cpu().numpy()
            w = want.reshape(n, n)
            d = got != w
            nb = int(d.sum())
            bad_total += nb
            msg = f"n={n} {kind}: status {int(st[0])}/{info['status']}  differing {nb}/{n * n}"
            if nb:
                rows, cols = np.nonzero(d.any(axis=1))[0], np.nonzero(d.any(axis=0))[0]
                rel = np.abs(got - w).max() / np.abs(w).max()
                msg += (f"  max rel {rel:.3e} nan {int(np.isnan(got).sum())} rows {rows.size} [{rows[:6].tolist()}..{rows[-3:].tolist()}]"
                        f" cols {cols.size} [{cols[:6].tolist()}..{cols[-3:].tolist()}]")
            print(msg, flush=True)
    inv.close()
    print("exact_debug:", "OK" if bad_total == 0 else "MISMATCH")
    return 1 if bad_total else 0


if __name__ == "__main__":
    sys.exit(main())
