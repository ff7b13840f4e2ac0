#!/usr/bin/env python3
"""Generate ``mirror_digest_N*.npz`` (run in the BUILD container; minutes of CPU).

For the orders whose blocked mirror is too slow to run inside a test -- the three- and
four-workgroup panels of N = 8200 and N = 16384 (BASELINE configs[4]) -- and for N = 4096
(configs[1]) as a cross-machine pin of the oracle build, this stores a DIGEST of the
oracle's two-level blocked mirror (oracle/gj_oracle.c: gjo_matrix_inv_32_blocked2w, the
CPU restatement of the reference's step loop mat_inv_32.cpp:317-362 in the HIP path's
accumulation order) on the seeded D_gate input:

    n, seed, widths (sub-panel width of every outer block), bw (outer block width),
    sha256 of the N*N fp32 output bytes, 4096 sampled flat indices and their values,
    the per-row sums of |x| in float64 (N values: localises a mismatch to rows).

The blocking is the one libmat_inv_32.so resolves for (n, batch 1) -- asked through the C ABI
with a NULL handle, no GPU needed -- and is stored, so that the GPU test can assert that the
library still resolves the same plan before it compares bits.
Data only: inputs are re-generated from the seed, outputs are numbers.
"""
import ctypes
import hashlib
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

from conftest import gate_matrix  # noqa: E402

CASES = ((4096, 10_000), (8200, 50_000), (16384, 70_000))  # (n, seed): the seeds of tests/test_gpu_parity.py


def resolved_plan(n, batch=1):
    from gpu_matrix_inversion_amd import _lib

    lib = _lib.load()
    w, bw, nb = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
    assert lib.mi32_resolve_blocking(None, n, batch, ctypes.byref(w), ctypes.byref(bw)) == 0
    widths = (ctypes.c_int * 128)()
    assert lib.mi32_resolve_panel_widths(None, n, batch, widths, 128, ctypes.byref(nb)) == 0
    return [int(widths[i]) for i in range(nb.value)], int(bw.value)


def digest_of(x):
    x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1)
    return np.frombuffer(hashlib.sha256(x.tobytes()).digest(), dtype=np.uint8).copy()


def main():
    import oracle as O

    only = [int(v) for v in sys.argv[1:]]
    for n, seed in CASES:
        if only and n not in only:
            continue
        a = gate_matrix(n, seed)
        widths, bw = resolved_plan(n)
        t0 = time.time()
        x, info = O.matrix_inv_32_blocked2(a, n, widths, bw, return_info=True)
        dt = time.time() - t0
        assert info["status"] == 0
        idx = np.random.default_rng(424242 + n).choice(n * n, 4096, replace=False).astype(np.int64)
        rowsum = np.abs(x.reshape(n, n).astype(np.float64)).sum(axis=1)
        name = f"mirror_digest_N{n}.npz"
        np.savez(os.path.join(HERE, name), n=np.array(n), seed=np.array(seed), widths=np.array(widths, np.int32),
                 bw=np.array(bw), sha256=digest_of(x), idx=idx, vals=x[idx].copy(), rowsum_abs=rowsum)
        print(f"wrote {name}: plan widths {sorted(set(widths))} x{len(widths)} bw {bw}, mirror {dt:.1f} s, "
              f"sha256 {bytes(digest_of(x)).hex()[:16]}...")


if __name__ == "__main__":
    main()
