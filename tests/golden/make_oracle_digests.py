#!/usr/bin/env python3
"""Generate ``oracle_digest_N*.npz`` (run in the BUILD container; minutes of CPU).

For the BASELINE orders N = 4096 (configs[1]), 8200 (three workgroups share the panel) and 16384
(configs[4]) this stores a DIGEST of the oracle's REFERENCE-ORDER result on the seeded D_gate input:
the step-by-step restatement of the reference's loop (mat_inv_32.cpp:317-362; oracle/gj_oracle.c
gjo_matrix_inv_32_inplace), computed through its cache-blocked evaluation gjo_matrix_inv_32_blocked_exact,
which tests/test_oracle.py proves bit-identical to the step-by-step form for every block width:

    n, seed, sha256 of the N*N fp32 output bytes (-0.0 stored as +0.0), 4096 sampled flat indices and
    their values, the per-row sums of |x| in float64 (N values: localises a mismatch to rows).

Nothing about the HIP path's blocking enters: from round 3 on the blocked HIP path evaluates the
sequential arithmetic and its result does not depend on the plan.
Data only: inputs are re-generated from the seed, outputs are numbers.
"""
import hashlib
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

from conftest import canonical_bytes, gate_matrix  # noqa: E402

CASES = ((4096, 10_000), (8200, 50_000), (16384, 70_000))  # (n, seed): the seeds of tests/test_gpu_parity.py


def digest_of(x):
    return np.frombuffer(hashlib.sha256(canonical_bytes(x)).digest(), dtype=np.uint8).copy()


def main():
    import oracle as O

    only = [int(v) for v in sys.argv[1:]]
    for n, seed in CASES:
        if only and n not in only:
            continue
        a = gate_matrix(n, seed)
        t0 = time.time()
        x, info = O.matrix_inv_32_blocked_exact(a, n, 128, return_info=True)
        dt = time.time() - t0
        assert info["status"] == 0
        if n <= 4096:  # the step-by-step form itself, where it finishes in seconds
            assert np.array_equal(x.view(np.uint32), O.matrix_inv_32_inplace(a, n).view(np.uint32))
        idx = np.random.default_rng(424242 + n).choice(n * n, 4096, replace=False).astype(np.int64)
        rowsum = np.abs(x.reshape(n, n).astype(np.float64)).sum(axis=1)
        name = f"oracle_digest_N{n}.npz"
        np.savez(os.path.join(HERE, name), n=np.array(n), seed=np.array(seed), sha256=digest_of(x), idx=idx,
                 vals=x[idx].copy(), rowsum_abs=rowsum)
        print(f"wrote {name}: oracle {dt:.1f} s, sha256 {bytes(digest_of(x)).hex()[:16]}...")


if __name__ == "__main__":
    main()
