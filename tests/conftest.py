import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

EPS32 = float(np.finfo(np.float32).eps) / 2  # unit roundoff 2^-24


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_files():
    """(input, float64 inverse) fixtures; the oracle_digest_* files are a different kind (see below)."""
    return sorted(p for p in glob.glob(os.path.join(GOLDEN, "*.npz"))
                  if not os.path.basename(p).startswith("oracle_digest_"))


def canonical_bytes(x):
    """The fp32 bytes of a result with -0.0 stored as +0.0: a zero multiplier is skipped by the reference
    (mat_inv_32.cpp:28) and multiplied through by the matrix cores, which can only differ in the sign of a zero."""
    x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1)
    return (x + np.float32(0.0)).tobytes()


def load_oracle_digest(n):
    """Digest of the oracle's reference-order result on gate_matrix(n, seed), written by
    tests/golden/make_oracle_digests.py in the build container: sha256 of the fp32 output bytes,
    4096 sampled entries, per-row sums of |x|."""
    d = np.load(os.path.join(GOLDEN, f"oracle_digest_N{n}.npz"), allow_pickle=False)
    return {k: d[k] for k in d.files}


def check_against_oracle_digest(x, dig):
    """Bit-exact comparison of a flat fp32 result with an oracle digest; on a mismatch the assertion
    message says how many sampled entries differ and which rows' |x| sums are off."""
    import hashlib

    x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1)
    n = int(dig["n"])
    assert x.size == n * n
    got = hashlib.sha256(canonical_bytes(x)).digest()
    if got == bytes(dig["sha256"]):
        return
    bad = int((x[dig["idx"]] != dig["vals"]).sum())
    rows = np.nonzero(np.abs(x.reshape(n, n).astype(np.float64)).sum(axis=1) != dig["rowsum_abs"])[0]
    raise AssertionError(f"N={n}: sha256 differs from the oracle's; {bad}/{dig['idx'].size} sampled entries differ, "
                         f"{rows.size} rows differ (first {rows[:8].tolist()})")


def load_golden(path):
    d = np.load(path, allow_pickle=False)
    return d["a"], d["inv64"]


def gate_matrix(n, seed):
    """D_gate of SURVEY.md 8(d): row-permuted U(-1,1) + sqrt(N) I, kappa ~ 6, forces ~N swaps."""
    rng = np.random.default_rng(seed)
    a = rng.uniform(-1.0, 1.0, (n, n)) + np.sqrt(n) * np.eye(n)
    return a[rng.permutation(n)].astype(np.float32)


def forward_tolerance(a, factor=2.0):
    """fp32 forward-error bound used by every parity test against a float64 inverse:
    max|X - X64| / max|X64| <= factor * kappa_inf(A) * 2^-24.
    (Measured for the oracle over all fixtures: <= 0.37 * kappa_inf * 2^-24.)"""
    a64 = np.asarray(a, dtype=np.float64)
    kappa = np.linalg.cond(a64, np.inf)
    return factor * kappa * EPS32


def rel_err(x, ref):
    x = np.asarray(x, dtype=np.float64).reshape(ref.shape)
    return float(np.abs(x - ref).max() / np.abs(ref).max())


@pytest.fixture(scope="session")
def oracle():
    import oracle as O

    O.build()
    return O
