#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ (run in the BUILD container only).

Two kinds of vectors, both data only (inputs + expected outputs):

1. ``ref_just_inv_K*.npz`` -- outputs of the REFERENCE ITSELF run here: the
   reference's CPU script /root/reference/matrix_inv_numpy.py is imported and its
   ``just_inv(K)`` (matrix_inv_numpy.py:39-46) is executed with NumPy's legacy
   global RNG seeded; ``numpy.linalg.inv`` is wrapped so the matrix the script
   built and the inverse it computed are captured (the script itself only prints
   ``TIME:``).  The call site being pinned is matrix_inv_numpy.py:44.

2. ``dist_*.npz`` / ``hollow_msvc_*.npz`` / ``c0_*.npz`` -- seeded inputs of the
   distributions SURVEY.md section 8(d) names, with ``numpy.linalg.inv`` evaluated in
   float64 on the fp32-cast input (the same third-party routine the reference
   calls; NumPy version recorded in each file).

/root/reference never travels to the GPU box; these .npz files do.
"""
import contextlib
import io
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)

REFERENCE = "/root/reference"


def save(name, **arrays):
    arrays["numpy_version"] = np.array(np.__version__)
    np.savez(os.path.join(HERE, name), **arrays)
    print("wrote", name, {k: getattr(v, "shape", None) for k, v in arrays.items()})


def capture_reference_just_inv(K, seed):
    """Run the reference's just_inv(K) and capture (input, output) of its inv call."""
    sys.dont_write_bytecode = True
    sys.path.insert(0, REFERENCE)
    import matrix_inv_numpy as ref  # the reference's own module

    captured = {}
    real_inv = np.linalg.inv

    def spy(a, *args, **kw):
        out = real_inv(a, *args, **kw)
        captured["a"] = np.array(a, dtype=np.float64)
        captured["inv"] = np.array(out, dtype=np.float64)
        return out

    np.random.seed(seed)
    np.linalg.inv = spy
    try:
        with contextlib.redirect_stdout(io.StringIO()) as so:
            ref.just_inv(K)
    finally:
        np.linalg.inv = real_inv
        sys.path.remove(REFERENCE)
    assert so.getvalue().startswith("TIME:"), so.getvalue()
    return captured["a"], captured["inv"]


def gate_matrix(n, seed):
    """D_gate: row-permuted U(-1,1) + sqrt(N) I, fp32 cast afterwards."""
    rng = np.random.default_rng(seed)
    a = rng.uniform(-1.0, 1.0, (n, n)) + np.sqrt(n) * np.eye(n)
    return a[rng.permutation(n)].astype(np.float32)


def main():
    # 1. the reference itself
    for K, seed in ((8, 11), (64, 12), (256, 13)):
        a64, inv64 = capture_reference_just_inv(K, seed)
        save(f"ref_just_inv_K{K}.npz", a64=a64, inv64_of_a64=inv64, a=a64.astype(np.float32),
             inv64=np.linalg.inv(a64.astype(np.float32).astype(np.float64)), seed=np.array(seed))

    # 2. distributions, fp64 inverse of the fp32-cast input
    sizes = (1, 2, 3, 5, 16, 63, 64, 65, 100, 256)
    for n in sizes:
        a = gate_matrix(n, 1000 + n)
        save(f"dist_gate_N{n}.npz", a=a, inv64=np.linalg.inv(a.astype(np.float64)))
    for n in (5, 16, 64, 100):
        rng = np.random.default_rng(2000 + n)
        a = rng.uniform(0.0, 100.0, (n, n)).astype(np.float32)  # PY:17, NP:40
        save(f"dist_ref100_N{n}.npz", a=a, inv64=np.linalg.inv(a.astype(np.float64)))
        rng = np.random.default_rng(3000 + n)
        a = rng.uniform(0.0, 1.0, (n, n)).astype(np.float32)  # MATLAB rand(N,N)
        save(f"dist_rand_N{n}.npz", a=a, inv64=np.linalg.inv(a.astype(np.float64)))
        rng = np.random.default_rng(4000 + n)
        a = rng.uniform(0.0, 100.0, (n, n))
        np.fill_diagonal(a, 0.0)  # NP:13-14 hollow
        a = a.astype(np.float32)
        save(f"dist_hollow_N{n}.npz", a=a, inv64=np.linalg.inv(a.astype(np.float64)))

    # the sweep driver's first two inputs: hollow rand()%10 from the unseeded MSVC
    # LCG (main_file.cpp:41-52); the stream continues from k=10 into k=20.
    import oracle as O

    state = 1
    for k in (10, 20):
        a, state = O.fill_hollow_msvc(k, state)
        save(f"hollow_msvc_K{k}.npz", a=a, inv64=np.linalg.inv(a.astype(np.float64)))

    # C0 = BASELINE configs[0]: seed-0 default_rng 256x256 U(0,100), fp32 cast
    rng = np.random.default_rng(0)
    a = rng.uniform(0.0, 100.0, (256, 256)).astype(np.float32)
    save("c0_u100_N256.npz", a=a, inv64=np.linalg.inv(a.astype(np.float64)))


if __name__ == "__main__":
    main()
