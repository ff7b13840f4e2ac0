"""CPU oracle for the fp32 Gauss-Jordan inversion hot path.

TEST INFRASTRUCTURE ONLY.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import this package, and only as the
checker.  The product package ``gpu_matrix_inversion_amd`` never does.

Parity pinning: see ``gj_oracle.h`` -- pinned by the reference's NumPy script
outputs captured into ``tests/golden/`` and by the reference's own acceptance
properties; the OpenCL C++ path cannot run in the build container (0 devices).
"""
from .oracle import (  # noqa: F401
    ARITH_FMA,
    ARITH_UNFUSED,
    PIVOT_REFERENCE_DEFECT,
    PIVOT_TRUE_PARTIAL,
    STATUS_BAD_SHAPE,
    STATUS_OK,
    STATUS_SINGULAR,
    build,
    fill_hollow_msvc,
    frobenius_metric,
    left_half_is_identity,
    matrix_inv_32,
    matrix_inv_32_blocked,
    matrix_inv_32_blocked2,
    matrix_inv_32_blocked_exact,
    matrix_inv_32_inplace,
    matrix_inv_64,
    matrix_inv_64_blocked,
    matrix_inversion_no_pivots,
    msvc_rand_stream,
    numpy_mirror_inv,
    residual_inf,
    residual_inf_left,
)
