"""MI355X-native fp32 Gauss-Jordan matrix inversion behind ``matrix_inv_32(vec, N)``.

Drop-in for the hot path of MarchesiGabriele/gpu_matrix_inversion: hand-written
HIP kernels for gfx950 in ``lib/libmat_inv_32.so`` (C ABI: include/mat_inv_32_c.h,
C++ drop-in: include/mat_inv_32.h) and this thin Python host mirror.
"""
from ._lib import (  # noqa: F401
    ALGO_AUTO,
    ALGO_BLOCKED,
    ALGO_SWEEP,
    MI32_BAD_SHAPE,
    MI32_OK,
    MI32_RUNTIME_ERROR,
    MI32_SINGULAR,
    Mi32Error,
    build_library,
)
from .api import Inverter, fp32_bench, fp64_bench, just_inv, matrix_multiply, last_timing, matrix_inv_32, matrix_inv_32_batched, matrix_inv_64, matrix_inversion_no_pivots  # noqa: F401
from .sharding import invert_distributed, invert_sharded, shard_range  # noqa: F401

__all__ = [
    "matrix_inv_32",
    "matrix_inv_32_batched",
    "matrix_inv_64",
    "matrix_inversion_no_pivots",
    "just_inv",
    "fp32_bench",
    "fp64_bench",
    "matrix_multiply",
    "last_timing",
    "Inverter",
    "shard_range",
    "invert_sharded",
    "invert_distributed",
    "build_library",
    "Mi32Error",
]
