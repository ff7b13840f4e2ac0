"""ctypes binding of ``lib/libmat_inv_32.so`` (the C ABI of include/mat_inv_32_c.h).

This is the binding a maintainer of the reference would write to call the
HIP library from Python.  There is deliberately no CPU
fallback: if the shared library is missing or cannot be loaded, every entry
point raises.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libmat_inv_32.so")
CSRC_DIR = os.path.join(_HERE, "csrc")

MI32_OK = 0
MI32_BAD_SHAPE = 1
MI32_SINGULAR = 2
MI32_RUNTIME_ERROR = 3

ALGO_AUTO = 0
ALGO_SWEEP = 1
ALGO_BLOCKED = 2
ALGO_NAMES = {"auto": ALGO_AUTO, "sweep": ALGO_SWEEP, "blocked": ALGO_BLOCKED}
KERNEL_CLASSES = ("init", "sweep_step", "panel", "update_in_block", "update_rank_bw", "finish", "panel_transpose")

# every symbol include/mat_inv_32_c.h declares
C_ABI_SYMBOLS = (
    "mi32_matrix_inv_32",
    "mi32_matrix_inv_32_batched",
    "mi32_matrix_inv_32_batched_multi",
    "mi32_shard_range",
    "mi32_create",
    "mi32_destroy",
    "mi32_set_stream",
    "mi32_set_algo",
    "mi32_set_blocking",
    "mi32_set_lookahead",
    "mi32_workspace_bytes",
    "mi32_reserve",
    "mi32_inv_device",
    "mi32_residual_device",
    "mi32_set_profiling",
    "mi32_get_profile",
    "mi32_last_timing",
    "mi32_bench_32",
    "mi32_bench_64",
    "mi32_matrix_multiply_64",
    "mi32_resolve_algo",
    "mi32_matrix_inv_64",
    "mi32_matrix_inversion_no_pivots",
    "mi32_set_pivoting",
    "mi32_inv_device_f64",
    "mi32_resolve_blocking_f64",
    "mi32_resolve_blocking",
    "mi32_resolve_panel_widths",
    "mi32_dominant_kernel",
    "mi32_last_error",
    "mi32_version",
)
# the C++ drop-in of include/mat_inv_32.h (Itanium-mangled matrix_inv_32(std::vector<float>, int))
CXX_DROPIN_SYMBOL = "_Z13matrix_inv_32St6vectorIfSaIfEEi"
# the fp64 twin of include/mat_inv_64.h: matrix_inversion_FP64(std::vector<double>, int)
CXX_FP64_SYMBOL = "_Z21matrix_inversion_FP64St6vectorIdSaIdEEi"
# the no-pivot variant of include/mat_inv_64.h: matrix_inversion_no_pivots(std::vector<double>, int)
CXX_NOPIVOT_SYMBOL = "_Z26matrix_inversion_no_pivotsSt6vectorIdSaIdEEi"
# the benchmark twin of include/mat_inv_bench.h: Res FP32_bench(std::vector<float>, int)
CXX_BENCH_SYMBOL = "_Z10FP32_benchSt6vectorIfSaIfEEi"
TIMES10_SLOTS = ("queue", "buffers", "build", "makeAug", "pivot", "row", "column", "compute", "getInverted", "total")


class Mi32Error(RuntimeError):
    pass


def build_library(force: bool = False) -> str:
    """Compile the HIP sources for gfx950 (hipcc cross-compiles without a GPU)."""
    if force and os.path.exists(LIB_PATH):
        os.remove(LIB_PATH)
    subprocess.check_call(["make", "-C", CSRC_DIR, "-s", "all"])
    if not os.path.exists(LIB_PATH):
        raise Mi32Error(f"build did not produce {LIB_PATH}")
    return LIB_PATH


_lib = None


def load() -> ctypes.CDLL:
    """Load the shared library; raises if it is absent (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise Mi32Error(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C gpu_matrix_inversion_amd/csrc`.  There is no CPU fallback."
        )
    try:
        # If torch is (or will be) in the process, its bundled HIP runtime must be the one this
        # library binds to (same SONAME libamdhip64.so.7), so that streams and device pointers are
        # shared: import torch first when it is importable.
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is optional for the pure ctypes/numpy path
        pass
    lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    fp = ctypes.POINTER(ctypes.c_float)
    ip = ctypes.POINTER(ctypes.c_int)
    vp = ctypes.c_void_p
    lib.mi32_matrix_inv_32.restype = ctypes.c_int
    lib.mi32_matrix_inv_32.argtypes = [fp, ctypes.c_size_t, ctypes.c_int, fp]
    lib.mi32_matrix_inv_32_batched.restype = ctypes.c_int
    lib.mi32_matrix_inv_32_batched.argtypes = [fp, ctypes.c_int, ctypes.c_int, fp, ip]
    lib.mi32_shard_range.restype = ctypes.c_int
    lib.mi32_shard_range.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ip, ip]
    lib.mi32_debug_drop_panel_group.restype = ctypes.c_int
    lib.mi32_debug_drop_panel_group.argtypes = [ctypes.c_int]
    lib.mi32_matrix_inv_32_batched_multi.restype = ctypes.c_int
    lib.mi32_matrix_inv_32_batched_multi.argtypes = [fp, ctypes.c_int, ctypes.c_int, fp, ip, ctypes.c_int]
    lib.mi32_create.restype = ctypes.c_int
    lib.mi32_create.argtypes = [ctypes.POINTER(vp), ctypes.c_int]
    lib.mi32_destroy.restype = ctypes.c_int
    lib.mi32_destroy.argtypes = [vp]
    lib.mi32_set_stream.restype = ctypes.c_int
    lib.mi32_set_stream.argtypes = [vp, vp]
    lib.mi32_set_algo.restype = ctypes.c_int
    lib.mi32_set_algo.argtypes = [vp, ctypes.c_int]
    lib.mi32_set_blocking.restype = ctypes.c_int
    lib.mi32_set_blocking.argtypes = [vp, ctypes.c_int, ctypes.c_int]
    lib.mi32_set_lookahead.restype = ctypes.c_int
    lib.mi32_set_lookahead.argtypes = [vp, ctypes.c_int]
    lib.mi32_workspace_bytes.restype = ctypes.c_size_t
    lib.mi32_workspace_bytes.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int]
    lib.mi32_reserve.restype = ctypes.c_int
    lib.mi32_reserve.argtypes = [vp, ctypes.c_int, ctypes.c_int]
    lib.mi32_inv_device.restype = ctypes.c_int
    lib.mi32_inv_device.argtypes = [vp, vp, ctypes.c_int, ctypes.c_int, vp, vp]
    lib.mi32_residual_device.restype = ctypes.c_int
    lib.mi32_residual_device.argtypes = [vp, vp, vp, ctypes.c_int, ctypes.c_int, vp]
    lib.mi32_set_profiling.restype = ctypes.c_int
    lib.mi32_set_profiling.argtypes = [vp, ctypes.c_int]
    lib.mi32_get_profile.restype = ctypes.c_int
    lib.mi32_get_profile.argtypes = [vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_longlong), ctypes.c_int]
    lib.mi32_bench_32.restype = ctypes.c_int
    lib.mi32_bench_32.argtypes = [fp, ctypes.c_size_t, ctypes.c_int, fp, ctypes.POINTER(ctypes.c_double)]
    lib.mi32_bench_64.restype = ctypes.c_int
    lib.mi32_bench_64.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.c_size_t, ctypes.c_int,
                                  ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), ctypes.c_int]
    lib.mi32_matrix_multiply_64.restype = ctypes.c_int
    lib.mi32_matrix_multiply_64.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), ctypes.c_size_t,
                                            ctypes.POINTER(ctypes.c_double)]
    lib.mi32_last_timing.restype = ctypes.c_int
    lib.mi32_last_timing.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
    lib.mi32_resolve_algo.restype = ctypes.c_int
    lib.mi32_resolve_algo.argtypes = [vp, ctypes.c_int, ctypes.c_int]
    lib.mi32_resolve_blocking.restype = ctypes.c_int
    lib.mi32_resolve_blocking.argtypes = [vp, ctypes.c_int, ctypes.c_int, ip, ip]
    dp = ctypes.POINTER(ctypes.c_double)
    lib.mi32_matrix_inv_64.restype = ctypes.c_int
    lib.mi32_matrix_inv_64.argtypes = [dp, ctypes.c_size_t, ctypes.c_int, dp]
    lib.mi32_matrix_inversion_no_pivots.restype = ctypes.c_int
    lib.mi32_matrix_inversion_no_pivots.argtypes = [dp, ctypes.c_size_t, ctypes.c_int, dp]
    lib.mi32_set_pivoting.restype = ctypes.c_int
    lib.mi32_set_pivoting.argtypes = [vp, ctypes.c_int]
    lib.mi32_inv_device_f64.restype = ctypes.c_int
    lib.mi32_inv_device_f64.argtypes = [vp, vp, ctypes.c_int, ctypes.c_int, vp, vp]
    lib.mi32_resolve_blocking_f64.restype = ctypes.c_int
    lib.mi32_resolve_blocking_f64.argtypes = [vp, ctypes.c_int, ip]
    lib.mi32_resolve_panel_widths.restype = ctypes.c_int
    lib.mi32_resolve_panel_widths.argtypes = [vp, ctypes.c_int, ctypes.c_int, ip, ctypes.c_int, ip]
    lib.mi32_dominant_kernel.restype = ctypes.c_char_p
    lib.mi32_dominant_kernel.argtypes = [ctypes.c_int]
    lib.mi32_last_error.restype = ctypes.c_char_p
    lib.mi32_last_error.argtypes = []
    lib.mi32_version.restype = ctypes.c_int
    lib.mi32_version.argtypes = []
    _lib = lib
    return lib


def check(rc: int, what: str) -> int:
    """Raise on MI32_RUNTIME_ERROR / MI32_BAD_SHAPE from a handle-level call."""
    if rc == MI32_RUNTIME_ERROR:
        raise Mi32Error(f"{what}: {load().mi32_last_error().decode()}")
    if rc == MI32_BAD_SHAPE:
        raise ValueError(f"{what}: bad shape or argument")
    return rc
