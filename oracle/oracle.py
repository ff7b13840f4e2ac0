"""ctypes front-end of ``libgj_oracle.so`` (TEST INFRASTRUCTURE ONLY).

Mirrors the reference's call shape ``matrix_inv_32(vec, N)``
(/root/reference/Matlab/mat_inv_32.h:4): flat row-major fp32 in, flat
row-major fp32 inverse out, an EMPTY array where the reference returns an
empty vector (mat_inv_32.cpp:206-215).
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

PIVOT_TRUE_PARTIAL = 0
PIVOT_REFERENCE_DEFECT = 1
ARITH_FMA = 0
ARITH_UNFUSED = 1
STATUS_OK = 0
STATUS_BAD_SHAPE = 1
STATUS_SINGULAR = 2

_lib = None


def build(force: bool = False) -> None:
    """Compile the C restatement (gcc, seconds)."""
    # make decides what is stale (a gj_oracle.c newer than the .so rebuilds it); where the sources travel
    # without a toolchain the prebuilt libraries are used as they are
    cmd = ["make", "-C", _HERE, "-s"] + (["-B"] if force else []) + ["all"]
    try:
        subprocess.check_call(cmd)
    except (OSError, subprocess.CalledProcessError):
        fast = os.path.join(_HERE, "libgj_oracle.so")
        gen = os.path.join(_HERE, "libgj_oracle_generic.so")
        if not (os.path.exists(fast) and os.path.exists(gen)):
            raise


def _cpu_has(*flags: str) -> bool:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    have = set(line.split(":", 1)[1].split())
                    return all(fl in have for fl in flags)
    except OSError:
        pass
    return False


def _load():
    global _lib
    if _lib is not None:
        return _lib
    name = "libgj_oracle.so" if _cpu_has("avx2", "fma") else "libgj_oracle_generic.so"
    path = os.path.join(_HERE, name)
    build()
    lib = ctypes.CDLL(path)
    fp = ctypes.POINTER(ctypes.c_float)
    ip = ctypes.POINTER(ctypes.c_int)
    lib.gjo_matrix_inv_32.restype = ctypes.c_int
    lib.gjo_matrix_inv_32.argtypes = [fp, ctypes.c_size_t, ctypes.c_int, fp, ctypes.c_int, ctypes.c_int, ip, fp]
    lib.gjo_matrix_inv_32_inplace.restype = ctypes.c_int
    lib.gjo_matrix_inv_32_inplace.argtypes = [fp, ctypes.c_size_t, ctypes.c_int, fp, ctypes.c_int, ip]
    lib.gjo_matrix_inv_32_blocked.restype = ctypes.c_int
    lib.gjo_matrix_inv_32_blocked.argtypes = [fp, ctypes.c_size_t, ctypes.c_int, fp, ctypes.c_int, ip]
    lib.gjo_matrix_inv_32_blocked2.restype = ctypes.c_int
    lib.gjo_matrix_inv_32_blocked2.argtypes = [fp, ctypes.c_size_t, ctypes.c_int, fp, ctypes.c_int, ctypes.c_int, ip]
    dp = ctypes.POINTER(ctypes.c_double)
    lib.gjo_matrix_inv_64_inplace.restype = ctypes.c_int
    lib.gjo_matrix_inv_64_inplace.argtypes = [dp, ctypes.c_size_t, ctypes.c_int, dp, ip]
    lib.gjo_matrix_inv_64_blocked.restype = ctypes.c_int
    lib.gjo_matrix_inv_64_blocked.argtypes = [dp, ctypes.c_size_t, ctypes.c_int, dp, ctypes.c_int, ip]
    lib.gjo_matrix_inv_64_nopivot.restype = ctypes.c_int
    lib.gjo_matrix_inv_64_nopivot.argtypes = [dp, ctypes.c_size_t, ctypes.c_int, dp]
    lib.gjo_matrix_inv_32_nopivot.restype = ctypes.c_int
    lib.gjo_matrix_inv_32_nopivot.argtypes = [fp, ctypes.c_size_t, ctypes.c_int, fp, ctypes.c_int]
    lib.gjo_matrix_inv_32_blocked_exact.restype = ctypes.c_int
    lib.gjo_matrix_inv_32_blocked_exact.argtypes = [fp, ctypes.c_size_t, ctypes.c_int, fp, ctypes.c_int, ip]
    lib.gjo_matrix_inv_32_blocked2w.restype = ctypes.c_int
    lib.gjo_matrix_inv_32_blocked2w.argtypes = [fp, ctypes.c_size_t, ctypes.c_int, fp, ip, ctypes.c_int, ctypes.c_int, ip]
    for nm in ("gjo_residual_inf", "gjo_residual_inf_left", "gjo_frobenius_metric"):
        getattr(lib, nm).restype = ctypes.c_double
        getattr(lib, nm).argtypes = [fp, fp, ctypes.c_int]
    lib.gjo_left_half_is_identity.restype = ctypes.c_int
    lib.gjo_left_half_is_identity.argtypes = [fp, ctypes.c_int]
    lib.gjo_msvc_rand.restype = ctypes.c_int
    lib.gjo_msvc_rand.argtypes = [ctypes.POINTER(ctypes.c_uint)]
    lib.gjo_fill_hollow_msvc.restype = None
    lib.gjo_fill_hollow_msvc.argtypes = [fp, ctypes.c_int, ctypes.POINTER(ctypes.c_uint)]
    _lib = lib
    return lib


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32).reshape(-1))


def _fp(a: np.ndarray):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def matrix_inv_32(vec, n: int, pivot_mode: int = PIVOT_TRUE_PARTIAL, arith_mode: int = ARITH_FMA,
                  return_info: bool = False):
    """Augmented-panel restatement (the reference's data flow).  Returns the flat
    inverse, or an empty array on a shape error.  ``return_info`` adds a dict with
    ``status``, ``pivots`` and the final augmented panel ``aug``."""
    lib = _load()
    v = _f32(vec)
    n = int(n)
    if n <= 0 or int(v.size // n) != n:
        empty = np.empty(0, dtype=np.float32)
        return (empty, {"status": STATUS_BAD_SHAPE}) if return_info else empty
    out = np.empty(n * n, dtype=np.float32)
    piv = np.empty(n, dtype=np.int32)
    aug = np.empty(2 * n * n, dtype=np.float32) if return_info else None
    st = lib.gjo_matrix_inv_32(_fp(v), v.size, n, _fp(out), pivot_mode, arith_mode,
                               piv.ctypes.data_as(ctypes.POINTER(ctypes.c_int)),
                               _fp(aug) if aug is not None else None)
    if return_info:
        return out, {"status": st, "pivots": piv, "aug": aug.reshape(n, 2 * n)}
    return out


def matrix_inv_32_inplace(vec, n: int, arith_mode: int = ARITH_FMA, return_info: bool = False):
    lib = _load()
    v = _f32(vec)
    n = int(n)
    if n <= 0 or int(v.size // n) != n:
        empty = np.empty(0, dtype=np.float32)
        return (empty, {"status": STATUS_BAD_SHAPE}) if return_info else empty
    out = np.empty(n * n, dtype=np.float32)
    piv = np.empty(n, dtype=np.int32)
    st = lib.gjo_matrix_inv_32_inplace(_fp(v), v.size, n, _fp(out), arith_mode,
                                       piv.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
    if return_info:
        return out, {"status": st, "pivots": piv}
    return out


def matrix_inv_64(vec, n: int, return_info: bool = False):
    """fp64 twin (the reference's matrix_inversion_FP64 call shape): flat row-major in, flat inverse out."""
    lib = _load()
    v = np.ascontiguousarray(np.asarray(vec, dtype=np.float64).reshape(-1))
    n = int(n)
    if n <= 0 or int(v.size // n) != n:
        empty = np.empty(0, dtype=np.float64)
        return (empty, {"status": STATUS_BAD_SHAPE}) if return_info else empty
    out = np.empty(n * n, dtype=np.float64)
    piv = np.empty(n, dtype=np.int32)
    dp = ctypes.POINTER(ctypes.c_double)
    st = lib.gjo_matrix_inv_64_inplace(v.ctypes.data_as(dp), v.size, n, out.ctypes.data_as(dp),
                                       piv.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
    if return_info:
        return out, {"status": st, "pivots": piv}
    return out


def matrix_inv_32_blocked(vec, n: int, w: int = 16, return_info: bool = False):
    lib = _load()
    v = _f32(vec)
    n = int(n)
    if n <= 0 or int(v.size // n) != n:
        empty = np.empty(0, dtype=np.float32)
        return (empty, {"status": STATUS_BAD_SHAPE}) if return_info else empty
    out = np.empty(n * n, dtype=np.float32)
    piv = np.empty(n, dtype=np.int32)
    st = lib.gjo_matrix_inv_32_blocked(_fp(v), v.size, n, _fp(out), int(w),
                                       piv.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
    if return_info:
        return out, {"status": st, "pivots": piv}
    return out


def matrix_inv_64_blocked(vec, n: int, bw: int, return_info: bool = False):
    """fp64 blocked mirror (outer blocks of bw columns, delayed rank-bw updates): the HIP fp64 blocked path's order."""
    lib = _load()
    v = np.ascontiguousarray(np.asarray(vec, dtype=np.float64).reshape(-1))
    n = int(n)
    if n <= 0 or int(v.size // n) != n:
        empty = np.empty(0, dtype=np.float64)
        return (empty, {"status": STATUS_BAD_SHAPE}) if return_info else empty
    out = np.empty(n * n, dtype=np.float64)
    piv = np.empty(n, dtype=np.int32)
    dp = ctypes.POINTER(ctypes.c_double)
    st = lib.gjo_matrix_inv_64_blocked(v.ctypes.data_as(dp), v.size, n, out.ctypes.data_as(dp), int(bw),
                                       piv.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
    return (out, {"status": st, "pivots": piv}) if return_info else out


def matrix_inversion_no_pivots(vec, n: int, return_info: bool = False):
    """The reference's no-pivot variant (matrix_inversion_no_pivots.cpp:10): fp64 for a float64 input (as the
    reference ships it), the same steps in fp32 for a float32 input.  Flat inverse, or an empty array on a shape error."""
    lib = _load()
    f64 = np.asarray(vec).dtype == np.float64
    dt = np.float64 if f64 else np.float32
    v = np.ascontiguousarray(np.asarray(vec, dtype=dt).reshape(-1))
    n = int(n)
    if n <= 0 or int(v.size // n) != n:
        empty = np.empty(0, dtype=dt)
        return (empty, {"status": STATUS_BAD_SHAPE}) if return_info else empty
    out = np.empty(n * n, dtype=dt)
    if f64:
        dp = ctypes.POINTER(ctypes.c_double)
        st = lib.gjo_matrix_inv_64_nopivot(v.ctypes.data_as(dp), v.size, n, out.ctypes.data_as(dp))
    else:
        st = lib.gjo_matrix_inv_32_nopivot(_fp(v), v.size, n, _fp(out), ARITH_FMA)
    return (out, {"status": st}) if return_info else out


def matrix_inv_32_blocked2(vec, n: int, w=16, bw: int = 256, return_info: bool = False):
    """Exact CPU mirror of the HIP blocked path's block structure (sub-panels w, outer blocks bw).
    w: one width, or a sequence with the sub-panel width of every outer block."""
    lib = _load()
    v = _f32(vec)
    n = int(n)
    if n <= 0 or int(v.size // n) != n:
        empty = np.empty(0, dtype=np.float32)
        return (empty, {"status": STATUS_BAD_SHAPE}) if return_info else empty
    out = np.empty(n * n, dtype=np.float32)
    piv = np.empty(n, dtype=np.int32)
    if isinstance(w, (list, tuple, np.ndarray)):
        ws = np.ascontiguousarray(w, dtype=np.int32)
        st = lib.gjo_matrix_inv_32_blocked2w(_fp(v), v.size, n, _fp(out),
                                             ws.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), int(ws.size), int(bw),
                                             piv.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
    else:
        st = lib.gjo_matrix_inv_32_blocked2(_fp(v), v.size, n, _fp(out), int(w), int(bw),
                                            piv.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
    if return_info:
        return out, {"status": st, "pivots": piv}
    return out


def matrix_inv_32_blocked_exact(vec, n: int, bw: int = 64, return_info: bool = False):
    """The SEQUENTIAL arithmetic (one fmaf per element and step, IEEE division of the pivot rows) evaluated block
    by block: bit-identical to ``matrix_inv_32_inplace`` for every ``bw`` -- the HIP blocked path's operation order
    from round 3 on, and the fast way to the reference-order result at the BASELINE sizes (N = 4096 in ~2 s)."""
    lib = _load()
    v = _f32(vec)
    n = int(n)
    if n <= 0 or int(v.size // n) != n:
        empty = np.empty(0, dtype=np.float32)
        return (empty, {"status": STATUS_BAD_SHAPE}) if return_info else empty
    out = np.empty(n * n, dtype=np.float32)
    piv = np.empty(n, dtype=np.int32)
    st = lib.gjo_matrix_inv_32_blocked_exact(_fp(v), v.size, n, _fp(out), int(bw),
                                             piv.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
    if return_info:
        return out, {"status": st, "pivots": piv}
    return out


def residual_inf(a, x, n: int) -> float:
    """||A X - I||_inf, double accumulation."""
    return float(_load().gjo_residual_inf(_fp(_f32(a)), _fp(_f32(x)), int(n)))


def residual_inf_left(a, x, n: int) -> float:
    """||X A - I||_inf (the side matrix_inv_pyopencl.py:341 checks)."""
    return float(_load().gjo_residual_inf_left(_fp(_f32(a)), _fp(_f32(x)), int(n)))


def frobenius_metric(a, x, n: int) -> float:
    """The reference's own metric sqrt(N) - ||A X||_F (matrix_multiply.cpp:193-200)."""
    return float(_load().gjo_frobenius_metric(_fp(_f32(a)), _fp(_f32(x)), int(n)))


def left_half_is_identity(aug, n: int) -> bool:
    return bool(_load().gjo_left_half_is_identity(_fp(_f32(aug)), int(n)))


def msvc_rand_stream(count: int, seed: int = 1) -> np.ndarray:
    lib = _load()
    st = ctypes.c_uint(seed)
    return np.array([lib.gjo_msvc_rand(ctypes.byref(st)) for _ in range(count)], dtype=np.int32)


def fill_hollow_msvc(k: int, state: int = 1):
    """Sweep-driver input of main_file.cpp:41-52; returns (matrix, next_state)."""
    lib = _load()
    a = np.empty(k * k, dtype=np.float32)
    st = ctypes.c_uint(state)
    lib.gjo_fill_hollow_msvc(_fp(a), int(k), ctypes.byref(st))
    return a.reshape(k, k), int(st.value)


# ---------------------------------------------------------------------------
# Pure-NumPy mirror of the unblocked step semantics (small cases only); used by
# the tests to cross-check the C restatement, never by anything else.
def numpy_mirror_inv(a: np.ndarray) -> np.ndarray:
    a = np.asarray(a, dtype=np.float32)
    n = a.shape[0]
    m = np.concatenate([a.copy(), np.eye(n, dtype=np.float32)], axis=1)
    for r in range(n):
        col = np.abs(m[r:, r])
        col = np.where(np.isnan(col), np.float32(-1), col)
        p = r + int(np.argmax(col))  # first maximum
        piv = m[p, r]
        if p != r:
            m[[r, p]] = m[[p, r]]
        m[r] = m[r] / piv
        rowr = m[r].astype(np.float64)
        for i in range(n):
            if i == r or m[i, r] == 0:
                continue
            # one rounding per element: exact product/sum in float64, then round to fp32.
            # (float64 holds the fp32*fp32 product exactly; the sum is rounded once to
            # double and once to float -- double rounding can differ from fmaf in rare
            # ties, so tests compare this mirror with a 1-ulp-per-step tolerance.)
            m[i] = (m[i].astype(np.float64) - np.float64(m[i, r]) * rowr).astype(np.float32)
    return m[:, n:].copy()
