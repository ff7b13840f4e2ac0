"""Python host side of the drop-in: the reference's call shapes over the C ABI.

* ``matrix_inv_32(vec, N)`` -- the library entry point
  (/root/reference/Matlab/mat_inv_32.h:4; MATLAB calls it as
  ``clib.matInv.matrix_inv_32(b, N)``, README.md:51): flat row-major fp32 in,
  flat row-major inverse out, EMPTY array for an invalid matrix (README.md:54).
* ``just_inv(K)`` -- the call shape of the reference's CPU script
  (/root/reference/matrix_inv_numpy.py:39-46): build a K x K U(0,100) matrix,
  time only the inversion with a monotonic clock, print ``TIME: <seconds>``.
* ``Inverter`` -- device-resident path for torch tensors (no host copies), used
  by bench.py and the multi-GPU driver.

PyTorch is plumbing only (device memory, streams); all arithmetic happens in
the HIP kernels of ``lib/libmat_inv_32.so``.
"""
from __future__ import annotations

import ctypes
import os
import time

import numpy as np

from . import _lib
from ._lib import ALGO_AUTO, ALGO_BLOCKED, ALGO_NAMES, ALGO_SWEEP, MI32_OK, MI32_SINGULAR, Mi32Error


def _algo_id(algo) -> int:
    if isinstance(algo, str):
        return ALGO_NAMES[algo.lower()]
    return int(algo)


def matrix_inv_32(matrix_vector, matrix_order: int) -> np.ndarray:
    """Drop-in for ``matrix_inv_32(std::vector<float>, int)``.

    Returns the flat row-major inverse (``float32``, ``N*N`` entries) or an empty
    array when the reference would return an empty vector: ``N <= 0``,
    ``int(len/N) != N`` (mat_inv_32.cpp:206-215), or a singular input (README.md:54;
    set ``MI32_SINGULAR_KEEP=1`` to get the shipped library's inf/NaN result instead).
    """
    lib = _lib.load()
    n = int(matrix_order)
    v = np.ascontiguousarray(np.asarray(matrix_vector, dtype=np.float32).reshape(-1))
    if n <= 0 or int(v.size // n) != n:
        return np.empty(0, dtype=np.float32)
    out = np.empty(n * n, dtype=np.float32)
    fp = ctypes.POINTER(ctypes.c_float)
    rc = lib.mi32_matrix_inv_32(v.ctypes.data_as(fp), v.size, n, out.ctypes.data_as(fp))
    if rc == MI32_OK:
        return out
    if rc == MI32_SINGULAR:
        return out if os.environ.get("MI32_SINGULAR_KEEP", "0") not in ("", "0") else np.empty(0, dtype=np.float32)
    if rc == _lib.MI32_RUNTIME_ERROR:
        raise Mi32Error(lib.mi32_last_error().decode())
    return np.empty(0, dtype=np.float32)


def matrix_inv_64(matrix_vector, matrix_order: int) -> np.ndarray:
    """Drop-in for the reference's ``matrix_inversion_FP64(std::vector<double>, int)`` (headers.h:9): flat
    row-major float64 in, flat inverse out, empty array for a bad shape or a singular input."""
    lib = _lib.load()
    n = int(matrix_order)
    v = np.ascontiguousarray(np.asarray(matrix_vector, dtype=np.float64).reshape(-1))
    if n <= 0 or int(v.size // n) != n:
        return np.empty(0, dtype=np.float64)
    out = np.empty(n * n, dtype=np.float64)
    dp = ctypes.POINTER(ctypes.c_double)
    rc = lib.mi32_matrix_inv_64(v.ctypes.data_as(dp), v.size, n, out.ctypes.data_as(dp))
    if rc == MI32_OK:
        return out
    if rc == MI32_SINGULAR:
        return out if os.environ.get("MI32_SINGULAR_KEEP", "0") not in ("", "0") else np.empty(0, dtype=np.float64)
    if rc == _lib.MI32_RUNTIME_ERROR:
        raise Mi32Error(lib.mi32_last_error().decode())
    return np.empty(0, dtype=np.float64)


def matrix_inversion_no_pivots(matrix_vector, matrix_order: int) -> np.ndarray:
    """Drop-in for the reference's ``matrix_inversion_no_pivots(std::vector<double>, int)`` (headers.h:11,
    matrix_inversion_no_pivots.cpp:10): Gauss-Jordan in double with the diagonal entry as every step's pivot --
    for diagonally dominant inputs.  Empty array for a bad shape or when a zero diagonal entry is met."""
    lib = _lib.load()
    n = int(matrix_order)
    v = np.ascontiguousarray(np.asarray(matrix_vector, dtype=np.float64).reshape(-1))
    if n <= 0 or int(v.size // n) != n:
        return np.empty(0, dtype=np.float64)
    out = np.empty(n * n, dtype=np.float64)
    dp = ctypes.POINTER(ctypes.c_double)
    rc = lib.mi32_matrix_inversion_no_pivots(v.ctypes.data_as(dp), v.size, n, out.ctypes.data_as(dp))
    if rc == MI32_OK:
        return out
    if rc == MI32_SINGULAR:
        return out if os.environ.get("MI32_SINGULAR_KEEP", "0") not in ("", "0") else np.empty(0, dtype=np.float64)
    if rc == _lib.MI32_RUNTIME_ERROR:
        raise Mi32Error(lib.mi32_last_error().decode())
    return np.empty(0, dtype=np.float64)


def matrix_inv_32_batched(a: np.ndarray, ngpus: int = 1):
    """Host batch (B, N, N) -> (inverses (B, N, N), status int32[B]).  ``ngpus`` != 1: the batch is sharded over that
    many GPUs of this node (0 = all visible) inside the library, one host thread and context per GPU
    (``mi32_matrix_inv_32_batched_multi``): no launcher, no torch.distributed."""
    lib = _lib.load()
    a = np.ascontiguousarray(a, dtype=np.float32)
    if a.ndim != 3 or a.shape[1] != a.shape[2] or a.shape[0] == 0 or a.shape[1] == 0:
        raise ValueError("expected a (B, N, N) array")
    b, n = a.shape[0], a.shape[1]
    out = np.empty_like(a)
    st = np.empty(b, dtype=np.int32)
    fp = ctypes.POINTER(ctypes.c_float)
    if ngpus != 1:
        rc = lib.mi32_matrix_inv_32_batched_multi(a.ctypes.data_as(fp), n, b, out.ctypes.data_as(fp),
                                                  st.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), int(ngpus))
        if rc == _lib.MI32_BAD_SHAPE:
            raise ValueError("more GPUs asked for than are visible")
    else:
        rc = lib.mi32_matrix_inv_32_batched(a.ctypes.data_as(fp), n, b, out.ctypes.data_as(fp),
                                            st.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
    if rc == _lib.MI32_RUNTIME_ERROR:
        raise Mi32Error(lib.mi32_last_error().decode())
    return out, st


def fp32_bench(matrix_vector, matrix_order: int):
    """The reference's ``Res FP32_bench(vector<float>, int)`` (FP32_bench.cpp:11): returns ``(inverse, times)``
    with ``times`` the ten durations of FP32_bench.cpp:256-443 in seconds, keyed by ``_lib.TIMES10_SLOTS``
    (queue, buffers, build, makeAug, pivot, row, column, compute, getInverted, total); ``(empty, {})`` where the
    reference returns an empty Res."""
    lib = _lib.load()
    n = int(matrix_order)
    v = np.ascontiguousarray(np.asarray(matrix_vector, dtype=np.float32).reshape(-1))
    if n <= 0 or int(v.size // n) != n:
        return np.empty(0, dtype=np.float32), {}
    out = np.empty(n * n, dtype=np.float32)
    times = (ctypes.c_double * 10)()
    fp = ctypes.POINTER(ctypes.c_float)
    rc = lib.mi32_bench_32(v.ctypes.data_as(fp), v.size, n, out.ctypes.data_as(fp), times)
    if rc == _lib.MI32_RUNTIME_ERROR:
        raise Mi32Error(lib.mi32_last_error().decode())
    if rc != MI32_OK:
        return np.empty(0, dtype=np.float32), {}
    return out, dict(zip(_lib.TIMES10_SLOTS, (float(t) for t in times)))


def fp64_bench(matrix_vector, matrix_order: int, pivoting: bool = True):
    """``Res FP64_bench`` / ``Res no_pivots_bench`` of the reference (headers.h:14,16): ``(inverse float64, times)``."""
    lib = _lib.load()
    n = int(matrix_order)
    v = np.ascontiguousarray(np.asarray(matrix_vector, dtype=np.float64).reshape(-1))
    if n <= 0 or int(v.size // n) != n:
        return np.empty(0, dtype=np.float64), {}
    out = np.empty(n * n, dtype=np.float64)
    times = (ctypes.c_double * 10)()
    dp = ctypes.POINTER(ctypes.c_double)
    rc = lib.mi32_bench_64(v.ctypes.data_as(dp), v.size, n, out.ctypes.data_as(dp), times, 1 if pivoting else 0)
    if rc == _lib.MI32_RUNTIME_ERROR:
        raise Mi32Error(lib.mi32_last_error().decode())
    if rc != MI32_OK:
        return np.empty(0, dtype=np.float64), {}
    return out, dict(zip(_lib.TIMES10_SLOTS, (float(t) for t in times)))


def matrix_multiply(matrice_a, matrice_b) -> float:
    """The reference's verification helper ``matrix_multiply`` (matrix_multiply.cpp:15): ``sqrt(N) - ||A B||_F`` with the
    product in double on the device; flat or square float64 operands of N*N entries each."""
    lib = _lib.load()
    a = np.ascontiguousarray(np.asarray(matrice_a, dtype=np.float64).reshape(-1))
    b = np.ascontiguousarray(np.asarray(matrice_b, dtype=np.float64).reshape(-1))
    if a.size != b.size:
        raise ValueError("operands of different size")
    err = ctypes.c_double()
    dp = ctypes.POINTER(ctypes.c_double)
    _lib.check(lib.mi32_matrix_multiply_64(a.ctypes.data_as(dp), b.ctypes.data_as(dp), a.size, ctypes.byref(err)),
               "mi32_matrix_multiply_64")
    return err.value


def last_timing():
    """(total_seconds, compute_seconds) of the last host-pointer call: the two numbers the
    reference prints (mat_inv_32.cpp:385-386)."""
    t, c = ctypes.c_double(), ctypes.c_double()
    _lib.load().mi32_last_timing(ctypes.byref(t), ctypes.byref(c))
    return t.value, c.value


def just_inv(K: int, seed=None, inv=None):
    """The reference CPU script's call shape (matrix_inv_numpy.py:39-46) on the GPU path:
    U(0,100) K x K matrix, time only the inversion, print ``TIME: <s>``.  Returns the
    elapsed seconds (the reference prints only).  ``inv`` lets the CPU-baseline harness
    time ``numpy.linalg.inv`` through the very same shape."""
    rng = np.random.default_rng(seed)
    a = rng.uniform(0, 100, (K, K)).astype(np.float32)
    fn = inv if inv is not None else (lambda m: matrix_inv_32(m.reshape(-1), K))
    start = time.monotonic()
    res = fn(a)
    end = time.monotonic()
    print(f"TIME: {end - start}")
    return end - start, a, res


class Inverter:
    """Device-resident inversion of torch CUDA(HIP) tensors through the C ABI handle."""

    def __init__(self, device=None, algo="auto", panel_width: int = 0, block_width: int = 0, pivoting: bool = True):
        import torch

        self._torch = torch
        if not torch.cuda.is_available():
            raise Mi32Error("no HIP device visible to torch; the product path has no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else
                                   (device.index if isinstance(device, torch.device) else int(device)))
        self._lib = _lib.load()
        h = ctypes.c_void_p()
        _lib.check(self._lib.mi32_create(ctypes.byref(h), self.device.index), "mi32_create")
        self._h = h
        self.algo = _algo_id(algo)
        _lib.check(self._lib.mi32_set_algo(self._h, self.algo), "mi32_set_algo")
        if panel_width or block_width:
            _lib.check(self._lib.mi32_set_blocking(self._h, panel_width, block_width), "mi32_set_blocking")
        if not pivoting:  # the reference's no-pivot variant (matrix_inversion_no_pivots.cpp:10); fp32: blocked from 512 rows on
            _lib.check(self._lib.mi32_set_pivoting(self._h, 0), "mi32_set_pivoting")

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mi32_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def _bind_stream(self):
        s = self._torch.cuda.current_stream(self.device)
        _lib.check(self._lib.mi32_set_stream(self._h, ctypes.c_void_p(s.cuda_stream)), "mi32_set_stream")

    def resolved_algo(self, n: int, batch: int = 1) -> int:
        return self._lib.mi32_resolve_algo(self._h, int(n), int(batch))

    def resolved_blocking(self, n: int, batch: int = 1):
        """(sub-panel width, outer block width) the blocked path uses for this shape."""
        w, bw = ctypes.c_int(), ctypes.c_int()
        _lib.check(self._lib.mi32_resolve_blocking(self._h, int(n), int(batch), ctypes.byref(w), ctypes.byref(bw)),
                   "mi32_resolve_blocking")
        return w.value, bw.value

    def resolved_blocking_f64(self, n: int) -> int:
        """Outer block width of the fp64 blocked path for this order; 0 where the unblocked sweep runs."""
        bw = ctypes.c_int()
        _lib.check(self._lib.mi32_resolve_blocking_f64(self._h, int(n), ctypes.byref(bw)), "mi32_resolve_blocking_f64")
        return bw.value

    def resolved_panel_widths(self, n: int, batch: int = 1):
        """Sub-panel width of every outer block (narrow while many rows are still candidates)."""
        nb = ctypes.c_int()
        buf = (ctypes.c_int * 128)()
        _lib.check(self._lib.mi32_resolve_panel_widths(self._h, int(n), int(batch), buf, 128, ctypes.byref(nb)),
                   "mi32_resolve_panel_widths")
        return [int(buf[i]) for i in range(min(nb.value, 128))]

    def dominant_kernel(self, n: int, batch: int = 1) -> str:
        return self._lib.mi32_dominant_kernel(self.resolved_algo(n, batch)).decode()

    def reserve(self, n: int, batch: int = 1):
        _lib.check(self._lib.mi32_reserve(self._h, int(n), int(batch)), "mi32_reserve")

    def inv(self, a, out=None, status=None):
        """a: (N,N) or (B,N,N) float32 (or float64: the fp64 twin, sweep path) contiguous tensor on this
        device.  Asynchronous on torch's current stream.  Returns (inverse, status int32[B] tensor)."""
        torch = self._torch
        if a.dtype not in (torch.float32, torch.float64) or not a.is_cuda:
            raise ValueError("expected a float32 or float64 tensor on the GPU")
        squeeze = a.dim() == 2
        a3 = a.unsqueeze(0) if squeeze else a
        if a3.dim() != 3 or a3.shape[1] != a3.shape[2] or a3.shape[0] == 0 or a3.shape[1] == 0:
            raise ValueError("expected (N,N) or (B,N,N)")
        a3 = a3.contiguous()
        b, n = a3.shape[0], a3.shape[1]
        if out is None:
            out = torch.empty_like(a3)
        else:
            out = out.view(b, n, n)
            if not out.is_contiguous() or out.data_ptr() == a3.data_ptr():
                raise ValueError("out must be contiguous and must not alias the input")
        if status is None:
            status = torch.empty(b, dtype=torch.int32, device=a3.device)
        self._bind_stream()
        fn = self._lib.mi32_inv_device if a.dtype == torch.float32 else self._lib.mi32_inv_device_f64
        _lib.check(fn(self._h, ctypes.c_void_p(a3.data_ptr()), n, b, ctypes.c_void_p(out.data_ptr()),
                      ctypes.c_void_p(status.data_ptr())), "mi32_inv_device")
        return (out[0] if squeeze else out), status

    def set_lookahead(self, enable: bool):
        _lib.check(self._lib.mi32_set_lookahead(self._h, 1 if enable else 0), "mi32_set_lookahead")

    def set_profiling(self, enable: bool):
        _lib.check(self._lib.mi32_set_profiling(self._h, 1 if enable else 0), "mi32_set_profiling")

    def get_profile(self):
        """{class: (milliseconds, launches)} since the last call (synchronises the recorded events)."""
        k = len(_lib.KERNEL_CLASSES)
        ms = (ctypes.c_double * k)()
        cnt = (ctypes.c_longlong * k)()
        _lib.check(self._lib.mi32_get_profile(self._h, ms, cnt, k), "mi32_get_profile")
        return {name: (ms[i], int(cnt[i])) for i, name in enumerate(_lib.KERNEL_CLASSES)}

    def residual(self, a, x):
        """Device-side check: returns a (B,3) float64 tensor [||AX-I||_inf, ||XA-I||_inf, sqrt(N)-||AX||_F]."""
        torch = self._torch
        a3 = (a.unsqueeze(0) if a.dim() == 2 else a).contiguous()
        x3 = (x.unsqueeze(0) if x.dim() == 2 else x).contiguous()
        b, n = a3.shape[0], a3.shape[1]
        out = torch.empty(b, 3, dtype=torch.float64, device=a3.device)
        self._bind_stream()
        _lib.check(self._lib.mi32_residual_device(self._h, ctypes.c_void_p(a3.data_ptr()),
                                                  ctypes.c_void_p(x3.data_ptr()), n, b,
                                                  ctypes.c_void_p(out.data_ptr())), "mi32_residual_device")
        return out
