#!/usr/bin/env python3
"""The reference's size sweep, on the HIP path.

Reference: the sweep drivers of /root/reference -- matrix_inv_pyopencl.py:358-371 (i = 10 .. 16000,
step 10 below 2000 and 1000 above, U(0,100) fp32 input, one line `N t_compute t_total err` per size,
err = sqrt(N) - sqrt(sum(C @ C)) with C = inv * A, PY:341-352) and main_file.cpp:27-84 (hollow
rand()%10 inputs, `k errore`).  Same loop, same line format, same metric; the inversion goes through
the drop-in `matrix_inv_32(vec, N)` (host pointers, so t_total includes the PCIe copies exactly like
the reference's "Tempo Totale Impiegato", and t_compute is its "Tempo Computazione").

    python tools/sweep_series.py [--max 16000] [--out series.txt] [--hollow] [--coarse] [--times10 times.txt]

--times10 FILE additionally writes one line `N t0 ... t9` per size: the reference's ten-slot timing vector
(FP32_bench.cpp:256-443: queue, buffers, build, makeAug, pivot, row, column, compute, getInverted, total) from
the benchmark twin FP32_bench / mi32_bench_32.
"""
import argparse
import math
import sys
import os

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpu_matrix_inversion_amd as g  # noqa: E402

try:  # only the error metric of the large sizes uses torch (the checker, not the inversion)
    import torch as _torch
    if not _torch.cuda.is_available():
        _torch = None
except Exception:  # pragma: no cover
    _torch = None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--max", type=int, default=16000)
    ap.add_argument("--out", default="-")
    ap.add_argument("--hollow", action="store_true", help="zero diagonal (matrix_inv_numpy.py:13-14)")
    ap.add_argument("--coarse", action="store_true", help="step 100 below 2000 instead of the reference's 10")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--times10", default="", help="also write the reference's 10-slot timing vector per size")
    args = ap.parse_args()
    out = sys.stdout if args.out == "-" else open(args.out, "w")
    tout = open(args.times10, "w") if args.times10 else None
    rng = np.random.default_rng(args.seed)
    i = 10
    while i < args.max:
        a = rng.uniform(0, 100, (i, i)).astype(np.float32)  # PY:17
        if args.hollow:
            np.fill_diagonal(a, 0.0)
        if tout is not None:
            x, t10 = g.fp32_bench(a.reshape(-1), i)
            tout.write(f"{i} " + " ".join(repr(t10[k]) for k in t10) + "\n" if t10 else f"{i}" + " nan" * 10 + "\n")
            tout.flush()
        else:
            x = g.matrix_inv_32(a.reshape(-1), i)
        t_total, t_compute = g.last_timing()
        if x.size == 0:
            out.write(f"{i} nan nan nan\n")
        else:
            if i >= 1500 and _torch is not None:  # the checker's two float64 products on the GPU (8 TFLOP at N = 16000)
                tx = _torch.from_numpy(x.reshape(i, i)).cuda().double()
                c = tx @ _torch.from_numpy(a).cuda().double()                 # PY:341  C = inv * A
                err = math.sqrt(i) - math.sqrt(abs(float((c @ c).sum())))      # PY:342-345 (matrix product c @ c)
                del tx, c
            else:
                c = x.reshape(i, i).astype(np.float64) @ a.astype(np.float64)  # PY:341  C = inv * A
                err = math.sqrt(i) - math.sqrt(abs(float(np.sum(c @ c))))     # PY:342-345 (matrix product c @ c)
            out.write(f"{i} {t_compute} {t_total} {err}\n")                 # PY:352
        out.flush()
        i += (100 if args.coarse else 10) if i < 2000 else 1000            # PY:365-368
    if out is not sys.stdout:
        out.close()
    if tout is not None:
        tout.close()


if __name__ == "__main__":
    main()
