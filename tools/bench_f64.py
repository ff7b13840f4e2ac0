#!/usr/bin/env python3
"""Diagnostic: fp64 inversion time (matrix_inversion_FP64 of the reference), sweep vs blocked, per block width."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import gpu_matrix_inversion_amd as g  # noqa: E402


def gate(n, seed):
    rng = np.random.default_rng(seed)
    a = rng.uniform(-1.0, 1.0, (n, n)) + np.sqrt(n) * np.eye(n)
    return a[rng.permutation(n)]


def main():
    sizes = [int(v) for v in sys.argv[1:]] or [1024, 2048, 4096, 8192]
    for n in sizes:
        a = torch.from_numpy(gate(n, n)).cuda()
        eye = torch.eye(n, dtype=torch.float64, device="cuda")
        for label, kw in (("sweep", dict(algo="sweep")), ("blocked bw64", dict(algo="auto", block_width=64)),
                          ("blocked bw128", dict(algo="auto", block_width=128)),
                          ("blocked bw256", dict(algo="auto", block_width=256))):
            if label == "sweep" and n > 4096:
                continue
            inv = g.Inverter(**kw)
            x, st = inv.inv(a)
            torch.cuda.synchronize()
            reps = 3 if n >= 4096 else 5
            t0 = time.perf_counter()
            for _ in range(reps):
                inv.inv(a, out=x)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            res = float((a @ x - eye).abs().sum(dim=1).max())
            inv.set_profiling(True)
            inv.get_profile()
            inv.inv(a, out=x)
            prof = {k: round(v[0], 2) for k, v in inv.get_profile().items() if v[1]}
            print(f"N={n:5d} {label:14s} {1e3 * dt:9.2f} ms  {2.0 * n ** 3 / dt / 1e12:6.2f} TFLOP/s  residual {res:.2e}  status {int(st[0])}  {prof}")
            inv.close()


if __name__ == "__main__":
    main()
