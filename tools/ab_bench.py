#!/usr/bin/env python3
"""Diagnostic (not part of the product): A/B timing of several builds of the library on the SAME box.

    python tools/ab_bench.py [--n 4096] [--batch 1] [--iters 20] [--rounds 2] libA.so libB.so ...

Every build is loaded in its own child process (round-robin, `rounds` times) and times `iters` device-resident
inversions with the wall clock around a synchronised loop; prints the median over the rounds per build.
"""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, time
sys.path.insert(0, {root!r})
import numpy as np, torch
from gpu_matrix_inversion_amd import _lib
_lib.LIB_PATH = {lib!r}
import gpu_matrix_inversion_amd as g
n, batch, iters = {n}, {batch}, {iters}
rng = np.random.default_rng(7)
a = np.stack([(rng.uniform(-1, 1, (n, n)) + np.sqrt(n) * np.eye(n))[rng.permutation(n)] for _ in range(batch)]).astype(np.float32)
a = torch.from_numpy(a).cuda()
inv = g.Inverter(algo={algo!r})
for _ in range(3):
    inv.inv(a)
torch.cuda.synchronize()
best = 1e9
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(iters):
        inv.inv(a)
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / iters)
print("MS", best * 1e3)
"""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=4096)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--algo", default="blocked")
    ap.add_argument("libs", nargs="+")
    args = ap.parse_args()
    res = {lib: [] for lib in args.libs}
    for _ in range(args.rounds):
        for lib in args.libs:
            code = CHILD.format(root=ROOT, lib=os.path.abspath(lib), n=args.n, batch=args.batch, iters=args.iters,
                                algo=args.algo)
            out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
            ms = [float(l.split()[1]) for l in out.stdout.splitlines() if l.startswith("MS")]
            if not ms:
                print(lib, "FAILED", out.stderr[-400:])
                continue
            res[lib].append(ms[0])
    for lib, v in res.items():
        v = sorted(v)
        print(f"{os.path.basename(lib):32s} n={args.n} batch={args.batch}: " + " ".join(f"{x:.3f}" for x in v) + " ms")


if __name__ == "__main__":
    main()
