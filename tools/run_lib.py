#!/usr/bin/env python3
"""Diagnostic: run `iters` device-resident inversions with a given build of the library (for rocprofv3 A/B runs).
    python tools/run_lib.py path/to/lib.so [n] [batch] [iters] [algo]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from gpu_matrix_inversion_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.abspath(sys.argv[1])
import gpu_matrix_inversion_amd as g  # noqa: E402

n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 1
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 5
algo = sys.argv[5] if len(sys.argv) > 5 else "blocked"
rng = np.random.default_rng(7)
a = np.stack([(rng.uniform(-1, 1, (n, n)) + np.sqrt(n) * np.eye(n))[rng.permutation(n)] for _ in range(batch)]).astype(np.float32)
a = torch.from_numpy(a).cuda()
inv = g.Inverter(algo=algo)
for _ in range(iters):
    inv.inv(a)
torch.cuda.synchronize()
