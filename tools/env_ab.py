#!/usr/bin/env python3
"""Diagnostic (not part of the product): time device-resident inversions under two values of an environment switch.

    python tools/env_ab.py MI32_LOOKAHEAD 0,1 4096,5120[,...] [batch]

Each (size, value) runs in a child process of its own (the library reads its switches once).
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import sys, time
sys.path.insert(0, %r)
import numpy as np, torch
import gpu_matrix_inversion_amd as g
n, batch = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(7)
a = np.stack([(rng.uniform(-1, 1, (n, n)) + np.sqrt(n) * np.eye(n))[rng.permutation(n)] for _ in range(batch)]).astype(np.float32)
a = torch.from_numpy(a).cuda()
inv = g.Inverter(algo="blocked")
for _ in range(3):
    inv.inv(a)
torch.cuda.synchronize()
best = 1e9
iters = max(3, int(20 * (4096 / n) ** 2 / batch))
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(iters):
        inv.inv(a)
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / iters)
print("MS", best * 1e3)
""" % ROOT


def main():
    var, values, sizes = sys.argv[1], sys.argv[2].split(","), [int(x) for x in sys.argv[3].split(",")]
    batch = sys.argv[4] if len(sys.argv) > 4 else "1"
    for n in sizes:
        for v in values:
            env = dict(os.environ)
            env[var] = v
            out = subprocess.run([sys.executable, "-c", CHILD, str(n), batch], env=env, capture_output=True, text=True)
            ms = [l for l in out.stdout.splitlines() if l.startswith("MS")]
            print("n %d %s=%s: %s" % (n, var, v, ("%.3f ms" % float(ms[0].split()[1])) if ms else "FAILED " + out.stderr[-300:]), flush=True)


if __name__ == "__main__":
    main()
