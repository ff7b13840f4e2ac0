import sys, os, time, statistics
sys.path.insert(0, os.getcwd())
import numpy as np
import gpu_matrix_inversion_amd as g
n = 4096
rng = np.random.default_rng(7)
a = (rng.uniform(-1, 1, (n, n)) + np.sqrt(n) * np.eye(n)).astype(np.float32).reshape(-1)
g.matrix_inv_32(a, n)
ts = []
for _ in range(9):
    t0 = time.perf_counter(); x = g.matrix_inv_32(a, n); ts.append(time.perf_counter() - t0)
print("e2e ms:", " ".join(f"{t*1e3:.1f}" for t in ts), "median", round(statistics.median(ts) * 1e3, 2))
