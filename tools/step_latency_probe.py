import sys, time, numpy as np, torch
sys.path.insert(0, ".")
import gpu_matrix_inversion_amd as g
def dom(n, seed, dt):
    rng = np.random.default_rng(seed); a = rng.uniform(-1, 1, (n, n)); a[np.arange(n), np.arange(n)] = np.abs(a).sum(axis=1) + 1; return a.astype(dt)
for dt in (np.float64, np.float32):
    for n in (512, 1024):
        a = torch.from_numpy(dom(n, n, dt)).cuda()
        for label, kw in (("sweep pivoting", dict(algo="sweep")), ("sweep no-pivot", dict(algo="sweep", pivoting=False))):
            inv = g.Inverter(**kw)
            x, st = inv.inv(a); torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(5): inv.inv(a, out=x)
            torch.cuda.synchronize()
            dtm = (time.perf_counter() - t) / 5
            print(dt.__name__, n, label, "%.2f ms  %.2f us/step" % (1e3 * dtm, 1e6 * dtm / n))
            inv.close()
