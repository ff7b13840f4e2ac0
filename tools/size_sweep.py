"""Diagnostic: time per inversion for a range of N, both algorithms (device-resident, median of reps)."""
import sys, time, statistics, numpy as np, torch
sys.path.insert(0, '.')
import gpu_matrix_inversion_amd as g

def gate(n, seed):
    rng = np.random.default_rng(seed)
    a = rng.uniform(-1, 1, (n, n)).astype(np.float32)
    a[np.arange(n), np.arange(n)] += np.float32(np.sqrt(n))
    return a[rng.permutation(n)]

sizes = [int(x) for x in sys.argv[1:]] or [16, 64, 128, 256, 512, 1024, 2048, 4096]
for algo in ("sweep", "blocked"):
    inv = g.Inverter(algo=algo)
    for n in sizes:
        if algo == "sweep" and n > 4096: continue
        a = torch.from_numpy(gate(n, n)).cuda()
        out = torch.empty_like(a); st = torch.empty(1, dtype=torch.int32, device="cuda")
        inv.reserve(n, 1)
        for _ in range(2): inv.inv(a, out=out.view(1, n, n), status=st)
        torch.cuda.synchronize()
        ts = []
        reps = 5 if n >= 2048 else 20
        for _ in range(reps):
            t0 = time.perf_counter(); inv.inv(a, out=out.view(1, n, n), status=st); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        t = statistics.median(ts)
        r = float(inv.residual(a, out)[0, 0])
        print(f"{algo:8s} n={n:5d} {t*1e3:9.3f} ms  {2*n**3/t/1e9:10.1f} GFLOP/s  {8.0*n*n*(n+1)/t/1e9:8.1f} GB/s(alg. sweep bytes)  residual={r:.1e}", flush=True)
    inv.close()
