"""Diagnostic: correctness (device residual + exact scaling property) and time at large N."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import gpu_matrix_inversion_amd as g

def gate(n, seed):
    rng = np.random.default_rng(seed)
    a = rng.uniform(-1, 1, (n, n)).astype(np.float32)
    a[np.arange(n), np.arange(n)] += np.float32(np.sqrt(n))
    return a[rng.permutation(n)]

for n in [int(x) for x in sys.argv[1:]]:
    a = torch.from_numpy(gate(n, n)).cuda()
    inv = g.Inverter(algo="blocked")
    inv.reserve(n, 1)
    x, st = inv.inv(a); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2): x, st = inv.inv(a)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 2
    r = inv.residual(a, x)
    x2, _ = inv.inv(a * 2.0)
    torch.cuda.synchronize()
    print(f"n={n} status={int(st.item())} time={dt*1e3:.1f} ms  {2*n**3/dt/1e12:.2f} TFLOP/s  residual={float(r[0,0]):.2e} left={float(r[0,1]):.2e} scale-exact={bool(torch.equal(x2*2.0, x))}", flush=True)
    inv.close()
