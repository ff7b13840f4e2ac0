#!/usr/bin/env python3
"""Diagnostic: randomized bit-exact parity of the HIP paths against the CPU oracle / blocked mirror.
    python tools/stress_parity.py [cases] [seed] [max_n]"""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import gpu_matrix_inversion_amd as g
import oracle as O

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
max_n = int(sys.argv[3]) if len(sys.argv) > 3 else 1400
rng = np.random.default_rng(seed)
blocked, sweep = g.Inverter(algo="blocked"), g.Inverter(algo="sweep")
bad = 0
t0 = time.time()
for c in range(cases):
    n = int(rng.integers(1, max_n + 1)) if c % 3 else int(rng.choice([1, 2, 15, 16, 17, 127, 128, 129, 255, 256, 257, 511, 512, 513, 1023, 1024, 1025]))
    n = min(n, max_n)
    batch = int(rng.integers(1, 5))
    kind = ["gate", "u100", "hollow", "int"][int(rng.integers(0, 4))]
    mats = []
    for b in range(batch):
        if kind == "gate":
            a = (rng.uniform(-1, 1, (n, n)) + np.sqrt(n) * np.eye(n))[rng.permutation(n)]
        elif kind == "u100":
            a = rng.uniform(0, 100, (n, n))
        elif kind == "hollow":
            a = rng.uniform(0, 100, (n, n)); np.fill_diagonal(a, 0.0)
        else:  # small integers: many exact ties in the pivot search
            a = rng.integers(0, 10, (n, n)).astype(np.float64); np.fill_diagonal(a, 0.0)
        mats.append(a.astype(np.float32))
    a = np.stack(mats)
    ta = torch.from_numpy(a).cuda()
    xb, stb = blocked.inv(ta)
    xs, sts = sweep.inv(ta)
    torch.cuda.synchronize()
    for b in range(batch):
        wb, ib = O.matrix_inv_32_inplace(a[b], n, return_info=True)   # both HIP paths follow the reference's order
        ws, isw = O.matrix_inv_32(a[b], n, return_info=True)
        okb = (int(stb[b]) == ib["status"]) and (ib["status"] != 0 or np.array_equal(xb[b].cpu().numpy().reshape(-1), wb))
        oks = (int(sts[b]) == isw["status"]) and (isw["status"] != 0 or np.array_equal(xs[b].cpu().numpy().reshape(-1), ws))
        if not (okb and oks):
            bad += 1
            print(f"MISMATCH case {c}: n={n} batch={batch} b={b} kind={kind} blocked_ok={okb} sweep_ok={oks} status {int(stb[b])}/{ib['status']} {int(sts[b])}/{isw['status']}", flush=True)
    if c % 5 == 4:
        print(f"{c + 1} cases, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
print("stress_parity:", "OK" if bad == 0 else f"{bad} MISMATCHES", f"({cases} cases, seed {seed})")
sys.exit(1 if bad else 0)
