"""Diagnostic: does splitting a batch over two streams (two contexts) overlap the MFMA-bound rank-bw phases of
one half with the latency/HBM-bound panel phases of the other?  python tools/two_stream_batch.py [n] [batch]"""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import gpu_matrix_inversion_amd as g

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 64
rng = np.random.default_rng(0)
a = torch.from_numpy((rng.uniform(-1, 1, (batch, n, n)) + np.sqrt(n) * np.eye(n)).astype(np.float32)).cuda()
out = torch.empty_like(a)

def run(parts, reps=5, offset_us=0):
    invs = [g.Inverter(algo="blocked") for _ in range(parts)]
    streams = [torch.cuda.Stream() for _ in range(parts)]
    sz = batch // parts
    sts = [torch.empty(sz, dtype=torch.int32, device="cuda") for _ in range(parts)]
    def once():
        for i in range(parts):
            with torch.cuda.stream(streams[i]):
                if i and offset_us: torch.cuda._sleep(int(offset_us * 2400 * i))   # ~2.4 GHz cycles
                invs[i].inv(a[i * sz:(i + 1) * sz], out=out[i * sz:(i + 1) * sz], status=sts[i])
    once(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): once()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    for inv in invs: inv.close()
    return dt

for parts, off in ((1, 0), (2, 0), (2, 150), (2, 300), (2, 450), (2, 600), (3, 0), (3, 250)):
    if batch % parts: continue
    dt = run(parts, offset_us=off)
    print(f"n={n} batch={batch} in {parts} stream(s), offset {off} us: {dt*1e3:.2f} ms  {batch/dt:.0f} matrices/s", flush=True)
