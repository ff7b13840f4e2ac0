#!/usr/bin/env python3
"""bench.py -- the hot path of BASELINE.json on N GPUs of one node.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic input resident in HBM:
by default config C1 of BASELINE.json -- ONE 4096 x 4096 fp32 inversion per GPU per step
(``--n/--batch`` select C2 = 64 x 2048^2 etc.).  With N GPUs every rank inverts its own
matrices (independent units, no data-path collective): weak scaling.  Rank 0 prints ONE JSON
line: value = matrices/s over all ranks = N * batch * K / T, T = max over ranks of the time of
exactly K steps bracketed by barrier + torch.cuda.synchronize().

Extra objects on the same line:
  roofline      -- the dominant kernel's algorithmic FLOP/s (or B/s) per launch: launch
                   durations come from HIP events recorded on the launch stream (inside the
                   library, mi32_set_profiling) over a second, instrumented pass of the same K
                   steps, so the un-instrumented pass that yields `value` is not perturbed.
                   (That pass runs with the look-ahead split off, so that every rank-bw update
                   is one full-size launch of the kernel the roofline is quoted for.)
  cpu_baseline  -- numpy.linalg.inv (the reference's CPU path, matrix_inv_numpy.py:44, through
                   our just_inv-shaped harness) on the same fp32 input on this box's host cores.
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

# the only published number for this metric -- N=4096 fp32, kernel loop only
# ("Tempo Computazione" 2.92434 s on an RX 5700, test_inversa_mat.mlx) = 0.342 matrices/s.
PUBLISHED_MATRICES_PER_S_N4096 = 1.0 / 2.92434
PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense fp32 MFMA = fp32 vector peak
PEAK_HBM_GBPS = 8000.0          # HBM3E spec


def gate_matrix(n, seed):
    """D_gate: row-permuted U(-1,1) + sqrt(N) I -- well conditioned, ~N row swaps."""
    rng = np.random.default_rng(seed)
    a = rng.uniform(-1.0, 1.0, (n, n)) + np.sqrt(n) * np.eye(n)
    return a[rng.permutation(n)].astype(np.float32)


def cpu_baseline(a, budget_s=20.0):
    """numpy.linalg.inv on the same fp32 matrix, all host cores NumPy's BLAS uses; bounded."""
    import gpu_matrix_inversion_amd as g

    n = a.shape[0]
    threads = os.cpu_count() or 1
    blas = "unknown"
    try:
        from threadpoolctl import threadpool_info

        info = [i for i in threadpool_info() if i.get("user_api") == "blas"]
        if info:
            threads = int(info[0].get("num_threads", threads))
            blas = f"{info[0].get('internal_api')} {info[0].get('version')}"
    except Exception:
        pass
    t0 = time.perf_counter()
    np.linalg.inv(a)  # warm-up: BLAS thread-pool start-up
    first = time.perf_counter() - t0
    times = []
    import contextlib
    import io

    while len(times) < 5 and (sum(times) + first) < budget_s:
        with contextlib.redirect_stdout(io.StringIO()):
            dt, _, _ = g.just_inv(n, inv=lambda m: np.linalg.inv(a))  # the reference script's call shape
        times.append(dt)
    med = statistics.median(times) if times else first
    return {
        "value": 1.0 / med,
        "unit": "matrices/s",
        "cores": threads,
        "kind": "reference",
        "what": "numpy.linalg.inv on the same fp32 input (the reference CPU path's call, "
                "matrix_inv_numpy.py:44, via the just_inv-shaped harness)",
        "sample": f"{len(times)} timed calls of one {n}x{n} fp32 inversion after 1 warm-up, median",
        "seconds_per_matrix": med,
        "gflops_2n3": 2.0 * n ** 3 / med / 1e9,
        "blas": blas,
        "numpy": np.__version__,
        "host_cpus": os.cpu_count(),
    }


def oracle_port_baseline(n_small=1024):
    """The scalar C oracle (our CPU restatement of the reference's own algorithm), 1 thread."""
    try:
        import oracle as O

        a = gate_matrix(n_small, 5)
        t0 = time.perf_counter()
        O.matrix_inv_32_inplace(a, n_small)
        dt = time.perf_counter() - t0
        return {"value": 1.0 / dt, "unit": "matrices/s", "cores": 1, "kind": "port",
                "sample": f"one {n_small}x{n_small} fp32 Gauss-Jordan inversion by oracle/gj_oracle.c "
                          f"(O(N^3): x{(4096 / n_small) ** 3:.0f} for N=4096)",
                "seconds_per_matrix": dt}
    except Exception as e:  # the oracle is optional for the bench
        return {"error": repr(e)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n", type=int, default=4096)
    ap.add_argument("--batch", type=int, default=1, help="matrices per GPU per step")
    ap.add_argument("--algo", default="auto", choices=["auto", "sweep", "blocked"])
    ap.add_argument("--panel-width", type=int, default=0)
    ap.add_argument("--block-width", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile-pass", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import gpu_matrix_inversion_amd as g

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    n, batch = args.n, args.batch
    host = np.stack([gate_matrix(n, 1000 * 1 + rank * batch + b) for b in range(batch)])
    a = torch.from_numpy(host).to(dev)
    out = torch.empty_like(a)
    status = torch.empty(batch, dtype=torch.int32, device=dev)
    inv = g.Inverter(device=dev, algo=args.algo, panel_width=args.panel_width, block_width=args.block_width)
    inv.reserve(n, batch)
    algo_id = inv.resolved_algo(n, batch)
    algo_name = {g.ALGO_SWEEP: "sweep", g.ALGO_BLOCKED: "blocked"}[algo_id]

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        inv.inv(a, out=out, status=status)
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        inv.inv(a, out=out, status=status)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    torch.cuda.synchronize()

    # correctness of what was just timed (outside the timed region)
    res = inv.residual(a, out)
    torch.cuda.synchronize()
    res_right = float(res[:, 0].max())
    res_left = float(res[:, 1].max())
    st_max = int(status.max())

    # instrumented pass: HIP events around every launch, on the launch stream
    roof = None
    breakdown = None
    if not args.no_profile_pass:
        inv.set_lookahead(False)  # one clean full-size launch of the rank-bw kernel per block
        inv.set_profiling(True)
        inv.get_profile()
        torch.cuda.synchronize()
        tp0 = time.perf_counter()
        for _ in range(args.steps):
            inv.inv(a, out=out, status=status)
        torch.cuda.synchronize()
        instrumented = time.perf_counter() - tp0
        prof = inv.get_profile()
        inv.set_profiling(False)
        inv.set_lookahead(True)
        breakdown = {k: {"ms_per_step": v[0] / args.steps, "launches_per_step": v[1] / args.steps,
                         "avg_us": (1e3 * v[0] / v[1]) if v[1] else None} for k, v in prof.items() if v[1]}
        # the roofline object below is for the kernel that carries the algorithmic flops (the fp32-MFMA
        # rank-bw update: 2 N^3 (1 - bw/N) of the 2 N^3).  For a SINGLE matrix the time-dominant kernel
        # is the latency-bound panel kernel (one workgroup, one barrier per pivot step): no roofline
        # applies to it; its share is reported here.
        tot_ms = sum(v[0] for v in prof.values()) or 1.0
        time_dominant = max(prof.items(), key=lambda kv: kv[1][0])
        breakdown["_time_dominant"] = {"class": time_dominant[0], "share_of_kernel_time": time_dominant[1][0] / tot_ms}
        if algo_id == g.ALGO_BLOCKED:
            ms, cnt = prof["update_rank_bw"]
            _, bw = inv.resolved_blocking(n, batch)
            # ALGORITHMIC flops of one rank-bw update launch: 2 * N * (N - bw) * bw per matrix
            # (sum over the N/bw launches = 2 N^3 (1 - bw/N): the block's own columns are done in-panel)
            flops = 2.0 * n * max(n - bw, 0) * bw * batch
            avg_s = (ms / cnt) * 1e-3 if cnt else float("nan")
            ach = flops / avg_s / 1e12 if cnt else None
            # HBM bytes per launch from the PMC passes (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, x1024),
            # collected offline with rocprofv3 --pmc (separate passes) on this exact config
            traffic = None
            pmc_path = os.path.join(ROOT, "profiles", "round1", "pmc_rank_bw2_n4096.json")
            if n == 4096 and batch == 1 and bw == 256 and os.path.exists(pmc_path):
                traffic = json.load(open(pmc_path)).get("hbm_bytes_per_launch_corrected")
            roof = {"bound": "mfma", "kernel": "gj_rank_bw2_kernel", "achieved": ach,
                    "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": (ach / PEAK_FP32_MFMA_TFLOPS) if ach else None,
                    "traffic": traffic, "avg_launch_us": avg_s * 1e6, "launches_per_step": cnt / args.steps,
                    "algorithmic_flops_per_launch": flops,
                    "share_of_step_time": (ms / args.steps) / (1e3 * instrumented / args.steps)}
        else:
            ms, cnt = prof["sweep_step"]
            bytes_per_launch = 8.0 * n * (n + 1) * batch  # one fp32 read + write of N rows x (N+1) live columns
            avg_s = (ms / cnt) * 1e-3 if cnt else float("nan")
            ach = bytes_per_launch / avg_s / 1e9 if cnt else None
            roof = {"bound": "hbm", "kernel": "gj_sweep_step_kernel", "achieved": ach, "peak": PEAK_HBM_GBPS,
                    "unit": "GB/s", "frac": (ach / PEAK_HBM_GBPS) if ach else None, "traffic": None,
                    "avg_launch_us": avg_s * 1e6, "launches_per_step": cnt / args.steps,
                    "algorithmic_bytes_per_launch": bytes_per_launch,
                    "share_of_step_time": (ms / args.steps) / (1e3 * instrumented / args.steps)}

    total_matrices = world * batch * args.steps
    value = total_matrices / elapsed
    line = {
        "metric": "matrices/sec, N=%d fp32 inversion (Gauss-Jordan, partial pivoting)" % n,
        "value": value,
        "unit": "matrices/s",
        "gflops": value * 2.0 * n ** 3 / 1e9,
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": (value / PUBLISHED_MATRICES_PER_S_N4096) if n == 4096 else None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": ("C1 (BASELINE configs[1]): single 4096x4096 fp32 inversion per GPU per step"
                         if (n == 4096 and batch == 1) else
                         f"batch of {batch} independent {n}x{n} fp32 matrices per GPU per step"),
            "n": n, "batch_per_gpu": batch, "algo": algo_name, "blocking": list(inv.resolved_blocking(n, batch)),
            "panel_width_per_block": (inv.resolved_panel_widths(n, batch) if algo_id == g.ALGO_BLOCKED else None),
            "distribution": "D_gate (row-permuted U(-1,1)+sqrt(N) I)",
            "parallelism": "independent matrices sharded over ranks, no data-path collective",
            "vs_baseline_denominator": "0.342 matrices/s: reference kernel loop, N=4096, RX 5700 (BASELINE.md)",
        },
        "residual_inf": res_right,
        "residual_inf_left": res_left,
        "status_max": st_max,
        "roofline": roof,
        "kernel_breakdown": breakdown,
    }
    if rank == 0 and not args.no_cpu_baseline:
        cb = cpu_baseline(host[0])
        line["cpu_baseline"] = cb
        line["cpu_baseline_oracle_port"] = oracle_port_baseline()
        line["speedup_vs_numpy_per_gpu"] = (value / world) / cb["value"] if cb.get("value") else None
    inv.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(line))


if __name__ == "__main__":
    main()
