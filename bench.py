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
  roofline      -- the kernel that carries the algorithmic flops (or bytes): per-launch
                   durations from HIP events recorded on the launch stream (inside the library,
                   mi32_set_profiling) over a second, instrumented pass of the same K steps, so the
                   un-instrumented pass that yields `value` is not perturbed.  (That pass runs with
                   the look-ahead split off: every rank-bw update is one full-size launch of the
                   kernel the roofline is quoted for.)  `traffic` comes from the rocprofv3 --pmc
                   passes committed under profiles/ -- only from an entry whose (algo, n, batch,
                   block width) matches this run.  `time_dominant` names the kernel class that takes
                   most of the step when it is not the roofline kernel.
  e2e           -- T_e2e: the same matrix through the host-pointer drop-in matrix_inv_32(vec, N)
                   (pageable host memory both ways), the figure "x NumPy wall-clock" is defined on;
                   the reference prints both timings too (mat_inv_32.cpp:385-386).
  cpu_baseline  -- numpy.linalg.inv (the reference's CPU path, matrix_inv_numpy.py:44, through our
                   just_inv-shaped harness) on the same fp32 input on this box's host cores: the
                   best of a few BLAS thread counts (stated), plus the 1-thread number.
  residuals     -- D_gate (gated at 1e-3) and, ungated, the reference's own input distributions
                   D_ref100 = U(0,100) and D_rand = U(0,1) next to NumPy's residual on the same input.
  distribution  -- N > 1 (skip with --no-distribute): the batch starts on rank 0 only and is scattered /
                   gathered over RCCL point-to-point sends; matrices/s including the transfers.
"""
import argparse
import contextlib
import io
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
PMC_TRAFFIC_FILE = os.path.join(ROOT, "profiles", "round3", "pmc_traffic.json")


def gate_matrix(n, seed):
    """D_gate: row-permuted U(-1,1) + sqrt(N) I -- well conditioned, ~N row swaps."""
    rng = np.random.default_rng(seed)
    a = rng.uniform(-1.0, 1.0, (n, n)) + np.sqrt(n) * np.eye(n)
    return a[rng.permutation(n)].astype(np.float32)


def _blas_info():
    try:
        from threadpoolctl import threadpool_info

        info = [i for i in threadpool_info() if i.get("user_api") == "blas"]
        if info:
            return int(info[0].get("num_threads", 0)), f"{info[0].get('internal_api')} {info[0].get('version')}"
    except Exception:
        pass
    return os.cpu_count() or 1, "unknown"


def _time_numpy_inv(a, reps):
    """Median seconds of numpy.linalg.inv(a) through the reference script's call shape (just_inv)."""
    import gpu_matrix_inversion_amd as g

    n = a.shape[0]
    times = []
    for _ in range(reps):
        with contextlib.redirect_stdout(io.StringIO()):
            dt, _, _ = g.just_inv(n, inv=lambda m: np.linalg.inv(a))
        times.append(dt)
    return statistics.median(times)


def cpu_baseline(a, budget_s=40.0):
    """numpy.linalg.inv on the same fp32 matrix: the best of a few BLAS thread counts and the 1-thread number
    (a 64-thread OpenBLAS on a 256-CPU host is not automatically its fastest setting); bounded in time."""
    n = a.shape[0]
    max_threads, blas = _blas_info()
    t_start = time.perf_counter()
    np.linalg.inv(a)  # warm-up: BLAS thread-pool start-up
    first = time.perf_counter() - t_start
    sweep = {}
    try:
        from threadpoolctl import threadpool_limits
    except Exception:
        threadpool_limits = None
    counts = [max_threads]
    if threadpool_limits is not None:
        counts = sorted({max_threads, max(1, max_threads // 2), min(max_threads, 16), min(max_threads, 8), 1}, reverse=True)
    for c in counts:
        if time.perf_counter() - t_start > budget_s and sweep:
            break
        ctx = threadpool_limits(limits=c, user_api="blas") if threadpool_limits is not None else contextlib.nullcontext()
        with ctx:
            np.linalg.inv(a)  # settle the pool at this size
            reps = 3 if c > 1 else 1
            sweep[c] = _time_numpy_inv(a, reps)
    best_c = min((c for c in sweep if c > 1), key=lambda c: sweep[c], default=max(sweep))
    med = sweep[best_c]
    return {
        "value": 1.0 / med,
        "unit": "matrices/s",
        "cores": best_c,
        "kind": "reference",
        "what": "numpy.linalg.inv on the same fp32 input (the reference CPU path's call, "
                "matrix_inv_numpy.py:44, via the just_inv-shaped harness); best of the BLAS thread counts tried",
        "sample": f"one {n}x{n} fp32 inversion, median of 3 timed calls per thread count after a warm-up "
                  f"(first call of the process: {first:.2f} s)",
        "seconds_per_matrix": med,
        "gflops_2n3": 2.0 * n ** 3 / med / 1e9,
        "seconds_by_blas_threads": {str(c): sweep[c] for c in sorted(sweep)},
        "one_thread_seconds": sweep.get(1),
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


def reference_distribution_residuals(inv, n, torch, dev, budget_s=25.0):
    """||A X - I||_inf on the reference's own input distributions (ungated at 1e-3: fp32 Gauss-Jordan cannot reach it
    there; gated in the tests relative to the reference-order elimination), ours next to the CPU oracle's
    (the reference's operation order, step by step) and NumPy's on the same matrix.  matrix_inv_pyopencl.py:17,
    matrix_inv_numpy.py:40 (U(0,100)) and the MATLAB live script's rand(N,N) (U(0,1))."""
    out = {}
    t_start = time.perf_counter()
    try:
        import oracle as O
    except Exception:  # the oracle is optional for the bench
        O = None
    for name, hi in (("D_ref100", 100.0), ("D_rand", 1.0)):
        a = np.random.default_rng(4242).uniform(0.0, hi, (n, n)).astype(np.float32)
        ta = torch.from_numpy(a).to(dev)
        x, st = inv.inv(ta)
        r = inv.residual(ta, x)
        torch.cuda.synchronize()
        entry = {"ours_residual_inf": float(r[0, 0]), "ours_residual_inf_left": float(r[0, 1]),
                 "ours_frobenius_metric": float(r[0, 2]), "status": int(st[0])}
        if O is not None and n <= 4096:
            # the reference-order elimination (gjo_matrix_inv_32_inplace's bits through its cache-blocked evaluation)
            xo = O.matrix_inv_32_blocked_exact(a, n, 128)
            ro = inv.residual(ta, torch.from_numpy(xo.reshape(n, n)).to(dev))
            torch.cuda.synchronize()
            entry["reference_order_gj_residual_inf"] = float(ro[0, 0])
            entry["reference_order_gj_residual_inf_left"] = float(ro[0, 1])
            entry["ours_equals_reference_order_gj_bit_for_bit"] = bool(np.array_equal(x.cpu().numpy().reshape(-1), xo))
        if time.perf_counter() - t_start < budget_s:
            xn = np.linalg.inv(a)
            rn = inv.residual(ta, torch.from_numpy(np.ascontiguousarray(xn, dtype=np.float32)).to(dev))
            torch.cuda.synchronize()
            entry["numpy_fp32_residual_inf"] = float(rn[0, 0])
            entry["numpy_fp32_residual_inf_left"] = float(rn[0, 1])
        out[name] = entry
    return out


def pmc_traffic(algo_name, n, batch, bw):
    """HBM bytes per launch of the roofline kernel from the committed PMC passes, only for the exact config."""
    try:
        table = json.load(open(PMC_TRAFFIC_FILE))
    except Exception:
        return None, None
    key = f"{algo_name}_n{n}_b{batch}" + (f"_bw{bw}" if algo_name == "blocked" else "")
    e = table.get(key)
    if not e:
        return None, None
    return e.get("hbm_bytes_per_launch_corrected"), {"file": os.path.relpath(PMC_TRAFFIC_FILE, ROOT), "key": key,
                                                     "fetch_bytes_x2": e.get("fetch_bytes_x2"),
                                                     "write_bytes": e.get("write_bytes")}


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
    ap.add_argument("--no-resident-batch", action="store_true",
                    help="skip the extra key: throughput of a resident batch of 4 and 8 matrices of this order")
    ap.add_argument("--no-e2e", action="store_true",
                    help="skip the host-pointer leg (matrix_inv_32(vec, N)): profiler passes only want the timed loop")
    ap.add_argument("--distribute", dest="distribute", action="store_true", default=True,
                    help="N > 1 (default on): also time the batch scattered from / gathered to rank 0 over RCCL (xGMI)")
    ap.add_argument("--no-distribute", dest="distribute", action="store_false")
    args = ap.parse_args()

    if args.algo != "auto":  # the host-pointer entry point (e2e leg) runs on the default context: same algorithm
        os.environ["MI32_ALGO"] = {"sweep": "1", "blocked": "2"}[args.algo]

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
    bw = inv.resolved_blocking(n, batch)[1] if algo_id == g.ALGO_BLOCKED else 0

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
    res_frob = float(res[:, 2].abs().max())
    st_max = int(status.max())

    # What the GPU does with this order when it is not waiting on one matrix's chain of pivot steps: the headline
    # configuration keeps 1 CU of 256 busy for most of its time (roofline.time_dominant), a resident batch of the same
    # matrices fills the rest.  Extra key, not the headline (BASELINE.json's metric is matrices/s).
    resident = None
    if rank == 0 and batch == 1 and n <= 4096 and not args.no_resident_batch:
        resident = {}
        for nb in (4, 8):
            ab = a.expand(nb, n, n).contiguous()
            ob = torch.empty_like(ab)
            sb = torch.empty(nb, dtype=torch.int32, device=dev)
            inv.inv(ab, out=ob, status=sb)  # warm-up (workspace growth)
            torch.cuda.synchronize()
            tb0 = time.perf_counter()
            for _ in range(3):
                inv.inv(ab, out=ob, status=sb)
            torch.cuda.synchronize()
            dtb = (time.perf_counter() - tb0) / 3
            resident[f"batch_{nb}"] = {"ms_per_batch": 1e3 * dtb, "matrices_per_s": nb / dtb,
                                       "tflops_2n3": nb * 2.0 * n ** 3 / dtb / 1e12,
                                       "identical_to_single": bool(torch.equal(ob[nb - 1], out[0])), "status_max": int(sb.max())}
            del ab, ob, sb

    # with distribution: the whole batch starts on rank 0, travels over RCCL point-to-point sends
    distribution = None
    if args.distribute and world > 1:
        try:
            full = None
            if rank == 0:
                full = torch.cat([a] + [torch.from_numpy(np.stack([gate_matrix(n, 1000 + r * batch + b) for b in range(batch)])).to(dev)
                                        for r in range(1, world)])
            bufs = {}
            tm_sum = {"scatter": 0.0, "compute": 0.0, "gather": 0.0}
            for i in range(args.warmup + args.steps):
                if i == args.warmup:
                    sync_all()
                    td0 = time.perf_counter()
                _, _, worst, tm = g.invert_distributed(full, lambda s: inv.inv(s), root=0, shard_buffers=bufs)
                if i >= args.warmup:
                    for k in tm_sum:
                        tm_sum[k] += tm[k]
            torch.cuda.synchronize()
            d_elapsed = time.perf_counter() - td0
            dist.barrier()
            t = torch.tensor([d_elapsed] + [tm_sum[k] for k in ("scatter", "compute", "gather")], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            d_elapsed = float(t[0])
            distribution = {
                "with_distribution_matrices_per_s": world * batch * args.steps / d_elapsed,
                "ms_per_step": 1e3 * d_elapsed / args.steps,
                "scatter_ms_per_step": 1e3 * float(t[1]) / args.steps,
                "compute_ms_per_step": 1e3 * float(t[2]) / args.steps,
                "gather_ms_per_step": 1e3 * float(t[3]) / args.steps,
                "worst_status": worst,
                "bytes_scattered_per_step": (world - 1) * batch * n * n * 4,
                "transport": "torch.distributed batch_isend_irecv on nccl (= grouped ncclSend/ncclRecv over xGMI), root = rank 0",
            }
            del full
        except Exception as exc:  # the compute-only line above stays valid without this leg
            distribution = {"error": f"{type(exc).__name__}: {exc}"[:300]}

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
        tot_ms = sum(v[0] for v in prof.values()) or 1.0
        time_dominant = max(prof.items(), key=lambda kv: kv[1][0])
        dominant = {"class": time_dominant[0], "share_of_kernel_time": time_dominant[1][0] / tot_ms,
                    "ms_per_step": time_dominant[1][0] / args.steps,
                    "bound": ("latency chain: one workgroup per matrix, one barrier per pivot step -- no roofline applies"
                              if time_dominant[0] == "panel" else "see roofline")}
        traffic, traffic_src = pmc_traffic(algo_name, n, batch, bw)
        if algo_id == g.ALGO_BLOCKED:
            ms, cnt = prof["update_rank_bw"]
            # ALGORITHMIC flops of one rank-bw update launch: 2 * N * (N - bw) * bw per matrix
            # (sum over the N/bw launches = 2 N^3 (1 - bw/N): the block's own columns are done in-panel)
            flops = 2.0 * n * max(n - bw, 0) * bw * batch
            avg_s = (ms / cnt) * 1e-3 if cnt else float("nan")
            ach = flops / avg_s / 1e12 if cnt else None
            roof = {"bound": "mfma", "kernel": "gj_rank_bw2_kernel", "achieved": ach,
                    "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": (ach / PEAK_FP32_MFMA_TFLOPS) if ach else None,
                    "traffic": traffic, "traffic_source": traffic_src,
                    "avg_launch_us": avg_s * 1e6, "launches_per_step": cnt / args.steps,
                    "algorithmic_flops_per_launch": flops,
                    "algorithmic_bytes_per_launch": 8.0 * n * max(n - bw, 0) * batch + 8.0 * n * bw * batch,
                    "share_of_step_time": (ms / args.steps) / (1e3 * instrumented / args.steps),
                    "time_dominant": dominant}
        else:
            ms, cnt = prof["sweep_step"]
            bytes_per_launch = 8.0 * n * (n + 1) * batch  # one fp32 read + write of N rows x (N+1) live columns
            avg_s = (ms / cnt) * 1e-3 if cnt else float("nan")
            ach = bytes_per_launch / avg_s / 1e9 if cnt else None
            roof = {"bound": "hbm", "kernel": "gj_sweep_step_kernel", "achieved": ach, "peak": PEAK_HBM_GBPS,
                    "unit": "GB/s", "frac": (ach / PEAK_HBM_GBPS) if ach else None,
                    "traffic": traffic, "traffic_source": traffic_src,
                    "avg_launch_us": avg_s * 1e6, "launches_per_step": cnt / args.steps,
                    "algorithmic_bytes_per_launch": bytes_per_launch,
                    "share_of_step_time": (ms / args.steps) / (1e3 * instrumented / args.steps),
                    "time_dominant": dominant,
                    "note": ("two working copies of %.0f MiB: %s the 256 MiB Infinity Cache"
                             % (n * n * 4 / 2 ** 20, "inside" if 2 * n * n * 4 <= 256 * 2 ** 20 else "beyond"))}

    # T_e2e through the host-pointer drop-in (rank 0 only; pageable host memory both ways)
    e2e = None
    if rank == 0 and not args.no_e2e:
        reps = 5 if n <= 4096 else 2
        times, computes = [], []
        flat = host.reshape(batch, -1)
        for i in range(reps + 1):
            te = time.perf_counter()
            if batch == 1:
                x = g.matrix_inv_32(flat[0], n)
            else:
                x, _ = g.matrix_inv_32_batched(host)
            dt = time.perf_counter() - te
            if i > 0:  # the first call creates the default context and its staging buffers
                times.append(dt)
                computes.append(g.last_timing()[1])
        e2e = {"e2e_ms": 1e3 * statistics.median(times), "compute_ms_inside": 1e3 * statistics.median(computes),
               "entry_point": "matrix_inv_32(vec, N)" if batch == 1 else "mi32_matrix_inv_32_batched",
               "matrices_per_s_e2e": batch / statistics.median(times),
               "what": "host vector in, host vector out (H2D + compute + D2H + status), median of %d calls; the reference's "
                       "'Tempo Totale Impiegato' / 'Tempo Computazione' pair (mat_inv_32.cpp:385-386)" % reps}
        del x

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
        "frobenius_metric_abs": res_frob,
        "status_max": st_max,
        "roofline": roof,
        "kernel_breakdown": breakdown,
        "e2e": e2e,
    }
    if distribution is not None:
        line["distribution"] = distribution
    if resident is not None:
        line["resident_batch_throughput"] = resident
    if rank == 0 and not args.no_cpu_baseline:
        cb = cpu_baseline(host[0])
        line["cpu_baseline"] = cb
        line["cpu_baseline_oracle_port"] = oracle_port_baseline()
        line["speedup_vs_numpy_per_gpu"] = (value / world) / cb["value"] if cb.get("value") else None
        if e2e:
            line["speedup_vs_numpy_e2e"] = e2e["matrices_per_s_e2e"] / cb["value"]  # NumPy: one matrix per call
        if n <= 8192:
            line["residuals_reference_distributions"] = reference_distribution_residuals(inv, n, torch, dev)
    inv.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(line))


if __name__ == "__main__":
    main()
