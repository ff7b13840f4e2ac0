"""CPU tests: the C-ABI library loads and exports every declared symbol, the host logic
(guards, sharding, NumPy-shaped harness) behaves like the reference's, and the product path
fails loudly -- never falls back -- when no GPU is present.  No compute calls on a GPU here."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

import gpu_matrix_inversion_amd as g
from gpu_matrix_inversion_amd import _lib


def _have_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def test_library_is_built_in_tree():
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    assert os.path.commonpath([ROOT, _lib.LIB_PATH]) == ROOT


def test_every_declared_symbol_is_exported():
    """Every function include/mat_inv_32_c.h declares is exported, plus the C++ drop-in."""
    hdr = open(os.path.join(ROOT, "include", "mat_inv_32_c.h")).read()
    declared = set(re.findall(r"\b(mi32_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.C_ABI_SYMBOLS), declared ^ set(_lib.C_ABI_SYMBOLS)
    lib = _lib.load()
    for sym in _lib.C_ABI_SYMBOLS:
        assert getattr(lib, sym) is not None
    # std::vector<float> matrix_inv_32(std::vector<float>, int) -- include/mat_inv_32.h
    assert getattr(lib, _lib.CXX_DROPIN_SYMBOL) is not None
    out = subprocess.run(["c++filt", _lib.CXX_DROPIN_SYMBOL], capture_output=True, text=True).stdout.strip()
    assert out == "matrix_inv_32(std::vector<float, std::allocator<float> >, int)"
    # std::vector<double> matrix_inversion_FP64(std::vector<double>, int) -- include/mat_inv_64.h (headers.h:9)
    assert getattr(lib, _lib.CXX_FP64_SYMBOL) is not None
    out = subprocess.run(["c++filt", _lib.CXX_FP64_SYMBOL], capture_output=True, text=True).stdout.strip()
    assert out == "matrix_inversion_FP64(std::vector<double, std::allocator<double> >, int)"
    # std::vector<double> matrix_inversion_no_pivots(std::vector<double>, int) -- include/mat_inv_64.h (headers.h:11)
    assert getattr(lib, _lib.CXX_NOPIVOT_SYMBOL) is not None
    out = subprocess.run(["c++filt", _lib.CXX_NOPIVOT_SYMBOL], capture_output=True, text=True).stdout.strip()
    assert out == "matrix_inversion_no_pivots(std::vector<double, std::allocator<double> >, int)"
    # Res FP32_bench(std::vector<float>, int) -- include/mat_inv_bench.h (headers.h:15, res_struct.h:4-6)
    assert getattr(lib, _lib.CXX_BENCH_SYMBOL) is not None
    out = subprocess.run(["c++filt", _lib.CXX_BENCH_SYMBOL], capture_output=True, text=True).stdout.strip()
    assert out == "FP32_bench(std::vector<float, std::allocator<float> >, int)"
    hb = open(os.path.join(ROOT, "include", "mat_inv_bench.h")).read()
    assert "Res FP32_bench(std::vector<float> matrix_vector, int matrix_order);" in hb
    # ... and the rest of the experiment project's header, declaration for declaration (headers.h:5-16)
    for decl, mangled in (
            ("double matrix_multiply(std::vector<double> matriceA, std::vector<double> matriceB);",
             "_Z15matrix_multiplySt6vectorIdSaIdEES1_"),
            ("std::vector<float> matrix_inversion_FP32(std::vector<float> matrix_vector, int matrix_order);",
             "_Z21matrix_inversion_FP32St6vectorIfSaIfEEi"),
            ("Res no_pivots_bench(std::vector<double> matrix_vector, int matrix_order);", "_Z15no_pivots_benchSt6vectorIdSaIdEEi"),
            ("Res FP64_bench(std::vector<double> matrix_vector, int matrix_order);", "_Z10FP64_benchSt6vectorIdSaIdEEi")):
        assert decl in hb, decl
        assert getattr(lib, mangled) is not None, mangled
    assert re.search(r"struct Res \{\s*std::vector<double> inversa64;\s*std::vector<double> times;\s*"
                     r"std::vector<float> inversa32;\s*\};", hb)


def test_dropin_header_matches_reference_declaration():
    """include/mat_inv_32.h keeps the reference's declaration verbatim (Matlab/mat_inv_32.h:4)."""
    hdr = open(os.path.join(ROOT, "include", "mat_inv_32.h")).read()
    assert "std::vector<float> matrix_inv_32(std::vector<float> matrix_vector, int matrix_order);" in hdr
    assert 'extern "C"' not in hdr


def test_code_object_is_gfx950_only(tmp_path):
    # llvm-objdump --offloading extracts the code objects NEXT TO its input: run it on a copy in a scratch
    # directory, not inside gpu_matrix_inversion_amd/lib/ (which travels to the GPU box)
    import shutil

    copy = tmp_path / os.path.basename(_lib.LIB_PATH)
    shutil.copy(_lib.LIB_PATH, copy)
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "--offloading", str(copy)],
                         capture_output=True, text=True, cwd=tmp_path).stdout
    archs = set(re.findall(r"gfx[0-9a-f]+", out))
    assert archs == {"gfx950"}, archs
    libdir = os.path.dirname(_lib.LIB_PATH)
    assert not [f for f in os.listdir(libdir) if ".hipv4-" in f or ".host-" in f], "extracted code objects left in lib/"


def test_shape_guards_need_no_gpu():
    """mat_inv_32.cpp:206-215 guards are answered before anything touches the device."""
    lib = _lib.load()
    fp = ctypes.POINTER(ctypes.c_float)
    buf = (ctypes.c_float * 8)()
    assert lib.mi32_matrix_inv_32(ctypes.cast(buf, fp), 4, 0, ctypes.cast(buf, fp)) == _lib.MI32_BAD_SHAPE
    assert lib.mi32_matrix_inv_32(ctypes.cast(buf, fp), 4, -1, ctypes.cast(buf, fp)) == _lib.MI32_BAD_SHAPE
    assert lib.mi32_matrix_inv_32(ctypes.cast(buf, fp), 3, 2, ctypes.cast(buf, fp)) == _lib.MI32_BAD_SHAPE
    assert lib.mi32_matrix_inv_32(ctypes.cast(buf, fp), 6, 2, ctypes.cast(buf, fp)) == _lib.MI32_BAD_SHAPE
    assert g.matrix_inv_32(np.ones(4), 0).size == 0
    assert g.matrix_inv_32(np.ones(3), 2).size == 0
    assert g.matrix_inv_32(np.ones(6), 2).size == 0
    assert g.matrix_inv_32(np.zeros(0), 3).size == 0
    assert g.matrix_inv_64(np.ones(6), 2).size == 0 and g.matrix_inv_64(np.ones(4), -1).size == 0


def test_c_abi_shard_range_equals_the_python_sharding():
    """mi32_matrix_inv_32_batched_multi partitions a host batch with mi32_shard_range: the same contiguous ceil-sized
    ranges as sharding.shard_range, ragged and empty shards included; every matrix in exactly one shard."""
    from gpu_matrix_inversion_amd.sharding import shard_range

    lib = _lib.load()
    lo, hi = ctypes.c_int(-1), ctypes.c_int(-1)
    for batch in (0, 1, 7, 8, 64, 512, 513):
        for ngpus in (1, 2, 3, 4, 8, 9):
            covered = []
            for g_ in range(ngpus):
                assert lib.mi32_shard_range(batch, ngpus, g_, ctypes.byref(lo), ctypes.byref(hi)) == _lib.MI32_OK
                assert (lo.value, hi.value) == shard_range(batch, ngpus, g_), (batch, ngpus, g_)
                covered += list(range(lo.value, hi.value))
            assert covered == list(range(batch))
    assert lib.mi32_shard_range(8, 0, 0, ctypes.byref(lo), ctypes.byref(hi)) == _lib.MI32_BAD_SHAPE
    assert lib.mi32_shard_range(8, 2, 2, ctypes.byref(lo), ctypes.byref(hi)) == _lib.MI32_BAD_SHAPE
    # the multi-GPU host batch answers its shape guards before it touches a device
    fp = ctypes.POINTER(ctypes.c_float)
    buf = (ctypes.c_float * 4)()
    assert lib.mi32_matrix_inv_32_batched_multi(ctypes.cast(buf, fp), 0, 1, ctypes.cast(buf, fp), None, 1) == _lib.MI32_BAD_SHAPE
    assert lib.mi32_matrix_inv_32_batched_multi(ctypes.cast(buf, fp), 2, 0, ctypes.cast(buf, fp), None, 1) == _lib.MI32_BAD_SHAPE
    assert lib.mi32_matrix_inv_32_batched_multi(None, 2, 1, ctypes.cast(buf, fp), None, 1) == _lib.MI32_BAD_SHAPE


def test_workspace_sizes():
    lib = _lib.load()
    n = 4096
    sweep = lib.mi32_workspace_bytes(n, 1, _lib.ALGO_SWEEP)
    blocked = lib.mi32_workspace_bytes(n, 1, _lib.ALGO_BLOCKED)
    # two working copies of the N x N matrix (the reference holds two N x 2N panels + N x N)
    assert 2 * n * n * 4 <= sweep < 2 * n * n * 4 + (1 << 20)
    # rows padded by 256 B, + five compact panels (32 x N), two multiplier panels (32 x 2N) and ten maps, + per block
    # (256 pivots) eight 256 x N arrays: the transposed multipliers, 2 x (u rows, pivot rows after their sub-panel,
    # multiplier matrix) double-buffered for the look-ahead, the parked rows of a split strip launch
    assert 2 * n * n * 4 <= blocked < 2 * n * (n + 64) * 4 + 8 * 256 * n * 4 + (5 << 20)
    assert lib.mi32_workspace_bytes(0, 1, 0) == 0
    # padding to a multiple of 128 in the blocked path
    assert lib.mi32_workspace_bytes(1000, 1, _lib.ALGO_BLOCKED) >= 2 * 1024 * 1024 * 4
    assert lib.mi32_dominant_kernel(_lib.ALGO_SWEEP) == b"gj_sweep_step_kernel"
    assert lib.mi32_dominant_kernel(_lib.ALGO_BLOCKED) == b"gj_rank_bw2_kernel"


@pytest.mark.skipif(_have_gpu(), reason="checks the no-GPU failure mode")
def test_no_silent_cpu_fallback():
    """Without a GPU the product path raises; it never computes on the CPU."""
    with pytest.raises(g.Mi32Error):
        g.matrix_inv_32(np.eye(2, dtype=np.float32).reshape(-1), 2)
    with pytest.raises(g.Mi32Error):
        g.Inverter()
    h = ctypes.c_void_p()
    assert _lib.load().mi32_create(ctypes.byref(h), 0) == _lib.MI32_RUNTIME_ERROR
    assert b"device" in _lib.load().mi32_last_error().lower()


def test_product_package_never_imports_the_oracle():
    src_dir = os.path.join(ROOT, "gpu_matrix_inversion_amd")
    for dirpath, _, files in os.walk(src_dir):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, f
                assert "gj_oracle" not in text.replace("oracle/gj_oracle.c", ""), f


def test_shard_range_partitions_the_batch():
    for batch in (0, 1, 7, 64, 511, 512):
        for world in (1, 2, 3, 4, 8):
            covered = []
            for r in range(world):
                lo, hi = g.shard_range(batch, world, r)
                assert 0 <= lo <= hi <= batch
                covered += list(range(lo, hi))
            assert covered == list(range(batch))
    assert g.shard_range(512, 8, 3) == (192, 256)  # C3: 64 matrices per GPU
    with pytest.raises(ValueError):
        g.shard_range(4, 2, 2)


def test_just_inv_call_shape(capsys):
    """just_inv mirrors matrix_inv_numpy.py:39-46: prints 'TIME: <s>' and times only the inverse
    (here with numpy.linalg.inv injected: C0 = BASELINE configs[0], 256x256 on the CPU)."""
    dt, a, res = g.just_inv(256, seed=0, inv=np.linalg.inv)
    out = capsys.readouterr().out
    assert re.match(r"^TIME: [0-9.e-]+\n$", out)
    assert a.shape == (256, 256) and a.dtype == np.float32 and a.min() >= 0 and a.max() <= 100
    gold = np.load(os.path.join(ROOT, "tests", "golden", "c0_u100_N256.npz"))
    assert np.array_equal(a, gold["a"])  # same seed-0 default_rng stream as the committed fixture
    assert np.abs(res - gold["inv64"]).max() / np.abs(gold["inv64"]).max() < 1e-4


_WORKER = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
import gpu_matrix_inversion_amd as g
import oracle as O

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
rng = np.random.default_rng(5)
B, n = 5, 24   # ragged: 3 + 2
a = np.stack([(rng.uniform(-1, 1, (n, n)) + np.sqrt(n) * np.eye(n))[rng.permutation(n)] for _ in range(B)]).astype(np.float32)
calls = []
def oracle_fn(shard):   # the test injects the checker as the per-shard function; the product never does
    calls.append(shard.shape[0])
    inv = np.stack([O.matrix_inv_32(m.numpy(), n).reshape(n, n) for m in shard])
    return torch.from_numpy(inv), torch.zeros(shard.shape[0], dtype=torch.int32)
inv, st, (lo, hi) = g.invert_sharded(torch.from_numpy(a), oracle_fn)
assert (lo, hi) == g.shard_range(B, world, rank)
assert calls == [hi - lo], calls                      # each rank inverts only its own shard
assert inv.shape == (B, n, n) and st.shape == (B,)
for b in range(B):
    assert O.residual_inf(a[b], inv[b].numpy(), n) < 1e-4
full = np.stack([O.matrix_inv_32(a[b], n).reshape(n, n) for b in range(B)])
assert np.array_equal(inv.numpy(), full)               # gathered result == unsharded result, bit for bit
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_sharded_batch_gloo_world2(tmp_path):
    """N>1 path on CPU: two gloo ranks, contiguous shards, all-gather of results and status."""
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert f"rank {r} ok" in o


def test_makefile_prerequisites_cover_every_included_header():
    """lib/libmat_inv_32.so is git-ignored yet shipped to the GPU box: a header-only edit that does not
    rebuild it would let the parity tests validate a stale binary.  Every `#include "..."` of the HIP
    sources (and of the headers they include) must be a prerequisite of the library target."""
    csrc = _lib.CSRC_DIR
    out = subprocess.run(["make", "-C", csrc, "-pn", "all"], capture_output=True, text=True).stdout
    m = re.search(r"^\.\./lib/libmat_inv_32\.so:(.*)$", out, re.M)
    assert m, "library rule not found in `make -pn`"
    prereq = {os.path.normpath(os.path.join(csrc, p)) for p in m.group(1).split()}
    seen, todo = set(), [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".hip")]
    while todo:
        f = todo.pop()
        if f in seen:
            continue
        seen.add(f)
        for inc in re.findall(r'^\s*#\s*include\s+"([^"]+)"', open(f).read(), re.M):
            for base in (os.path.dirname(f), os.path.join(ROOT, "include")):
                cand = os.path.normpath(os.path.join(base, inc))
                if os.path.exists(cand):
                    todo.append(cand)
                    break
            else:
                raise AssertionError(f"{f} includes {inc}: not found")
    missing = {os.path.normpath(f) for f in seen} - prereq
    assert not missing, f"not prerequisites of the library: {sorted(missing)}"
    # and the oracle: `make` decides staleness (oracle.build() always calls it)
    o = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-pn", "all"], capture_output=True, text=True).stdout
    assert re.search(r"^libgj_oracle\.so:.*gj_oracle\.c.*gj_oracle\.h", o, re.M)


def test_status_codes_and_boundary_rule_are_documented():
    """include/mat_inv_32_c.h states the invalid-matrix rule the tests hold both paths to."""
    h = open(os.path.join(ROOT, "include", "mat_inv_32_c.h")).read()
    assert "non-finite entry" in h and "NaN-filled" in h


_DIST_WORKER = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
import gpu_matrix_inversion_amd as g
import oracle as O

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
rng = np.random.default_rng(5)
B, n = 7, 24   # ragged over 3 ranks: 3 + 3 + 1; over 2: 4 + 3
a = np.stack([(rng.uniform(-1, 1, (n, n)) + np.sqrt(n) * np.eye(n))[rng.permutation(n)] for _ in range(B)]).astype(np.float32)
a[5] = 1.0      # one singular member: its status must reach root, the worst status every rank
calls = []
def oracle_fn(shard):   # the test injects the checker as the per-shard function; the product never does
    calls.append(shard.shape[0])
    res = [O.matrix_inv_32(m.numpy(), n, return_info=True) for m in shard]
    return (torch.from_numpy(np.stack([r[0].reshape(n, n) for r in res])),
            torch.tensor([r[1]["status"] for r in res], dtype=torch.int32))
bufs = {{}}
for rep in range(2):    # second call re-uses the buffers
    inv, st, worst, tm = g.invert_distributed(torch.from_numpy(a) if rank == 0 else None, oracle_fn, root=0,
                                              shard_buffers=bufs)
    lo, hi = g.shard_range(B, world, rank)
    assert calls[-1] == hi - lo                      # each rank inverts only its own shard
    assert worst == 2                                 # all_reduce(MAX) of the status words
    assert set(tm) == {{"scatter", "compute", "gather"}}
    if rank == 0:
        assert inv.shape == (B, n, n) and st.tolist() == [0, 0, 0, 0, 0, 2, 0]
        full = np.stack([O.matrix_inv_32(a[b], n).reshape(n, n) for b in range(B)])
        ok = [b for b in range(B) if b != 5]
        assert np.array_equal(inv.numpy()[ok], full[ok])   # distributed result == unsharded result, bit for bit
    else:
        assert inv.shape == (hi - lo, n, n) and st.shape == (hi - lo,)
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
"""


@pytest.mark.parametrize("world", [2, 3])
def test_distributed_batch_from_root_gloo(tmp_path, world):
    """The xGMI distribution path on CPU ranks (gloo): the batch lives on rank 0 only, shards travel by grouped
    point-to-point sends, results and status words come back the same way, the worst status is all-reduced."""
    script = tmp_path / "dist_worker.py"
    script.write_text(_DIST_WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29540 + world), WORLD_SIZE=str(world))
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert f"rank {r} ok" in o
