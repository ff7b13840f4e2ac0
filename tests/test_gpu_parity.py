"""GPU parity tests (run with ``-m gpu`` on an MI355X).  Everything goes through the C ABI of
libmat_inv_32.so; the oracle is only the checker.

Tolerances (fp32, stated once):
  * sweep path:   bit-identical to the CPU oracle (same operation order, explicit fmaf, IEEE divide)
  * blocked path: bit-identical to the SAME oracle (round 3: the blocked path evaluates the reference's own operation
                  order -- one fmaf per element and pivot step from the old value, IEEE division of the pivot rows;
                  v_mfma_f32_32x32x2_f32 with the old value as C operand is that fmaf chain), whatever the blocking;
                  both paths against float64 inverses: max|X-X64|/max|X64| <= 2 * kappa_inf(A) * 2^-24
  * residual gate (BASELINE.json): ||A X - I||_inf < 1e-3 on D_gate at every size
"""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import (EPS32, ROOT, check_against_oracle_digest, forward_tolerance, gate_matrix, golden_files, load_golden,
                      load_oracle_digest, rel_err)

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import gpu_matrix_inversion_amd as g  # noqa: E402
from gpu_matrix_inversion_amd import _lib  # noqa: E402


@pytest.fixture(scope="module")
def inv_sweep():
    inv = g.Inverter(algo="sweep")
    yield inv
    inv.close()


@pytest.fixture(scope="module")
def inv_blocked():
    inv = g.Inverter(algo="blocked")
    yield inv
    inv.close()


def run(inv, a):
    ta = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
    x, st = inv.inv(ta)
    torch.cuda.synchronize()
    return x.cpu().numpy(), st.cpu().numpy()


def oracle_inverse(oracle, a, n, return_info=False):
    """The reference-order result (mat_inv_32.cpp:317-362): the step-by-step restatement itself up to N = 1024, above
    that its cache-blocked evaluation, which tests/test_oracle.py proves bit-identical to it for every block width."""
    if n <= 1024:
        return oracle.matrix_inv_32_inplace(a, n, return_info=return_info)
    return oracle.matrix_inv_32_blocked_exact(a, n, 128, return_info=return_info)


def dist_matrix(kind, n, seed):
    rng = np.random.default_rng(seed)
    if kind == "gate":
        return gate_matrix(n, seed)
    if kind == "ref100":  # matrix_inv_pyopencl.py:17, matrix_inv_numpy.py:40
        return rng.uniform(0, 100, (n, n)).astype(np.float32)
    if kind == "rand":    # MATLAB rand(N,N)
        return rng.uniform(0, 1, (n, n)).astype(np.float32)
    if kind == "hollow":  # matrix_inv_numpy.py:13-14, main_file.cpp:46-48
        a = rng.uniform(0, 100, (n, n))
        np.fill_diagonal(a, 0.0)
        return a.astype(np.float32)
    raise ValueError(kind)


SIZES = [1, 2, 3, 4, 5, 16, 63, 64, 65, 100, 127, 128, 129, 255, 256, 257, 300, 512, 1000]


@pytest.mark.parametrize("n", SIZES)
def test_sweep_bit_identical_to_oracle(oracle, inv_sweep, n):
    for kind in ("gate", "ref100", "hollow"):
        if kind == "hollow" and n == 1:
            continue
        a = dist_matrix(kind, n, 7000 + n)
        want, info = oracle.matrix_inv_32(a, n, return_info=True)
        got, st = run(inv_sweep, a)
        assert st[0] == info["status"] == 0
        assert np.array_equal(got.reshape(-1), want), (kind, n, np.abs(got.reshape(-1) - want).max())


@pytest.mark.parametrize("n", SIZES)
def test_blocked_bit_identical_to_oracle(oracle, inv_blocked, n):
    for kind in ("gate", "ref100", "rand", "hollow"):
        if kind == "hollow" and n == 1:
            continue
        a = dist_matrix(kind, n, 8000 + n)
        want, info = oracle.matrix_inv_32_inplace(a, n, return_info=True)
        got, st = run(inv_blocked, a)
        assert st[0] == info["status"] == 0
        assert np.array_equal(got.reshape(-1), want), (kind, n, np.abs(got.reshape(-1) - want).max())


@pytest.mark.parametrize("w,bw", [(32, 128), (16, 128), (8, 128), (4, 256), (8, 384), (16, 512), (32, 256)])
def test_blocked_other_blockings(oracle, w, bw):
    inv = g.Inverter(algo="blocked", panel_width=w, block_width=bw)
    try:
        for n in (200, 640):
            a = dist_matrix("gate", n, 8100 + n + w)
            assert inv.resolved_blocking(n, 1) == (w, bw if bw <= ((n + 127) & ~127) else ((n + 127) & ~127))
            want = oracle.matrix_inv_32_inplace(a, n)  # the blocking never changes a bit
            got, st = run(inv, a)
            assert st[0] == 0
            assert np.array_equal(got.reshape(-1), want), (w, bw, n)
    finally:
        inv.close()


@pytest.mark.parametrize("path", golden_files(), ids=lambda p: os.path.basename(p)[:-4])
def test_golden_vectors_both_paths(oracle, inv_sweep, inv_blocked, path):
    a, inv64 = load_golden(path)
    n = a.shape[0]
    tol = forward_tolerance(a)
    for inv in (inv_sweep, inv_blocked):
        got, st = run(inv, a)
        assert st[0] == 0
        assert rel_err(got, inv64) <= tol
        assert oracle.residual_inf(a, got, n) <= max(4 * n * tol, 1e-6)


def test_dropin_host_entry_point(oracle):
    """matrix_inv_32(vec, N): the reference's call shape through the host-pointer C ABI."""
    n = 96
    a = gate_matrix(n, 31)
    got = g.matrix_inv_32(a.reshape(-1), n)
    assert got.dtype == np.float32 and got.shape == (n * n,)
    assert oracle.residual_inf(a, got, n) < 1e-4
    total, compute = g.last_timing()
    assert total >= compute > 0
    # the reference's guards (mat_inv_32.cpp:206-215)
    assert g.matrix_inv_32(a.reshape(-1), 0).size == 0
    assert g.matrix_inv_32(a.reshape(-1)[:-1], n).size == 0
    # integer-division quirk: a tail of fewer than N extra floats is accepted and ignored
    tail = np.concatenate([a.reshape(-1), np.full(n - 1, 99.0, np.float32)])
    assert np.array_equal(g.matrix_inv_32(tail, n), got)
    # N = 1
    assert g.matrix_inv_32(np.array([4.0], np.float32), 1)[0] == 0.25
    # invalid (singular) matrix -> empty vector (README.md:54)
    sing = np.ones((8, 8), np.float32)
    assert g.matrix_inv_32(sing.reshape(-1), 8).size == 0
    os.environ["MI32_SINGULAR_KEEP"] = "1"
    try:
        assert g.matrix_inv_32(sing.reshape(-1), 8).size == 64
    finally:
        del os.environ["MI32_SINGULAR_KEEP"]


def test_cxx_dropin_links_and_runs(oracle, tmp_path):
    """A C++ caller compiled against include/mat_inv_32.h (the reference's header, unchanged)
    links to libmat_inv_32.so and gets the same answer as the C ABI."""
    src = tmp_path / "caller.cpp"
    src.write_text(r'''
#include <cstdio>
#include <vector>
#include "mat_inv_32.h"
#include "mat_inv_64.h"
int main() {
    const int n = 3;
    std::vector<float> a = {2, 1, 0,  1, 3, 1,  0, 1, 4};
    std::vector<float> x = matrix_inv_32(a, n);
    if (x.size() != 9) { std::printf("EMPTY\n"); return 1; }
    for (float v : x) std::printf("%.9g\n", v);
    std::vector<float> bad = matrix_inv_32(a, 2);   // 9/2 = 4 != 2
    std::vector<float> neg = matrix_inv_32(a, -1);
    std::vector<float> sing = matrix_inv_32(std::vector<float>(16, 1.0f), 4);
    std::printf("sizes %zu %zu %zu\n", bad.size(), neg.size(), sing.size());
    // the fp64 twin, by the reference's own name (matrix_inversion/headers.h:9)
    std::vector<double> ad(a.begin(), a.end());
    std::vector<double> xd = matrix_inversion_FP64(ad, n);
    if (xd.size() != 9) { std::printf("EMPTY64\n"); return 1; }
    for (double v : xd) std::printf("%.17g\n", v);
    std::printf("sizes64 %zu %zu\n", matrix_inversion_FP64(ad, 2).size(), matrix_inv_64(std::vector<double>(16, 1.0), 4).size());
    return 0;
}
''')
    exe = tmp_path / "caller"
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                           "-L", libdir, "-lmat_inv_32", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = out.stdout.strip().splitlines()
    a = np.array([2, 1, 0, 1, 3, 1, 0, 1, 4], np.float32)
    want = oracle.matrix_inv_32(a, 3)
    got = np.array([float(v) for v in lines[:9]], np.float32)
    assert np.allclose(got, want, rtol=1e-6, atol=1e-7)
    assert lines[9] == "sizes 0 0 0"
    got64 = np.array([float(v) for v in lines[10:19]], np.float64)
    assert np.array_equal(got64, oracle.matrix_inv_64(a.astype(np.float64), 3))
    assert lines[19] == "sizes64 0 0"


def test_batched_with_a_singular_member(oracle, inv_sweep, inv_blocked):
    n, B = 160, 6
    mats = np.stack([gate_matrix(n, 900 + b) for b in range(B)])
    mats[3] = 1.0  # rank-1: singular
    for inv, mirror in ((inv_sweep, lambda m: oracle.matrix_inv_32(m, n)),
                        (inv_blocked, lambda m: oracle.matrix_inv_32_inplace(m, n))):
        got, st = run(inv, mats)
        assert list(st) == [0, 0, 0, 2, 0, 0]
        for b in range(B):
            if b == 3:
                continue
            assert np.array_equal(got[b].reshape(-1), mirror(mats[b])), b
    out, st = g.matrix_inv_32_batched(mats)
    assert list(st) == [0, 0, 0, 2, 0, 0]
    assert oracle.residual_inf(mats[5], out[5], n) < 1e-4


@pytest.mark.parametrize("bad", [np.nan, np.inf, -np.inf], ids=["nan", "inf", "-inf"])
def test_nonfinite_input_is_status_singular_on_both_paths(oracle, inv_sweep, inv_blocked, bad):
    """The boundary rule (README.md:54 "in case of invalid matrix an empty vector is returned"): a non-finite
    entry anywhere in the input, like a zero / NaN / infinite pivot, is status 2 on BOTH paths -- exactly the
    oracle's status -- and `{}` from matrix_inv_32."""
    n = 40
    for pos in ((5, 5), (0, 39), (39, 0)):
        a = gate_matrix(n, 77)
        a[pos] = bad
        want = oracle.matrix_inv_32(a, n, return_info=True)[1]["status"]
        assert want == oracle.STATUS_SINGULAR
        for inv in (inv_sweep, inv_blocked):
            _, st = run(inv, a)
            assert st[0] == want, (pos, bad)
        assert g.matrix_inv_32(a.reshape(-1), n).size == 0
    # fp64 twin
    a = gate_matrix(n, 78).astype(np.float64)
    a[3, 9] = bad
    _, st = inv_sweep.inv(torch.from_numpy(a).cuda())
    assert int(st[0]) == oracle.matrix_inv_64(a, n, return_info=True)[1]["status"] == oracle.STATUS_SINGULAR
    assert g.matrix_inv_64(a.reshape(-1), n).size == 0


def test_zero_and_rank_deficient_are_status_singular(oracle, inv_sweep, inv_blocked):
    n = 40
    for a in (np.zeros((n, n), np.float32), np.ones((n, n), np.float32)):
        want = oracle.matrix_inv_32(a, n, return_info=True)[1]["status"]
        for inv in (inv_sweep, inv_blocked):
            _, st = run(inv, a)
            assert st[0] == want == oracle.STATUS_SINGULAR
    # an overflowing pivot chain: huge entries make an infinite intermediate -> flagged, never "OK with inf"
    a = gate_matrix(n, 79) * np.float32(3e38)
    for inv in (inv_sweep, inv_blocked):
        x, st = run(inv, a)
        assert st[0] == oracle.STATUS_SINGULAR or np.isfinite(x).all()


def test_a_batch_keeps_valid_members_when_one_is_nonfinite(oracle, inv_blocked):
    n, B = 96, 5
    mats = np.stack([gate_matrix(n, 950 + b) for b in range(B)])
    mats[2, 10, 11] = np.nan
    got, st = run(inv_blocked, mats)
    assert list(st) == [0, 0, 2, 0, 0]
    for b in (0, 1, 3, 4):
        assert np.array_equal(got[b].reshape(-1), oracle.matrix_inv_32_inplace(mats[b], n))


@pytest.mark.parametrize("n", [1, 2, 5, 64, 100, 257, 700])
def test_fp64_sweep_bit_identical_to_fp64_oracle(oracle, inv_sweep, n):
    """The fp64 twin (matrix_inversion_FP64 of the reference): same launches on doubles, bit-identical to the
    oracle's fp64 restatement; device-resident (torch.float64) and through the host-pointer C ABI."""
    for kind in ("gate", "ref100", "hollow"):
        if kind == "hollow" and n == 1:
            continue
        a = dist_matrix(kind, n, 9000 + n).astype(np.float64)
        want, info = oracle.matrix_inv_64(a, n, return_info=True)
        ta = torch.from_numpy(a).cuda()
        x, st = inv_sweep.inv(ta)
        torch.cuda.synchronize()
        assert int(st[0]) == info["status"] == 0
        assert np.array_equal(x.cpu().numpy().reshape(-1), want), (kind, n)
    got = g.matrix_inv_64(a.reshape(-1), n)   # the host-pointer twin, AUTO: blocked from N = 256 on
    bw = _default_f64_block_width(n)
    want_host = want if bw == 0 else oracle.matrix_inv_64_blocked(a, n, bw)
    assert got.dtype == np.float64 and np.array_equal(got, want_host)
    assert g.matrix_inv_64(np.ones((4, 4)).reshape(-1), 4).size == 0   # singular -> empty, like the reference


def _default_f64_block_width(n):
    inv = g.Inverter(algo="auto")
    try:
        return inv.resolved_blocking_f64(n)
    finally:
        inv.close()


@pytest.mark.parametrize("n", [256, 257, 300, 512, 700, 1000, 1500])
def test_fp64_blocked_bit_identical_to_fp64_blocked_mirror(oracle, n):
    """The fp64 blocked path (windowed fused steps + one rank-bw update per block on v_mfma_f64_16x16x4_f64,
    matrix_inversion_FP64 of the reference from N = 256 on): bit-identical to the oracle's fp64 blocked mirror
    (gjo_matrix_inv_64_blocked: same block width, fma chains starting from the old value), for the default block
    width and the two others; and within 1e-11 of the unblocked fp64 result."""
    for bw_req in (0, 64, 256):
        inv = g.Inverter(algo="auto", block_width=bw_req)
        try:
            bw = inv.resolved_blocking_f64(n)
            assert bw in (64, 128, 256) and (bw_req == 0 or bw == min(bw_req, 256) or n < bw_req)
            for kind in ("gate", "ref100", "hollow"):
                a = dist_matrix(kind, n, 9500 + n).astype(np.float64)
                want, info = oracle.matrix_inv_64_blocked(a, n, bw, return_info=True)
                x, st = inv.inv(torch.from_numpy(a).cuda())
                torch.cuda.synchronize()
                assert int(st[0]) == info["status"] == 0
                got = x.cpu().numpy().reshape(-1)
                assert np.array_equal(got, want), (kind, n, bw, np.abs(got - want).max())
                ref = oracle.matrix_inv_64(a, n)
                assert np.abs(got - ref).max() <= 1e-9 * np.abs(ref).max()
        finally:
            inv.close()


def test_fp64_blocked_4096_residual_and_speed(oracle):
    """N = 4096 in double through the blocked path: residual at fp64 level, bit-identical to the mirror."""
    n = 4096
    a = gate_matrix(n, 60_001).astype(np.float64)
    inv = g.Inverter(algo="auto")
    try:
        bw = inv.resolved_blocking_f64(n)
        ta = torch.from_numpy(a).cuda()
        x, st = inv.inv(ta)
        torch.cuda.synchronize()
        res = float((ta @ x - torch.eye(n, dtype=torch.float64, device="cuda")).abs().sum(dim=1).max())
        assert int(st[0]) == 0 and res < 1e-10, res
        want = oracle.matrix_inv_64_blocked(a, n, bw)
        assert np.array_equal(x.cpu().numpy().reshape(-1), want)
        # a singular / non-finite member of a batch is flagged, the others are untouched
        b3 = torch.stack([ta[:512, :512].contiguous(), torch.ones(512, 512, dtype=torch.float64, device="cuda"),
                          ta[512:1024, 512:1024].contiguous()])
        _, st3 = inv.inv(b3)
        torch.cuda.synchronize()
        assert st3.tolist() == [0, 2, 0]
    finally:
        inv.close()


def test_fp64_1024_residual(inv_sweep):
    n = 1024
    a = torch.from_numpy(gate_matrix(n, 60_000).astype(np.float64)).cuda()
    x, st = inv_sweep.inv(a)
    torch.cuda.synchronize()
    res = float((a @ x - torch.eye(n, dtype=torch.float64, device="cuda")).abs().sum(dim=1).max())
    assert int(st[0]) == 0 and res < 1e-11, res


def test_device_residual_matches_oracle(oracle, inv_blocked):
    n = 300
    a = dist_matrix("ref100", n, 5)
    x, _ = run(inv_blocked, a)
    ta, tx = torch.from_numpy(a).cuda(), torch.from_numpy(x).cuda()
    r = inv_blocked.residual(ta, tx).cpu().numpy()[0]
    assert r[0] == pytest.approx(oracle.residual_inf(a, x, n), rel=1e-9)
    assert r[1] == pytest.approx(oracle.residual_inf_left(a, x, n), rel=1e-9)
    assert r[2] == pytest.approx(oracle.frobenius_metric(a, x, n), rel=1e-6, abs=1e-12)


def test_non_default_stream_and_determinism(inv_blocked):
    n = 384
    a = torch.from_numpy(gate_matrix(n, 3)).cuda()
    x0, _ = inv_blocked.inv(a)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        x1, _ = inv_blocked.inv(a)
    s.synchronize()
    torch.cuda.synchronize()
    assert torch.equal(x0, x1)


# ---- BASELINE.json full sizes: size-independent properties + the residual gate ----------

def _full_size_properties(inv, n, batch, seed):
    rng = np.random.default_rng(seed)
    mats = torch.from_numpy(np.stack([gate_matrix(n, seed + b) for b in range(batch)])).cuda()
    x, st = inv.inv(mats)
    res = inv.residual(mats, x)
    # (1) power-of-two scaling is exact in every operation: inv(2A) == inv(A)/2 bit for bit
    x2, _ = inv.inv(mats * 2.0)
    # (2) row-permuted input: partial pivoting picks the same rows, so inv(P A) == inv(A)[:, perm] bit for bit
    perm = torch.from_numpy(rng.permutation(n)).cuda()
    xp, _ = inv.inv(mats[:, perm, :].contiguous())
    # (3) involution: inv(inv(A)) ~= A
    xx, _ = inv.inv(x)
    torch.cuda.synchronize()
    assert int(st.max()) == 0
    assert float(res[:, 0].max()) < 1e-3 and float(res[:, 1].max()) < 1e-3, res
    assert torch.equal(x2 * 2.0, x)
    assert torch.equal(xp, x[:, :, perm])
    back = float((xx - mats).abs().max() / mats.abs().max())
    assert back < 1e-4, back
    return float(res[:, 0].max())


def test_c1_single_4096_blocked(inv_blocked):
    r = _full_size_properties(inv_blocked, 4096, 1, 10_000)
    print("C1 blocked residual", r)


def test_c1_single_4096_sweep_and_blocked_agree(inv_sweep, inv_blocked):
    n = 4096
    a = torch.from_numpy(gate_matrix(n, 10_001)).cuda()
    xs, st = inv_sweep.inv(a)
    xb, _ = inv_blocked.inv(a)
    rs = inv_sweep.residual(a, xs)
    torch.cuda.synchronize()
    assert int(st.item()) == 0 and float(rs[0, 0]) < 1e-3
    rel = float((xs - xb).abs().max() / xs.abs().max())
    assert rel < 1e-5, rel


def test_blocked_4200_two_workgroup_panel_bit_identical_to_oracle(oracle, inv_blocked):
    """N = 4200 pads to 4224 rows: more than one workgroup holds at 4 rows per lane, so the first sub-panels
    run on a panel shared by two workgroups (per-step exchange of the local winners through global memory);
    the arithmetic is unchanged: bit-identical to the reference-order oracle."""
    n = 4200
    a = gate_matrix(n, 40_000)
    w, bw = inv_blocked.resolved_blocking(n, 1)
    widths = inv_blocked.resolved_panel_widths(n, 1)
    assert (w, bw) == (16, 256) and set(widths) == {16} and len(widths) == 17
    got, st = run(inv_blocked, a)
    want = oracle_inverse(oracle, a, n)
    assert st[0] == 0 and np.array_equal(got.reshape(-1), want)


def test_blocked_4200_narrow_subpanel_schedule_bit_identical_to_oracle(oracle):
    """The same matrix with the multi-workgroup panel off (a batch too large for it takes this path): while more
    than 4096 rows are candidates the single panel workgroup holds 8 rows per lane and only W = 8 columns fit
    in registers; from the second outer block on W = 16."""
    n = 4200
    a = gate_matrix(n, 40_000)
    os.environ["MI32_MULTI_PANEL"] = "0"
    try:
        inv = g.Inverter(algo="blocked")
        try:
            widths = inv.resolved_panel_widths(n, 1)
            assert widths[0] == 8 and set(widths[1:]) == {16} and len(widths) == 17
            got, st = run(inv, a)
            want = oracle_inverse(oracle, a, n)
            assert st[0] == 0 and np.array_equal(got.reshape(-1), want)
        finally:
            inv.close()
    finally:
        del os.environ["MI32_MULTI_PANEL"]


def test_blocked_2300_unfused_then_fused_blocks_bit_identical_to_oracle(oracle, inv_blocked):
    """N = 2300 (2304 padded rows): the first outer block still has more than 2048 candidate rows (panel and
    in-block update as separate launches), every later block runs the fused launches -- both modes and the
    hand-over between them in one inversion, bit-identical to the reference-order oracle."""
    n = 2300
    a = dist_matrix("ref100", n, 41_000)
    got, st = run(inv_blocked, a)
    want = oracle_inverse(oracle, a, n)
    assert st[0] == 0 and np.array_equal(got.reshape(-1), want)


def _default_inverter():
    return g.Inverter(algo="auto")   # what matrix_inv_32 runs: AUTO plan, look-ahead where it pays


def test_c1_4096_default_plan_bit_identical_to_oracle(oracle, inv_sweep, monkeypatch):
    """BASELINE configs[1] itself: N = 4096, the default plan (bw 256, W 16; four rows per lane in the unfused
    panels of the first half, fused launches in the second, the pivot rows' strips riding in the panel launches; one
    stream -- the look-ahead starts above 4096 rows) against the reference-order oracle run live on this box -- the
    step-by-step restatement itself, 12 s of CPU -- against its digest committed from the build container
    (tests/golden/make_oracle_digests.py), against the sweep path, and with the look-ahead forced on."""
    dig = load_oracle_digest(4096)
    n, seed = 4096, int(dig["seed"])
    a = gate_matrix(n, seed)
    inv = _default_inverter()
    try:
        assert inv.resolved_algo(n, 1) == g.ALGO_BLOCKED
        got, st = run(inv, a)
        monkeypatch.setenv("MI32_LOOKAHEAD_MIN", "2048")   # the two-stream schedule must give the same bits
        got1, st1 = run(inv, a)
    finally:
        inv.close()
    gots, sts = run(inv_sweep, a)
    want = oracle.matrix_inv_32_inplace(a, n)
    assert st[0] == 0 and st1[0] == 0 and sts[0] == 0
    assert np.array_equal(got.reshape(-1), want)
    assert np.array_equal(got1.reshape(-1), want)
    assert np.array_equal(gots.reshape(-1), want)    # sweep(4096) == oracle, bit for bit
    check_against_oracle_digest(got, dig)


@pytest.mark.parametrize("n", [3072, 3500, 4300, 6100])
def test_lookahead_sizes_bit_identical_to_oracle(oracle, n, monkeypatch):
    """The look-ahead schedule outside 4096: forced on below its default range (MI32_LOOKAHEAD_MIN, read per call: 3072
    padded rows -- the half runs on half of the CUs, with a whole CU's LDS per workgroup -- and a ragged size), its
    default lower end (4300 -> 4352 padded rows) and 6100 (6144 padded rows: the first panels are shared by two
    workgroups, the half runs on three quarters of the CUs) -- with the second stream and without it, bit for bit the
    reference-order oracle."""
    a = gate_matrix(n, 7000 + n)
    if n < 4097:
        monkeypatch.setenv("MI32_LOOKAHEAD_MIN", "2048")
    inv = _default_inverter()
    try:
        got, st = run(inv, a)
        inv.set_lookahead(False)
        got1, st1 = run(inv, a)
    finally:
        inv.close()
    want, info = oracle_inverse(oracle, a, n, return_info=True)
    assert st[0] == st1[0] == info["status"] == 0
    assert np.array_equal(got.reshape(-1), want)
    assert np.array_equal(got1.reshape(-1), want)


@pytest.mark.parametrize("kind", ["ref100", "rand", "hollow"])
def test_c1_4096_reference_distributions_bit_identical_to_oracle(oracle, kind):
    """N = 4096 on the reference's own input distributions (U(0,100) of matrix_inv_pyopencl.py:17 / matrix_inv_numpy.py:40,
    MATLAB rand, the hollow variant of main_file.cpp:46-48): far worse conditioned than D_gate -- different pivot
    sequences, large growth -- and still bit for bit the reference-order oracle."""
    n = 4096
    a = dist_matrix(kind, n, 4096_000 + len(kind))
    inv = _default_inverter()
    try:
        got, st = run(inv, a)
    finally:
        inv.close()
    want, info = oracle_inverse(oracle, a, n, return_info=True)
    assert st[0] == info["status"] == 0
    assert np.array_equal(got.reshape(-1), want), (kind, np.abs(got.reshape(-1) - want).max())


@pytest.mark.parametrize("n", [1024, 2048, 4096])
@pytest.mark.parametrize("kind", ["ref100", "rand"])
def test_residual_on_the_references_own_inputs_against_reference_order_elimination(oracle, inv_blocked, kind, n):
    """on the reference's own input distributions (U(0,100): matrix_inv_pyopencl.py:17, matrix_inv_numpy.py:40;
    rand: test_inversa_mat.mlx) fp32 Gauss-Jordan cannot reach 1e-3, so the gate is relative: the blocked HIP result's
    ||A X - I||_inf must be <= 2 x the residual of the reference-order elimination (one fmaf per element and step,
    mat_inv_32.cpp:28-38).  The matrices are bench.py's (default_rng(4242)).  From round 3 on the two are the same
    bits, so the ratio is exactly 1."""
    rng = np.random.default_rng(4242)
    a = (rng.uniform(0, 100, (n, n)) if kind == "ref100" else rng.uniform(0, 1, (n, n))).astype(np.float32)
    got, st = run(inv_blocked, a)
    want, info = oracle_inverse(oracle, a, n, return_info=True)
    assert st[0] == info["status"] == 0
    r_got, r_ref = oracle.residual_inf(a, got, n), oracle.residual_inf(a, want, n)
    print(f"N={n} {kind}: ||AX-I||inf blocked HIP {r_got:.4e}, reference-order oracle {r_ref:.4e}")
    assert r_got <= 2.0 * r_ref
    assert np.array_equal(got.reshape(-1), want)


@pytest.mark.parametrize("n", [2048, 4096])
def test_forward_error_against_float64_inverse_at_baseline_sizes(inv_sweep, inv_blocked, n):
    """An oracle-independent check at the BASELINE sizes: both HIP paths against numpy.linalg.inv of the same matrix in
    float64, computed on this box: max|X - X64| / max|X64| <= 2 kappa_inf(A) 2^-24, on D_gate and on U(0,100)."""
    for kind in ("gate", "ref100"):
        a = dist_matrix(kind, n, 123_000 + n)
        a64 = a.astype(np.float64)
        inv64 = np.linalg.inv(a64)
        kappa = np.abs(a64).sum(axis=1).max() * np.abs(inv64).sum(axis=1).max()
        tol = 2.0 * kappa * EPS32
        for inv in (inv_sweep, inv_blocked):
            got, st = run(inv, a)
            assert st[0] == 0
            err = rel_err(got, inv64)
            print(f"N={n} {kind}: forward error {err:.3e}, bound {tol:.3e} (kappa_inf {kappa:.3e})")
            assert err <= tol, (kind, n)


@pytest.mark.parametrize("n", [8200, 16384])
def test_shared_panel_sizes_bit_identical_to_oracle_digest(n):
    """N = 8200 (8320 padded rows: three workgroups share the first panels) and N = 16384 = BASELINE configs[4]
    (four workgroups, bw = 512): bit-exact against the digest of the reference-order oracle (sha256 of all N^2
    outputs + 4096 sampled entries + per-row |x| sums; the oracle takes minutes on a CPU at 16384, so it ran in the
    build container: tests/golden/make_oracle_digests.py)."""
    dig = load_oracle_digest(n)
    a = gate_matrix(n, int(dig["seed"]))
    inv = _default_inverter()
    try:
        got, st = run(inv, a)
    finally:
        inv.close()
    assert st[0] == 0
    check_against_oracle_digest(got, dig)


def test_8200_live_oracle(oracle):
    """The same N = 8200 comparison against the oracle run live on this box (1.1 TFLOP of CPU work on its
    AVX2/OpenMP build): no dependence on a committed digest."""
    n = 8200
    a = gate_matrix(n, 50_001)
    inv = _default_inverter()
    try:
        got, st = run(inv, a)
    finally:
        inv.close()
    want = oracle_inverse(oracle, a, n)
    assert st[0] == 0 and np.array_equal(got.reshape(-1), want)


def test_three_workgroup_panel_8200(inv_blocked):
    """N = 8200 (8320 padded rows): the first sub-panels are shared by three workgroups: the size-independent
    exact properties and the residual gate (bit-exactness: the two tests above)."""
    r = _full_size_properties(inv_blocked, 8200, 1, 50_000)
    print("8200 (three-workgroup panel) residual", r)


def test_shared_panel_lost_partner_is_flagged_and_poisoned(inv_blocked):
    """The time-out path of the shared panels.  mi32_debug_drop_panel_group(1) (host side only) never launches
    the last workgroup of a shared panel -- what a foreign kernel holding the CUs would cause.  The present
    workgroup must give the partner up after a bounded wait, flag the matrix MI32_RUNTIME_ERROR, every later
    launch must skip it, and the caller must get NaN, not numbers.  The next call is healthy again."""
    n = 4200   # 4224 padded rows: two workgroups share the first panels
    a = gate_matrix(n, 40_000)
    inv = g.Inverter(algo="blocked")
    try:
        lib = _lib.load()
        lib.mi32_debug_drop_panel_group(1)
        try:
            x, st = run(inv, a)
        finally:
            lib.mi32_debug_drop_panel_group(0)
        assert st[0] == _lib.MI32_RUNTIME_ERROR
        assert np.isnan(x).all()
        x2, st2 = run(inv, a)
        assert st2[0] == 0 and np.isfinite(x2).all()
        ta, tx = torch.from_numpy(a).cuda(), torch.from_numpy(x2).cuda()
        assert float(inv.residual(ta, tx)[0, 0]) < 1e-3
    finally:
        inv.close()


def test_two_inverters_on_two_threads_never_give_a_wrong_finite_answer(oracle):
    """Two contexts inverting N = 4200 on one device at the same time: both use shared (two-workgroup) panels and
    the look-ahead, and compete for the CUs those need.  Each result must be either the oracle's bits with status 0
    or a clean MI32_RUNTIME_ERROR with a NaN-filled inverse (a panel gave a partner up) -- never finite wrong numbers."""
    import threading

    n = 4200
    mats = [gate_matrix(n, 41_000 + i) for i in range(2)]
    want = [oracle_inverse(oracle, m, n) for m in mats]
    results = [None, None]

    def worker(i):
        inv = g.Inverter(algo="auto")
        try:
            s = torch.cuda.Stream()
            outs = []
            with torch.cuda.stream(s):
                ta = torch.from_numpy(mats[i]).cuda()
                for _ in range(3):
                    x, st = inv.inv(ta)
                    s.synchronize()
                    outs.append((x.cpu().numpy(), int(st[0])))
            results[i] = outs
        finally:
            inv.close()

    ts = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for i in range(2):
        assert results[i] is not None
        for x, st in results[i]:
            if st == 0:
                assert np.array_equal(x.reshape(-1), want[i]), i
            else:
                assert st == _lib.MI32_RUNTIME_ERROR and np.isnan(x).all(), (i, st)


def test_multi_gpu_host_batch_entry_point_oversubscribed(oracle):
    """mi32_matrix_inv_32_batched_multi (the C ABI's multi-GPU host batch: one context and host thread per GPU, every
    GPU copies and inverts its own contiguous shard, worst status returned).  On this one-GPU box the logical GPUs
    are mapped onto device 0 (MI32_MULTI_OVERSUBSCRIBE=1): the threading, the ragged shards (7 matrices over 3 and
    over 4 GPUs), a singular member and the status reduction run as on a node; every inverse must be bit-identical
    to the single-context result and to the oracle."""
    n, B = 300, 7
    mats = np.stack([gate_matrix(n, 88_000 + b) for b in range(B)])
    mats[4] = 1.0  # singular member, lands in the middle shard
    single, st_single = g.matrix_inv_32_batched(mats)
    assert list(st_single) == [0, 0, 0, 0, 2, 0, 0]
    lib = _lib.load()
    fp, ip = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int)
    # more GPUs than visible without the switch: refused
    out = np.empty_like(mats)
    st = np.empty(B, np.int32)
    assert lib.mi32_matrix_inv_32_batched_multi(mats.ctypes.data_as(fp), n, B, out.ctypes.data_as(fp), st.ctypes.data_as(ip),
                                                torch.cuda.device_count() + 1) == _lib.MI32_BAD_SHAPE
    os.environ["MI32_MULTI_OVERSUBSCRIBE"] = "1"
    try:
        for ngpus in (1, 2, 3, 4, 7, 9):
            out, st = g.matrix_inv_32_batched(mats, ngpus=ngpus)
            assert list(st) == [0, 0, 0, 0, 2, 0, 0], ngpus
            ok = [b for b in range(B) if b != 4]
            assert np.array_equal(out[ok], single[ok]), ngpus
        for b in (0, 3, 6):
            assert np.array_equal(out[b].reshape(-1), oracle.matrix_inv_32_inplace(mats[b], n))
        # worst status is the return value; status may be NULL
        rc = lib.mi32_matrix_inv_32_batched_multi(mats.ctypes.data_as(fp), n, B, out.ctypes.data_as(fp), None, 3)
        assert rc == _lib.MI32_SINGULAR
        total, compute = g.last_timing()
        assert total >= compute > 0
    finally:
        del os.environ["MI32_MULTI_OVERSUBSCRIBE"]
    # all visible devices (ngpus = 0), no switch
    out0, st0 = g.matrix_inv_32_batched(mats, ngpus=0)
    assert list(st0) == [0, 0, 0, 0, 2, 0, 0] and np.array_equal(out0[[0, 6]], single[[0, 6]])


@pytest.mark.parametrize("n", [300, 1300, 3200])
def test_blocked_scaled_inputs_take_the_full_division_and_stay_bit_identical(oracle, inv_blocked, n):
    """The pivot-row strips divide through a shortened form of hipcc's own division when both operands lie in
    [2^-47, 2^48) and through the full expansion otherwise (mi32_blocked.hip, strip_div).  Matrices scaled by 2^-56 /
    2^+56 put every strip on the full expansion (tiny pivots / tiny numerators), a mixed one -- half the rows scaled
    -- takes both branches inside one strip; each must still be the oracle's bits, and a power-of-two scaling must
    carry through exactly: inv(2^k A) == 2^-k inv(A)."""
    a = gate_matrix(n, 66_000 + n)
    base, st = run(inv_blocked, a)
    assert st[0] == 0
    for k in (-56, 56):
        sc = np.float32(2.0) ** k
        got, st = run(inv_blocked, a * sc)
        assert st[0] == 0
        assert np.array_equal(got, base / sc), k
        if n <= 1300:
            assert np.array_equal(got.reshape(-1), oracle_inverse(oracle, a * sc, n)), k
    mixed = a.copy()
    mixed[::2] *= np.float32(2.0) ** -52   # every other row tiny: pivots and numerators on both sides of the range
    got, st = run(inv_blocked, mixed)
    want, info = oracle_inverse(oracle, mixed, n, return_info=True)
    assert st[0] == info["status"] == 0
    assert np.array_equal(got.reshape(-1), want)


def test_sweep_2048_bit_identical_to_oracle(oracle, inv_sweep):
    n = 2048
    a = gate_matrix(n, 20_000)
    got, st = run(inv_sweep, a)
    want = oracle.matrix_inv_32_inplace(a, n)
    assert st[0] == 0 and np.array_equal(got.reshape(-1), want)


def test_c2_real_shape_64_x_2048(oracle):
    """BASELINE configs[2] at its real shape: 64 x 2048^2 on one GPU, default plan -> outer block width 128 and the
    two-stream batch split (32 + 32).  Residual gate on all 64, the exact size-independent properties on the whole
    batch, and bit-exactness against the reference-order oracle for four of the 64 (two from each half)."""
    n, B = 2048, 64
    inv = _default_inverter()
    try:
        assert inv.resolved_algo(n, B) == g.ALGO_BLOCKED
        res = _full_size_properties(inv, n, B, 30_000)
        print("C2 (64 x 2048^2) worst residual", res)
        mats = torch.from_numpy(np.stack([gate_matrix(n, 30_000 + i) for i in range(B)])).cuda()
        x, st = inv.inv(mats)
        torch.cuda.synchronize()
        assert int(st.max()) == 0
        for b in (0, 31, 32, 63):
            want = oracle_inverse(oracle, gate_matrix(n, 30_000 + b), n)
            assert np.array_equal(x[b].cpu().numpy().reshape(-1), want), b
    finally:
        inv.close()


def test_c2_subset_8_of_2048(inv_blocked):
    r = _full_size_properties(inv_blocked, 2048, 8, 30_000)
    print("8 x 2048^2 (bw 256, unsplit) residual", r)


def test_split_batch_is_bit_identical_to_the_unsplit_batch(oracle, inv_blocked):
    """A GPU-filling batch runs as two halves on two streams (both with the blocking of the whole batch): every
    matrix's inverse and status must be exactly what the single-stream run gives, odd batch sizes included -- and the
    reference-order oracle's bits (this shape runs the 128-column block strip kernel of bw = 128 batches)."""
    n, B = 1024, 67   # 67 x 1024^2 elements >= 64 Mi: split into 34 + 33
    mats = torch.from_numpy(np.stack([gate_matrix(n, 80_000 + b) for b in range(B)])).cuda()
    mats[5] = 1.0     # one singular member, in the first half
    mats[60] = 0.0    # and one in the second
    x_split, st_split = inv_blocked.inv(mats)
    torch.cuda.synchronize()
    os.environ["MI32_BATCH_SPLIT"] = "0"
    try:
        x_one, st_one = inv_blocked.inv(mats)
        torch.cuda.synchronize()
    finally:
        del os.environ["MI32_BATCH_SPLIT"]
    want = [0] * B
    want[5] = want[60] = 2
    assert st_split.tolist() == want and st_one.tolist() == want
    ok = [b for b in range(B) if b not in (5, 60)]
    assert torch.equal(x_split[ok], x_one[ok])
    r = inv_blocked.residual(mats[ok], x_split[ok])
    assert float(r[:, 0].max()) < 1e-3
    for b in (0, 33, 34, 66):
        want_b = oracle_inverse(oracle, gate_matrix(n, 80_000 + b), n)
        assert np.array_equal(x_split[b].cpu().numpy().reshape(-1), want_b), b


def test_small_resident_batch_is_split_and_bit_identical(oracle, inv_blocked):
    """Five matrices of 3700 rows (>= 64 Mi elements: split 3 + 2 over the two streams; unfused four-rows-per-lane
    panels in the first blocks, whose launches leave most of the chip to the other half): the same bits as the
    single-stream run and as the reference-order oracle."""
    n, B = 3700, 5
    mats_np = [dist_matrix("ref100" if b % 2 else "gate", n, 81_000 + b) for b in range(B)]
    mats = torch.from_numpy(np.stack(mats_np)).cuda()
    x_split, st_split = inv_blocked.inv(mats)
    torch.cuda.synchronize()
    os.environ["MI32_BATCH_SPLIT"] = "0"
    try:
        x_one, st_one = inv_blocked.inv(mats)
        torch.cuda.synchronize()
    finally:
        del os.environ["MI32_BATCH_SPLIT"]
    assert st_split.tolist() == [0] * B and st_one.tolist() == [0] * B
    assert torch.equal(x_split, x_one)
    for b in (1, 4):
        want_b = oracle_inverse(oracle, mats_np[b], n)
        assert np.array_equal(x_split[b].cpu().numpy().reshape(-1), want_b), b


def test_c4_single_16384_maximum_size(inv_blocked):
    """C4 = the largest order the blocked path takes (four-workgroup panel for the first 12288 pivots).  The fp64
    residual product is done by torch (checker only, 8.8 TFLOP); plus the exact power-of-two scaling property.
    (Bit-exactness against the oracle: test_shared_panel_sizes_bit_identical_to_oracle_digest.)"""
    n = 16384
    a = torch.from_numpy(gate_matrix(n, 70_000)).cuda()
    x, st = inv_blocked.inv(a)
    x2, st2 = inv_blocked.inv(a * 2.0)
    torch.cuda.synchronize()
    assert int(st[0]) == 0 and int(st2[0]) == 0
    assert torch.equal(x2 * 2.0, x)
    del x2
    r = a.double() @ x.double()
    r.diagonal().sub_(1.0)
    res = float(r.abs().sum(dim=1).max())
    print("C4 residual", res)
    # BASELINE.json gates ||A X - I||_inf < 1e-3 at N = 4096.  At N = 16384 the reference's own operation order (one
    # fp32 fmaf per element and pivot step, 16384 steps) leaves 1.36e-3 on this matrix, and the HIP result IS that
    # result bit for bit (test_shared_panel_sizes_bit_identical_to_oracle_digest[16384]): the bound here only says
    # "fp32 Gauss-Jordan accuracy", parity is what the digest test pins.
    assert res < 2e-3, res


def test_host_pointer_entry_points_are_thread_safe(oracle):
    """matrix_inv_32 and matrix_inversion_FP64 (host-pointer twins) share the default context's staging buffers:
    two threads calling them concurrently, with sizes that force the buffers to grow, must each get exactly the
    single-threaded answer (one mutex serialises all host-pointer entry points)."""
    import threading

    jobs32 = [(n, gate_matrix(n, 600 + n)) for n in (64, 300, 150, 520, 90)]
    jobs64 = [(n, gate_matrix(n, 700 + n).astype(np.float64)) for n in (200, 50, 410, 120, 330)]
    want32 = [g.matrix_inv_32(a.reshape(-1), n).copy() for n, a in jobs32]
    want64 = [g.matrix_inv_64(a.reshape(-1), n).copy() for n, a in jobs64]
    errors = []

    def worker(fn, jobs, want):
        try:
            for _ in range(6):
                for (n, a), w in zip(jobs, want):
                    got = fn(a.reshape(-1), n)
                    if not np.array_equal(got, w):
                        errors.append((fn.__name__, n))
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    ts = [threading.Thread(target=worker, args=(g.matrix_inv_32, jobs32, want32)),
          threading.Thread(target=worker, args=(g.matrix_inv_64, jobs64, want64)),
          threading.Thread(target=worker, args=(g.matrix_inv_32, jobs32[::-1], want32[::-1]))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors[:5]
    total, compute = g.last_timing()
    assert total >= compute > 0


def test_device_call_without_status_buffer(inv_blocked):
    """d_status = NULL is allowed by the C ABI: the context keeps the status words itself."""
    n = 200
    a = torch.from_numpy(gate_matrix(n, 12)).cuda()
    want, _ = inv_blocked.inv(a)
    out = torch.empty_like(a)
    inv_blocked._bind_stream()
    rc = inv_blocked._lib.mi32_inv_device(inv_blocked._h, ctypes.c_void_p(a.data_ptr()), n, 1,
                                          ctypes.c_void_p(out.data_ptr()), None)
    torch.cuda.synchronize()
    assert rc == 0 and torch.equal(out, want)


def test_fp32_bench_twin_fills_the_references_ten_timing_slots(oracle, tmp_path):
    """Res FP32_bench(vector<float>, int) (FP32_bench.cpp:11, res_struct.h:4-6): same inverse as matrix_inv_32 and
    the reference's ten durations -- queue, buffers, build, makeAug, pivot, row, column, compute, getInverted,
    total (FP32_bench.cpp:256-443) -- through Python and through a C++ caller of include/mat_inv_bench.h."""
    n = 700
    a = dist_matrix("ref100", n, 4)
    want = g.matrix_inv_32(a.reshape(-1), n)
    got, t = g.fp32_bench(a.reshape(-1), n)
    assert np.array_equal(got, want)
    assert list(t) == list(_lib.TIMES10_SLOTS) and all(v >= 0 for v in t.values())
    assert t["build"] == 0.0 and t["row"] == 0.0                      # AOT code object; fixRow has no launch
    assert t["makeAug"] > 0 and t["pivot"] > 0 and t["column"] > 0 and t["getInverted"] > 0
    assert t["pivot"] + t["column"] + t["makeAug"] <= t["compute"] * 1.05
    assert t["buffers"] + t["compute"] + t["getInverted"] <= t["total"] * 1.05
    assert t["compute"] < t["total"] < 5.0
    assert g.fp32_bench(a.reshape(-1), n + 1)[1] == {}                    # bad shape -> empty Res
    assert g.fp32_bench(np.ones(64, np.float32), 8)[1] == {}             # invalid matrix -> empty Res
    src = tmp_path / "bench_caller.cpp"
    src.write_text(r'''
#include <cstdio>
#include "mat_inv_bench.h"
int main() {
    std::vector<float> a = {2, 1, 0,  1, 3, 1,  0, 1, 4};
    Res r = FP32_bench(a, 3);
    std::printf("%zu %zu %zu\n", r.inversa32.size(), r.times.size(), r.inversa64.size());
    for (float v : r.inversa32) std::printf("%.9g\n", v);
    Res bad = FP32_bench(a, 2);
    std::printf("bad %zu %zu\n", bad.inversa32.size(), bad.times.size());
    return 0;
}
''')
    exe = tmp_path / "bench_caller"
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                           "-L", libdir, "-lmat_inv_32", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = out.stdout.strip().splitlines()
    assert lines[0] == "9 10 0" and lines[-1] == "bad 0 0"
    got3 = np.array([float(v) for v in lines[1:10]], np.float32)
    assert np.allclose(got3, oracle.matrix_inv_32(np.array([2, 1, 0, 1, 3, 1, 0, 1, 4], np.float32), 3), rtol=1e-6)


def test_sweep_series_line_format_and_error_magnitude(tmp_path):
    """tools/sweep_series.py = the reference's size sweep (matrix_inv_pyopencl.py:358-371: N = 10, 20, ..., one line
    `N t_compute t_total err` per size, err = sqrt(N) - sqrt(sum(C @ C)), PY:341-352) on a short range: line format,
    sizes, timings and the magnitude of the reference's own error metric."""
    out = tmp_path / "series.txt"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sweep_series.py"), "--max", "310", "--out", str(out),
                        "--times10", str(tmp_path / "times.txt")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    rows = [ln.split() for ln in out.read_text().strip().splitlines()]
    assert [int(x[0]) for x in rows] == list(range(10, 310, 10))       # PY:361-366: step 10 below 2000
    for x in rows:
        assert len(x) == 4
        nn, t_compute, t_total, err = int(x[0]), float(x[1]), float(x[2]), float(x[3])
        assert 0 < t_compute <= t_total < 5.0
        assert abs(err) < 1e-2 * np.sqrt(nn), (nn, err)   # fp32 Gauss-Jordan on U(0,100): |err| << sqrt(N)
    trows = [ln.split() for ln in (tmp_path / "times.txt").read_text().strip().splitlines()]
    assert len(trows) == len(rows) and all(len(x) == 11 for x in trows)


def _dominant(n, seed, dtype):
    rng = np.random.default_rng(seed)
    a = rng.uniform(-1.0, 1.0, (n, n))
    a[np.arange(n), np.arange(n)] = np.abs(a).sum(axis=1) + 1.0
    return a.astype(dtype)


@pytest.mark.parametrize("n", [1, 3, 64, 257, 600, 1000, 2300])
def test_no_pivot_variant_bit_identical_to_oracle(oracle, n):
    """matrix_inversion_no_pivots (headers.h:11, matrix_inversion_no_pivots.cpp:10): the diagonal entry is every
    step's pivot.  fp64 through the host-pointer twin, fp32 / fp64 device-resident through a context with
    pivoting off: bit-identical to the oracle's no-pivot restatement; a zero diagonal entry -> status 2 / {}."""
    a64 = _dominant(n, 500 + n, np.float64)
    want64 = oracle.matrix_inversion_no_pivots(a64, n)
    got = g.matrix_inversion_no_pivots(a64.reshape(-1), n)
    assert got.dtype == np.float64 and np.array_equal(got, want64)
    inv = g.Inverter(algo="auto", pivoting=False)
    try:
        # fp32: blocked from 512 rows on (the W x W diagonal block is the whole "panel", every other row is taken
        # through the steps by the update tiles); fp64: the sweep kernels at every size
        assert inv.resolved_algo(n, 1) == (g.ALGO_SWEEP if n < 512 else g.ALGO_BLOCKED)
        x, st = inv.inv(torch.from_numpy(a64).cuda())
        torch.cuda.synchronize()
        assert int(st[0]) == 0 and np.array_equal(x.cpu().numpy().reshape(-1), want64)
        a32 = a64.astype(np.float32)
        x, st = inv.inv(torch.from_numpy(a32).cuda())
        torch.cuda.synchronize()
        assert int(st[0]) == 0
        assert np.array_equal(x.cpu().numpy().reshape(-1), oracle.matrix_inversion_no_pivots(a32, n))
        if n >= 3:
            h = a32.copy()
            h[1, 1] = 0.0
            h[1, 0] = 0.0        # keeps the (1,1) entry exactly zero after step 0
            want = oracle.matrix_inversion_no_pivots(h, n, return_info=True)[1]["status"]
            _, st = inv.inv(torch.from_numpy(h).cuda())
            torch.cuda.synchronize()
            assert int(st[0]) == want == oracle.STATUS_SINGULAR
            assert g.matrix_inversion_no_pivots(h.astype(np.float64).reshape(-1), n).size == 0
    finally:
        inv.close()
    assert g.matrix_inversion_no_pivots(a64.reshape(-1), n + 1).size == 0   # the shape guards


def test_no_pivot_blocked_4096_bit_identical_and_timed(oracle):
    """The no-pivot variant (matrix_inversion_no_pivots.cpp:10) in fp32 at N = 4096 through the blocked path -- with the
    look-ahead and without -- against the oracle's step-by-step no-pivot restatement, and how long it takes next to
    the sweep kernels (one launch per pivot step)."""
    import time

    n = 4096
    a = _dominant(n, 4096, np.float32)
    want = oracle.matrix_inversion_no_pivots(a, n)
    ta = torch.from_numpy(a).cuda()
    times = {}
    for algo in ("blocked", "sweep"):
        inv = g.Inverter(algo=algo, pivoting=False)
        try:
            x, st = inv.inv(ta)
            torch.cuda.synchronize()
            assert int(st[0]) == 0
            assert np.array_equal(x.cpu().numpy().reshape(-1), want), algo
            t0 = time.perf_counter()
            for _ in range(3):
                inv.inv(ta)
            torch.cuda.synchronize()
            times[algo] = (time.perf_counter() - t0) / 3 * 1e3
            if algo == "blocked":
                inv.set_lookahead(False)
                x1, _ = inv.inv(ta)
                torch.cuda.synchronize()
                assert torch.equal(x1, x)
        finally:
            inv.close()
    print(f"no-pivot fp32 N=4096: blocked {times['blocked']:.2f} ms, sweep {times['sweep']:.2f} ms")
    assert times["blocked"] < times["sweep"]


def test_bench_py_contract_line():
    """bench.py prints ONE JSON line with the driver's contract keys, the roofline and cpu_baseline objects and the
    round-2 additions (e2e through matrix_inv_32, time-dominant kernel class, reference distributions), on a small run."""
    import json

    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--n", "640", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "e2e"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["dtype"] == "f32" and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["higher_is_better"] is True and "workload" in d["config"]
    assert abs(d["value"] - 1e3 * 1 / d["ms_per_step"]) < 1e-6 * d["value"]
    rf = d["roofline"]
    assert rf["bound"] in ("mfma", "hbm") and rf["peak"] > 0 and 0 < rf["frac"] < 1 and "time_dominant" in rf
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    cb = d["cpu_baseline"]
    assert cb["kind"] == "reference" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    assert d["e2e"]["e2e_ms"] >= d["e2e"]["compute_ms_inside"] > 0 and d["speedup_vs_numpy_e2e"] > 0
    assert d["residual_inf"] < 1e-3 and d["status_max"] == 0
    assert set(d["residuals_reference_distributions"]) == {"D_ref100", "D_rand"}


def test_experiment_header_twins(oracle, tmp_path):
    """The rest of the reference's experiment header (headers.h:5-16) on the HIP path: FP64_bench / no_pivots_bench
    (Res.inversa64 + the ten timing slots), matrix_inversion_FP32 (the experiment twin of matrix_inv_32) and the
    verification helper matrix_multiply (sqrt(N) - ||A B||_F, product in double), through Python and through a C++
    caller of include/mat_inv_bench.h."""
    n = 300
    a = dist_matrix("gate", n, 77).astype(np.float64)
    x, t = g.fp64_bench(a.reshape(-1), n)
    assert np.array_equal(x, g.matrix_inv_64(a.reshape(-1), n)) and list(t) == list(_lib.TIMES10_SLOTS)
    assert t["pivot"] > 0 and t["column"] > 0 and t["compute"] < t["total"]          # blocked fp64: panel steps + rank-bw updates
    dom = np.abs(a) + np.diag(np.abs(a).sum(axis=1) + 1.0)
    y, t2 = g.fp64_bench(dom.reshape(-1), n, pivoting=False)
    assert np.array_equal(y, oracle.matrix_inversion_no_pivots(dom, n)) and t2["pivot"] == 0.0 and t2["column"] > 0
    # matrix_multiply: the reference's acceptance metric on (inverse, matrix), against numpy in float64
    err = g.matrix_multiply(x, a)
    want = np.sqrt(n) - np.linalg.norm(x.reshape(n, n) @ a, "fro")
    assert abs(err - want) <= 1e-9 and abs(err) < 1e-9
    r = np.random.default_rng(5)
    p_, q_ = r.normal(size=(n, n)), r.normal(size=(n, n))
    assert g.matrix_multiply(p_, q_) == pytest.approx(np.sqrt(n) - np.linalg.norm(p_ @ q_, "fro"), rel=1e-12)
    with pytest.raises(ValueError):
        g.matrix_multiply(np.ones(10), np.ones(10))          # 10 is not a perfect square
    src = tmp_path / "exp_caller.cpp"
    src.write_text(r'''
#include <cstdio>
#include "mat_inv_bench.h"
int main() {
    std::vector<double> a = {2, 1, 0,  1, 3, 1,  0, 1, 4};
    Res r = FP64_bench(a, 3), q = no_pivots_bench(a, 3);
    std::vector<float> f = matrix_inversion_FP32(std::vector<float>(a.begin(), a.end()), 3);
    std::printf("%zu %zu %zu %zu %zu\n", r.inversa64.size(), r.times.size(), q.inversa64.size(), q.times.size(), f.size());
    std::printf("%.3e\n", matrix_multiply(r.inversa64, a));
    std::printf("%zu\n", matrix_inversion_FP32(std::vector<float>(16, 1.0f), 4).size());
    return 0;
}
''')
    exe = tmp_path / "exp_caller"
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                           "-L", libdir, "-lmat_inv_32", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = out.stdout.strip().splitlines()
    assert lines[0] == "9 10 9 10 9" and abs(float(lines[1])) < 1e-12 and lines[2] == "0"
