"""CPU tests of the oracle itself: pinned against the golden vectors (outputs of the
reference's NumPy script + seeded numpy.linalg.inv fixtures) and against the reference's own
acceptance properties.  No GPU, no product code."""
import os

import numpy as np
import pytest

from conftest import forward_tolerance, gate_matrix, golden_files, load_golden, rel_err


@pytest.mark.parametrize("path", golden_files(), ids=lambda p: os.path.basename(p)[:-4])
def test_oracle_matches_golden(oracle, path):
    a, inv64 = load_golden(path)
    n = a.shape[0]
    x, info = oracle.matrix_inv_32(a, n, return_info=True)
    assert info["status"] == oracle.STATUS_OK
    # (1) entry-wise agreement with the float64 inverse within the fp32 forward bound
    assert rel_err(x, inv64) <= forward_tolerance(a)
    # (2) the reference's exact-identity acceptance check (matrix_inversion_FP32.cpp:814-835)
    assert oracle.left_half_is_identity(info["aug"], n)
    # (3) A * inv(A) ~= I, both sides, and the reference's Frobenius metric
    tol_res = 4 * n * forward_tolerance(a)  # residual <= ||A|| * ||dX||: generous row-sum bound
    assert oracle.residual_inf(a, x, n) <= max(tol_res, 1e-6)
    assert oracle.residual_inf_left(a, x, n) <= max(tol_res, 1e-6)
    assert abs(oracle.frobenius_metric(a, x, n)) <= max(tol_res, 1e-6)


def test_reference_script_fixture_is_reference_output():
    """ref_just_inv_K*.npz hold what the reference's own just_inv computed (matrix_inv_numpy.py:44):
    the captured float64 inverse really inverts the captured float64 input."""
    for p in golden_files():
        if "ref_just_inv" not in p:
            continue
        d = np.load(p)
        a64, inv = d["a64"], d["inv64_of_a64"]
        assert np.abs(a64 @ inv - np.eye(a64.shape[0])).max() < 1e-8
        assert a64.min() >= 0.0 and a64.max() <= 100.0  # U(0,100), matrix_inv_numpy.py:40


@pytest.mark.parametrize("K", [8, 64, 256])
def test_oracle_matches_the_reference_scripts_own_output(oracle, K):
    """The direct pin: the oracle, fed the fp32 cast of the matrix the reference's just_inv(K) built
    (matrix_inv_numpy.py:40-41), against the inverse the reference itself computed from the float64
    matrix (matrix_inv_numpy.py:44; captured by tests/golden/make_golden.py).  Tolerance: the fp32
    forward bound 2 kappa_inf 2^-24 for the oracle's own rounding plus kappa_inf 2^-24 for the fp32
    cast of the input (|dA| <= 2^-24 |A| moves the exact inverse by <= kappa 2^-24 relative)."""
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"ref_just_inv_K{K}.npz"))
    a, ref_out = d["a"], d["inv64_of_a64"]
    x, info = oracle.matrix_inv_32(a, K, return_info=True)
    assert info["status"] == oracle.STATUS_OK
    assert rel_err(x, ref_out) <= forward_tolerance(a, factor=3.0)
    # and every other restatement the GPU paths are compared with, against the same reference output
    for y in (oracle.matrix_inv_32_inplace(a, K), oracle.matrix_inv_32_blocked2(a, K, 16, 256)):
        assert rel_err(y, ref_out) <= forward_tolerance(a, factor=3.0)


def test_nonfinite_input_and_bad_pivots_are_status_singular(oracle):
    """The boundary rule (README.md:54): a zero / NaN / infinite pivot, or any non-finite input entry,
    is an invalid matrix -> status SINGULAR from every restatement (the HIP paths are held to the same)."""
    n = 40
    base = gate_matrix(n, 77)
    cases = {"nan": np.nan, "inf": np.inf, "-inf": -np.inf}
    for name, v in cases.items():
        a = base.copy()
        a[5, 7] = v
        for fn in (lambda m: oracle.matrix_inv_32(m, n, return_info=True),
                   lambda m: oracle.matrix_inv_32_inplace(m, n, return_info=True),
                   lambda m: oracle.matrix_inv_32_blocked2(m, n, 16, 256, return_info=True)):
            assert fn(a)[1]["status"] == oracle.STATUS_SINGULAR, name
        _, info = oracle.matrix_inv_64(a.astype(np.float64), n, return_info=True)
        assert info["status"] == oracle.STATUS_SINGULAR
    for a in (np.zeros((n, n), np.float32), np.ones((n, n), np.float32)):
        assert oracle.matrix_inv_32(a, n, return_info=True)[1]["status"] == oracle.STATUS_SINGULAR
    assert oracle.matrix_inv_32(base, n, return_info=True)[1]["status"] == oracle.STATUS_OK


def test_fast_and_portable_oracle_builds_are_bit_identical(oracle):
    """libgj_oracle.so runs the blocked mirrors' rank-k updates 64 elements at a time on AVX2 FMA lanes
    and over OpenMP threads; libgj_oracle_generic.so one element at a time through libm's fmaf.  Every
    output element is its own k-ascending fmaf chain either way: same bits."""
    import ctypes

    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")
    libs = [ctypes.CDLL(os.path.join(here, nm)) for nm in ("libgj_oracle.so", "libgj_oracle_generic.so")]
    fp, ip = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int)
    for n, bw, w in ((130, 128, 16), (300, 256, 8), (700, 256, 16), (1100, 512, 32)):
        a = np.ascontiguousarray(gate_matrix(n, 5 * n).reshape(-1))
        ws = np.full(16, w, np.int32)
        outs = []
        for lib in libs:
            lib.gjo_matrix_inv_32_blocked2w.restype = ctypes.c_int
            lib.gjo_matrix_inv_32_blocked2w.argtypes = [fp, ctypes.c_size_t, ctypes.c_int, fp, ip, ctypes.c_int,
                                                        ctypes.c_int, ip]
            out = np.empty(n * n, np.float32)
            assert lib.gjo_matrix_inv_32_blocked2w(a.ctypes.data_as(fp), a.size, n, out.ctypes.data_as(fp),
                                                   ws.ctypes.data_as(ip), ws.size, bw, None) == 0
            outs.append(out)
        assert np.array_equal(outs[0], outs[1]), (n, bw, w)
        # ... and the blocked evaluation of the sequential arithmetic (AVX2 fnmadd lanes vs fmaf through libm)
        outs = []
        for lib in libs:
            lib.gjo_matrix_inv_32_blocked_exact.restype = ctypes.c_int
            lib.gjo_matrix_inv_32_blocked_exact.argtypes = [fp, ctypes.c_size_t, ctypes.c_int, fp, ctypes.c_int, ip]
            out = np.empty(n * n, np.float32)
            assert lib.gjo_matrix_inv_32_blocked_exact(a.ctypes.data_as(fp), a.size, n, out.ctypes.data_as(fp), 128, None) == 0
            outs.append(out)
        assert np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32)), n


def test_oracle_digest_4096_reproduced_here(oracle):
    """The committed digest of the reference-order result at N = 4096 (tests/golden/make_oracle_digests.py) is
    reproduced by whatever oracle build runs on this machine: what the GPU tests compare with is the same
    function of its input everywhere."""
    from conftest import check_against_oracle_digest, load_oracle_digest

    dig = load_oracle_digest(4096)
    a = gate_matrix(4096, int(dig["seed"]))
    x = oracle.matrix_inv_32_blocked_exact(a, 4096, 64)   # another block width than the generator's: same bits
    check_against_oracle_digest(x, dig)


def test_round2_delayed_update_order_was_less_accurate_on_the_references_inputs(oracle):
    """Why the blocked HIP path changed its operation order in round 3.  Rounds 1-2 applied a block of pivots to the
    other columns through the block's COMPOSITE transformation (gjo_matrix_inv_32_blocked2: the transformed panel
    times the pivot rows as they stood before the block).  On the reference's own ill-conditioned inputs (U(0,100),
    matrix_inv_pyopencl.py:17; rand, test_inversa_mat.mlx) that is ~2x less accurate than the reference's own
    order -- one fmaf per element and pivot step (mat_inv_32.cpp:28-38) -- which the blocked evaluation
    gjo_matrix_inv_32_blocked_exact reproduces bit for bit."""
    n = 1024
    for kind, hi in (("ref100", 100.0), ("rand", 1.0)):
        a = np.random.default_rng(4242).uniform(0, hi, (n, n)).astype(np.float32)
        seq = oracle.matrix_inv_32_inplace(a, n)
        exact = oracle.matrix_inv_32_blocked_exact(a, n, 256)
        composite = oracle.matrix_inv_32_blocked2(a, n, 16, 256)
        r_seq, r_comp = oracle.residual_inf(a, seq, n), oracle.residual_inf(a, composite, n)
        assert np.array_equal(seq.view(np.uint32), exact.view(np.uint32))
        assert r_comp > 1.5 * r_seq, (kind, r_seq, r_comp)


@pytest.mark.parametrize("n", [1, 2, 3, 7, 16, 33, 64, 100, 130, 257])
def test_inplace_form_is_bit_identical_to_augmented(oracle, n):
    """The N x N in-place layout the HIP kernels use stores exactly the values of the
    reference's [A|I] panel."""
    for seed, a in ((n, gate_matrix(n, n)),
                    (n + 1, np.random.default_rng(n + 1).uniform(0, 100, (n, n)).astype(np.float32))):
        for arith in (oracle.ARITH_FMA, oracle.ARITH_UNFUSED):
            x, i1 = oracle.matrix_inv_32(a, n, arith_mode=arith, return_info=True)
            y, i2 = oracle.matrix_inv_32_inplace(a, n, arith_mode=arith, return_info=True)
            assert np.array_equal(i1["pivots"], i2["pivots"])
            assert np.array_equal(x, y), (n, seed, arith)


@pytest.mark.parametrize("n", [1, 2, 5, 16, 40, 64])
def test_c_restatement_matches_numpy_mirror(oracle, n):
    a = np.random.default_rng(100 + n).uniform(-1, 1, (n, n)).astype(np.float32)
    x = oracle.matrix_inv_32(a, n).reshape(n, n)
    m = oracle.numpy_mirror_inv(a)
    # the mirror rounds through float64 (double rounding): allow a few ulps of drift per step
    assert np.abs(x - m).max() <= 64 * n * np.finfo(np.float32).eps * np.abs(x).max()


@pytest.mark.parametrize("n,w", [(16, 16), (48, 16), (100, 16), (130, 8), (64, 4), (257, 16)])
def test_blocked_restatement_within_tolerance(oracle, n, w):
    a = gate_matrix(n, 5000 + n)
    x, i1 = oracle.matrix_inv_32_inplace(a, n, return_info=True)
    y, i2 = oracle.matrix_inv_32_blocked(a, n, w, return_info=True)
    assert np.array_equal(i1["pivots"], i2["pivots"])  # well-separated pivots: same choices
    inv64 = np.linalg.inv(a.astype(np.float64))
    assert rel_err(y, inv64) <= forward_tolerance(a)
    assert oracle.residual_inf(a, y, n) < 1e-4


@pytest.mark.parametrize("n", [1, 2, 3, 7, 16, 33, 64, 100, 130, 257, 700])
def test_blocked_exact_is_bit_identical_to_the_step_by_step_restatement(oracle, n):
    """The blocked evaluation the HIP blocked path follows from round 3 on (multipliers kept, pivot-row strip, one
    fmaf chain per element from the old value) is the reference-order elimination bit for bit -- for every block
    width, on the well-conditioned gate matrices and on the reference's own ill-conditioned U(0,100) / rand inputs
    (matrix_inv_pyopencl.py:17, test_inversa_mat.mlx), sign of zeros included."""
    mats = [gate_matrix(n, 7000 + n),
            np.random.default_rng(7100 + n).uniform(0, 100, (n, n)).astype(np.float32),
            np.random.default_rng(7200 + n).uniform(0, 1, (n, n)).astype(np.float32)]
    if n >= 7:
        h = mats[1].copy()
        np.fill_diagonal(h, 0.0)  # hollow (matrix_inv_numpy.py:13-14): pivoting from step 0, exact-zero multipliers
        h[n // 2, : n // 3] = 0.0
        mats.append(h)
    for a in mats:
        x, i1 = oracle.matrix_inv_32_inplace(a, n, return_info=True)
        for bw in (1, 5, 16, 64, 256):
            y, i2 = oracle.matrix_inv_32_blocked_exact(a, n, bw, return_info=True)
            assert i1["status"] == i2["status"]
            assert np.array_equal(i1["pivots"], i2["pivots"])
            assert np.array_equal(x.view(np.uint32), y.view(np.uint32)), (n, bw)


def test_blocked_exact_singular_inputs_agree_with_the_step_by_step_restatement(oracle):
    n = 40
    a = np.random.default_rng(3).uniform(-1, 1, (n, n)).astype(np.float32)
    a[:, 11] = 0.0  # a zero pivot in step 11: inf / NaN from there on
    x, i1 = oracle.matrix_inv_32_inplace(a, n, return_info=True)
    y, i2 = oracle.matrix_inv_32_blocked_exact(a, n, 16, return_info=True)
    assert i1["status"] == i2["status"] == oracle.STATUS_SINGULAR
    assert np.array_equal(i1["pivots"], i2["pivots"])
    assert np.array_equal(x, y, equal_nan=True)  # the same inf / NaN pattern (the sign bit of a NaN is not specified)


def test_shape_guards_return_empty(oracle):
    """mat_inv_32.cpp:206-215: N <= 0 -> {}, int(size/N) != N -> {} (integer division, so a tail
    of fewer than N extra floats is accepted and ignored)."""
    assert oracle.matrix_inv_32(np.ones(4, np.float32), 0).size == 0
    assert oracle.matrix_inv_32(np.ones(4, np.float32), -3).size == 0
    assert oracle.matrix_inv_32(np.ones(3, np.float32), 2).size == 0   # 3/2 = 1 != 2
    assert oracle.matrix_inv_32(np.ones(6, np.float32), 2).size == 0   # 6/2 = 3 != 2
    assert oracle.matrix_inv_32(np.zeros(0, np.float32), 1).size == 0
    a = np.array([4, 7, 2, 6, 99], np.float32)                          # 5/2 = 2: tail ignored
    x = oracle.matrix_inv_32(a, 2)
    assert np.allclose(x.reshape(2, 2), np.linalg.inv(a[:4].reshape(2, 2)), rtol=1e-6)


def test_singular_and_nan_inputs_are_flagged(oracle):
    a = np.array([[1, 2, 3], [2, 4, 6], [1, 0, 1]], np.float32)
    _, info = oracle.matrix_inv_32(a, 3, return_info=True)
    assert info["status"] == oracle.STATUS_SINGULAR
    z = np.zeros((4, 4), np.float32)
    _, info = oracle.matrix_inv_32(z, 4, return_info=True)
    assert info["status"] == oracle.STATUS_SINGULAR
    b = np.eye(3, dtype=np.float32)
    b[1, 1] = np.nan
    _, info = oracle.matrix_inv_32(b, 3, return_info=True)
    assert info["status"] == oracle.STATUS_SINGULAR


def test_pivot_ties_pick_lowest_row(oracle):
    # column 0 has |.| = 2 in rows 1 and 3: the first maximum (row 1) must win
    a = np.array([[1, 0, 0, 0], [-2, 1, 0, 0], [0, 0, 1, 0], [2, 0, 0, 1]], np.float32)
    _, info = oracle.matrix_inv_32(a, 4, return_info=True)
    assert info["pivots"][0] == 1


def test_msvc_rand_stream_and_hollow_fill(oracle):
    """The sweep driver's inputs (main_file.cpp:41-52) from the unseeded MSVC LCG."""
    assert list(oracle.msvc_rand_stream(5)) == [41, 18467, 6334, 26500, 19169]
    a, state = oracle.fill_hollow_msvc(10, 1)
    assert np.all(np.diag(a) == 0) and a.min() >= 0 and a.max() <= 9
    assert a[0, 1] == 41 % 10 and a[0, 2] == 18467 % 10  # the diagonal does not draw from rand()
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "hollow_msvc_K10.npz"))["a"]
    assert np.array_equal(a, g)


def test_reference_pivot_defect_emulation(oracle):
    """the reference's maxPivotKernel inspects a shrinking subset of rows inside the
    work-group that contains r.  The lock-step emulation reproduces the survey's counts: the chosen
    pivot differs from the true arg-max in most steps, yet any non-zero pivot still inverts."""
    n = 256
    a = np.random.default_rng(42).uniform(0, 100, (n, n)).astype(np.float32)
    xt, it = oracle.matrix_inv_32(a, n, return_info=True)
    xd, idf = oracle.matrix_inv_32(a, n, pivot_mode=oracle.PIVOT_REFERENCE_DEFECT, return_info=True)
    differ = int(np.sum(it["pivots"] != idf["pivots"]))
    assert differ > n // 2                      # survey: 244/256 on its input
    assert idf["pivots"][0] == it["pivots"][0]  # r % 256 == 0: every row is inspected
    assert np.all(idf["pivots"][86:] == np.arange(86, n))  # r % 256 >= 86: only row r itself
    rt, rd = oracle.residual_inf(a, xt, n), oracle.residual_inf(a, xd, n)
    assert rt < rd                              # true partial pivoting is the better inverse
    assert oracle.left_half_is_identity(idf["aug"], n)


def test_metrics_on_known_case(oracle):
    a = np.array([[2, 0], [0, 4]], np.float32)
    x = np.array([[0.5, 0], [0, 0.25]], np.float32)
    assert oracle.residual_inf(a, x, 2) == 0.0
    assert oracle.residual_inf_left(a, x, 2) == 0.0
    assert abs(oracle.frobenius_metric(a, x, 2)) < 1e-15
    bad = np.array([[0.5, 0.1], [0, 0.25]], np.float32)
    assert oracle.residual_inf(a, bad, 2) == pytest.approx(0.2, rel=1e-6)
    assert oracle.residual_inf_left(a, bad, 2) == pytest.approx(0.4, rel=1e-6)


def test_fp64_twin_against_numpy_and_fp32_oracle():
    """gjo_matrix_inv_64_inplace (the reference's matrix_inversion_FP64 restated): agrees with
    numpy.linalg.inv to fp64 Gauss-Jordan accuracy, makes the same pivot choices as the fp32 oracle on
    the same (fp32-representable) input, and keeps the reference's guards."""
    import oracle as O

    for n, seed in ((1, 1), (5, 2), (64, 3), (100, 4), (257, 5)):
        rng = np.random.default_rng(seed)
        a = (rng.uniform(-1, 1, (n, n)) + np.sqrt(n) * np.eye(n))[rng.permutation(n)].astype(np.float32)
        x64, info64 = O.matrix_inv_64(a.astype(np.float64), n, return_info=True)
        _, info32 = O.matrix_inv_32_inplace(a, n, return_info=True)
        assert info64["status"] == 0
        want = np.linalg.inv(a.astype(np.float64))
        assert np.abs(x64.reshape(n, n) - want).max() / np.abs(want).max() < 1e-12
        assert np.array_equal(info64["pivots"], info32["pivots"])
    assert O.matrix_inv_64(np.ones(6), 2).size == 0 and O.matrix_inv_64(np.ones(4), 0).size == 0
    _, info = O.matrix_inv_64(np.ones((4, 4)), 4, return_info=True)
    assert info["status"] == O.STATUS_SINGULAR


def _dominant(n, seed, dtype):
    """Strictly diagonally dominant: Gauss-Jordan needs no pivoting (the no-pivot variant's domain)."""
    rng = np.random.default_rng(seed)
    a = rng.uniform(-1.0, 1.0, (n, n))
    a[np.arange(n), np.arange(n)] = np.abs(a).sum(axis=1) + 1.0
    return a.astype(dtype)


@pytest.mark.parametrize("n", [1, 2, 7, 64, 130])
def test_no_pivot_restatement(oracle, n):
    """matrix_inversion_no_pivots (matrix_inversion_no_pivots.cpp:10): on a diagonally dominant input true partial
    pivoting never swaps either, so the no-pivot restatement must give the pivoting one's bits; against float64
    NumPy within the forward bound; a zero diagonal entry -> SINGULAR (the reference returns {}, :670)."""
    a64 = _dominant(n, 300 + n, np.float64)
    x, info = oracle.matrix_inversion_no_pivots(a64, n, return_info=True)
    assert info["status"] == oracle.STATUS_OK
    assert np.array_equal(x, oracle.matrix_inv_64(a64, n))
    assert np.abs(x.reshape(n, n) - np.linalg.inv(a64)).max() <= 1e-12 * np.abs(np.linalg.inv(a64)).max() * n
    a32 = a64.astype(np.float32)
    y, info = oracle.matrix_inversion_no_pivots(a32, n, return_info=True)
    assert info["status"] == oracle.STATUS_OK and y.dtype == np.float32
    assert np.array_equal(y, oracle.matrix_inv_32_inplace(a32, n))
    if n >= 2:
        h = a64.copy()
        h[0, 0] = 0.0   # needs a swap at step 0: the no-pivot variant divides by zero
        assert oracle.matrix_inversion_no_pivots(h, n, return_info=True)[1]["status"] == oracle.STATUS_SINGULAR
        assert oracle.matrix_inv_64(h, n, return_info=True)[1]["status"] == oracle.STATUS_OK
