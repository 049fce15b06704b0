"""ComplexF64 methods of the HIP path (dhqr_*_c64, through the C ABI) against the CPU oracle, the
complex golden fixtures and the reference's own ComplexF64 assertions.

Reference: the test suite runs every shape for T in (Float64, ComplexF64) (test/runtests.jl:43) and
its only known-answer test is the ComplexF64 partialdot (test/partialdot.jl:12-20).
Tolerances as in test_gpu_parity.py: |dH|, |dalpha| <= 1e-11 * max|H|, ||A-QR||_F/||A||_F < 1e-12,
and ||A^H A x - A^H b|| < 8 * (same for LAPACK QR).
"""
import glob
import os

import numpy as np
import pytest
import scipy.linalg as sl

pytestmark = pytest.mark.gpu


def TOL(H):
    """element-wise tolerance of a factorisation against the oracle's, relative to max|H|: 8 n eps (n = columns = number of
    dependent reflector steps; summation order differs from the reference's @simd, never bitwise).  Measured agreement is
    1e-15 ... 1e-14, so a wrong summation (a dropped term, a float accumulator) fails."""
    return 8.0 * max(int(H.shape[1]), 8) * np.finfo(np.float64).eps
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REF_SHAPES = [(110, 100), (220, 200), (440, 400), (880, 800), (1100, 1000), (2200, 2000), (4400, 4000)]


def test_partialdot_kat_reference_host(pkg):
    # test/partialdot.jl:12-20, through the entry point the Julia module binds (host vectors)
    rng = np.random.default_rng(0)
    for N in range(1, 21):
        a = rng.random(N) + 1j * rng.random(N)
        b = rng.random(N) + 1j * rng.random(N)
        for i in range(N):
            got = pkg.partialdot(a, b, i, N)
            assert got == pytest.approx(np.vdot(a[i:], b[i:]), rel=np.sqrt(np.finfo(float).eps))
    assert pkg.partialdot(a, b, 5, 5) == 0


def test_partialdot_device_long(pkg, orc):
    import torch
    n = 1_000_003
    a = orc.rand_vector_c(n, 5)
    b = orc.rand_vector_c(n, 6) - (0.5 + 0.5j)
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    for lo, hi in ((0, n), (17, n - 5), (n - 1, n)):
        got = pkg.partialdot(ta, tb, lo, hi)
        want = np.vdot(a[lo:hi], b[lo:hi])
        assert abs(got - want) <= 1e-12 * np.sum(np.abs(a[lo:hi]) * np.abs(b[lo:hi]))


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "zqr_*.npz"))))
def test_golden_fixtures_complex_host_dropin(pkg, orc, path):
    """qr!(A::Matrix{ComplexF64}) / `\\` with host arrays (dhqr_qr_c64 / dhqr_ldiv_c64)."""
    g = np.load(path)
    m, n, seed = int(g["m"]), int(g["n"]), int(g["seed"])
    A = orc.rand_matrix_c(m, n, seed)
    A0 = A.copy()
    H = pkg.qr_(A)
    assert H.A is A and H.α.dtype == np.complex128  # in place, complex alpha (src:306-309)
    scale = np.abs(g["H"]).max()
    assert np.abs(A - g["H"]).max() <= TOL(g["H"]) * scale
    assert np.abs(H.α - g["alpha"]).max() <= TOL(g["H"]) * scale
    b = orc.rand_vector_c(m, seed + 1)
    b0 = b.copy()
    x = pkg.ldiv(H, b)
    assert np.array_equal(b, b0), "H \\ b must not modify b (src:318)"
    assert np.abs(x - g["x"]).max() <= 1e-9 * np.abs(g["x"]).max()
    QR = orc.form_qr_c(np.asfortranarray(A), H.α)
    assert np.linalg.norm(A0 - QR) / np.linalg.norm(A0) < 1e-12


# every kernel variant: 256 / 512 / 1024 threads per column (<=1024, <=4096, >4096 rows), m == n, tiny
@pytest.mark.parametrize("m,n", [(5, 3), (2, 2), (64, 64), (111, 100), (1500, 40), (5000, 24), (9001, 12)])
def test_device_path_vs_oracle(pkg, orc, m, n):
    import torch
    A = pkg.rand_colmajor_c(m, n, 3, "cuda:0")
    A0 = A.cpu().numpy().copy()
    assert np.array_equal(A0, orc.rand_matrix_c(m, n, 3))  # same generator on device and host
    H = pkg.qr_(A)
    torch.cuda.synchronize()
    Ho, ao = orc.householder_c(A0)
    scale = np.abs(Ho).max()
    Hd, ad = H.A.cpu().numpy(), H.α.cpu().numpy()
    assert np.abs(Hd - Ho).max() <= TOL(Ho) * scale
    assert np.abs(ad - ao).max() <= TOL(Ho) * scale
    QR = orc.form_qr_c(np.asfortranarray(Hd), ad)
    assert np.linalg.norm(A0 - QR) / np.linalg.norm(A0) < 1e-12
    # solve on the device, b untouched
    b = torch.from_numpy(orc.rand_vector_c(m, 4)).cuda()
    b0 = b.clone()
    x = pkg.ldiv(H, b)
    assert torch.equal(b, b0)
    xo = orc.solve_c(Ho, ao, b0.cpu().numpy())
    assert np.abs(x.cpu().numpy() - xo).max() <= 1e-11 * max(1.0, np.linalg.cond(A0)) * np.abs(xo).max()


_ORACLE_STAT = {}  # (m, n, seed) -> (stdliberr, oracle_err): one O(m n^2) oracle run per draw, shared by the tests below


def _normal_residual(A, x, b):
    """the reference's statistic as written, in double: norm(A' * A * x .- A' * b) (test/runtests.jl:51,62)"""
    Ah = A.conj().T
    return float(np.linalg.norm(Ah @ (A @ x) - Ah @ b))


def _draw(orc, m, n, seed, with_oracle):
    """(A, b, stdliberr, oracle_err): x1 from LAPACK's QR as in test/runtests.jl:49; oracle_err = the same statistic for
    the ORACLE's x (the reference's algorithm restated on the CPU) -- what the reference itself would score on this draw"""
    A = orc.rand_matrix_c(m, n, seed)
    b = orc.rand_vector_c(m, seed + 1)
    key = (m, n, seed)
    if key not in _ORACLE_STAT or (with_oracle and _ORACLE_STAT[key][1] is None):
        q, r = np.linalg.qr(A)
        x1 = sl.solve_triangular(r, q.conj().T @ b)
        stdliberr = _normal_residual(A, x1, b)
        oerr = None
        if with_oracle:
            Ho, ao = orc.householder_c(A.copy(order="F"))
            oerr = _normal_residual(A, orc.solve_c(Ho, ao, b), b)
        _ORACLE_STAT[key] = (stdliberr, oerr)
    return (A, b) + _ORACLE_STAT[key]


def _gpu_statistic(pkg, A, b, nb):
    H = pkg.qr_(A.copy(order="F"), nb=nb)
    return _normal_residual(A, np.asarray(pkg.ldiv(H, b)), b)


def _check_against_reference_bound(pkg, orc, m, n, seed, nb):
    """test/runtests.jl:62 LITERALLY for one draw: norm(A' * A * x2 .- A' * b) < 8 stdliberr, residuals in double as written
    there -- asserted for every shape and draw since round 4.  Round 3 needed medians and an extended-precision evaluator
    here; what the measurements of this round showed (profiles/r04_c64_ratio_table_*.txt, the five draws of 4400 x 4000):
      * the statistic is a property of x, not evaluation noise (extended-precision evaluation moves it by a few per cent);
      * the bound divides by LAPACK's own statistic, which moves 9 x between draws, and the ORACLE -- the reference's algorithm
        restated on the CPU -- itself scores 1.8 ... 7.8 x LAPACK on the GPU box's host and 8.8 / 15.6 on another host
        (profiles/r04_c64_oracle_ratio_cpu_container.json): the reference does not meet its own bound on every draw;
      * with a plain-double solve the GPU paths scored like the oracle (0.2 ... 14 x LAPACK), and the ORACLE'S solve on the
        GPU's factor scored half of that: the O(mn) solve, not the O(mn^2) factorisation, dominated.  With b carried in
        double-double through Q'b and the back substitution (k_zqtb_col_dd, k_zbacksub_*_dd) the GPU paths score 0.09 ...
        1.2 x LAPACK on the same draws.
    For the two largest shapes the restated reference's score on the same draw is printed beside the GPU's, and the GPU's
    statistic must not exceed it."""
    big = n >= 2000
    A, b, stdliberr, oerr = _draw(orc, m, n, seed, with_oracle=big)
    gerr = _gpu_statistic(pkg, A, b, nb)
    ratio = gerr / stdliberr
    if big:
        print(f"{m}x{n} seed {seed} nb={nb}: GPU {gerr:.2e} = {ratio:.2f} x LAPACK, restated reference (oracle) {oerr:.2e} = "
              f"{oerr / stdliberr:.2f} x LAPACK (bound 8)")
        assert gerr < oerr, (gerr, oerr)
    else:
        print(f"{m}x{n} seed {seed} nb={nb}: ratio {ratio:.2f} (reference bound 8)")
    assert ratio < 8, ratio
    return ratio


@pytest.mark.parametrize("m,n", REF_SHAPES)
def test_reference_single_draw_seed0_default_path(pkg, orc, m, n):
    """test/runtests.jl:62 LITERALLY: one draw (seed 0), residuals in double, `< 8 stdliberr`, for the path qr_(A) takes
    by default (nb=None: blocked from n >= 256); on the two largest shapes with the restated reference's own score beside
    it (see _check_against_reference_bound)."""
    _check_against_reference_bound(pkg, orc, m, n, 0, None)


@pytest.mark.parametrize("seed", [0, 2, 4, 6, 8])
@pytest.mark.parametrize("nb", [0, 64])
def test_reference_single_draws_largest_shape(pkg, orc, seed, nb):
    """every draw of the reference's largest shape as its own literal `< 8` check, both paths (nb = 0: the reference's
    operation order; nb = 64: blocked).  Round 3 recorded two of these ten as xfail; none is left."""
    _check_against_reference_bound(pkg, orc, 4400, 4000, seed, nb)


@pytest.mark.parametrize("m,n", REF_SHAPES)
def test_reference_acceptance_inequality_complex(pkg, orc, m, n):
    """test/runtests.jl:42-63 with T = ComplexF64 and x from the GPU path in the reference's operation order (nb = 0;
    the blocked default has its own test below), residuals in double as the reference evaluates them.  Shapes with
    n >= 2000: three draws, each against the literal bound."""
    for seed in ((0, 2, 4) if n >= 2000 else (0,)):
        _check_against_reference_bound(pkg, orc, m, n, seed, 0)
    if n >= 2000:
        # largest shapes: pin the GPU factor (last seed) against LAPACK zgeqrf directly (rows of R equal up to the
        # unit phase of alpha_j, see tests/test_oracle_complex.py)
        A = orc.rand_matrix_c(m, n, 4)
        H = pkg.qr_(A.copy(order="F"), nb=0)
        (qr_raw, _tau), _ = sl.qr(A, mode="raw")
        R = np.triu(H.A, 1)[:n] + np.diag(H.α)
        Rl = np.triu(qr_raw)[:n]
        ph = np.diag(R) / np.diag(Rl)
        assert np.abs(np.abs(ph) - 1.0).max() < 1e-10
        assert np.abs(R - ph[:, None] * Rl).max() < 1e-10 * np.abs(R).max()
        v2 = (np.abs(np.tril(H.A)) ** 2).sum(axis=0)
        assert np.abs(v2 - 2.0).max() < 1e-12


@pytest.mark.parametrize("m,n", [(300, 200), (1100, 1000), (2100, 2048), (5000, 130)])
def test_blocked_complex_vs_oracle(pkg, orc, m, n):
    """nb = 64: panels by the unblocked complex kernels, trailing update by the FP64 MFMA kernels on the real embedding of
    the 64 complex reflectors (dhqr_factor_c64_nb): the reference's factorisation, element by element"""
    import torch
    A = pkg.rand_colmajor_c(m, n, 3, "cuda:0")
    A0 = A.cpu().numpy().copy()
    H = pkg.qr_(A, nb=64)
    torch.cuda.synchronize()
    Hd, ad = H.A.cpu().numpy(), H.α.cpu().numpy()
    Ho, ao = orc.householder_c(A0)  # the reference restatement at every size (2100 x 2048: ~1 min on the host cores)
    scale = np.abs(Ho).max()
    # two backward-stable orderings of the same factorisation agree element-wise to ~ kappa(A) * eps, not to eps: the
    # nearly square 2100 x 2048 case has kappa = 5.8e3 and differs by 6e-12 (H) / 2.3e-11 (alpha) in its LAST columns
    # while ||A - QR|| / ||A|| and the solution are equally accurate for both (profiles/r02_c64_blocked_accuracy.txt)
    kappa = np.linalg.cond(A0) if n > 1000 else 1.0
    tol = max(TOL(Ho), 64 * kappa * np.finfo(float).eps)
    assert np.abs(Hd - Ho).max() <= tol * scale, np.abs(Hd - Ho).max() / scale
    assert np.abs(ad - ao).max() <= tol * scale
    QR = orc.form_qr_c(np.asfortranarray(Hd), ad)
    assert np.linalg.norm(A0 - QR) / np.linalg.norm(A0) < 1e-12
    b = torch.from_numpy(orc.rand_vector_c(m, 4)).cuda()
    x = pkg.ldiv(H, b).cpu().numpy()
    xr = np.linalg.lstsq(A0, b.cpu().numpy(), rcond=None)[0]
    assert np.abs(x - xr).max() <= 1e-8 * np.abs(xr).max()


@pytest.mark.parametrize("ranks,m,n", [(1, 300, 200), (2, 300, 200), (3, 700, 650), (2, 1100, 1000), (8, 1500, 1030),
                                        (3, 9000, 200), (4, 64, 64), (5, 70, 3)])
def test_complex_column_split_logical_ranks_one_gpu(pkg, orc, ranks, m, n):
    """qr!(A; ndev) for ComplexF64 (dhqr_mg_qr_c64 -> zcs_factor, dhqr_zdist.h): cyclic blocks of 64 complex columns over
    `ranks` rank threads on cuda:0 (peer-copy transport), one broadcast per panel, look-ahead by the owner of the next
    panel; against the oracle element by element, then the reference's solve on the host-format result.  Shapes: fewer
    panels than ranks, a partial last panel, panels taller than the pipelined kernel's 8192 rows (no look-ahead there)."""
    mg = pkg.MultiGpuQR(devices=[0] * ranks)
    try:
        A0 = orc.rand_matrix_c(m, n, 3)
        A = np.asfortranarray(A0.copy())
        H, alpha = mg.qr_(A)
        assert H is A
        Ho, ao = orc.householder_c(A0)
        scale = np.abs(Ho).max()
        kappa = np.linalg.cond(A0) if n >= 1000 else 1.0
        tol = max(TOL(Ho), 64 * kappa * np.finfo(float).eps)
        assert np.abs(H - Ho).max() <= tol * scale, np.abs(H - Ho).max() / scale
        assert np.abs(alpha - ao).max() <= tol * scale
        QR = orc.form_qr_c(np.asfortranarray(H), alpha)
        assert np.linalg.norm(A0 - QR) / np.linalg.norm(A0) < 1e-12
        b = orc.rand_vector_c(m, 4)
        x = pkg.ldiv(pkg.DistributedHouseholderQRStruct(H, alpha), b)
        xr = np.linalg.lstsq(A0, b, rcond=None)[0]
        assert np.abs(np.asarray(x) - xr).max() <= 1e-8 * np.abs(xr).max()
        if ranks > 1 and n > 64:
            cnt = mg.comm_counters(0)
            assert cnt["n_bcast"] == (n + 63) // 64, cnt  # ONE broadcast per panel (src:141-143 fans out per column)
            # what travels: alpha (64 complex) + T, T' and status (2 * 128^2 + 128 + 16 doubles) + the factored panel as
            # rows x 64 COMPLEX -- not its real embedding (2 rows x 128), which every rank forms itself; the last panel
            # (applied to nothing) sends its alpha only
            tail = 2 * 128 * 128 + 128 + 16
            want = sum(8 * (128 + ((tail + 2 * (m - 64 * k) * 64) if 64 * (k + 1) < n else 0)) for k in range((n + 63) // 64))
            assert cnt["bytes_bcast"] == want, (cnt, want)
        # the handle's `\` for complex128: Q'b and the back substitution distributed over the same ranks (dhqr_mg_ldiv_c64
        # -> zcs_solve, src:226-282), against the oracle's solve; the host inputs are not modified
        Hk, bk = H.copy(), b.copy()
        xd = mg.ldiv(H, alpha, b)
        xo = orc.solve_c(Ho, ao, b)
        assert np.abs(xd - xo).max() <= 1e-9 * np.abs(xo).max(), np.abs(xd - xo).max() / np.abs(xo).max()
        assert np.array_equal(H, Hk) and np.array_equal(b, bk)
        if ranks > 1 and n > 64:
            c0 = mg.comm_counters(0)
            mg.ldiv(H, alpha, b)
            c1 = mg.comm_counters(0)
            npan = (n + 63) // 64
            assert c1["n_bcast"] - c0["n_bcast"] == 3 * npan and c1["n_allreduce"] - c0["n_allreduce"] == npan, (c0, c1)
        # a second factorisation on the same handle gives the same bits (buffers, events, mailboxes reused)
        A2 = np.asfortranarray(A0.copy())
        H2, alpha2 = mg.qr_(A2)
        assert np.array_equal(H2, H) and np.array_equal(alpha2, alpha)
    finally:
        mg.close()


def test_complex_darray_front_end_single_gpu(pkg, orc):
    """qr!(A::DArray{ComplexF64}) at world size 1 (dhqr_cs_qr_darray_c64: contiguous host block in / out, the cyclic layout
    and the SPMD driver inside) = the single-GPU blocked factorisation, bit for bit; several processes: tests/test_distributed_cpu.py"""
    import importlib
    D = importlib.import_module("dhqr_amd.distributed")
    api = importlib.import_module("dhqr_amd.api")
    comm = D.Communicator.from_torch(api.get_context(0))
    m, n = 700, 650
    A0 = orc.rand_matrix_c(m, n, 9)
    blk = A0.copy(order="F")
    alpha = pkg.qr_darray_c64_(blk, m, n, comm)
    H1 = A0.copy(order="F")
    H = pkg.qr_(H1, nb=64)
    assert np.array_equal(blk, np.asarray(H.A)) and np.array_equal(alpha, np.asarray(H.α))
    Ho, ao = orc.householder_c(A0)
    assert np.abs(blk - Ho).max() <= TOL(Ho) * np.abs(Ho).max()
    with pytest.raises(ValueError):
        pkg.qr_darray_c64_(np.zeros((m, n - 1), dtype=complex, order="F"), m, n, comm)
    # `qrA \ b` through dhqr_cs_ldiv_darray_c64 (world size 1) against the oracle
    b = orc.rand_vector_c(m, 10)
    x = pkg.ldiv_darray_(blk, m, n, alpha, b, comm)
    xo = orc.solve_c(Ho, ao, b)
    assert np.abs(x - xo).max() <= 1e-9 * np.abs(xo).max()
    comm.close()


@pytest.mark.parametrize("ranks,m,n", [(2, 300, 200), (3, 700, 650), (8, 1500, 1030), (3, 200, 3)])
def test_complex_darray_qr_and_ldiv_logical_ranks_one_gpu(pkg, orc, ranks, m, n):
    """test/runtests.jl:71-82 with T = ComplexF64: `qrA = qr!(A::DArray); x = qrA \\ b` through dhqr_cs_qr_darray_c64 and
    dhqr_cs_ldiv_darray_c64 with `ranks` rank threads sharing cuda:0 (callback transport), every rank holding its contiguous
    column block; factor and x against the complex oracle on every rank"""
    import ctypes
    from dist_helpers import gpu_thread_ranks
    A0 = orc.rand_matrix_c(m, n, 83)
    b = orc.rand_vector_c(m, 84)
    Ho, ao = orc.householder_c(A0)
    xo = orc.solve_c(Ho, ao, b)
    scale = np.abs(Ho).max()
    kappa = np.linalg.cond(A0) if n >= 1000 else 1.0
    tol = max(TOL(Ho), 64 * kappa * np.finfo(float).eps)

    def rank_fn(rank, comm, L):
        lo, hi = ctypes.c_int64(), ctypes.c_int64()
        L.dhqr_cs_contiguous_range(n, ranks, rank, ctypes.byref(lo), ctypes.byref(hi))
        lo, hi = lo.value, hi.value
        blk = np.array(A0[:, lo:hi], order="F")
        al = pkg.qr_darray_c64_(blk, m, n, comm)
        if hi > lo:
            assert np.abs(blk - Ho[:, lo:hi]).max() <= tol * scale
        assert np.abs(al - ao).max() <= tol * scale
        keep = blk.copy()
        x = pkg.ldiv_darray_(blk, m, n, al, b, comm)
        assert np.array_equal(blk, keep)
        assert np.abs(x - xo).max() <= 1e-9 * np.abs(xo).max()
        return x

    xs = gpu_thread_ranks(ranks, rank_fn)
    for r in range(1, ranks):
        assert np.array_equal(xs[r], xs[0])


@pytest.mark.parametrize("dtype", ["complex128", "float64"])
@pytest.mark.parametrize("ranks,m,n,seed", [(2, 440, 400, 0), (3, 1100, 1000, 0), (2, 4400, 4000, 0), (3, 4400, 4000, 8)])
def test_reference_distributed_acceptance(pkg, orc, dtype, ranks, m, n, seed):
    """test/runtests.jl:71-82, the reference's "distributed + threaded" check, LITERALLY: A3 = DArray of A over the workers
    (contiguous column blocks), qrA = qr!(A3), x3 = qrA \\ b, norm(A' * A * x3 .- A' * b) < 8 stdliberr -- for T in (Float64,
    ComplexF64) through the entry points a Julia worker binds (dhqr_cs_qr_darray_*, dhqr_cs_ldiv_darray_*), `ranks` rank
    threads sharing cuda:0 (callback transport)"""
    import ctypes
    from dist_helpers import gpu_thread_ranks
    cplx = dtype == "complex128"
    A = orc.rand_matrix_c(m, n, seed) if cplx else orc.rand_matrix(m, n, seed)
    b = orc.rand_vector_c(m, seed + 1) if cplx else orc.rand_vector(m, seed + 1)
    q, r = np.linalg.qr(A)
    x1 = sl.solve_triangular(r, q.conj().T @ b)
    stdliberr = _normal_residual(A, x1, b)

    def rank_fn(rank, comm, L):
        lo, hi = ctypes.c_int64(), ctypes.c_int64()
        L.dhqr_cs_contiguous_range(n, ranks, rank, ctypes.byref(lo), ctypes.byref(hi))
        blk = np.array(A[:, lo.value:hi.value], order="F")
        if cplx:
            al = pkg.qr_darray_c64_(blk, m, n, comm)
        else:
            al = np.zeros(n)
            rc = L.dhqr_cs_qr_darray_f64(comm.handle, blk.ctypes.data_as(ctypes.c_void_p), m, n, m, al.ctypes.data_as(ctypes.c_void_p))
            assert rc == 0, L.dhqr_last_error()
        return pkg.ldiv_darray_(blk, m, n, al, b, comm)

    xs = gpu_thread_ranks(ranks, rank_fn)
    ratio = _normal_residual(A, xs[0], b) / stdliberr
    print(f"distributed {dtype} {m}x{n} over {ranks} ranks, seed {seed}: ratio {ratio:.2f} (reference bound 8)")
    assert ratio < 8, ratio


@pytest.mark.parametrize("m,n", REF_SHAPES)
def test_reference_acceptance_inequality_complex_blocked(pkg, orc, m, n):
    """test/runtests.jl:42-63 with T = ComplexF64 through the BLOCKED path (host drop-in, nb = 64): the literal bound,
    residuals in DOUBLE as the reference evaluates them (round 3 needed an extended-precision evaluator and a median here).
    Shapes with n >= 2000: five draws."""
    for seed in ((0, 2, 4, 6, 8) if n >= 2000 else (0,)):
        _check_against_reference_bound(pkg, orc, m, n, seed, 64)


def test_zero_pivot_complex(pkg, orc):
    # angle(0) == 0 => alpha = -||a|| and a proper reflection (src:9); the Real method gives alpha = 0
    A = np.asfortranarray(np.array([[0.0, 1.0], [3.0j, 2.0], [4.0, 5.0j]], dtype=complex))
    A0 = A.copy()
    H = pkg.qr_(A)
    assert H.α[0] == pytest.approx(-5.0)
    Ho, ao = orc.householder_c(A0)
    assert np.abs(A - Ho).max() < 1e-14 and np.abs(H.α - ao).max() < 1e-14


def test_complex_argument_errors(pkg):
    A = np.asfortranarray(np.ones((8, 4), dtype=complex))
    with pytest.raises(ValueError):
        pkg.qr_(A, nb=128)  # ComplexF64 panels are 64 wide (128 real columns)
    with pytest.raises(pkg.DHQRError):
        pkg.qr_(np.asfortranarray(np.ones((3, 5), dtype=complex)))  # m < n
