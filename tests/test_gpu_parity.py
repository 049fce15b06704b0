"""Parity of the HIP path (through the C ABI) with the CPU oracle on the same seeded inputs.

Tolerances: the reference accumulates with @simd (order unspecified, src:45); the GPU uses
wavefront trees / MFMA chains, so element-wise agreement is to rounding, not bitwise:
  |dH|, |dalpha| <= 8 n eps * max|H|  (TOL below: 1.8e-13 at n = 100, 7e-12 at n = 4000),  ||A-QR||_F/||A||_F < 1e-12
(the north-star tolerance), and the reference's own acceptance inequality
  ||A'A x - A'b|| < 8 * (same for LAPACK QR)      (test/runtests.jl:61-63).
"""
import glob
import os

import ctypes

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


def _factor_dev(pkg, m, n, seed, nb):
    import torch
    A = pkg.rand_colmajor(m, n, seed, "cuda:0")
    A0 = A.clone()
    H = pkg.qr_(A, nb=nb)
    torch.cuda.synchronize()
    return H, A0


@pytest.mark.parametrize("nb", [0, 128])
@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "qr_*.npz"))))
def test_golden_fixtures(pkg, path, nb):
    g = np.load(path)
    m, n, seed = int(g["m"]), int(g["n"]), int(g["seed"])
    H, A0 = _factor_dev(pkg, m, n, seed, nb)
    scale = np.abs(g["H"]).max()
    assert np.abs(H.A.cpu().numpy() - g["H"]).max() <= TOL(g["H"]) * scale
    assert np.abs(H.α.cpu().numpy() - g["alpha"]).max() <= TOL(g["H"]) * scale
    b = pkg.rand_vector_device(m, seed + 1, "cuda:0")
    b0 = b.clone()
    x = pkg.ldiv(H, b)
    assert (b == b0).all(), "H \\ b must not modify b (src:318)"
    assert np.abs(x.cpu().numpy() - g["x"]).max() <= 1e-9 * np.abs(g["x"]).max()
    assert pkg.residual(H, A0) < 1e-12


# every kernel variant of the unblocked path: register-resident (<=512 ... <=8192 rows), the
# two-pass tall kernel (>8192 rows), even/odd m (16-byte vs scalar loads), m == n
@pytest.mark.parametrize("m,n", [(5, 3), (64, 64), (111, 100), (500, 40), (1000, 64), (2000, 48), (4000, 32),
                                 (8192, 24), (9001, 16), (20000, 12), (33, 33),
                                 # several reflectors per pass (k_rankk_fused): down the workgroup-size ladder, and
                                 # continuing from the tall-column kernels once a column fits (even / odd m)
                                 (1030, 600), (8210, 48), (8201, 30),
                                 # columns of 8192 < rows <= 16384 (k_rankk_tall: 512 threads x 24 / 32 elements, every
                                 # reflector streamed), the hand-over from the one-reflector kernels above 16384 rows
                                 # and to k_rankk_fused at 8192, even / odd m
                                 (12288, 64), (16390, 48), (16384, 33), (12001, 37), (9000, 40),
                                 # columns of 16384 < rows <= 32768 (k_rankk_xtall: one column in a workgroup's registers,
                                 # 512 threads x 48 / 64 elements, reflectors streamed twice per step), even / odd m, the
                                 # hand-over from the one-reflector kernels above 32768 rows and to k_rankk_tall at 16384
                                 (32768, 24), (20000, 64), (24577, 21), (32790, 40), (16400, 36),
                                 # columns above 32768 rows: one reflector per launch until a column fits a K-pass kernel
                                 (40000, 24), (65536, 16)])
def test_unblocked_vs_oracle(pkg, orc, m, n):
    H, A0 = _factor_dev(pkg, m, n, 3, 0)
    Ho, ao = orc.householder(orc.rand_matrix(m, n, 3))
    scale = np.abs(Ho).max()
    assert np.abs(H.A.cpu().numpy() - Ho).max() <= TOL(Ho) * scale
    assert np.abs(H.α.cpu().numpy() - ao).max() <= TOL(Ho) * scale
    assert pkg.residual(H, A0) < 1e-12


@pytest.mark.parametrize("m,n", [(8000, 64), (8192, 30), (7000, 40), (6200, 33), (6144, 40), (4000, 60), (3000, 50), (1000, 120), (7001, 35), (8190, 8190)])
def test_unblocked_more_than_five_reflectors_per_pass(pkg, orc, m, n, monkeypatch):
    """Columns of at most 8192 rows take as many reflectors per pass as the CU can hold (six at 6145 ... 8192 rows through the
    16-elements-per-thread instantiations, 16-byte path only; six / seven / eight at <= 6144 / 4096 / 3072 rows) while more
    than 4096 columns are left -- here from the first column on (DHQR_TUNE rankk_max_min_cols=0)"""
    if n > 1000:
        monkeypatch.delenv("DHQR_TUNE", raising=False)  # the shipped threshold on a square matrix
    else:
        monkeypatch.setenv("DHQR_TUNE", "rankk_max_min_cols=0")
    api = pkg.api
    old = api._contexts.pop(0, None)
    try:
        H, A0 = _factor_dev(pkg, m, n, 5, 0)
        if n <= 1000:
            Ho, ao = orc.householder(orc.rand_matrix(m, n, 5))
            scale = np.abs(Ho).max()
            assert np.abs(H.A.cpu().numpy() - Ho).max() <= TOL(Ho) * scale
            assert np.abs(H.α.cpu().numpy() - ao).max() <= TOL(Ho) * scale
        assert pkg.residual(H, A0) < 1e-12
    finally:
        api._contexts.pop(0, None)
        if old is not None:
            api._contexts[0] = old


@pytest.mark.parametrize("K", [1, 2, 3, 4, 5])
@pytest.mark.parametrize("m,n", [(2100, 300), (8203, 41), (517, 517), (8192, 40), (5000, 64), (12290, 23), (20010, 19)])
def test_unblocked_reflectors_per_pass(pkg, orc, m, n, K, monkeypatch):
    """DHQR_RANKK = 1..5 reflectors per pass over the trailing columns: the same factorisation as the oracle's"""
    monkeypatch.setenv("DHQR_RANKK", str(K))  # read by dhqr_create
    api = pkg.api
    old = api._contexts.pop(0, None)
    try:
        H, A0 = _factor_dev(pkg, m, n, 5, 0)
        Ho, ao = orc.householder(orc.rand_matrix(m, n, 5))
        scale = np.abs(Ho).max()
        assert np.abs(H.A.cpu().numpy() - Ho).max() <= TOL(Ho) * scale
        assert np.abs(H.α.cpu().numpy() - ao).max() <= TOL(Ho) * scale
        assert pkg.residual(H, A0) < 1e-12
    finally:
        api._contexts.pop(0, None)
        if old is not None:
            api._contexts[0] = old


@pytest.mark.parametrize("m,n", [(12288, 64), (16390, 48), (9000, 40), (8192, 40), (8000, 60), (6000, 60), (20000, 50)])
def test_unblocked_few_workgroups_own_many_columns(pkg, orc, m, n, monkeypatch):
    """DHQR_TUNE rankk_wgs=8: three bulk workgroups walk through all trailing columns of a pass (on 256 CUs the shapes above give
    every workgroup ONE column) -- the column-to-column pipeline of k_rankk_tall (next reflector streaming into LDS while the
    current one is applied, next column prefetched) and of k_rankk_fused against the oracle"""
    monkeypatch.setenv("DHQR_TUNE", "rankk_wgs=8,rankk_max_min_cols=0")  # six to eight per pass at <= 8192 rows whatever the number of columns
    api = pkg.api
    old = api._contexts.pop(0, None)
    try:
        H, A0 = _factor_dev(pkg, m, n, 9, 0)
        Ho, ao = orc.householder(orc.rand_matrix(m, n, 9))
        scale = np.abs(Ho).max()
        assert np.abs(H.A.cpu().numpy() - Ho).max() <= TOL(Ho) * scale
        assert np.abs(H.α.cpu().numpy() - ao).max() <= TOL(Ho) * scale
        assert pkg.residual(H, A0) < 1e-12
    finally:
        api._contexts.pop(0, None)
        if old is not None:
            api._contexts[0] = old


@pytest.mark.parametrize("pipe", [0, 2])
@pytest.mark.parametrize("m,n", [(2100, 300), (8192, 40), (5000, 64), (300, 40), (4097, 33)])
def test_unblocked_lead_one_workgroup_or_pipelined(pkg, orc, m, n, pipe, monkeypatch):
    """DHQR_RANKK_PIPE = 0 (the lead of a pass is ONE workgroup) / 2 (always K workgroups handing their reflectors on through
    flags; the default, 1, picks per launch): the same factorisation as the oracle's either way"""
    monkeypatch.setenv("DHQR_RANKK_PIPE", str(pipe))  # read by dhqr_create
    api = pkg.api
    old = api._contexts.pop(0, None)
    try:
        H, A0 = _factor_dev(pkg, m, n, 7, 0)
        Ho, ao = orc.householder(orc.rand_matrix(m, n, 7))
        scale = np.abs(Ho).max()
        assert np.abs(H.A.cpu().numpy() - Ho).max() <= TOL(Ho) * scale
        assert np.abs(H.α.cpu().numpy() - ao).max() <= TOL(Ho) * scale
        assert pkg.residual(H, A0) < 1e-12
    finally:
        api._contexts.pop(0, None)
        if old is not None:
            api._contexts[0] = old


@pytest.mark.parametrize("m,n", [(2000, 512), (8190, 384), (9000, 256)])
def test_blocked_with_panels_through_the_column_kernels(pkg, orc, m, n, monkeypatch):
    """DHQR_PANEL=1: every 128-column panel is factored by the unblocked column kernels (k_rankk_fused passes for panels of
    at most 8192 rows, the one-reflector kernels above) and packed into the block reflector -- same factorisation"""
    monkeypatch.setenv("DHQR_PANEL", "1")  # read by dhqr_create
    api = pkg.api
    old = api._contexts.pop(0, None)
    try:
        H, A0 = _factor_dev(pkg, m, n, 6, 128)
        Ho, ao = orc.householder(orc.rand_matrix(m, n, 6))
        scale = np.abs(Ho).max()
        assert np.abs(H.A.cpu().numpy() - Ho).max() <= TOL(Ho) * scale
        assert np.abs(H.α.cpu().numpy() - ao).max() <= TOL(Ho) * scale
        assert pkg.residual(H, A0) < 1e-12
    finally:
        api._contexts.pop(0, None)
        if old is not None:
            api._contexts[0] = old


@pytest.mark.parametrize("m,n", [(128, 128), (129, 129), (300, 128), (300, 200), (1000, 999), (2050, 1030),
                                 (1153, 600), (9000, 300),
                                 # n % 128 == 0 with >= 4 panels: the two-panel (K = 256) wide-update driver,
                                 # even / odd panel counts, square and tall
                                 (512, 512), (640, 640), (1536, 1024), (2048, 2048), (3000, 1152), (1281, 896)])
def test_blocked_vs_oracle(pkg, orc, m, n):
    H, A0 = _factor_dev(pkg, m, n, 4, 128)
    Ho, ao = orc.householder(orc.rand_matrix(m, n, 4))
    scale = np.abs(Ho).max()
    assert np.abs(H.A.cpu().numpy() - Ho).max() <= TOL(Ho) * scale
    assert np.abs(H.α.cpu().numpy() - ao).max() <= TOL(Ho) * scale
    assert pkg.residual(H, A0) < 1e-12


@pytest.mark.parametrize("nb", [0, 128])
@pytest.mark.parametrize("m,n", REF_SHAPES)
def test_reference_acceptance_inequality(pkg, orc, m, n, nb):
    """The reference's only assertion (test/runtests.jl:42-63) with x from the GPU path."""
    A = orc.rand_matrix(m, n, 0)
    b = orc.rand_vector(m, 1)
    q, r = np.linalg.qr(A)
    x1 = sl.solve_triangular(r, q.T @ b)
    stdliberr = np.linalg.norm(A.T @ (A @ x1) - A.T @ b)
    H, _ = _factor_dev(pkg, m, n, 0, nb)
    x = pkg.ldiv(H, pkg.rand_vector_device(m, 1, "cuda:0")).cpu().numpy()
    assert np.linalg.norm(A.T @ (A @ x) - A.T @ b) < 8 * stdliberr
    xo = orc.solve(*orc.householder(A), b)
    assert np.abs(x - xo).max() <= 1e-8 * np.abs(xo).max()


def test_host_in_host_out_drop_in(pkg, orc):
    """qr!(A::Matrix) / H \\ b through the host entry points (dhqr_qr_f64 / dhqr_ldiv_f64)."""
    m, n = 440, 400
    A = orc.rand_matrix(m, n, 9)
    b = orc.rand_vector(m, 10)
    Ho, ao = orc.householder(A)
    for nb in (0, 128):
        Ah = A.copy(order="F")
        H = pkg.qr_(Ah, nb=nb)
        assert H.A is Ah  # in place like qr!
        assert np.abs(Ah - Ho).max() <= TOL(Ho) * np.abs(Ho).max()
        assert np.abs(H.α - ao).max() <= TOL(Ho) * np.abs(Ho).max()
        x = pkg.ldiv(H, b)
        assert np.abs(x - orc.solve(Ho, ao, b)).max() <= 1e-9 * np.abs(x).max()
    # row-major caller: still factored "in place"
    Ac = np.ascontiguousarray(A)
    pkg.qr_(Ac, nb=128)
    assert np.abs(Ac - Ho).max() <= TOL(Ho) * np.abs(Ho).max()


def test_zero_pivot_matches_reference(pkg, orc):
    import torch
    A = np.asfortranarray(np.array([[0.0, 1.0], [3.0, 2.0], [4.0, 5.0]]))
    Ho, ao = orc.householder(A)
    Ah = A.copy(order="F")
    H = pkg.qr_(Ah, nb=0)
    assert np.allclose(Ah, Ho, atol=1e-15) and np.allclose(H.α, ao, atol=1e-15)


def test_error_paths(pkg):
    import torch
    A = pkg.empty_colmajor(4, 8, "cuda:0")  # m < n unsupported (reference yields NaN; we reject)
    with pytest.raises(pkg.DHQRError) as e:
        pkg.qr_(A)
    assert e.value.code == pkg._lib.EINVAL
    with pytest.raises(pkg.DHQRError):
        pkg.qr_(pkg.empty_colmajor(16, 8, "cuda:0"), nb=64)  # only nb in {0, 128}
    with pytest.raises(ValueError):
        pkg.qr_(torch.zeros((16, 8), dtype=torch.float64, device="cuda:0"))  # row-major tensor


@pytest.mark.parametrize("n,nb", [(8192, 0), (8192, 128), (16384, 128), (32768, 128)])
def test_full_size_properties(pkg, n, nb):
    """BASELINE-size checks (32768 = configs[2], the headline configuration) through size-independent properties:
    ||A-QR||/||A|| < 1e-12, ||v_j||^2 == 2 for every column, and Q'(Q b) == b."""
    import torch
    A = pkg.rand_colmajor(n, n, 0, "cuda:0")
    H = pkg.qr_(A, nb=nb)
    v2 = torch.cat([torch.tril(H.A[:, c:c + 4096], diagonal=-c).pow(2).sum(dim=0) for c in range(0, n, 4096)])
    assert (v2 - 2.0).abs().max().item() < 1e-11
    A0 = pkg.rand_colmajor(n, n, 0, "cuda:0")
    assert pkg.residual(H, A0) < 1e-12
    del A0
    b = pkg.rand_vector_device(n, 5, "cuda:0")
    B = b.clone().reshape(n, 1)
    pkg.apply_q_(H, B, trans=False)
    pkg.apply_q_(H, B, trans=True)
    assert (B.reshape(-1) - b).abs().max().item() < 1e-11


@pytest.mark.parametrize("m,n", [(700, 520), (1300, 1290), (300, 128), (700, 384)])
def test_column_cyclic_driver_single_rank(pkg, orc, m, n):
    """The SPMD multi-GPU entry points (dhqr_cs_*) at world size 1 (no process group): same driver code the
    N-GPU bench runs, checked against the oracle incl. residual and solve.  b arrives as a DEVICE tensor: its copy into
    the work vector runs on torch's stream and must be ordered before the library reads it (the front-end makes torch's
    current stream the library's caller stream; without that the solve read a half-written vector)."""
    import torch
    q = pkg.ColumnCyclicQR(m, n)
    q.fill(8)
    q.factor()
    H, alpha = q.local_numpy()
    Ho, ao = orc.householder(orc.rand_matrix(m, n, 8))
    scale = np.abs(Ho).max()
    assert np.abs(H - Ho).max() <= TOL(Ho) * scale
    assert np.abs(alpha - ao).max() <= TOL(Ho) * scale
    assert q.residual(8) < 1e-12
    b = orc.rand_vector(m, 9)
    x = q.solve(torch.tensor(b, device="cuda:0")).cpu().numpy()
    xo = orc.solve(Ho, ao, b)
    assert np.abs(x - xo).max() <= 1e-9 * np.abs(xo).max()


@pytest.mark.parametrize("env", [{}, {"DHQR_LANE_SIDE": "0"}, {"DHQR_NN_SPLIT": "1"}])
def test_lane_schedule_switches_on_a_small_matrix_with_quads(pkg, orc, monkeypatch, env):
    """quad steps forced onto a 2600 x 2560 matrix (five quads of pairs, heads, the K = 512 cross term; narrow V'C products
    through slot-sized k_gemm_tn workgroups with the two reflector blocks in blockIdx.z), with and without the lane's side
    stream and the row chunks of the wide subtraction launches -- every variant is the oracle's factorisation.  (Round 4's
    losing variants -- whole-CU narrow products, the early head, the head folded into the wide launches -- are deleted.)"""
    monkeypatch.setenv("DHQR_QUAD_MIN_COLS", "0")  # read by dhqr_create of the rank context
    monkeypatch.setenv("DHQR_PAIR_MIN_N", "0")
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    m, n = 2600, 2560
    mg = pkg.MultiGpuQR(devices=[0])
    try:
        mg.alloc(m, n).fill(21)
        mg.factor()
        H, alpha = mg.download()
        assert mg.residual(21) < 1e-12
        Ho, ao = orc.householder(orc.rand_matrix(m, n, 21))
        scale = np.abs(Ho).max()
        assert np.abs(H - Ho).max() <= TOL(Ho) * scale and np.abs(alpha - ao).max() <= TOL(Ho) * scale
    finally:
        mg.close()


@pytest.mark.parametrize("min_tiles", [3, 1000000])
def test_wide_tn_split_model_on_a_small_matrix(pkg, orc, monkeypatch, min_tiles):
    """the decomposition of the wide k_gemm_tn2 launches (normally for >= 32 column tiles = matrices beyond 4096
    columns) forced onto a 2304^2 matrix -- stream-K (tile-major fine units, a contiguous range per workgroup,
    k_reduce_pieces; DHQR_TUNE tn_min_tiles=3) and the column-tile x row-slab units every smaller launch uses
    (tn_min_tiles beyond reach): same factorisation as the oracle's"""
    monkeypatch.setenv("DHQR_TUNE", f"tn_min_tiles={min_tiles}")  # read by dhqr_create of the rank context
    m = n = 2304
    mg = pkg.MultiGpuQR(devices=[0])
    try:
        mg.alloc(m, n).fill(15)
        mg.factor()
        H, alpha = mg.download()
        assert mg.residual(15) < 1e-12
        Ho, ao = orc.householder(orc.rand_matrix(m, n, 15))
        scale = np.abs(Ho).max()
        assert np.abs(H - Ho).max() <= TOL(Ho) * scale and np.abs(alpha - ao).max() <= TOL(Ho) * scale
    finally:
        mg.close()


@pytest.mark.parametrize("ranks,m,n", [(2, 1500, 1300), (2, 2304, 2304), (3, 2000, 1700), (8, 4096, 4096)])
def test_multi_device_handle_logical_ranks_one_gpu(pkg, orc, ranks, m, n):
    """dhqr_mg_* with `ranks` rank threads all on cuda:0 (RCCL cannot put two ranks on one device, so the
    in-process peer-copy transport carries the panel broadcasts): the real streams / events / look-ahead /
    two-panel updates / device-side verification of the N-GPU driver on a 1-GPU box, against the oracle."""
    mg = pkg.MultiGpuQR(devices=[0] * ranks)
    try:
        assert mg.transport == "local-peer-copy"
        mg.alloc(m, n).fill(11)
        A0, _ = mg.download()
        Ah = orc.rand_matrix(m, n, 11)
        assert np.array_equal(A0, Ah)
        mg.factor()
        H, alpha = mg.download()
        res = mg.residual(11)
        assert res < 1e-12, res
        if m <= 2304:
            Ho, ao = orc.householder(Ah)
            scale = np.abs(Ho).max()
            assert np.abs(H - Ho).max() <= TOL(Ho) * scale, np.abs(H - Ho).max()
            assert np.abs(alpha - ao).max() <= TOL(Ho) * scale
            b = orc.rand_vector(m, 12)
            x = mg.solve(b)
            xo = orc.solve(Ho, ao, b)
            assert np.abs(x - xo).max() <= 1e-9 * np.abs(xo).max()
        st = [mg.stats(r) for r in range(ranks)]
        assert sum(s_["panels_fallback"] for s_ in st) == 0
        assert sum(s_["panels_fast"] for s_ in st) >= n // 128 - 1
        # a second factorisation on the same handle (buffers, events and mailboxes are reused)
        mg.fill(12).factor()
        assert mg.residual(12) < 1e-12
    finally:
        mg.close()


@pytest.mark.parametrize("rung", [1, 0])
def test_multi_device_host_drop_in_and_rejected_panel_gpu(pkg, orc, rung, monkeypatch):
    """qr!(A; ndev) host-in / host-out on 2 logical ranks, with a panel the device-side verification must reject: it
    is redone by TSQR-HR (DHQR_TSQR_RUNG=1: no column-by-column fallback) or by the column kernels (default)"""
    if rung:
        monkeypatch.setenv("DHQR_TSQR_RUNG", "1")  # read by dhqr_create of the rank contexts
    m, n = 1100, 768
    A0 = orc.rand_matrix(m, n, 22)
    A0[:, 300] = A0[:, 299] * (1.0 + 1e-9)   # second panel of the second pair
    mg = pkg.MultiGpuQR(devices=[0, 0])
    try:
        A = A0.copy(order="F")
        A, al = mg.qr_(A)
        QR = orc.form_qr(np.asfortranarray(A), al)
        assert np.linalg.norm(A0 - QR) / np.linalg.norm(A0) < 1e-13
        nfb = sum(mg.stats(r)["panels_fallback"] for r in range(2))
        assert (nfb == 0) if rung else (nfb >= 1)
        A1 = orc.rand_matrix(900, 520, 23)
        Ho, ao = orc.householder(A1)
        F, al = mg.qr_(A1.copy(order="F"))
        assert np.abs(F - Ho).max() <= TOL(Ho) * np.abs(Ho).max()
        b = orc.rand_vector(900, 24)
        x = mg.ldiv(F, al, b)
        xo = orc.solve(Ho, ao, b)
        assert np.abs(x - xo).max() <= 1e-9 * np.abs(xo).max()
    finally:
        mg.close()


def _rccl_one_rank(rank, P, m, n):
    """RCCL bootstrap exactly as a multi-process job does it (unique id -> ncclCommInitRank) on the one GPU of the
    box: world size 1, so the collectives are trivial, but librccl.so is dlopen()ed and the communicator is real"""
    import ctypes
    import torch
    import __graft_entry__ as g
    from oracle import dhqr_oracle as orc
    torch.cuda.set_device(0)
    pkg = g.import_package()
    L = pkg._lib.lib()
    idbuf = (ctypes.c_char * 128)()
    pkg._lib.check(L.dhqr_comm_unique_id(idbuf))
    assert any(idbuf.raw)
    comm = pkg.Communicator.from_torch(pkg.get_context(0))
    assert (comm.nranks, comm.rank) == (1, 0)
    q = pkg.ColumnCyclicQR(m, n, comm=comm)
    q.fill(3).factor()
    H, alpha = q.local_numpy()
    Ho, ao = orc.householder(orc.rand_matrix(m, n, 3))
    assert np.abs(H - Ho).max() <= TOL(Ho) * np.abs(Ho).max()
    return True


def test_rccl_loads_and_world_size_one_process_group():
    from dist_helpers import run_ranks
    run_ranks(_rccl_one_rank, 1, 700, 520)


def test_tall_skinny_single_gpu(pkg, orc):
    """tall panel shapes (many row chunks per panel step): 70000 x 200, blocked and unblocked"""
    m, n = 70000, 200
    Ho, ao = orc.householder(orc.rand_matrix(m, n, 13))
    for nb in (0, 128):
        H, A0 = _factor_dev(pkg, m, n, 13, nb)
        scale = np.abs(Ho).max()
        assert np.abs(H.A.cpu().numpy() - Ho).max() <= TOL(Ho) * scale
        assert np.abs(H.α.cpu().numpy() - ao).max() <= TOL(Ho) * scale
        assert pkg.residual(H, A0) < 1e-12


def test_tsqr_tree_r_factor_gpu(pkg):
    """csrc/dhqr_tsqr.h alone on the device: R of tall panels (odd leaf counts, a short last leaf) against LAPACK's
    R up to the sign of each row"""
    import ctypes
    import torch
    ctx = pkg.get_context(0)
    L = pkg._lib.lib()
    for rows in (200, 3000, 70000):
        Pd = pkg.rand_colmajor(rows, 128, 100 + rows, "cuda:0")
        R = pkg.empty_colmajor(128, 128, torch.device("cuda", 0))
        ctx.use_torch_stream()
        pkg._lib.check(L.dhqr_tsqr_r_f64(ctx.handle, ctypes.c_void_p(Pd.data_ptr()), rows, rows, ctypes.c_void_p(R.data_ptr())))
        torch.cuda.synchronize()
        Rg = R.cpu().numpy()
        Rl = np.linalg.qr(Pd.cpu().numpy(), mode="r")
        sg = np.sign(np.diag(Rg)) * np.sign(np.diag(Rl))
        assert np.abs(np.tril(Rg, -1)).max() == 0.0 and np.all(np.abs(sg) == 1.0)
        assert np.abs(sg[:, None] * Rg - Rl).max() <= 1e-12 * np.abs(Rl).max()


@pytest.mark.parametrize("m,n", [(1500, 640), (9000, 300), (33000, 256)])
def test_tsqr_hr_as_the_source_of_every_panel_gpu(pkg, orc, m, n):
    """dhqr_set_r_source(ctx, 3): every R-first panel goes through TSQR-HR (tree R, explicit Q, reflectors from Q);
    the reference's factorisation element by element, like the default path"""
    import torch
    ctx = pkg.get_context(0)
    ctx.set_r_source(3)
    try:
        ctx.reset_stats()
        H, A0 = _factor_dev(pkg, m, n, 31, 128)
        fast, fb = ctx.panel_counters()
        assert fb == 0 and fast >= 1 and ctx.tsqr_count() == fast
        Ho, ao = orc.householder(orc.rand_matrix(m, n, 31))
        scale = np.abs(Ho).max()
        assert np.abs(H.A.cpu().numpy() - Ho).max() <= TOL(Ho) * scale
        assert np.abs(H.α.cpu().numpy() - ao).max() <= TOL(Ho) * scale
        assert pkg.residual(H, A0) < 1e-12
    finally:
        ctx.set_r_source(1)


def test_fast_panel_path_is_used_and_falls_back(pkg, orc):
    """The R-first panel path must (a) actually run on well-conditioned input and (b) detect an
    ill-conditioned panel (two nearly dependent columns) and redo it -- by TSQR-HR at full accuracy, or, with that
    rung switched off, by the column-by-column kernels -- and still deliver a backward-stable factorisation."""
    import torch
    ctx = pkg.get_context(0)
    m, n = 1500, 640
    ctx.reset_stats()
    H, A0 = _factor_dev(pkg, m, n, 21, 128)
    fast, fb = ctx.panel_counters()
    assert fast >= 4 and fb == 0, (fast, fb)
    assert pkg.residual(H, A0) < 1e-12
    # near-dependent columns inside the second panel
    A = pkg.rand_colmajor(m, n, 22, "cuda:0")
    A[:, 200] = A[:, 199] * (1.0 + 1e-9)
    A0 = A.clone()
    Ah = A0.cpu().numpy()
    ctx.set_tsqr_rung(True)
    try:
        ctx.reset_stats()
        H = pkg.qr_(A.clone(), nb=128)
        torch.cuda.synchronize()
        fast, fb = ctx.panel_counters()
        assert fb == 0 and fast == 5 and ctx.tsqr_count() == 1, (fast, fb, ctx.tsqr_count())
        assert pkg.residual(H, A0) < 1e-13
        v2 = (torch.tril(H.A) ** 2).sum(dim=0)
        assert float((v2 - 2).abs().max()) < 1e-12
    finally:
        ctx.set_tsqr_rung(False)  # the single-GPU default: straight to the column kernels
    ctx.reset_stats()
    H = pkg.qr_(A, nb=128)
    torch.cuda.synchronize()
    fast, fb = ctx.panel_counters()
    assert fb >= 1 and ctx.tsqr_count() == 0, (fast, fb)
    assert pkg.residual(H, A0) < 1e-12
    Ho, ao = orc.householder(np.asfortranarray(Ah))
    # R agrees with the oracle where it is well determined (|R| not tiny)
    Rg = np.triu(H.A.cpu().numpy()[:n], 1) + np.diag(H.α.cpu().numpy())
    Ro = np.triu(Ho[:n], 1) + np.diag(ao)
    assert np.abs(np.abs(Rg[:199, :199]) - np.abs(Ro[:199, :199])).max() <= 1e-10 * np.abs(Ro).max()


@pytest.mark.parametrize("kind", ["randn", "graded", "rank_deficient"])
def test_robustness_across_input_classes(pkg, kind):
    """Backward stability must not depend on which panel path (fast / CholeskyQR2 retry / robust
    column-by-column) ends up being used."""
    import torch
    m, n = 2000, 768
    g = torch.Generator(device="cuda:0").manual_seed(77)
    X = torch.randn((n, m), generator=g, dtype=torch.float64, device="cuda:0")
    if kind == "graded":      # columns scaled over 10 orders of magnitude: kappa(panel) up to 1e10
        X *= torch.logspace(0, -10, n, dtype=torch.float64, device="cuda:0")[:, None]
    elif kind == "rank_deficient":   # exact rank 700 < n: trailing panels are numerically singular
        B = torch.randn((700, m), generator=g, dtype=torch.float64, device="cuda:0")
        C = torch.randn((n, 700), generator=g, dtype=torch.float64, device="cuda:0")
        X = C @ B
    A = X.t()                 # column-major (m x n) view
    A0 = A.clone()
    ctx = pkg.get_context(0)
    ctx.reset_stats()
    H = pkg.qr_(A, nb=128)
    torch.cuda.synchronize()
    fast, fb = ctx.panel_counters()
    res = pkg.residual(H, A0)
    print(f"\n{kind}: fast panels {fast}, fallback panels {fb}, residual {res:.2e}")
    assert torch.isfinite(H.A).all()
    assert res < 1e-12


@pytest.mark.parametrize("m,n", [(20000, 512), (8192, 1024), (4097, 300)])
def test_row_split_driver_single_rank(pkg, orc, m, n):
    """BASELINE configs[4] path (dhqr_rs_*, rows split over the ranks) at world size 1, incl. a partial last panel."""
    import torch
    q = pkg.RowSplitQR(m, n)
    q.fill(41)
    q.factor()
    H, alpha = q.local_numpy()
    Ho, ao = orc.householder(orc.rand_matrix(m, n, 41))
    scale = np.abs(Ho).max()
    assert np.abs(H - Ho).max() <= TOL(Ho) * scale
    assert np.abs(alpha - ao).max() <= TOL(Ho) * scale
    assert q.residual(41) < 1e-12
    b = orc.rand_vector(m, 42)
    x = q.solve(torch.tensor(b, device="cuda:0")).cpu().numpy()
    xo = orc.solve(Ho, ao, b)
    assert np.abs(x - xo).max() <= 1e-9 * np.abs(xo).max()


def test_row_split_full_size_properties(pkg):
    """BASELINE configs[4] at its FULL shape, 262144 x 4096, through the row-split driver at world size 1 (every all-reduce
    issued, moving nothing): size-independent properties -- ||A-QR||/||A|| < 1e-12 and ||v_j||^2 == 2 for every column"""
    import torch
    m, n = 262144, 4096
    q = pkg.RowSplitQR(m, n)
    q.fill(0)
    q.factor()
    torch.cuda.synchronize()
    v2 = torch.cat([torch.tril(q.A[:, c:c + 512], diagonal=-c).pow(2).sum(dim=0) for c in range(0, n, 512)])
    assert (v2 - 2.0).abs().max().item() < 1e-11
    assert q.residual(0) < 1e-12
    cnt = q.comm.counters()
    assert cnt["n_allreduce"] >= n // 128  # the cross-partition partial dots: at least a Gram matrix per panel


def test_column_split_full_size_eight_logical_ranks(pkg):
    """BASELINE configs[3] at its FULL size, 32768 x 32768 over 8 ranks -- here 8 rank threads sharing cuda:0 (peer-copy
    transport; every stream, event and mailbox of the 8-GPU program live): ||A-QR||/||A|| < 1e-12, every panel on the
    device-verified fast path, one broadcast per panel"""
    n = 32768
    mg = pkg.MultiGpuQR(devices=[0] * 8)
    try:
        mg.alloc(n, n).fill(0)
        c0 = mg.comm_counters(0)
        mg.factor()
        c1 = mg.comm_counters(0)
        assert n // 128 - 1 <= c1["n_bcast"] - c0["n_bcast"] <= n // 128  # ONE broadcast per panel (src:141-143: one per column and process)
        assert mg.residual(0) < 1e-12
        fast = sum(mg.stats(r)["panels_fast"] for r in range(8))
        fb = sum(mg.stats(r)["panels_fallback"] for r in range(8))
        assert fb == 0 and fast >= n // 128 - 2, (fast, fb)
    finally:
        mg.close()


def test_row_split_full_size_eight_logical_ranks(pkg):
    """BASELINE configs[4] at its FULL shape AND its partition: 262144 x 4096 with the rows split over 8 ranks (32768 rows
    each) -- here 8 rank threads sharing cuda:0 through dhqr_mg_rs_* (in-process transport; every channel, event and
    mailbox of the 8-GPU program live).  Size-independent properties: ||A-QR||/||A|| < 1e-12, no panel off the fast path,
    and the cross-partition partial dots really are all-reduced: per pair of panels one Gram all-reduce per panel, one for
    S = V'V / the cross term, one for the stacked V'C partial dots (DESIGN section 5 "Row split")"""
    m, n = 262144, 4096
    mg = pkg.MultiGpuQR(devices=[0] * 8)
    try:
        mg.rs_alloc(m, n).rs_fill(0)
        c0 = mg.comm_counters(0)
        mg.rs_factor()
        c1 = mg.comm_counters(0)
        n_ar = c1["n_allreduce"] - c0["n_allreduce"]
        assert n // 128 * 2 <= n_ar <= n // 128 * 5, n_ar  # >= a Gram and an S all-reduce per panel; no per-column fallback
        assert c1["bytes_allreduce"] - c0["bytes_allreduce"] >= (n // 128) * 128 * 128 * 8
        assert mg.rs_residual(0) < 1e-12
        assert sum(mg.stats(r)["panels_fallback"] for r in range(8)) == 0
    finally:
        mg.close()


@pytest.mark.parametrize("ranks,m,n,tsqr", [(2, 6000, 512, 0), (2, 16384, 1024, 0), (8, 16384, 2048, 0), (3, 3000, 1100, 0),
                                            (2, 6000, 512, 1), (8, 16384, 1024, 1), (3, 3000, 1100, 1)])
def test_row_split_logical_ranks_one_gpu(pkg, orc, ranks, m, n, tsqr, monkeypatch):
    """dhqr_mg_rs_* with `ranks` rank threads on cuda:0 (in-process transport): diagonal blocks move from rank to
    rank when n exceeds a rank's rows ((8, 16384, 2048): 2048 rows per rank), partial last panel ((3, 3000, 1100)).
    tsqr = 1: every panel through TSQR-HR across the ranks (local trees, gather, cross-rank tree up and down)."""
    if tsqr:
        monkeypatch.setenv("DHQR_TSQR", "1")  # read by dhqr_create of the rank contexts
    mg = pkg.MultiGpuQR(devices=[0] * ranks)
    try:
        mg.rs_alloc(m, n).rs_fill(43)
        A0, _ = mg.rs_download()
        Ah = orc.rand_matrix(m, n, 43)
        assert np.array_equal(A0, Ah)
        mg.rs_factor()
        H, alpha = mg.rs_download()
        Ho, ao = orc.householder(Ah)
        scale = np.abs(Ho).max()
        assert np.abs(H - Ho).max() <= TOL(Ho) * scale, np.abs(H - Ho).max()
        assert np.abs(alpha - ao).max() <= TOL(Ho) * scale
        assert mg.rs_residual(43) < 1e-12
        b = orc.rand_vector(m, 44)
        x = mg.rs_solve(b)
        xo = orc.solve(Ho, ao, b)
        assert np.abs(x - xo).max() <= 1e-9 * np.abs(xo).max()
        assert sum(mg.stats(r)["panels_fallback"] for r in range(ranks)) == 0
    finally:
        mg.close()


@pytest.mark.parametrize("rung", [1, 0])
def test_row_split_rejected_panel_gpu(pkg, orc, rung, monkeypatch):
    """a numerically rank-deficient panel: the fast path is refused on every rank (same all-reduced S); the panel is
    redone by TSQR-HR across the ranks (row-split default on P > 1) or column by column across the ranks
    (DHQR_TSQR_RUNG=0)"""
    if not rung:
        monkeypatch.setenv("DHQR_TSQR_RUNG", "0")
    m, n = 8192, 512
    A0 = orc.rand_matrix(m, n, 22)
    A0[:, 300] = A0[:, 299] * (1.0 + 1e-9)
    mg = pkg.MultiGpuQR(devices=[0, 0])
    try:
        mg.rs_alloc(m, n).rs_upload(A0).rs_factor()
        H, al = mg.rs_download()
        QR = orc.form_qr(np.asfortranarray(H), al)
        assert np.linalg.norm(A0 - QR) / np.linalg.norm(A0) < 1e-13
        st = mg.stats(0)
        assert (st["panels_fallback"] == 0 and st["panels_fast"] == 4) if rung else (st["panels_fallback"] >= 1)
    finally:
        mg.close()


@pytest.mark.parametrize("m,n", [(300, 128), (1000, 333)])
def test_explicit_q_and_r(pkg, orc, m, n):
    """get_r / get_q / apply_q_: R upper triangular with diag == alpha, Q'Q == I, Q R == A -- and
    element by element against the oracle: Q = the reference's reflectors applied to [I; 0] (orc.form_qr of the reflectors
    with R = I), apply_q_ on [R; 0] = the oracle's Q R"""
    import torch
    A = pkg.rand_colmajor(m, n, 9, "cuda:0")
    A0 = A.clone()
    H = pkg.qr_(A)
    R = pkg.get_r(H)
    Q = pkg.get_q(H)
    torch.cuda.synchronize()
    assert R.shape == (n, n) and Q.shape == (m, n)
    assert torch.equal(torch.triu(R), R) and torch.equal(R.diagonal(), H.α)
    eye = torch.eye(n, dtype=torch.float64, device="cuda:0")
    assert (Q.t() @ Q - eye).abs().max().item() < 1e-12
    assert ((Q @ R - A0).norm() / A0.norm()).item() < 1e-12
    Ho, ao = orc.householder(A0.cpu().numpy())
    Ro = np.triu(Ho, 1)[:n] + np.diag(ao)
    assert np.abs(R.cpu().numpy() - Ro).max() <= TOL(Ho) * np.abs(Ro).max()
    Qo = orc.form_qr(np.asfortranarray(np.tril(Ho)), np.ones(n))  # H_1 ... H_n [I; 0] in the reference's order
    assert np.abs(Q.cpu().numpy() - Qo).max() <= TOL(Ho)
    B = pkg.empty_colmajor(m, n, "cuda:0")
    B.zero_()
    B[:n] = R
    pkg.apply_q_(H, B, trans=False)  # Q [R; 0]
    QRo = orc.form_qr(Ho, ao)
    assert np.abs(B.cpu().numpy() - QRo).max() <= TOL(Ho) * np.abs(QRo).max()
    pkg.apply_q_(H, B, trans=True)   # Q' (Q [R; 0]) = [R; 0]
    Bn = B.cpu().numpy()
    assert np.abs(Bn[:n] - Ro).max() <= TOL(Ho) * np.abs(Ro).max() and np.abs(Bn[n:]).max() <= TOL(Ho) * np.abs(Ro).max()


def test_panel_kernels_keep_every_panel_on_the_fast_path(pkg, orc):
    """k_chol_inv / k_recon_top / k_build_t must give the reference's factorisation with every panel on the
    device-verified fast path (a silent fallback to the step kernels would hide a broken kernel)."""
    import ctypes
    import torch
    ctx = pkg.Context(0)
    try:
        for (m, n) in [(1536, 1024), (700, 384)]:
            A = pkg.rand_colmajor(m, n, 4, "cuda:0")
            alpha = torch.zeros(n, dtype=torch.float64, device="cuda:0")
            f0, b0 = ctx.panel_counters()
            ctx.use_torch_stream()
            pkg._lib.check(pkg._lib.lib().dhqr_factor_f64(ctx.handle, ctypes.c_void_p(A.data_ptr()), m, n, m,
                                                          ctypes.c_void_p(alpha.data_ptr()), 128))
            torch.cuda.synchronize()
            f1, b1 = ctx.panel_counters()
            assert (f1 - f0, b1 - b0) == (n // 128, 0)
            Ho, ao = orc.householder(orc.rand_matrix(m, n, 4))
            scale = np.abs(Ho).max()
            assert np.abs(A.cpu().numpy() - Ho).max() <= TOL(Ho) * scale
            assert np.abs(alpha.cpu().numpy() - ao).max() <= TOL(Ho) * scale
    finally:
        ctx.close()


def test_matrix_without_columns_is_a_noop(pkg):
    """qr!(zeros(m, 0)) returns an empty alpha and `\\` an empty x in the reference (its loops do not run,
    src:127, src:217); the C ABI treats n == 0 the same way for host and device, Float64 and ComplexF64."""
    import torch
    for dt in (np.float64, np.complex128):
        A = np.zeros((5, 0), dtype=dt, order="F")
        H = pkg.qr_(A)
        assert H.α.shape == (0,) and H.α.dtype == dt
        x = pkg.ldiv(H, np.ones(5, dtype=dt))
        assert x.shape == (0,)
    Ad = torch.zeros((0, 7), dtype=torch.float64, device="cuda:0").t()  # 7 x 0, column-major
    Hd = pkg.qr_(Ad)
    assert tuple(Hd.α.shape) == (0,)
    b = torch.ones(7, dtype=torch.float64, device="cuda:0")
    assert tuple(pkg.ldiv(Hd, b).shape) == (0,)


def test_darray_front_end_single_gpu(pkg, orc):
    """qr_darray_ (contiguous column blocks in / out, the reference's DArray layout) at world size 1; the
    multi-rank scatter/gather is covered under gloo in tests/test_distributed_cpu.py"""
    import torch
    m, n = 700, 520
    A = pkg.rand_colmajor(m, n, 61, "cuda:0")
    q, alpha = pkg.qr_darray_(A, n)
    torch.cuda.synchronize()
    Ho, ao = orc.householder(orc.rand_matrix(m, n, 61))
    scale = np.abs(Ho).max()
    assert np.abs(A.cpu().numpy() - Ho).max() <= TOL(Ho) * scale
    assert np.abs(alpha.cpu().numpy() - ao).max() <= TOL(Ho) * scale
    b = orc.rand_vector(m, 62)
    x = q.solve(torch.tensor(b, device="cuda:0")).cpu().numpy()
    xo = orc.solve(Ho, ao, b)
    assert np.abs(x - xo).max() <= 1e-9 * np.abs(xo).max()
    # `qrA \ b` through the one-call host entry point a Julia worker binds (dhqr_cs_ldiv_darray_f64)
    F, al = np.asfortranarray(A.cpu().numpy()), alpha.cpu().numpy()[:n].copy()
    x2 = pkg.ldiv_darray_(F, m, n, al, b, q.comm)
    assert np.abs(x2 - xo).max() <= 1e-9 * np.abs(xo).max()


@pytest.mark.parametrize("ranks,m,n", [(2, 700, 520), (3, 1100, 1000), (8, 1500, 1030), (3, 300, 2)])
def test_darray_qr_and_ldiv_logical_ranks_one_gpu(pkg, orc, ranks, m, n):
    """The reference's only distributed test (test/runtests.jl:71-82): `qrA = qr!(A::DArray); x = qrA \\ b`, through the two
    entry points a Julia worker binds -- dhqr_cs_qr_darray_f64 and dhqr_cs_ldiv_darray_f64 -- with `ranks` rank threads
    sharing cuda:0 (callback transport: device-to-device copies), every rank holding ITS contiguous column block
    (DistributedArrays' layout; ranks without columns included).  Factor and x against the oracle on every rank."""
    from dist_helpers import gpu_thread_ranks
    A0 = orc.rand_matrix(m, n, 81)
    b = orc.rand_vector(m, 82)
    Ho, ao = orc.householder(A0)
    xo = orc.solve(Ho, ao, b)
    scale = np.abs(Ho).max()

    def rank_fn(rank, comm, L):
        lo, hi = ctypes.c_int64(), ctypes.c_int64()
        L.dhqr_cs_contiguous_range(n, ranks, rank, ctypes.byref(lo), ctypes.byref(hi))
        lo, hi = lo.value, hi.value
        blk = np.array(A0[:, lo:hi], order="F") if hi > lo else np.zeros((m, 1), order="F")
        al = np.zeros(n)
        rc = L.dhqr_cs_qr_darray_f64(comm.handle, blk.ctypes.data_as(ctypes.c_void_p), m, n, m, al.ctypes.data_as(ctypes.c_void_p))
        assert rc == 0, L.dhqr_last_error()
        if hi > lo:
            assert np.abs(blk - Ho[:, lo:hi]).max() <= TOL(Ho) * scale
        assert np.abs(al - ao).max() <= TOL(Ho) * scale
        fb = np.array(blk[:, : hi - lo], order="F")
        keep = fb.copy()
        x = pkg.ldiv_darray_(fb, m, n, al, b, comm)
        assert np.array_equal(fb, keep)
        assert np.abs(x - xo).max() <= 1e-9 * np.abs(xo).max()
        return x

    xs = gpu_thread_ranks(ranks, rank_fn)
    for r in range(1, ranks):
        assert np.array_equal(xs[r], xs[0])  # x is replicated


# ---------------------------------------------------------------- BASELINE-size pins
@pytest.mark.parametrize("n,nb", [(8192, 0), (8192, 128), (16384, 128)])
def test_full_size_r_and_tau_pinned_against_lapack(pkg, n, nb):
    """R (strict upper part of H + alpha on the diagonal) and tau_j = v_jj^2 of a FULL-SIZE factorisation against
    LAPACK dgeqrf on the host cores (scipy): for real matrices the reference's factorisation equals LAPACK's with
    v_lapack = v / v_jj (for m == n LAPACK does not reflect the last column: tau_n = 0 and
    R[n,n] has the opposite sign).  Tolerance: 8 n eps * max|R| element-wise (1.5e-11 at 8192; measured 4e-15) -- the
    blocked MFMA path sums in a different order from dgeqrf and cond(A) ~ n for U[0,1) entries."""
    import scipy.linalg as sl
    import torch
    A = pkg.rand_colmajor(n, n, 0, "cuda:0")
    Ah = np.asfortranarray(A.cpu().numpy())
    H = pkg.qr_(A, nb=nb)
    torch.cuda.synchronize()
    Hh, al = H.A.cpu().numpy(), H.α.cpu().numpy()
    del A, H
    torch.cuda.empty_cache()
    (qr_l, tau), _ = sl.qr(Ah, mode="raw", overwrite_a=True, check_finite=False)
    scale = np.abs(np.diag(qr_l)).max()
    dR = np.abs(np.triu(Hh, 1) - np.triu(qr_l, 1)).max()
    dd = np.abs(al[:-1] - np.diag(qr_l)[:-1]).max()
    vjj = np.diag(Hh)
    dt = np.abs(vjj[:-1] ** 2 - tau[:-1]).max()
    print(f"n={n} nb={nb}: |dR|={dR / scale:.2e} |d diag|={dd / scale:.2e} |v_jj^2 - tau|={dt:.2e} (relative to max|R_jj|={scale:.1f})")
    tol = 8.0 * n * np.finfo(np.float64).eps
    assert dR <= tol * scale and dd <= tol * scale and dt <= tol
    assert abs(abs(al[-1]) - abs(qr_l[-1, -1])) <= 1e-8 * scale  # same magnitude, sign convention differs (see above)
    # v itself: v_lapack = v / v_jj on a sample of columns
    for j in (0, n // 3, n - 2):
        v = Hh[j:, j] / Hh[j, j]
        assert np.abs(v[1:] - qr_l[j + 1:, j]).max() <= 1e-9 * max(1.0, np.abs(qr_l[j + 1:, j]).max())


def test_unblocked_8192_elementwise_against_oracle(pkg, orc):
    """BASELINE configs[1] in full: every entry of H and alpha of the 8192 x 8192 unblocked factorisation against the
    oracle's restatement of src:122-148,198-213 run on the host cores (OpenMP over the trailing columns like @batch;
    ~0.7 TFLOP, tens of seconds).  Tolerance 8 n eps * max|H| = 1.5e-11 (8192 dependent reflectors, different summation
    order; measured 1e-14)."""
    import torch
    n = 8192
    A = pkg.rand_colmajor(n, n, 0, "cuda:0")
    Ah = np.asfortranarray(A.cpu().numpy())
    H = pkg.qr_(A, nb=0)
    torch.cuda.synchronize()
    Hh, al = H.A.cpu().numpy(), H.α.cpu().numpy()
    del A, H
    Ho, ao = orc.householder(Ah)
    scale = np.abs(Ho).max()
    eH, ea = np.abs(Hh - Ho).max() / scale, np.abs(al - ao).max() / scale
    print(f"8192^2 unblocked vs oracle: |dH|={eH:.2e} |dalpha|={ea:.2e}")
    assert eH <= TOL(Ho) and ea <= TOL(Ho)
