"""The WHOLE library (csrc/dhqr_api.hip + every kernel header, unmodified) compiled with the host clang++
against tests/simt/fake/hip/hip_runtime.h (fiber mode) and driven through the real C ABI on the CPU.

What this covers that the per-kernel emulation tests do not: the HOST logic -- unblocked / blocked /
look-ahead / two-panel drivers, panel dispatch (R-first fast path, CholeskyQR2 retry, column-by-column
fallback, partial last panel), workspace management, solve / apply-Q / residual pipelines, the ComplexF64
entry points, argument validation -- on tiny problems, against the oracle.  "Device" pointers are numpy
buffers (hipMalloc is malloc in the emulated runtime).  TEST INFRASTRUCTURE ONLY: the emulated library is
built into a temporary directory, the product package cannot load it and never falls back to it.
"""
import ctypes
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

P = ctypes.c_void_p


def _ptr(a):
    return a.ctypes.data_as(P)


@pytest.fixture(scope="module")
def emu(emulated_so):
    from dist_helpers import load_emulated_library
    return load_emulated_library(emulated_so)  # prototypes taken from the product binding (_lib.SIGNATURES)


_SLOW = pytest.mark.skipif(os.environ.get("DHQR_SLOW") != "1", reason="extra configuration; set DHQR_SLOW=1")


def _ctx(L, **env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        h = P()
        assert L.dhqr_create(ctypes.byref(h), 0) == 0, L.dhqr_last_error()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return h


def _counters(L, h):
    a, b = ctypes.c_int64(), ctypes.c_int64()
    assert L.dhqr_get_panel_counters(h, ctypes.byref(a), ctypes.byref(b)) == 0
    return a.value, b.value


def _factor(L, h, A0, nb):
    m, n = A0.shape
    A = A0.copy(order="F")
    al = np.zeros(n)
    rc = L.dhqr_factor_f64(h, _ptr(A), m, n, m, _ptr(al), nb)
    assert rc == 0, L.dhqr_last_error()
    return A, al


def _check(orc, A0, A, al, tol=1e-12):
    Ho, ao = orc.householder(A0)
    scale = np.abs(Ho).max()
    assert np.abs(A - Ho).max() <= tol * scale
    assert np.abs(al - ao).max() <= tol * scale
    return Ho, ao


# shapes: unblocked (fused register kernels); blocked with a partial last panel (robust column kernels for
# the 2-column panel); K = 2 (no look-ahead); K = 4 look-ahead driver
@pytest.mark.parametrize("m,n,nb", [(60, 20, 0), (201, 33, 0), (200, 130, 128), (300, 256, 128), (700, 512, 128)])
def test_factor_drivers_vs_oracle(emu, orc, m, n, nb):
    h = _ctx(emu)
    A0 = orc.rand_matrix(m, n, 3)
    A, al = _factor(emu, h, A0, nb)
    _check(orc, A0, A, al)
    if nb == 128:
        fast, fb = _counters(emu, h)
        # the R-first path takes full 128-column panels with at least 256 rows; none had to be redone
        assert fast == sum(1 for k in range(n // 128) if m - 128 * k >= 256) and fb == 0
    assert emu.dhqr_destroy(h) == 0


@pytest.mark.parametrize("m,n,Ks", [(200, 64, (2, 5)), (131, 37, (3,)), (70, 70, (4,)),
                                    (8300, 9, (3,)),  # columns of more than 8192 rows: k_rankk_tall, then k_rankk_fused
                                    (8000, 14, (5,)),  # six per pass at 6145 ... 8192 rows: 512 threads x 16 elements, four reflectors in registers
                                    (7000, 9, (5,)), (4000, 17, (5,)), (3000, 19, (5,)),  # 448 x 16 (six), 512 x 8 (seven), 384 x 8 (eight)
                                    (16420, 7, (3,)),  # more than 16384 rows: k_rankk_xtall (48 elements per thread), then k_rankk_tall
                                    (24590, 6, (2, 5))])  # k_rankk_xtall with 64 elements per thread
def test_unblocked_k_reflectors_per_pass_is_the_same_arithmetic(emu, orc, m, n, Ks):
    """nb = 0: k_rankk_fused applies K reflectors in one pass over every trailing column (1/K of the HBM traffic) --
    element by element the operations of K k_rank1_fused launches; only the summation order of the dot products differs
    (they are summed from the pass's first row, and the first column's norm by the lead workgroup instead of k_reflector)"""
    A0 = orc.rand_matrix(m, n, 14)
    res = {}
    for K in (1,) + Ks:
        # (DHQR_RANKK=5 also takes up to 8 per pass where the CU can hold them -- here from the first column on)
        h = _ctx(emu, DHQR_RANKK=K, DHQR_TUNE=f"rankk_wgs={2 + K % 2},rankk_max_min_cols=0")  # the lead + 7 or 11 persistent bulk workgroups of 256 threads
        A, al = _factor(emu, h, A0, 0)
        _check(orc, A0, A, al)
        res[K] = (A, al)
        emu.dhqr_destroy(h)
    scale = np.abs(res[1][0]).max()
    for K in Ks:
        assert np.abs(res[K][0] - res[1][0]).max() <= 1e-13 * scale and np.abs(res[K][1] - res[1][1]).max() <= 1e-13 * scale


def test_two_panel_driver_and_switches(emu, orc):
    """two-panel groups (K = 256 wide updates) engage for n >= DHQR_PAIR_MIN_N; DHQR_PAIR=0 keeps single-panel
    groups, DHQR_LOOKAHEAD=0 the simple host-verified loop -- same factorisation, every panel on the fast path"""
    A0 = orc.rand_matrix(640, 512, 4)
    for env in ({"DHQR_PAIR_MIN_N": 512}, {"DHQR_PAIR": 0}, {"DHQR_LOOKAHEAD": 0}):
        h = _ctx(emu, **env)
        A, al = _factor(emu, h, A0, 128)
        _check(orc, A0, A, al)
        assert _counters(emu, h) == (4, 0)
        emu.dhqr_destroy(h)




def test_quad_steps_two_pairs_in_one_k512_update(emu, orc):
    """1280 columns = 5 pairs with DHQR_QUAD_MIN_COLS=0: steps quad (0,1), quad (2,3), pair 4 -- the K = 512 update
    (k_gemm_nn_quad: row tiles above the second pair run half the K loop; 64-row tiles in the lane's narrow update, the
    streamed-C path for interior tiles), the 256 x 256 cross term, the head of a wide step for the quad's second pair --
    against the oracle, and the same factorisation as pairs only (DHQR_QUAD=0) to rounding"""
    A0 = orc.rand_matrix(1290, 1280, 8)
    res = []
    for env in ({"DHQR_PAIR_MIN_N": 512, "DHQR_QUAD_MIN_COLS": 0}, {"DHQR_PAIR_MIN_N": 512, "DHQR_QUAD": 0}):
        h = _ctx(emu, **env)
        A, al = _factor(emu, h, A0, 128)
        _check(orc, A0, A, al)
        assert _counters(emu, h) == (9, 0)  # the last panel (138 rows) is below the fast path's height
        emu.dhqr_destroy(h)
        res.append((A, al))
    scale = np.abs(res[1][0]).max()
    assert np.abs(res[0][0] - res[1][0]).max() <= 1e-12 * scale


def test_wide_subtraction_launches_in_chunks(emu, orc):
    """nn_row_chunks: a wide subtraction launch of few column tiles issued as several launches over ROW ranges (one row tile
    per chunk is enough here; the path of the row split and of a rank's local block).  Chunks regroup the same tiles; a
    chunk's last tile takes the edge path (C tile loaded up front instead of streamed in during the K loop: the same sum in
    another order), so equal to rounding"""
    A0 = orc.rand_matrix(650, 640, 12)
    res = []
    for env in ({"DHQR_PAIR_MIN_N": 0, "DHQR_QUAD_MIN_COLS": 0},
                {"DHQR_PAIR_MIN_N": 0, "DHQR_QUAD_MIN_COLS": 0, "DHQR_TUNE": "nn_chunk_tiles=1"},
                {"DHQR_PAIR_MIN_N": 0, "DHQR_QUAD": 0, "DHQR_TUNE": "nn_chunk_tiles=1", "DHQR_NN_SPLIT": 3}):
        h = _ctx(emu, **env)
        A, al = _factor(emu, h, A0, 128)
        _check(orc, A0, A, al)
        emu.dhqr_destroy(h)
        res.append(A)
    scale = np.abs(res[0]).max()
    assert np.abs(res[1] - res[0]).max() <= 1e-13 * scale and np.abs(res[2] - res[0]).max() <= 1e-12 * scale


@pytest.mark.parametrize("bad", [200, 300, 400])
def test_rejected_panel_inside_a_quad_step(emu, orc, bad):
    """640 columns: quad (a, b)(c, d) + one panel to its right.  A nearly dependent column pair in panel b / c / d is
    rejected on the device; the committed panels of the quad, which had only reached the blocks the lane needed, are
    applied to the rest before the run resumes (cs_factor) -- backward stable, every other panel on the fast path"""
    h = _ctx(emu, DHQR_PAIR_MIN_N=0, DHQR_QUAD_MIN_COLS=0)
    m, n = 700, 640
    A0 = orc.rand_matrix(m, n, 23)
    A0[:, bad] = A0[:, bad - 1] * (1.0 + 1e-9)
    A, al = _factor(emu, h, A0, 128)
    fast, fb = _counters(emu, h)
    assert fb >= 1 and fast + fb >= 4, (fast, fb)  # the last panel (188 rows) is below the fast path's height
    QR = orc.form_qr(np.asfortranarray(A), al)
    assert np.linalg.norm(A0 - QR) / np.linalg.norm(A0) < 1e-13
    emu.dhqr_destroy(h)


def test_wide_update_on_the_persistent_tn_kernel(emu, orc):
    """1024 columns: the first pair's wide update covers 4 column tiles (> 2), i.e. its V'C pass runs on the persistent
    k_gemm_tn2 (workgroups looping over their (column tile, row slab) units) and its subtraction on the 128 x 128-tile
    kernel with edge tiles in the last row tile -- against the oracle"""
    A0 = orc.rand_matrix(1030, 1024, 6)
    h = _ctx(emu)
    A, al = _factor(emu, h, A0, 128)
    _check(orc, A0, A, al)
    emu.dhqr_destroy(h)


@pytest.mark.parametrize("m,n", [(1040, 896), pytest.param(1161, 1152, marks=_SLOW)])
def test_wide_tn_stream_k(emu, orc, m, n):
    """stream-K decomposition of the wide k_gemm_tn2 launches (normally from 32 column tiles on; DHQR_TUNE tn_min_tiles=3
    brings it to a small matrix): 128-row fine units numbered tile-major, a contiguous range per workgroup, a tile's
    partial sums = the workgroups that share it (k_reduce_pieces) -- against the oracle, even and odd m (16-byte / scalar
    loads), and against the column-tile x row-slab units (tn_min_tiles beyond reach) to rounding"""
    A0 = orc.rand_matrix(m, n, 16)
    res = []
    for env in ({"DHQR_TUNE": "tn_min_tiles=3"}, {"DHQR_TUNE": "tn_min_tiles=1000000"}):
        h = _ctx(emu, **env)
        A, al = _factor(emu, h, A0, 128)
        _check(orc, A0, A, al)
        emu.dhqr_destroy(h)
        res.append(A)
    assert np.abs(res[0] - res[1]).max() <= 1e-12 * np.abs(res[1]).max()


@pytest.mark.parametrize("m,env", [(301, {}),                          # odd m: scalar (VEC = 1) loads everywhere
                                   (300, {"DHQR_PANEL": 2}),           # row-split step kernels for every panel
                                   pytest.param(300, {"DHQR_PANEL": 1}, marks=_SLOW),  # one workgroup per column
                                   (300, {"DHQR_CHOLQR_PASSES": 2})])  # CholeskyQR2 in the fast path
def test_panel_implementations_and_switches(emu, orc, m, env):
    h = _ctx(emu, **env)
    A0 = orc.rand_matrix(m, 256, 13)
    A, al = _factor(emu, h, A0, 128)
    _check(orc, A0, A, al)
    emu.dhqr_destroy(h)


def _tsqr_count(L, h):
    a = ctypes.c_int64()
    assert L.dhqr_get_tsqr_count(h, ctypes.byref(a)) == 0
    return a.value


@pytest.mark.parametrize("rows", [100, 300, pytest.param(700, marks=_SLOW)])
def test_tsqr_tree_r_factor(emu, rows):
    """csrc/dhqr_tsqr.h alone: R of a rows x 128 panel through leaves of 256 rows and the pairwise reduction (odd leaf
    counts, a short last leaf, fewer rows than columns) against LAPACK's R"""
    h = _ctx(emu)
    rng = np.random.default_rng(rows)
    Pm = np.asfortranarray(rng.random((rows, 128)))
    R = np.zeros((128, 128), order="F")
    assert emu.dhqr_tsqr_r_f64(h, _ptr(Pm), rows, rows, _ptr(R)) == 0, emu.dhqr_last_error()
    assert emu.dhqr_synchronize(h) == 0
    Rl = np.linalg.qr(Pm, mode="r")
    k = min(rows, 128)
    assert np.abs(np.tril(R, -1)).max() == 0.0
    # R is unique up to the sign of each row; the tree's signs depend on the leaf structure (every node applies the
    # reference's rule alpha = -sign(pivot) norm to ITS pivot), the replay of dhqr_recon.h restores the reference's
    sg = np.sign(np.diag(R)[:k]) * np.sign(np.diag(Rl)[:k])
    assert np.all(np.abs(sg) == 1.0)
    assert np.abs(sg[:, None] * R[:k] - Rl[:k]).max() <= 1e-13 * np.abs(Rl).max()
    if rows < 128:
        assert np.abs(R[rows:]).max() == 0.0
    emu.dhqr_destroy(h)


@pytest.mark.parametrize("m,n", [(300, 128), pytest.param(300, 256, marks=_SLOW), pytest.param(700, 512, marks=_SLOW)])
def test_tsqr_as_the_r_source_of_every_panel(emu, orc, m, n):
    """DHQR_TSQR=1: every R-first panel takes R from the tree, then the same replay / reconstruction"""
    h = _ctx(emu, DHQR_TSQR=1)
    A0 = orc.rand_matrix(m, n, 5)
    A, al = _factor(emu, h, A0, 128)
    _check(orc, A0, A, al)
    fast, fb = _counters(emu, h)
    assert fb == 0 and fast >= 1 and _tsqr_count(emu, h) == fast
    emu.dhqr_destroy(h)


@pytest.mark.parametrize("delta", [pytest.param(1e-6, marks=_SLOW), 1e-12])
def test_tsqr_rung_of_the_fallback_ladder(emu, orc, delta):
    """two nearly dependent columns inside panel 1 (kappa(panel) ~ 1e7 / 1e13): the Gram/Cholesky path must refuse the
    panel (||v||^2 check, before anything is written); TSQR-HR -- Householder R from the tree, reflectors from the
    explicit orthonormal Q -- takes it at full accuracy; the column-by-column kernels are never needed"""
    h = _ctx(emu, DHQR_TSQR_RUNG=1)
    m, n = 600, 384
    A0 = orc.rand_matrix(m, n, 22)
    A0[:, 200] = A0[:, 199] + delta * A0[:, 200]
    A, al = _factor(emu, h, A0, 128)
    fast, fb = _counters(emu, h)
    assert fb == 0 and fast == 3 and _tsqr_count(emu, h) == 1, (fast, fb, _tsqr_count(emu, h))
    QR = orc.form_qr(np.asfortranarray(A), al)
    assert np.linalg.norm(A0 - QR) / np.linalg.norm(A0) < 1e-14
    v2 = (np.tril(A) ** 2).sum(axis=0)
    assert np.abs(v2 - 2.0).max() < 1e-12
    emu.dhqr_destroy(h)


def test_ill_conditioned_panel_falls_back_and_stays_stable(emu, orc):
    """the same kind of panel without the TSQR rung (the single-GPU default): the column-by-column kernels redo it,
    the result is backward stable"""
    h = _ctx(emu)
    m, n = 600, 384
    A0 = orc.rand_matrix(m, n, 22)
    A0[:, 200] = A0[:, 199] * (1.0 + 1e-9)
    A, al = _factor(emu, h, A0, 128)
    fast, fb = _counters(emu, h)
    assert fb >= 1 and fast + fb >= 3 and _tsqr_count(emu, h) == 0
    QR = orc.form_qr(np.asfortranarray(A), al)
    assert np.linalg.norm(A0 - QR) / np.linalg.norm(A0) < 1e-13
    emu.dhqr_destroy(h)


def test_solve_apply_q_and_residual_entry_points(emu, orc):
    h = _ctx(emu)
    m, n = 300, 200
    A0 = orc.rand_matrix(m, n, 5)
    A, al = _factor(emu, h, A0, 128)
    Ho, ao = orc.householder(A0)
    # solve_householder!(b, H, alpha): b is overwritten, x = b[0:n]
    b = orc.rand_vector(m, 6)
    bb = b.copy()
    assert emu.dhqr_solve_f64(h, _ptr(A), m, n, m, _ptr(al), _ptr(bb)) == 0, emu.dhqr_last_error()
    xo = orc.solve(Ho, ao, b)
    assert np.abs(bb[:n] - xo).max() <= 1e-10 * np.abs(xo).max()
    # host drop-in `H \\ b` must not touch b
    x = np.zeros(n)
    b2 = b.copy()
    assert emu.dhqr_ldiv_f64(h, _ptr(A), m, n, m, _ptr(al), _ptr(b2), _ptr(x)) == 0
    assert np.array_equal(b2, b) and np.abs(x - xo).max() <= 1e-10 * np.abs(xo).max()
    # ||A - QR|| / ||A|| and Q'(Q B) == B
    work = np.zeros((m, n), order="F")
    rel = ctypes.c_double()
    assert emu.dhqr_residual_f64(h, _ptr(A), m, n, m, _ptr(al), _ptr(np.asfortranarray(A0)), m, _ptr(work),
                                 ctypes.byref(rel)) == 0
    assert rel.value < 1e-14
    B0 = orc.rand_matrix(m, 3, 7)
    B = B0.copy(order="F")
    assert emu.dhqr_apply_q_f64(h, _ptr(A), m, n, m, _ptr(B), 3, m, 0) == 0
    assert emu.dhqr_apply_q_f64(h, _ptr(A), m, n, m, _ptr(B), 3, m, 1) == 0
    assert np.abs(B - B0).max() < 1e-13
    # host-in / host-out qr!(A) (own device copy, odd m -> padded leading dimension)
    A3 = orc.rand_matrix(81, 30, 12)
    A2 = A3.copy(order="F")
    al2 = np.zeros(30)
    assert emu.dhqr_qr_f64(h, _ptr(A2), 81, 30, 81, _ptr(al2), 0) == 0
    _check(orc, A3, A2, al2)
    emu.dhqr_destroy(h)


@pytest.mark.parametrize("m,n,env", [(300, 200, {}), (520, 384, {}), (777, 130, {}), (260, 257, {"DHQR_TUNE": "qtb_vec=2"}),
                                     (1100, 1000, {"DHQR_TUNE": "qtb_vec=2"}), (640, 128, {"DHQR_TUNE": "qtb_vec=1"}), (70, 50, {}),
                                     (129, 129, {})])
def test_pipelined_solve_kernels(emu, orc, m, n, env):
    """dhqr_solve_f64 through dhqr_qtb.h (batched Gram / T' pre-pass on the factor in place, one k_qtb_step launch per
    panel with the last-workgroup reduction, the flag-pipelined back substitution) against the oracle's solve
    (src:215-294) and against the round-1 path (DHQR_SOLVE_PIPE=0): partial last panel, odd heights, both load widths"""
    A0 = orc.rand_matrix(m, n, 31)
    Ho, ao = orc.householder(A0)
    b = orc.rand_vector(m, 32)
    xo = orc.solve(Ho, ao, b)
    H = np.asfortranarray(Ho)
    xs = []
    for pipe in (1, 0):
        h = _ctx(emu, DHQR_SOLVE_PIPE=pipe, **env)
        bb = b.copy()
        assert emu.dhqr_solve_f64(h, _ptr(H), m, n, m, _ptr(ao), _ptr(bb)) == 0, emu.dhqr_last_error()
        assert emu.dhqr_synchronize(h) == 0, emu.dhqr_last_error()
        assert np.abs(bb[:n] - xo).max() <= 1e-10 * np.abs(xo).max(), (pipe, np.abs(bb[:n] - xo).max())
        if pipe and m > n:  # b <- Q'b below the triangle: rows n..m of the oracle's Q'b (the reference leaves them in b, src:284-294)
            qtb = b.copy()
            for j in range(n):
                s = Ho[j:, j] @ qtb[j:]
                qtb[j:] -= Ho[j:, j] * s
            assert np.abs(bb[n:] - qtb[n:]).max() <= 1e-12 * max(1.0, np.abs(qtb).max())
        xs.append(bb[:n].copy())
        emu.dhqr_destroy(h)
    assert np.abs(xs[0] - xs[1]).max() <= 1e-11 * np.abs(xo).max()


# (every instantiation and its edges; the remaining shapes with DHQR_SLOW=1 -- ~15 s each on the emulator, so this is about
# redundancy, not suite time: profiles/r06_pytest_cpu.txt has the durations; the GPU suite runs all of them)
SMALL_SHAPES = [(1, 1), (5, 3), (111, 100), (128, 128), (130, 20), (224, 224), (256, 192)] + [
    pytest.param(m, n, marks=_SLOW) for m, n in ((33, 33), (64, 64), (220, 200), (224, 208), (256, 17), (129, 129))]


@pytest.mark.parametrize("m,n", SMALL_SHAPES)
def test_small_route_single_workgroup_kernels(emu, orc, m, n):
    """csrc/dhqr_small.h: qr! and `\\` of a matrix that fits one compute unit's registers in ONE launch each (k_small_qr_d: the
    reference's column-by-column algorithm with the matrix in registers; k_small_ldiv: Q'b + back substitution with the
    factor streamed through LDS) -- device pointers and the host-array entry points on the pinned staging buffer, every
    instantiation and its edges, against the oracle; nb is ignored on this route"""
    h = _ctx(emu, DHQR_SMALL=1)
    A0 = orc.rand_matrix(m, n, 41)
    Ho, ao = orc.householder(A0)
    for nb in (0, 128):
        A, al = _factor(emu, h, A0, nb)
        _check(orc, A0, A, al)
    fa, fb = _counters(emu, h)
    assert fa == 0 and fb == 0  # no panel went through the blocked drivers
    A2 = A0.copy(order="F")
    al2 = np.zeros(n)
    assert emu.dhqr_qr_f64(h, _ptr(A2), m, n, m, _ptr(al2), 128) == 0, emu.dhqr_last_error()
    assert np.array_equal(A2, A) and np.array_equal(al2, al)  # same kernel, same arithmetic
    b = orc.rand_vector(m, 42)
    xo = orc.solve(Ho, ao, b)
    bb = b.copy()
    assert emu.dhqr_solve_f64(h, _ptr(A), m, n, m, _ptr(al), _ptr(bb)) == 0, emu.dhqr_last_error()
    assert emu.dhqr_synchronize(h) == 0
    assert np.abs(bb[:n] - xo).max() <= 1e-10 * np.abs(xo).max()
    qtb = b.copy()
    for j in range(n):
        s_ = Ho[j:, j] @ qtb[j:]
        qtb[j:] -= Ho[j:, j] * s_
    if m > n:  # the reference leaves Q'b below the triangle (src:284-294)
        assert np.abs(bb[n:] - qtb[n:]).max() <= 1e-12 * max(1.0, np.abs(qtb).max())
    x = np.zeros(n)
    b2 = b.copy()
    assert emu.dhqr_ldiv_f64(h, _ptr(A), m, n, m, _ptr(al), _ptr(b2), _ptr(x)) == 0
    assert np.array_equal(b2, b) and np.array_equal(x, bb[:n])
    # a padded leading dimension on the host side
    ld = m + 3
    A3 = np.zeros((ld, n), order="F")
    A3[:m] = A0
    al3 = np.zeros(n)
    assert emu.dhqr_qr_f64(h, _ptr(A3), m, n, ld, _ptr(al3), 0) == 0
    assert np.array_equal(A3[:m], A) and np.all(A3[m:] == 0.0)
    emu.dhqr_destroy(h)


def test_small_route_off_and_out_of_range(emu, orc):
    """dhqr_set_small_route(ctx, 0) and shapes beyond the instantiations take the general drivers (seen in the launch-group
    statistics: the small route is ONE group of the reflector-apply category, the blocked drivers time panels)"""
    h = _ctx(emu, DHQR_SMALL=1)

    def groups(A0, nb=128):
        assert emu.dhqr_reset_stats(h) == 0 and emu.dhqr_set_profiling(h, 1) == 0
        A, al = _factor(emu, h, A0, nb)
        _check(orc, A0, A, al)
        st = emu.Stats()
        assert emu.dhqr_get_stats(h, ctypes.byref(st)) == 0
        assert emu.dhqr_set_profiling(h, 0) == 0
        return st.n_rank1, st.n_panel

    A0 = orc.rand_matrix(200, 130, 3)
    assert groups(A0) == (1, 0)
    assert emu.dhqr_set_small_route(h, 0) == 0
    r1, pn = groups(A0)
    assert r1 == 0 and pn >= 2          # two panels through the blocked driver
    assert emu.dhqr_set_small_route(h, 1) == 0
    r1, pn = groups(orc.rand_matrix(300, 225, 4))    # 225 columns: beyond every instantiation
    assert r1 == 0 and pn >= 2
    assert groups(orc.rand_matrix(224, 208, 4), nb=0) == (1, 0)
    emu.dhqr_destroy(h)


def test_small_route_flag_form_gives_up_instead_of_hanging(emu, orc):
    """above 128 rows the small route's column loop runs on LDS flags with BOUNDED waits (dhqr_small.h): with no poll
    allowed (DHQR_TUNE small_spin_limit=0) every wait that is not already satisfied gives up, the kernel answers NaN in
    alpha -- and the host-array entry point factors once more with the barrier form, so the caller still gets the
    factorisation; the device-resident entry point reports what the kernel said"""
    A0 = orc.rand_matrix(150, 40, 12)
    h = _ctx(emu, DHQR_SMALL=1, DHQR_TUNE="small_spin_limit=0")
    A, al = A0.copy(order="F"), np.zeros(40)
    assert emu.dhqr_factor_f64(h, _ptr(A), 150, 40, 150, _ptr(al), 0) == 0, emu.dhqr_last_error()
    assert np.isnan(al).all()                      # every lane of the emulator is a fiber of its own: some wait always has to poll
    A2, al2 = A0.copy(order="F"), np.zeros(40)
    assert emu.dhqr_qr_f64(h, _ptr(A2), 150, 40, 150, _ptr(al2), 0) == 0, emu.dhqr_last_error()
    _check(orc, A0, A2, al2)
    emu.dhqr_destroy(h)
    h = _ctx(emu, DHQR_SMALL=1, DHQR_TUNE="small_flags=0")   # the barrier form by request
    A3, al3 = A0.copy(order="F"), np.zeros(40)
    assert emu.dhqr_factor_f64(h, _ptr(A3), 150, 40, 150, _ptr(al3), 0) == 0, emu.dhqr_last_error()
    _check(orc, A0, A3, al3)
    emu.dhqr_destroy(h)


def test_complex_entry_points(emu, orc):
    h = _ctx(emu)
    m, n = 150, 90
    A0 = orc.rand_matrix_c(m, n, 8)
    A = A0.copy(order="F")
    al = np.zeros(n, dtype=complex)
    assert emu.dhqr_factor_c64(h, _ptr(A), m, n, m, _ptr(al)) == 0, emu.dhqr_last_error()
    Ho, ao = orc.householder_c(A0)
    assert np.abs(A - Ho).max() <= 1e-12 * np.abs(Ho).max() and np.abs(al - ao).max() <= 1e-12 * np.abs(Ho).max()
    b = orc.rand_vector_c(m, 9)
    bb = b.copy()
    assert emu.dhqr_solve_c64(h, _ptr(A), m, n, m, _ptr(al), _ptr(bb)) == 0
    xo = orc.solve_c(Ho, ao, b)
    assert np.abs(bb[:n] - xo).max() <= 1e-10 * np.abs(xo).max()
    out = (ctypes.c_double * 2)()
    assert emu.dhqr_partialdot_c64(h, _ptr(A0[:, 0].copy()), _ptr(A0[:, 1].copy()), 7, m, out) == 0
    assert complex(out[0], out[1]) == pytest.approx(np.vdot(A0[7:, 0], A0[7:, 1]), rel=1e-13)
    emu.dhqr_destroy(h)


def test_complex_blocked_through_the_real_embedding(emu, orc):
    """dhqr_factor_c64_nb(nb = 64): panels by the unblocked complex kernels, trailing update by the Float64 MFMA kernels
    on the real embedding of the 64 complex reflectors; last panel partial (8 columns).  Same factorisation."""
    h = _ctx(emu)
    m, n = 300, 200
    A0 = orc.rand_matrix_c(m, n, 18)
    A = A0.copy(order="F")
    al = np.zeros(n, dtype=complex)
    assert emu.dhqr_factor_c64_nb(h, _ptr(A), m, n, m, _ptr(al), 64) == 0, emu.dhqr_last_error()
    Ho, ao = orc.householder_c(A0)
    assert np.abs(A - Ho).max() <= 1e-12 * np.abs(Ho).max() and np.abs(al - ao).max() <= 1e-12 * np.abs(Ho).max()
    A2, al2 = A0.copy(order="F"), np.zeros(n, dtype=complex)
    assert emu.dhqr_qr_c64_nb(h, _ptr(A2), m, n, m, _ptr(al2), 64) == 0
    assert np.array_equal(A2, A) and np.array_equal(al2, al)
    assert emu.dhqr_factor_c64_nb(h, _ptr(A2), m, n, m, _ptr(al2), 32) == -1  # unsupported width
    emu.dhqr_destroy(h)


def test_argument_validation_and_empty_matrix(emu):
    h = _ctx(emu)
    A = np.zeros((4, 8), order="F")
    al = np.zeros(8)
    assert emu.dhqr_factor_f64(h, _ptr(A), 4, 8, 4, _ptr(al), 128) == -1          # m < n
    assert b"m >= n" in emu.dhqr_last_error()
    assert emu.dhqr_factor_f64(h, _ptr(A), 8, 4, 4, _ptr(al), 128) == -1          # lda < m
    assert emu.dhqr_factor_f64(h, _ptr(A), 8, 4, 8, _ptr(al), 64) == -1           # nb not in {0, 128}
    assert emu.dhqr_factor_f64(h, None, 8, 4, 8, _ptr(al), 0) == -1               # null matrix
    assert emu.dhqr_factor_f64(h, _ptr(A), 8, 0, 8, None, 0) == 0                 # no columns: no-op (src:127)
    assert emu.dhqr_solve_f64(h, _ptr(A), 8, 0, 8, None, None) == 0
    assert emu.dhqr_factor_f64(None, _ptr(A), 8, 4, 8, _ptr(al), 0) == -1         # null context
    emu.dhqr_destroy(h)
