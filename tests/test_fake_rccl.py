"""The RCCL transport (COMM_RCCL of csrc/dhqr_comm.h) executed on the CPU: the EMULATED library dlopen()s a stand-in
librccl (tests/simt/fake/rccl/fake_rccl.cpp, via DHQR_RCCL_LIB) and its rank threads sit on distinct emulated devices
(SIMT_DEVICES), so everything that only runs on a multi-GPU node otherwise is exercised against the oracle:

  * dhqr_mg_create -> ncclCommInitAll twice (main channel + the row-split lane's channel), comm_tune_bcast (the timed
    trial of ncclBroadcast vs scatter + all-gather), rccl_bcast_sag (grouped ncclSend / ncclRecv + in-place ncclAllGather),
    its ring fallback when the count does not divide by the rank count, ncclAllReduce in residual / solve / row split;
  * dhqr_comm_create_rank -> ncclCommInitRank from a unique id, the second id shipped over the first communicator;
  * both communicators of the row-split driver in use at once (the stand-in turns "different order on different ranks"
    into a timeout error instead of a hang).

Replaces the reference's fan-out / reduction (src/DistributedHouseholderQR.jl:115-120,141-143,262-266).
TEST INFRASTRUCTURE ONLY: the product never loads the stand-in (DHQR_RCCL_LIB is set inside this module only).
"""
import ctypes
import os
import shutil
import subprocess
import threading

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P_ = ctypes.c_void_p
_SLOW = pytest.mark.skipif(os.environ.get("DHQR_SLOW") != "1", reason="extra configuration; set DHQR_SLOW=1")


def _ptr(a):
    return a.ctypes.data_as(P_)


def build_fake_rccl(outdir):
    so = os.path.join(str(outdir), "libfake_rccl.so")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-pthread",
                           os.path.join(ROOT, "tests", "simt", "fake", "rccl", "fake_rccl.cpp"), "-o", so])
    return so


@pytest.fixture(scope="module")
def rk(emulated_so, tmp_path_factory):
    """(emulated library bound to the stand-in RCCL, the stand-in itself).  A private COPY of the emulated library: its
    RCCL binding is decided once per loaded image, and other test modules share the original."""
    from dist_helpers import load_emulated_library
    d = tmp_path_factory.mktemp("fake_rccl")
    fake = build_fake_rccl(d)
    so2 = shutil.copy(emulated_so, os.path.join(str(d), "libdhqr_emulated_rccl.so"))
    old = {k: os.environ.get(k) for k in ("DHQR_RCCL_LIB", "SIMT_DEVICES", "FAKE_RCCL_TIMEOUT_S")}
    os.environ.update(DHQR_RCCL_LIB=fake, SIMT_DEVICES="8", FAKE_RCCL_TIMEOUT_S="120")
    L = load_emulated_library(so2)
    F = ctypes.CDLL(fake)
    F.fake_rccl_stats.argtypes = [ctypes.POINTER(ctypes.c_int64), ctypes.c_int]
    F.fake_rccl_stats.restype = None
    yield L, F
    for k, v in old.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


def _stats(F, reset=True):
    out = (ctypes.c_int64 * 16)()
    F.fake_rccl_stats(out, 1 if reset else 0)
    names = ("bcast", "allreduce", "allgather", "send", "recv", "groups", "comms", "live", "max_live", "timeouts", "mismatches")
    return dict(zip(names, list(out)))


class _env:
    def __init__(self, **kv):
        self.kv = {k: str(v) for k, v in kv.items()}

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        os.environ.update(self.kv)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _mg(L, ndev):
    h = P_()
    devs = (ctypes.c_int32 * ndev)(*range(ndev))  # distinct devices: the product default transport (RCCL) forms
    assert L.dhqr_mg_create(ctypes.byref(h), devs, ndev) == 0, L.dhqr_last_error()
    t = ctypes.c_int32()
    assert L.dhqr_mg_info(h, None, ctypes.byref(t), None, None) == 0
    assert t.value == 1, "expected DHQR_COMM_RCCL"
    return h


# (ndev, m, n, env): ring / scatter+all-gather forced / chosen by the timed trial; 3 ranks: the panel length does not
# divide by 3 -> ring fallback inside the scatter+all-gather algorithm; 8 ranks with 2 cyclic blocks: six ranks own nothing
@pytest.mark.parametrize("ndev,m,n,env", [
    (2, 520, 384, {"DHQR_BCAST": "ring"}),
    (2, 520, 384, {"DHQR_BCAST": "ring", "DHQR_LANE_SIDE": 0}),
    pytest.param(2, 700, 512, {"DHQR_BCAST": "sag", "DHQR_TUNE": "sag_min=1"}, marks=_SLOW),
    (2, 520, 384, {"DHQR_TUNE": "sag_min=1"}),
    (3, 640, 522, {"DHQR_BCAST": "sag", "DHQR_TUNE": "sag_min=1"}),
    (8, 600, 512, {"DHQR_BCAST": "sag", "DHQR_TUNE": "sag_min=1"}),
    pytest.param(8, 2400, 2304, {"DHQR_BCAST": "sag", "DHQR_TUNE": "sag_min=1"}, marks=_SLOW),
])
def test_column_split_over_rccl(rk, orc, ndev, m, n, env):
    L, F = rk
    _stats(F)
    with _env(**env):
        h = _mg(L, ndev)
    s0 = _stats(F, reset=False)
    assert s0["comms"] == ndev and s0["live"] == ndev  # ONE communicator per rank: the RCCL default (DHQR_LANE_CHANNEL=1 adds the lane's)
    algo, t0, t1 = ctypes.c_int32(), ctypes.c_double(), ctypes.c_double()
    assert L.dhqr_mg_get_bcast_tuning(h, ctypes.byref(algo), ctypes.byref(t0), ctypes.byref(t1)) == 0
    if "DHQR_BCAST" in env:
        assert algo.value == (1 if env["DHQR_BCAST"] == "sag" else 0) and t0.value == 0.0
    else:
        assert t0.value > 0.0 and t1.value > 0.0  # the trial ran both algorithms (collective over the rank threads)
        assert s0["allgather"] >= 8 * ndev and s0["bcast"] >= 8 * ndev
    assert L.dhqr_mg_alloc_f64(h, m, n) == 0, L.dhqr_last_error()
    assert L.dhqr_mg_fill_uniform_f64(h, 3) == 0
    A0 = orc.rand_matrix(m, n, 3)
    _stats(F)
    assert L.dhqr_mg_factor_f64(h) == 0, L.dhqr_last_error()
    s1 = _stats(F)
    npanels = (n + 127) // 128
    assert s1["bcast"] + s1["allgather"] >= ndev * npanels  # one broadcast per panel on every rank
    if algo.value == 1 and ndev != 3:
        # every panel went root -> peers by grouped point-to-point sends, then one in-place all-gather per rank
        assert s1["allgather"] >= ndev * npanels and s1["send"] >= (ndev - 1) * npanels and s1["recv"] == s1["send"]
    elif algo.value == 0:
        assert s1["allgather"] == 0
    if ndev == 3 and algo.value == 1:
        assert s1["bcast"] > 0 and s1["allgather"] > 0  # counts that do not divide by 3 fall back to ncclBroadcast
    assert s1["timeouts"] == 0 and s1["mismatches"] == 0
    H, al = np.zeros((m, n), order="F"), np.zeros(n)
    assert L.dhqr_mg_download_f64(h, _ptr(H), m, _ptr(al)) == 0
    Ho, ao = orc.householder(A0)
    scale = np.abs(Ho).max()
    assert np.abs(H - Ho).max() <= 1e-12 * scale and np.abs(al - ao).max() <= 1e-12 * scale
    rel = ctypes.c_double()
    assert L.dhqr_mg_residual_f64(h, 3, ctypes.byref(rel)) == 0, L.dhqr_last_error()
    assert rel.value < 1e-14
    b, x = orc.rand_vector(m, 9), np.zeros(n)
    assert L.dhqr_mg_solve_f64(h, _ptr(b), _ptr(x)) == 0, L.dhqr_last_error()
    xo = orc.solve(Ho, ao, b)
    assert np.abs(x - xo).max() <= 1e-10 * np.abs(xo).max()
    s2 = _stats(F)
    assert s2["allreduce"] > 0 and s2["timeouts"] == 0 and s2["mismatches"] == 0  # residual + back-substitution sums
    assert L.dhqr_mg_destroy(h) == 0
    assert _stats(F)["live"] == 0  # every communicator destroyed


# Quad steps at P > 1 (cs_plan: two consecutive pairs -- owned by DIFFERENT ranks -- applied in one K = 512 pass): every rank
# builds the 256 x 256 cross term from the broadcast panels itself; the group after a quad gets the quad from its owner's lane,
# the quad's second pair is factored from the head of the previous wide step.  The ranks agree on the plan through one
# all-reduce per factorisation (cs_agree_quads).  bad > 0: a nearly dependent column pair inside the quad is rejected on the
# device and the run resumes (the committed panels of the quad are broadcast again and applied to the rest).
@pytest.mark.parametrize("ndev,m,n,bad", [(2, 780, 768, 0), (2, 700, 640, 300), pytest.param(2, 1290, 1280, 0, marks=_SLOW),
                                          pytest.param(3, 1560, 1536, 0, marks=_SLOW), pytest.param(3, 700, 640, 400, marks=_SLOW)])
def test_column_split_quad_steps_over_rccl(rk, orc, ndev, m, n, bad):
    L, F = rk
    A0 = orc.rand_matrix(m, n, 8)
    if bad:
        A0[:, bad] = A0[:, bad - 1] * (1.0 + 1e-9)
    res = []
    for env in ({"DHQR_QUAD_MIN_COLS": 0, "DHQR_BCAST": "ring"}, {"DHQR_QUAD": 0, "DHQR_BCAST": "ring"}):
        with _env(**env):
            h = _mg(L, ndev)
        A, al = A0.copy(order="F"), np.zeros(n)
        _stats(F)
        assert L.dhqr_mg_qr_f64(h, _ptr(A), m, n, m, _ptr(al)) == 0, L.dhqr_last_error()
        s1 = _stats(F)
        assert s1["timeouts"] == 0 and s1["mismatches"] == 0
        assert s1["allreduce"] == (ndev if "DHQR_QUAD_MIN_COLS" in env else 0)  # the plan agreement, once per factorisation
        if bad:
            QR = orc.form_qr(np.asfortranarray(A), al)
            assert np.linalg.norm(A0 - QR) / np.linalg.norm(A0) < 1e-13
        else:
            Ho, ao = orc.householder(A0)
            scale = np.abs(Ho).max()
            assert np.abs(A - Ho).max() <= 1e-12 * scale and np.abs(al - ao).max() <= 1e-12 * scale
        res.append(A)
        assert L.dhqr_mg_destroy(h) == 0
    if not bad:  # quads and pairs: the same factorisation to rounding
        assert np.abs(res[0] - res[1]).max() <= 1e-12 * np.abs(res[1]).max()


# ComplexF64 column split (dhqr_zdist.h): cyclic blocks of 64 complex columns, one ncclBroadcast per panel issued on the
# look-ahead stream of every rank; 3 ranks / 3 panels with a partial last one; DHQR_SLOW=1: 2 ranks / 4 panels, 8 ranks with 3 panels (five ranks own nothing)
@pytest.mark.parametrize("ndev,m,n,algo", [(3, 200, 150, "ring"), pytest.param(2, 260, 200, "ring", marks=_SLOW),
                                           pytest.param(8, 200, 130, "ring", marks=_SLOW), pytest.param(2, 200, 150, "sag", marks=_SLOW)])
def test_complex_column_split_over_rccl(rk, orc, ndev, m, n, algo):
    L, F = rk
    with _env(DHQR_BCAST=algo, DHQR_TUNE="sag_min=1"):
        h = _mg(L, ndev)
    A0 = orc.rand_matrix_c(m, n, 21)
    A, al = A0.copy(order="F"), np.zeros(n, dtype=complex)
    _stats(F)
    assert L.dhqr_mg_qr_c64(h, _ptr(A), m, n, m, _ptr(al)) == 0, L.dhqr_last_error()
    s1 = _stats(F)
    if algo == "ring":
        assert s1["bcast"] == ndev * ((n + 63) // 64), s1  # one broadcast per panel on every rank
    else:  # scatter + all-gather where the unit divides by the rank count, ncclBroadcast otherwise
        assert s1["allgather"] > 0 and s1["bcast"] + s1["allgather"] == ndev * ((n + 63) // 64), s1
    assert s1["timeouts"] == 0 and s1["mismatches"] == 0
    Ho, ao = orc.householder_c(A0)
    scale = np.abs(Ho).max()
    assert np.abs(A - Ho).max() <= 1e-12 * scale and np.abs(al - ao).max() <= 1e-12 * scale
    # the same bits as one rank (every column sees the same panel operands in the same order)
    h1 = P_()
    assert L.dhqr_create(ctypes.byref(h1), 0) == 0
    A1, al1 = A0.copy(order="F"), np.zeros(n, dtype=complex)
    assert L.dhqr_qr_c64_nb(h1, _ptr(A1), m, n, m, _ptr(al1), 64) == 0, L.dhqr_last_error()
    assert np.abs(A1 - A).max() <= 1e-13 * scale
    L.dhqr_destroy(h1)
    # `H \ b` over the same ranks (dhqr_mg_ldiv_c64 -> zcs_solve, b in double-double): per panel two broadcasts of b's tail
    # (high / low parts, Q'b), one all-reduce gathering the partial dots and one broadcast of the solved block, src:226-282
    b, x = orc.rand_vector_c(m, 22), np.zeros(n, dtype=complex)
    bk = b.copy()
    _stats(F)
    assert L.dhqr_mg_ldiv_c64(h, _ptr(A), m, n, m, _ptr(al), _ptr(b), _ptr(x)) == 0, L.dhqr_last_error()
    s2 = _stats(F)
    np_ = (n + 63) // 64
    assert s2["allreduce"] == ndev * np_ and s2["bcast"] + s2["allgather"] == ndev * 3 * np_, s2
    assert s2["timeouts"] == 0 and s2["mismatches"] == 0
    xo = orc.solve_c(Ho, ao, bk)
    assert np.array_equal(b, bk) and np.abs(x - xo).max() <= 1e-10 * np.abs(xo).max()
    assert L.dhqr_mg_destroy(h) == 0
    assert _stats(F)["live"] == 0


# row split: by default ONE RCCL communicator carries the lane's and the wide stream's collectives (issued in the same host
# order on every rank); DHQR_LANE_CHANNEL=1: the lane's collectives on a second communicator while the wide stream uses the
# first; tsqr: every panel through the cross-rank tree (gather of the R factors by one all-reduce)
@pytest.mark.parametrize("ndev,m,n,env", [
    (2, 1024, 384, {}),
    (2, 1024, 384, {"DHQR_LANE_CHANNEL": 1}),
    (3, 900, 300, {"DHQR_LANE_CHANNEL": 1}),
    pytest.param(3, 900, 300, {}, marks=_SLOW),
    (8, 1100, 256, {}),
    pytest.param(8, 1100, 256, {"DHQR_LANE_CHANNEL": 1}, marks=_SLOW),
    pytest.param(2, 384, 128, {"DHQR_TSQR": 1}, marks=_SLOW),
    pytest.param(8, 2304, 640, {"DHQR_LANE_CHANNEL": 1}, marks=_SLOW),
])
def test_row_split_over_rccl_two_communicators(rk, orc, ndev, m, n, env):
    L, F = rk
    _stats(F)
    with _env(**env):
        h = _mg(L, ndev)
    lane = env.get("DHQR_LANE_CHANNEL", 0) != 0
    assert _stats(F, reset=False)["comms"] == (2 if lane else 1) * ndev
    assert L.dhqr_mg_rs_alloc_f64(h, m, n) == 0, L.dhqr_last_error()
    assert L.dhqr_mg_rs_fill_uniform_f64(h, 31) == 0
    A0 = orc.rand_matrix(m, n, 31)
    _stats(F)
    assert L.dhqr_mg_rs_factor_f64(h) == 0, L.dhqr_last_error()
    s1 = _stats(F)
    assert s1["allreduce"] >= ndev * 2 * (n // 128) and s1["bcast"] >= ndev * (n // 128)  # Gram + S (+ partial dots) per panel
    assert s1["timeouts"] == 0 and s1["mismatches"] == 0
    H, al = np.zeros((m, n), order="F"), np.zeros(n)
    assert L.dhqr_mg_rs_transfer_f64(h, _ptr(H), m, _ptr(al), 0) == 0
    Ho, ao = orc.householder(A0)
    scale = np.abs(Ho).max()
    assert np.abs(H - Ho).max() <= 1e-12 * scale, np.abs(H - Ho).max()
    assert np.abs(al - ao).max() <= 1e-12 * scale
    rel = ctypes.c_double()
    assert L.dhqr_mg_rs_residual_f64(h, 31, ctypes.byref(rel)) == 0, L.dhqr_last_error()
    assert rel.value < 1e-14
    b, x = orc.rand_vector(m, 32), np.zeros(n)
    assert L.dhqr_mg_rs_solve_f64(h, _ptr(b), _ptr(x)) == 0, L.dhqr_last_error()
    xo = orc.solve(Ho, ao, b)
    assert np.abs(x - xo).max() <= 1e-10 * np.abs(xo).max()
    assert L.dhqr_mg_destroy(h) == 0
    assert _stats(F)["live"] == 0


@_SLOW
def test_row_split_rejected_panel_over_rccl(rk, orc):
    """a rejected panel: every rank takes the same decision from the all-reduced S, the panel is redone by TSQR-HR across
    the ranks (three more collectives), the pair is re-applied -- with the collectives going through RCCL"""
    L, F = rk
    h = _mg(L, 2)
    m, n = 1024, 384
    A0 = orc.rand_matrix(m, n, 22)
    A0[:, 200] = A0[:, 199] * (1.0 + 1e-9)
    assert L.dhqr_mg_rs_alloc_f64(h, m, n) == 0
    assert L.dhqr_mg_rs_transfer_f64(h, _ptr(np.asfortranarray(A0)), m, None, 1) == 0
    assert L.dhqr_mg_rs_factor_f64(h) == 0, L.dhqr_last_error()
    H, al = np.zeros((m, n), order="F"), np.zeros(n)
    assert L.dhqr_mg_rs_transfer_f64(h, _ptr(H), m, _ptr(al), 0) == 0
    QR = orc.form_qr(np.asfortranarray(H), al)
    assert np.linalg.norm(A0 - QR) / np.linalg.norm(A0) < 1e-13
    assert L.dhqr_mg_destroy(h) == 0
    s = _stats(F)
    assert s["timeouts"] == 0 and s["mismatches"] == 0 and s["live"] == 0


def _spmd_rank(L, P, r, idbuf, m, n, split, out, errs, lane=False):
    """one rank of the multi-PROCESS form (a torchrun rank / a Julia worker), here a thread: its own context on its own
    device, dhqr_comm_create_rank from the shared unique id, then the SPMD entry points"""
    try:
        h, cm = P_(), P_()
        assert L.dhqr_create(ctypes.byref(h), r) == 0, L.dhqr_last_error()
        assert L.dhqr_comm_create_rank(ctypes.byref(cm), h, P, r, idbuf) == 0, L.dhqr_last_error()
        a, b = ctypes.c_int32(), ctypes.c_int32()
        assert L.dhqr_comm_rccl_nranks(cm, ctypes.byref(a), ctypes.byref(b)) == 0
        assert (a.value, b.value) == (P, P if lane else 0)  # what RCCL itself reports for the main and the lane channel
        al = np.zeros(n)
        rel = ctypes.c_double()
        if split == "cs":
            ncl = L.dhqr_cs_local_cols(n, P, r)
            A = np.zeros((m, max(ncl, 1)), order="F")
            assert L.dhqr_cs_fill_uniform_f64(cm, _ptr(A), m, n, m, 5) == 0, L.dhqr_last_error()
            assert L.dhqr_cs_factor_f64(cm, _ptr(A), m, n, m, _ptr(al)) == 0, L.dhqr_last_error()
            W, A0 = np.zeros((m, max(ncl, 1)), order="F"), np.zeros((m, max(ncl, 1)), order="F")
            assert L.dhqr_cs_residual_f64(cm, _ptr(A), m, n, m, _ptr(al), 5, _ptr(W), _ptr(A0), ctypes.byref(rel)) == 0, L.dhqr_last_error()
        else:
            row0, mloc = ctypes.c_int64(), ctypes.c_int64()
            L.dhqr_rs_row_range(m, P, r, ctypes.byref(row0), ctypes.byref(mloc))
            ml = mloc.value
            A = np.zeros((max(ml, 1), n), order="F")
            ld = max(ml, 1)
            assert L.dhqr_rs_fill_uniform_f64(cm, _ptr(A), m, n, ld, 5) == 0, L.dhqr_last_error()
            assert L.dhqr_rs_factor_f64(cm, _ptr(A), m, n, ld, _ptr(al)) == 0, L.dhqr_last_error()
            W, A0 = np.zeros((max(ml, 1), n), order="F"), np.zeros((max(ml, 1), n), order="F")
            assert L.dhqr_rs_residual_f64(cm, _ptr(A), m, n, ld, _ptr(al), 5, _ptr(W), _ptr(A0), ctypes.byref(rel)) == 0, L.dhqr_last_error()
        out[r] = (A, al.copy(), rel.value)
        assert L.dhqr_comm_destroy(cm) == 0
        assert L.dhqr_destroy(h) == 0
    except BaseException as e:  # surfaced by the caller
        import traceback
        errs.append(f"rank {r}: {traceback.format_exc()}")


@pytest.mark.parametrize("P,m,n,split,lane", [(2, 520, 384, "cs", 0), pytest.param(3, 900, 384, "cs", 0, marks=_SLOW), (3, 600, 256, "rs", 1),
                                                pytest.param(3, 600, 256, "rs", 0, marks=_SLOW), pytest.param(2, 1024, 256, "rs", 1, marks=_SLOW)])
def test_spmd_ranks_bootstrap_from_unique_id(rk, orc, P, m, n, split, lane):
    """dhqr_comm_unique_id -> dhqr_comm_create_rank on every rank (lane = 1, DHQR_LANE_CHANNEL=1: ncclCommInitRank twice, the
    lane's id travels over the first communicator), then the dhqr_cs_* / dhqr_rs_* entry points a torchrun rank / Julia worker calls."""
    L, F = rk
    _stats(F)
    idbuf = (ctypes.c_char * 128)()
    assert L.dhqr_comm_unique_id(idbuf) == 0, L.dhqr_last_error()
    out, errs = {}, []
    th = [threading.Thread(target=_spmd_rank, args=(L, P, r, idbuf, m, n, split, out, errs, bool(lane))) for r in range(P)]
    with _env(**({"DHQR_LANE_CHANNEL": 1} if lane else {})):  # read when the communicators are created
        for t in th:
            t.start()
        for t in th:
            t.join()
    assert not errs, "\n".join(errs)
    Ho, ao = orc.householder(orc.rand_matrix(m, n, 5))
    scale = np.abs(Ho).max()
    for r in range(P):
        A, al, rel = out[r]
        assert np.abs(al - ao).max() <= 1e-12 * scale and rel < 1e-14
        if split == "cs":
            cols = [c for c in range(n) if (c // 256) % P == r]  # block-cyclic, block = a pair of panels
            if cols:  # a rank beyond the last cyclic block owns nothing and still takes part in every collective
                assert np.abs(A[:, :len(cols)] - Ho[:, cols]).max() <= 1e-12 * scale
    s = _stats(F)
    assert s["comms"] == (2 if lane else 1) * P and s["live"] == 0 and s["timeouts"] == 0 and s["mismatches"] == 0


def test_stand_in_reports_mismatch_and_different_order_instead_of_hanging(rk):
    """the checks that make the stand-in worth having: ranks that disagree on a collective get an error on every rank;
    two communicators used in different orders by different ranks time out with a message (FAKE_RCCL_TIMEOUT_S)"""
    L, F = rk
    R = ctypes.CDLL(os.environ["DHQR_RCCL_LIB"])
    comms = (P_ * 2)()
    comms2 = (P_ * 2)()
    devs = (ctypes.c_int * 2)(0, 1)
    assert R.ncclCommInitAll(comms, 2, devs) == 0 and R.ncclCommInitAll(comms2, 2, devs) == 0
    assert R.ncclCommInitAll((P_ * 2)(), 2, (ctypes.c_int * 2)(0, 0)) == 5  # duplicate device: ncclInvalidUsage
    bufs = [np.arange(8, dtype=np.float64) + r for r in range(2)]
    rc = [None, None]

    def call(r, fn):
        rc[r] = fn(r)

    def run(fn):
        th = [threading.Thread(target=call, args=(r, fn)) for r in range(2)]
        [t.start() for t in th]
        [t.join() for t in th]
        return list(rc)

    bc = lambda r, cnt, cm: R.ncclBroadcast(_ptr(bufs[r]), _ptr(bufs[r]), ctypes.c_size_t(cnt), 8, 0, P_(cm[r]), None)
    assert run(lambda r: bc(r, 8, comms)) == [0, 0] and np.array_equal(bufs[1], np.arange(8.0))
    _stats(F)
    assert run(lambda r: bc(r, 8 - 4 * r, comms)) == [4, 4]  # counts differ: ncclInvalidArgument on both
    assert _stats(F)["mismatches"] == 2
    with _env(FAKE_RCCL_TIMEOUT_S=0.5):
        got = run(lambda r: bc(r, 8, comms if r == 0 else comms2))  # rank 0 on communicator 1, rank 1 on communicator 2
    assert got == [3, 3] and _stats(F)["timeouts"] >= 1
    for c in list(comms) + list(comms2):
        R.ncclCommDestroy(P_(c))
