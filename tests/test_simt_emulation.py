"""CPU execution of UNMODIFIED HIP kernel sources on the SIMT emulator of tests/simt/ (one OS thread per
HIP thread, ThreadSanitizer build):
  * the single-workgroup panel kernels (csrc/dhqr_recon.h): one-barrier-per-step Cholesky / replay, blocked
    triangular inverses with five barriers;
  * the unblocked factorisation kernels for Float64 and ComplexF64 (dhqr_rank1.h, dhqr_complex.h) and the
    solve kernels (dhqr_solve.h, dhqr_complex.h), launched in the library's per-column sequence;
  * the FP64-MFMA trailing-update GEMMs (dhqr_gemm.h) with v_mfma_f64_16x16x4_f64 emulated wave-
    synchronously: tile edges, split-K slabs, scalar/16-byte loads, narrow reflector blocks, the
    XCD-aware 1-D launch order, and the LDS double-buffer protocol under ThreadSanitizer.  Checks the numerics against numpy / the oracle and that ThreadSanitizer reports no
data race (= no missing barrier in the LDS staging protocol).  Test infrastructure only; needs the
host clang++ of the ROCm toolchain (for -fsanitize=thread and the clang vector extensions).
"""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIMT = os.path.join(ROOT, "tests", "simt")
CSRC = os.path.join(ROOT, "distributedhouseholderqr.jl_amd", "csrc")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
N = 128

pytestmark = pytest.mark.skipif(not os.path.exists(CLANG), reason="host clang++ (ROCm llvm) not found")


@pytest.fixture(scope="module")
def emu(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("simt") / "emu_recon_tsan")
    subprocess.check_call([CLANG, "-std=c++20", "-O1", "-g", "-fsanitize=thread", "-Wno-unknown-attributes",
                           "-I", os.path.join(SIMT, "fake"), "-I", CSRC, os.path.join(SIMT, "emu_recon.cpp"),
                           "-o", exe, "-lpthread"])
    return exe


@pytest.fixture(scope="module")
def emu_paths(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("simt") / "emu_paths_tsan")
    subprocess.check_call([CLANG, "-std=c++20", "-O1", "-g", "-fsanitize=thread", "-Wno-unknown-attributes",
                           "-I", os.path.join(SIMT, "fake"), "-I", CSRC, os.path.join(SIMT, "emu_paths.cpp"),
                           "-o", exe, "-lpthread"])
    return exe


@pytest.fixture(scope="module")
def emu_gemm(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("simt") / "emu_gemm_tsan")
    subprocess.check_call([CLANG, "-std=c++20", "-O1", "-g", "-fsanitize=thread", "-Wno-unknown-attributes",
                           "-Wno-psabi", "-I", os.path.join(SIMT, "fake"), "-I", CSRC,
                           os.path.join(SIMT, "emu_gemm.cpp"), "-o", exe, "-lpthread"])
    return exe


def _run(exe, *args):
    r = subprocess.run([exe, *map(str, args)], capture_output=True, text=True, timeout=900)
    assert "ThreadSanitizer" not in r.stderr, r.stderr[:3000]
    assert r.returncode == 0, r.stderr[:3000]


def _put(path, M):  # column-major raw float64
    np.asfortranarray(M).T.tofile(path)


def _get(path, shape=(N, N)):
    return np.fromfile(path).reshape(shape[::-1]).T


def test_rig_detects_a_missing_barrier(emu, tmp_path):
    out = str(tmp_path / "o.bin")
    _run(emu, "sync", 0, out)
    o = np.fromfile(out)
    assert np.array_equal(o, (np.arange(128) + 64) % 128 + 64.0)  # LDS neighbour + wave_sum(1) == 64
    r = subprocess.run([emu, "racy", "0", out], capture_output=True, text=True, timeout=300)
    assert "ThreadSanitizer: data race" in r.stderr


@pytest.mark.parametrize("variant", [5])
def test_cholesky_and_inverse(emu, orc, tmp_path, variant):
    P = orc.rand_matrix(300, N, 5)
    G = P.T @ P
    f = {k: str(tmp_path / f"{k}.bin") for k in ("G", "R", "X", "flag", "Rp", "R2", "X2")}
    _put(f["G"], G)
    _run(emu, "chol", variant, f["G"], "-", 1, f["R"], f["X"], f["flag"])
    R, negX = _get(f["R"]), _get(f["X"])
    Rn = np.linalg.cholesky(G).T
    assert np.abs(R - Rn).max() < 1e-12 * np.abs(Rn).max()
    assert np.abs(-negX - np.linalg.inv(Rn)).max() < 1e-12 * np.abs(np.linalg.inv(Rn)).max()
    assert np.array_equal(np.tril(R, -1), np.zeros((N, N)))
    assert np.array_equal(np.fromfile(f["flag"]), [0.0, 0.0])
    if True:  # second CholeskyQR pass (R <- R * Rprev) and the breakdown flag: once is enough
        Q1 = P @ np.linalg.inv(Rn)
        _put(f["G"], Q1.T @ Q1)
        _put(f["Rp"], Rn)
        _run(emu, "chol", variant, f["G"], f["Rp"], 0, f["R2"], f["X2"], f["flag"])
        R2 = _get(f["R2"])
        Rfull = np.linalg.qr(P, mode="r")
        Rfull *= np.sign(np.diag(Rfull))[:, None]
        assert np.abs(R2 - Rfull).max() < 1e-12 * np.abs(Rfull).max()
        Gbad = G.copy()
        Gbad[40, 40] = -1.0
        _put(f["G"], Gbad)
        _run(emu, "chol", variant, f["G"], "-", 0, f["R2"], f["X2"], f["flag"])
        assert np.fromfile(f["flag"])[0] == 1.0


@pytest.mark.parametrize("variant", [5])
def test_replay_of_top_block(emu, orc, tmp_path, variant):
    rows = 300
    P = orc.rand_matrix(rows, N, 6)
    Ho, ao = orc.householder(P)                      # the reference algorithm on the whole panel
    R = np.linalg.qr(P, mode="r")                    # any row signs are accepted by the kernel
    f = {k: str(tmp_path / f"{k}.bin") for k in ("P", "R", "al", "Rref", "Mi")}
    _put(f["P"], P[:N])
    _put(f["R"], R)
    _run(emu, "recon", variant, f["P"], f["R"], f["al"], f["Rref"], f["Mi"])
    alpha, Rref, negMinv = np.fromfile(f["al"]), _get(f["Rref"]), _get(f["Mi"])
    scale = np.abs(Ho).max()
    assert np.abs(alpha - ao).max() < 1e-12 * scale
    assert np.abs(Rref - np.triu(Ho[:N], 1)).max() < 1e-12 * scale
    # what the library does next (one GEMM): V = tril((P - alpha E) M^{-1}) for ALL rows of the panel
    PE = P.copy()
    PE[:N] -= np.diag(alpha)
    V = np.tril(PE @ (-negMinv))
    assert np.abs(V - np.tril(Ho)).max() < 1e-12 * scale
    assert np.array_equal(np.tril(negMinv, -1), np.zeros((N, N)))  # -M^{-1} is upper triangular


def test_fused_panel_top(emu, orc, tmp_path):
    """k_panel_top: Cholesky of the Gram matrix and the replay of the top block in ONE loop (one barrier per column for
    both), then -M^{-1}: the reference's alpha, R and (after the library's GEMM) V; ThreadSanitizer-clean"""
    rows = 300
    P = orc.rand_matrix(rows, N, 6)
    Ho, ao = orc.householder(P)
    f = {k: str(tmp_path / f"{k}.bin") for k in ("G", "P", "al", "Rref", "Mi", "flag")}
    _put(f["G"], P.T @ P)
    _put(f["P"], P[:N])
    _run(emu, "top", 0, f["G"], f["P"], f["al"], f["Rref"], f["Mi"], f["flag"])
    alpha, Rref, negMinv = np.fromfile(f["al"]), _get(f["Rref"]), _get(f["Mi"])
    scale = np.abs(Ho).max()
    assert np.abs(alpha - ao).max() < 1e-12 * scale
    assert np.abs(Rref - np.triu(Ho[:N], 1)).max() < 1e-12 * scale
    PE = P.copy()
    PE[:N] -= np.diag(alpha)
    V = np.tril(PE @ (-negMinv))
    assert np.abs(V - np.tril(Ho)).max() < 1e-12 * scale
    assert np.array_equal(np.tril(negMinv, -1), np.zeros((N, N)))
    assert np.array_equal(np.fromfile(f["flag"]), [0.0, 0.0])


@pytest.mark.parametrize("variant,ncols", [(5, 128), (5, 77), (5, 1)])
def test_block_reflector_t(emu, orc, tmp_path, variant, ncols):
    Ho, _ = orc.householder(orc.rand_matrix(300, ncols, 7))
    V = np.zeros((300, N))
    V[:, :ncols] = np.tril(Ho)
    S = V.T @ V
    f = {k: str(tmp_path / f"{k}.bin") for k in ("S", "T", "Tt")}
    _put(f["S"], S)
    _run(emu, "buildt", variant, f["S"], ncols, f["T"], f["Tt"])
    T, Tt = _get(f["T"]), _get(f["Tt"])
    Tn = np.linalg.inv(np.eye(N) + np.triu(S, 1))    # compact WY: T^{-1} = I + striu(V'V)
    assert np.abs(T - Tn).max() < 1e-13
    assert np.array_equal(Tt, T.T)
    # H_1 ... H_ncols == I - V T V'
    Q = np.eye(300)
    for j in reversed(range(ncols)):
        Q -= np.outer(V[:, j], V[:, j] @ Q)
    assert np.abs(Q - (np.eye(300) - V @ T @ V.T)).max() < 1e-13


# ------------------------------------------------------------------ unblocked path / solve kernels
@pytest.mark.parametrize("m,n,threads", [(70, 9, 256), (71, 6, 256), (40, 5, 0), (41, 4, 0)])
def test_unblocked_f64_kernels(emu_paths, orc, tmp_path, m, n, threads):
    """k_reflector + k_rank1_fused<256,8,VEC> (threads=256) / k_rank1_generic<1024,VEC> (threads=0); even m
    takes the 16-byte (VEC=2) loads, odd m the scalar ones -- one launch per column like the library"""
    A = orc.rand_matrix(m, n, 3)
    f = {k: str(tmp_path / f"{k}.bin") for k in ("A", "H", "al")}
    _put(f["A"], A)
    _run(emu_paths, "f64", m, n, f["A"], f["H"], f["al"], threads)
    Ho, ao = orc.householder(A)
    assert np.abs(_get(f["H"], (m, n)) - Ho).max() < 1e-13 * np.abs(Ho).max()
    assert np.abs(np.fromfile(f["al"]) - ao).max() < 1e-13 * np.abs(Ho).max()


@pytest.mark.parametrize("m,n,threads", [(70, 7, 256), (33, 5, 512), (20, 4, 1024)])
def test_unblocked_c64_kernels(emu_paths, orc, tmp_path, m, n, threads):
    A = orc.rand_matrix_c(m, n, 3)
    f = {k: str(tmp_path / f"{k}.bin") for k in ("A", "H", "al")}
    A.T.copy().view(np.float64).tofile(f["A"])  # column-major interleaved (re, im)
    _run(emu_paths, "c64", m, n, f["A"], f["H"], f["al"], threads)
    H = np.fromfile(f["H"]).view(np.complex128).reshape(n, m).T
    al = np.fromfile(f["al"]).view(np.complex128)
    Ho, ao = orc.householder_c(A)
    assert np.abs(H - Ho).max() < 1e-13 * np.abs(Ho).max()
    assert np.abs(al - ao).max() < 1e-13 * np.abs(Ho).max()


def test_solve_kernels_c64_and_f64(emu_paths, orc, tmp_path):
    f = {k: str(tmp_path / f"{k}.bin") for k in ("H", "al", "b", "x")}
    # ComplexF64: Q^H b column by column, then the 32-row blocked back substitution (two blocks + remainder)
    m, n = 90, 70
    A = orc.rand_matrix_c(m, n, 8)
    b = orc.rand_vector_c(m, 9)
    Ho, ao = orc.householder_c(A)
    Ho.T.copy().view(np.float64).tofile(f["H"])
    ao.view(np.float64).tofile(f["al"])
    b.view(np.float64).tofile(f["b"])
    _run(emu_paths, "zsolve", m, n, f["H"], f["al"], f["b"], f["x"])
    x = np.fromfile(f["x"]).view(np.complex128)
    xo = orc.solve_c(Ho, ao, b)
    assert np.abs(x - xo).max() < 1e-11 * np.abs(xo).max()
    # Float64 back substitution (64-row blocks) on b = Q'b
    m, n = 150, 100
    A = orc.rand_matrix(m, n, 10)
    b = orc.rand_vector(m, 11)
    Ho, ao = orc.householder(A)
    qtb = b.copy()
    for j in range(n):
        qtb[j:] -= Ho[j:, j] * (Ho[j:, j] @ qtb[j:])
    _put(f["H"], Ho)
    ao.tofile(f["al"])
    qtb.tofile(f["b"])
    _run(emu_paths, "backsub", m, n, f["H"], f["al"], f["b"], f["x"])
    xo = orc.solve(Ho, ao, b)
    assert np.abs(np.fromfile(f["x"]) - xo).max() < 1e-11 * np.abs(xo).max()


# ------------------------------------------------------------------ FP64 MFMA trailing-update GEMMs
@pytest.mark.parametrize("vec,nbv,rows,ncols,rps", [(2, 128, 100, 200, 48), (1, 128, 101, 130, 64),
                                                    (2, 64, 64, 128, 64), (2, 32, 96, 70, 32)])
def test_gemm_tn_split_k(emu_gemm, tmp_path, vec, nbv, rows, ncols, rps):
    """W = V'C per row slab (k_gemm_tn<VEC,1,NBV>): partial column tile, last slab shorter than a K-tile,
    padding rows of V / C (set to 7 / 9) must never be read into the result"""
    rng = np.random.default_rng(1)
    ldv, ldc = rows + rows % 2 + 2, rows + rows % 2 + 4
    V = np.full((ldv, 128), 7.0)
    V[:rows] = 0.0
    V[:rows, :nbv] = rng.standard_normal((rows, nbv))
    C = np.full((ldc, ncols), 9.0)
    C[:rows] = rng.standard_normal((rows, ncols))
    f = {k: str(tmp_path / f"{k}.bin") for k in ("V", "C", "o")}
    _put(f["V"], V)
    _put(f["C"], C)
    _run(emu_gemm, "tn", vec, nbv, rows, ncols, ldv, ldc, rps, f["V"], f["C"], f["o"])
    ns = (rows + rps - 1) // rps
    out = np.fromfile(f["o"]).reshape(ns, ncols, 128)  # [slab][column][p]
    W = out.sum(axis=0).T[:nbv]
    assert np.abs(W - V[:rows, :nbv].T @ C[:rows]).max() < 1e-12


@pytest.mark.parametrize("vec,rows,ncols,rps", [(2, 200, 150, 80), (1, 131, 129, 64)])
def test_gemm_tn_two_panels(emu_gemm, tmp_path, vec, rows, ncols, rps):
    """W = [V_a V_b]' C in one pass over C (k_gemm_tn2: 512 threads, 128 x 256 output tile): partial column tile,
    short last slab, scalar / 16-byte loads, padding rows never read"""
    rng = np.random.default_rng(3)
    ldv, ldc = rows + rows % 2 + 2, rows + rows % 2 + 4
    V = np.full((ldv, 256), 7.0)
    V[:rows] = rng.standard_normal((rows, 256))
    V[:128, 128:] = 0.0  # the second panel starts 128 rows further down
    C = np.full((ldc, ncols), 9.0)
    C[:rows] = rng.standard_normal((rows, ncols))
    f = {k: str(tmp_path / f"{k}.bin") for k in ("V", "C", "o")}
    _put(f["V"], V)
    _put(f["C"], C)
    _run(emu_gemm, "tn2", vec, 256, rows, ncols, ldv, ldc, rps, f["V"], f["C"], f["o"])
    ns = (rows + rps - 1) // rps
    out = np.fromfile(f["o"]).reshape(ns, ncols, 256)  # [slab][column][p]
    W = out.sum(axis=0).T
    assert np.abs(W - V[:rows].T @ C[:rows]).max() < 1e-12


@pytest.mark.parametrize("vec,rows,ncols,groups,wgs", [(2, 420, 300, 1, 7), (2, 420, 300, 8, 19), (2, 610, 150, 4, 13),
                                                        (1, 333, 260, 2, 9), (2, 200, 140, 8, 24), (2, 96, 130, 8, 11)])
def test_gemm_tn_two_panels_stream_k_row_groups(emu_gemm, tmp_path, vec, rows, ncols, groups, wgs):
    """stream-K k_gemm_tn2 + k_reduce_pieces with ROW GROUPS (tn2_sk_group_of: workgroup b belongs to group (b % 8) % R and
    works inside that group's row range): every (tile, slab) unit exactly once whatever the number of workgroups -- grids
    that are not multiples of 8, groups with different numbers of workgroups, a last group with fewer slabs, groups with
    no slab at all (96 rows = 3 slabs, 8 groups)"""
    rng = np.random.default_rng(5)
    ldv, ldc = rows + rows % 2 + 2, rows + rows % 2 + 4
    V = np.full((ldv, 256), 7.0)
    V[:rows] = rng.standard_normal((rows, 256))
    C = np.full((ldc, ncols), 9.0)
    C[:rows] = rng.standard_normal((rows, ncols))
    f = {k: str(tmp_path / f"{k}.bin") for k in ("V", "C", "o")}
    _put(f["V"], V)
    _put(f["C"], C)
    _run(emu_gemm, "tn2sk", vec, groups, rows, ncols, ldv, ldc, wgs, f["V"], f["C"], f["o"])
    Y = np.fromfile(f["o"]).reshape(ncols, 256).T
    assert np.abs(Y - V[:rows].T @ C[:rows]).max() < 1e-11


@pytest.mark.parametrize("vec,kw,rows,ncols,swz", [(2, 128, 200, 150, 0), (1, 128, 131, 129, 0),
                                                   (2, 256, 256, 128, 0), (2, 128, 300, 260, 1),
                                                   (2, 128, 200, 150, 2), (2, 256, 192, 256, 2)])  # swz 2: 64-row tiles
def test_gemm_nn_sub(emu_gemm, tmp_path, vec, kw, rows, ncols, swz):
    """C -= V W (k_gemm_nn_sub<VEC,KW>): edge tiles in both directions, K = 256 (two-panel update), the
    XCD-aware 1-D launch (swz = 1: every tile exactly once, surplus workgroups exit) and the 64-row tiles of the lane's
    narrow products (swz = 2 in this harness)"""
    rng = np.random.default_rng(2)
    ldv, ldc = rows + rows % 2 + 2, rows + rows % 2 + 4
    V = np.full((ldv, kw), 5.0)
    V[:rows] = rng.standard_normal((rows, kw))
    W = rng.standard_normal((kw, ncols))
    C = np.full((ldc, ncols), 3.0)
    C[:rows] = rng.standard_normal((rows, ncols))
    f = {k: str(tmp_path / f"{k}.bin") for k in ("V", "W", "C", "Co")}
    _put(f["V"], V)
    _put(f["W"], W)
    _put(f["C"], C)
    _run(emu_gemm, "nn", vec, kw, rows, ncols, ldv, ldc, swz, f["V"], f["W"], f["C"], f["Co"])
    Co = _get(f["Co"], (ldc, ncols))
    ref = C.copy()
    ref[:rows] -= V[:rows] @ W
    assert np.abs(Co - ref).max() < 1e-12   # the padding rows of C keep their value


# (rows, ncols, swz): interior tiles below the second pair (streamed C, K = 512), the two row tiles above it (half the K
# loop, V2 never touched), edge tiles both ways, the XCD-aware 1-D launch (9 x 4 tiles: two 8 x 8 blocks; the size at which
# the driver switches it on, 2100 x 2100, behind DHQR_SLOW=1), 64-row tiles (the lane's narrow quad update)
@pytest.mark.parametrize("rows,ncols,swz", [(512, 256, 0), (700, 300, 0), (1100, 400, 1), (450, 256, 2),
                                            pytest.param(2100, 2100, 1, marks=pytest.mark.skipif(
                                                os.environ.get("DHQR_SLOW") != "1", reason="5 min; DHQR_SLOW=1"))])
def test_gemm_nn_quad(emu_gemm, tmp_path, rows, ncols, swz):
    """C -= [V1 | V2] [W1; W2] (k_gemm_nn_quad): V2 starts 256 rows below V1; padding rows of V / C never used or changed"""
    rng = np.random.default_rng(4)
    skip = 256
    ldv, ldc = rows + 2, rows + 4
    V1 = np.full((ldv, 256), 5.0)
    V1[:rows] = rng.standard_normal((rows, 256))
    V2 = np.full((ldv, 256), 6.0)  # storage row r = operand row r + skip
    V2[:rows - skip] = rng.standard_normal((rows - skip, 256))
    W = rng.standard_normal((512, ncols))
    C = np.full((ldc, ncols), 3.0)
    C[:rows] = rng.standard_normal((rows, ncols))
    f = {k: str(tmp_path / f"{k}.bin") for k in ("V1", "V2", "W", "C", "Co")}
    _put(f["V1"], V1)
    _put(f["V2"], V2)
    _put(f["W"], W)
    _put(f["C"], C)
    _run(emu_gemm, "quad", 2, 512, rows, ncols, ldv, ldc, swz, f["V1"], f["V2"], f["W"], f["C"], f["Co"], skip)
    Co = _get(f["Co"], (ldc, ncols))
    ref = C.copy()
    ref[:rows] -= V1[:rows] @ W[:256]
    ref[skip:rows] -= V2[:rows - skip] @ W[256:]
    assert np.abs(Co - ref).max() < 1e-11


# ------------------------------------------------------------------ memory safety of the new generations
def test_new_panel_kernels_under_address_sanitizer(orc, tmp_path):
    """the panel kernels once more in an AddressSanitizer build of the emulator: no access outside
    the LDS arrays (static globals with red zones) or the global buffers (heap allocations)"""
    exe = str(tmp_path / "emu_recon_asan")
    subprocess.check_call([CLANG, "-std=c++20", "-O1", "-g", "-fsanitize=address", "-Wno-unknown-attributes",
                           "-I", os.path.join(SIMT, "fake"), "-I", CSRC, os.path.join(SIMT, "emu_recon.cpp"),
                           "-o", exe, "-lpthread"])
    P = orc.rand_matrix(300, N, 6)
    Ho, _ = orc.householder(P)
    f = {k: str(tmp_path / f"{k}.bin") for k in ("S", "P", "R", "G", "o1", "o2", "o3")}
    V = np.tril(Ho)
    _put(f["S"], V.T @ V)
    _put(f["P"], P[:N])
    _put(f["R"], np.linalg.qr(P, mode="r"))
    _put(f["G"], P.T @ P)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")
    for args in (("buildt", 5, f["S"], 128, f["o1"], f["o2"]),
                 ("buildt", 5, f["S"], 77, f["o1"], f["o2"]),
                 ("recon", 5, f["P"], f["R"], f["o1"], f["o2"], f["o3"]),
                 ("chol", 5, f["G"], "-", 1, f["o1"], f["o2"], f["o3"])):
        r = subprocess.run([exe, *map(str, args)], capture_output=True, text=True, timeout=600, env=env)
        assert "AddressSanitizer" not in r.stderr and r.returncode == 0, (args[:2], r.stderr[:2000])
