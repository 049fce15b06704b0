"""Kernel-level checks on a real MI355X, all through the C ABI (ctypes)."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available()
    return torch


def test_native_library_loaded(pkg, torch_cuda):
    ctx = pkg.get_context(0)
    assert ctx.handle
    maps = open("/proc/self/maps").read()
    assert "libdhqr.so" in maps, "HIP extension not loaded: tests would be meaningless"


def test_mfma_layout_probe(pkg, torch_cuda):
    """Pins the f64 16x16x4 fragment maps the GEMM kernels rely on (asymmetric A, B)."""
    torch = torch_cuda
    rng = np.random.default_rng(5)
    A = rng.standard_normal((16, 4))
    B = rng.standard_normal((4, 16))
    da = torch.tensor(A.reshape(-1), device="cuda:0")
    db = torch.tensor(B.reshape(-1), device="cuda:0")
    out = torch.zeros(256, dtype=torch.float64, device="cuda:0")
    BL, bh = pkg.bench_context(0)  # the probe lives in libdhqr_bench.so (same kernels source), not in the product library
    torch.cuda.synchronize()
    pkg.bench_check(BL, BL.dhqr_debug_mfma_probe(bh, ctypes.c_void_p(da.data_ptr()), ctypes.c_void_p(db.data_ptr()),
                                               ctypes.c_void_p(out.data_ptr())))
    D = A @ B
    got = out.cpu().numpy().reshape(64, 4)
    want = np.empty((64, 4))
    for lane in range(64):
        for g in range(4):
            want[lane, g] = D[(lane >> 4) + 4 * g, lane & 15]
    assert np.allclose(got, want, rtol=1e-14, atol=1e-14), "f64 MFMA C/D layout differs from documentation"


def test_fill_matches_oracle_generator(pkg, orc, torch_cuda):
    A = pkg.rand_colmajor(37, 11, 42, "cuda:0")
    assert np.array_equal(A.cpu().numpy(), orc.rand_matrix(37, 11, 42))
    # block-cyclic local fill: rank 1 of 3, column block 4, local 37 x 8 of a 37 x 22 global matrix
    lay = pkg.BlockCyclicColumns(22, 4, 3)
    ncl = lay.local_ncols(1)
    Al = pkg.rand_colmajor(37, ncl, 42, "cuda:0", global_m=37, colblock=4, nranks=3, rank=1).cpu().numpy()
    G = orc.rand_matrix(37, 22, 42)
    for jl in range(ncl):
        assert np.array_equal(Al[:, jl], G[:, lay.global_col(1, jl)])


def test_partialdot_kat(pkg, torch_cuda):
    # test/partialdot.jl:12-20 on the device reduction, plus a long vector
    rng = np.random.default_rng(0)
    for N in list(range(1, 21)) + [1000, 100003]:
        a, b = rng.random(N), rng.random(N)
        for i in sorted(set([0, N // 3, N - 1] if N > 20 else range(N))):
            got = pkg.partialdot(a, b, i, N)
            assert got == pytest.approx(np.dot(a[i:], b[i:]), rel=1e-13)


@pytest.mark.parametrize("rows,ncols,ldc_pad,trans", [
    (256, 128, 0, 1), (384, 300, 0, 1), (1000, 517, 2, 1), (130, 1, 0, 1), (2048, 640, 0, 0),
    (777, 129, 1, 1),  # odd rows / odd ld: scalar-load path
])
def test_panel_apply_vs_fp64_matmul(pkg, torch_cuda, rows, ncols, ldc_pad, trans):
    """C <- (I - V op(T) V') C against plain torch fp64 matmul on the same device data."""
    torch = torch_cuda
    L = pkg._lib.lib()
    g = torch.Generator(device="cuda:0").manual_seed(rows * 7 + ncols)
    ldv = L.dhqr_panel_ldv(rows)
    nel = L.dhqr_panel_buffer_elems(rows)
    vt = torch.zeros(nel, dtype=torch.float64, device="cuda:0")
    V = torch.rand((rows, 128), generator=g, dtype=torch.float64, device="cuda:0") - 0.5
    V = torch.tril(V)  # packed V has zeros above the diagonal
    T = torch.triu(torch.rand((128, 128), generator=g, dtype=torch.float64, device="cuda:0") - 0.5)
    vt[: ldv * 128].view(128, ldv)[:, :rows] = V.t()
    vt[ldv * 128: ldv * 128 + 128 * 128].view(128, 128).copy_(T.t())           # T, column-major
    vt[ldv * 128 + 128 * 128: ldv * 128 + 2 * 128 * 128].view(128, 128).copy_(T)  # T', column-major
    ldc = rows + ldc_pad
    Cbuf = torch.rand((ncols, ldc), generator=g, dtype=torch.float64, device="cuda:0")
    C = Cbuf.t()[:rows, :]
    C0 = C.clone()
    pad0 = Cbuf.t()[rows:, :].clone()
    ctx = pkg.get_context(0)
    ctx.use_torch_stream()
    pkg._lib.check(L.dhqr_panel_apply_f64(ctx.handle, ctypes.c_void_p(vt.data_ptr()), rows,
                                          ctypes.c_void_p(C.data_ptr()), ncols, ldc, trans))
    torch.cuda.synchronize()
    Top = T.t() if trans else T
    want = C0 - V @ (Top @ (V.t() @ C0))
    err = (C - want).abs().max().item() / want.abs().max().item()
    assert err < 1e-12, err
    if ldc_pad:  # pad rows untouched
        assert torch.equal(Cbuf.t()[rows:, :], pad0)


def test_microbenchmarks_report(pkg, torch_cuda):
    tf = pkg.bench_mfma_tflops(0)
    gb = pkg.bench_stream_gbps(1 << 30, 0)
    print(f"\nFP64 MFMA issue-bound: {tf:.1f} TFLOP/s; streaming copy: {gb:.0f} GB/s")
    assert tf > 20.0 and gb > 1000.0


@pytest.mark.parametrize("m,n,vec", [(440, 400, 0), (1100, 1000, 2), (2207, 2000, 0), (4400, 4000, 1), (4400, 4000, 2),
                                     (2048, 2048, 0), (3000, 129, 0), (20000, 640, 2), (130, 130, 0)])
def test_pipelined_solve_vs_oracle_and_round1_path(pkg, orc, torch_cuda, m, n, vec, monkeypatch):
    """dhqr_solve_f64 through dhqr_qtb.h (batched Gram / T' pre-pass in place, one k_qtb_step launch per panel, the
    flag-pipelined back substitution) on the oracle's factor: x against the oracle's solve (src:215-294) and against the
    round-1 path (DHQR_SOLVE_PIPE=0); rows n..m of b hold Q'b like the reference leaves them (src:284-294)"""
    torch = torch_cuda
    L = pkg._lib.lib()
    A0 = orc.rand_matrix(m, n, 77)
    Ho, ao = orc.householder(A0)
    b = orc.rand_vector(m, 78)
    xo = orc.solve(Ho, ao, b)
    H = torch.tensor(np.asfortranarray(Ho).T.copy(), device="cuda:0").T  # column-major device copy
    assert H.stride() == (1, m)
    al = torch.tensor(ao, device="cuda:0")
    xs = []
    for pipe in (1, 0):
        monkeypatch.setenv("DHQR_SOLVE_PIPE", str(pipe))
        if vec:
            monkeypatch.setenv("DHQR_TUNE", f"qtb_vec={vec}")
        ctx = pkg.Context(0)  # the switches are read by dhqr_create
        bb = torch.tensor(b, device="cuda:0")
        torch.cuda.synchronize()
        pkg._lib.check(L.dhqr_solve_f64(ctx.handle, ctypes.c_void_p(H.data_ptr()), m, n, m, ctypes.c_void_p(al.data_ptr()),
                                   ctypes.c_void_p(bb.data_ptr())))
        ctx.synchronize()
        got = bb.cpu().numpy()
        assert np.abs(got[:n] - xo).max() <= 1e-9 * np.abs(xo).max(), (pipe, np.abs(got[:n] - xo).max())
        xs.append(got)
        ctx.close()
    if m > n:
        assert np.abs(xs[0][n:] - xs[1][n:]).max() <= 1e-11 * max(1.0, np.abs(xs[1]).max())
    assert np.abs(xs[0][:n] - xs[1][:n]).max() <= 1e-10 * np.abs(xo).max()


def test_expired_pipeline_wait_is_reported(pkg, orc, torch_cuda, monkeypatch):
    """a hand-over wait that gives up lets its kernel finish with wrong numbers; the context's error word must
    reach the caller.  DHQR_TUNE=spin_limit=1 (read by dhqr_create) makes every inter-workgroup wait expire at once:
    the ComplexF64 panel pipeline and the pipelined back substitution both have to report through dhqr_synchronize, and
    the blocked Float64 driver through its own status read."""
    torch = torch_cuda
    L = pkg._lib.lib()
    monkeypatch.setenv("DHQR_TUNE", "spin_limit=1")
    ctx = pkg.Context(0)
    try:
        # (1) k_backsub_pipe: 32 blocks, every workgroup but the first waits for its predecessors
        m, n = 4400, 4096
        A = pkg.rand_colmajor(m, n, 3, "cuda:0")
        al = torch.ones(n, dtype=torch.float64, device="cuda:0")
        bb = pkg.rand_vector_device(m, 4, "cuda:0")
        torch.cuda.synchronize()
        pkg._lib.check(L.dhqr_solve_f64(ctx.handle, ctypes.c_void_p(A.data_ptr()), m, n, m, ctypes.c_void_p(al.data_ptr()),
                                   ctypes.c_void_p(bb.data_ptr())))
        with pytest.raises(pkg.DHQRError) as e:
            ctx.synchronize()
        assert "pipeline" in str(e.value)
        ctx.synchronize()  # the word is cleared once reported
        # (1b) ADVICE r5: the synchronous host-array entry point reads the word itself (it used to return DHQR_OK and a wrong x)
        Ah = np.asfortranarray(A.cpu().numpy())
        alh, bh, xh = np.ones(n), bb.cpu().numpy().copy(), np.zeros(n)
        P = ctypes.c_void_p
        rc = L.dhqr_ldiv_f64(ctx.handle, Ah.ctypes.data_as(P), m, n, m, alh.ctypes.data_as(P), bh.ctypes.data_as(P),
                             xh.ctypes.data_as(P))
        assert rc != 0 and b"pipeline" in L.dhqr_last_error()
        ctx.synchronize()
        # (2) k_zpanel_pipe (blocked ComplexF64, 64-column panels)
        Z = torch.view_as_complex(pkg.rand_colmajor(2 * 1024, 512, 5, "cuda:0").T.contiguous().view(512, 1024, 2)).T
        assert Z.stride() == (1, 1024)
        za = torch.zeros(512, dtype=torch.complex128, device="cuda:0")
        torch.cuda.synchronize()
        pkg._lib.check(L.dhqr_factor_c64_nb(ctx.handle, ctypes.c_void_p(Z.data_ptr()), 1024, 512, 1024,
                                       ctypes.c_void_p(za.data_ptr()), 64))
        with pytest.raises(pkg.DHQRError):
            ctx.synchronize()
    finally:
        ctx.close()
    monkeypatch.delenv("DHQR_TUNE")
    ctx = pkg.Context(0)  # a fresh context waits again
    try:
        bb = pkg.rand_vector_device(m, 4, "cuda:0")
        pkg._lib.check(L.dhqr_solve_f64(ctx.handle, ctypes.c_void_p(A.data_ptr()), m, n, m, ctypes.c_void_p(al.data_ptr()),
                                   ctypes.c_void_p(bb.data_ptr())))
        ctx.synchronize()
    finally:
        ctx.close()


@pytest.mark.parametrize("m,n", [(4400, 4000), (2207, 2000), (1100, 300)])
def test_solve_with_kept_t_factors(pkg, orc, torch_cuda, m, n, monkeypatch):
    """A blocked dhqr_factor_f64 leaves T' of every panel in the context; dhqr_solve_f64 on the same matrix skips its Gram /
    T' pre-pass when alpha still is the factorisation's (device-side check).  (1) x with kept T == x without (DHQR_KEEP_T=0)
    to rounding, both against the oracle; (2) ANOTHER factor written to the same address is noticed (its alpha differs):
    the solve falls back to the pre-pass and is still right."""
    torch = torch_cuda
    L = pkg._lib.lib()
    P = ctypes.c_void_p
    b = orc.rand_vector(m, 91)
    xs = {}
    for keep in (1, 0):
        monkeypatch.setenv("DHQR_KEEP_T", str(keep))
        ctx = pkg.Context(0)
        try:
            A = pkg.rand_colmajor(m, n, 90, "cuda:0")
            al = torch.zeros(n, dtype=torch.float64, device="cuda:0")
            torch.cuda.synchronize()
            pkg._lib.check(L.dhqr_factor_f64(ctx.handle, P(A.data_ptr()), m, n, m, P(al.data_ptr()), 128))
            bb = torch.tensor(b, device="cuda:0")
            pkg._lib.check(L.dhqr_solve_f64(ctx.handle, P(A.data_ptr()), m, n, m, P(al.data_ptr()), P(bb.data_ptr())))
            ctx.synchronize()
            xs[keep] = bb[:n].cpu().numpy()
            if keep:
                # (2) the same device buffers now receive a DIFFERENT factorisation (the oracle's, of another matrix)
                A2 = orc.rand_matrix(m, n, 92)
                H2, a2 = orc.householder(A2)
                A.copy_(torch.tensor(np.asfortranarray(H2).T.copy(), device="cuda:0").T)
                al.copy_(torch.tensor(a2, device="cuda:0"))
                bb = torch.tensor(b, device="cuda:0")
                torch.cuda.synchronize()
                pkg._lib.check(L.dhqr_solve_f64(ctx.handle, P(A.data_ptr()), m, n, m, P(al.data_ptr()), P(bb.data_ptr())))
                ctx.synchronize()
                x2 = orc.solve(H2, a2, b)
                assert np.abs(bb[:n].cpu().numpy() - x2).max() <= 1e-9 * np.abs(x2).max()
        finally:
            ctx.close()
    Ho, ao = orc.householder(orc.rand_matrix(m, n, 90))
    xo = orc.solve(Ho, ao, b)
    for keep in (1, 0):
        assert np.abs(xs[keep] - xo).max() <= 1e-9 * np.abs(xo).max(), keep
    assert np.abs(xs[1] - xs[0]).max() <= 1e-10 * np.abs(xo).max()


def test_trim_releases_and_the_context_keeps_working(pkg, orc, torch_cuda):
    """dhqr_trim (the host-in / host-out entry points keep an 8 GiB device copy at 32768^2 until dhqr_destroy): the
    cached device matrix, staging buffers and solve workspaces are released, and the next calls allocate again"""
    torch = torch_cuda
    ctx = pkg.Context(0)
    try:
        L = pkg._lib.lib()
        m, n = 3000, 2000
        A0 = orc.rand_matrix(m, n, 55)
        al = np.zeros(n)
        free0 = torch.cuda.mem_get_info()[0]
        A = A0.copy(order="F")
        pkg._lib.check(L.dhqr_qr_f64(ctx.handle, A.ctypes.data_as(ctypes.c_void_p), m, n, m, al.ctypes.data_as(ctypes.c_void_p), 128))
        free1 = torch.cuda.mem_get_info()[0]
        assert free0 - free1 >= m * n * 8  # the device copy stays in the context
        ctx.trim()
        free2 = torch.cuda.mem_get_info()[0]
        assert free2 - free1 >= m * n * 8
        A2 = A0.copy(order="F")
        al2 = np.zeros(n)
        pkg._lib.check(L.dhqr_qr_f64(ctx.handle, A2.ctypes.data_as(ctypes.c_void_p), m, n, m, al2.ctypes.data_as(ctypes.c_void_p), 128))
        assert np.array_equal(A, A2) and np.array_equal(al, al2)
        b = orc.rand_vector(m, 56)
        x = np.zeros(n)
        pkg._lib.check(L.dhqr_ldiv_f64(ctx.handle, A2.ctypes.data_as(ctypes.c_void_p), m, n, m, al2.ctypes.data_as(ctypes.c_void_p),
                                       b.ctypes.data_as(ctypes.c_void_p), x.ctypes.data_as(ctypes.c_void_p)))
        Ho, ao = orc.householder(A0)
        xo = orc.solve(Ho, ao, b)
        assert np.abs(x - xo).max() <= 1e-9 * np.abs(xo).max()
    finally:
        ctx.close()


def test_persistent_solve_kernel_is_repeated_when_it_cannot_get_its_workgroups(pkg, torch_cuda):
    """VERDICT r5 weak #7 / ADVICE r5: k_qtb_persist's workgroups wait for each other in both directions, so all of them must
    be resident; a competing kernel can prevent that.  tools/solve_retry_probe.py (a process of its own: see its header)
    holds most of the compute units with another stream's kernel (libdhqr_bench.so's test hook: 150 KB of LDS per workgroup,
    nothing fits beside it) while a solve forces the persistent kernel: where its 128 workgroups do not fit on the
    remaining CUs their bounded waits expire -- and the first synchronising call must repeat the solve with the per-step
    kernels and return the right x, not an error."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "solve_retry_probe.py")], capture_output=True, text=True,
                       timeout=280)
    rows = [json.loads(ln) for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(rows) == 4, (p.stdout[-2000:], p.stderr[-2000:])
    for r in rows:
        assert r["rc"] == 0, r
        assert r["max_abs_diff_to_undisturbed_x"] <= 1e-9 * r["max_abs_x"], r
    assert rows[-1]["retries"] >= 1, f"no fill level made the persistent kernel expire: {rows}"


SMALL_SHAPES = [(1, 1), (5, 3), (33, 33), (64, 64), (110, 100), (111, 100), (128, 128), (130, 20), (220, 200), (224, 208),
                (256, 192), (256, 17), (129, 129)]


@pytest.fixture
def small_route(pkg):
    """the product default (csrc/dhqr_small.h) on the shared context for the duration of one test (conftest.py switches it
    off for the rest of the suite, which exercises the general drivers on small shapes)"""
    ctx = pkg.get_context(0)
    ctx.set_small_route(True)
    yield ctx
    ctx.set_small_route(False)


@pytest.mark.parametrize("m,n", [(220, 200), (224, 224), (256, 192), (129, 7), (200, 129)])
def test_small_route_flag_form_equals_barrier_form(pkg, orc, torch_cuda, monkeypatch, m, n):
    """above 128 rows the column loop of the small route runs on LDS flags (k_small_qr_d<.., FLAGS>: no barrier, eight
    buffers, a different builder wave per column); DHQR_TUNE small_flags=0 restores the barrier per column.  The same
    arithmetic in another order of EVENTS: bit-identical factors -- and with small_spin_limit=0 on the device the kernel
    either gets through (a wave seldom has to wait twice) or answers NaN, and the host-array entry point then returns the
    barrier form's bits"""
    torch = torch_cuda
    L = pkg._lib.lib()
    A0 = orc.rand_matrix(m, n, 77)
    out = {}
    for tune in ("small_flags=1", "small_flags=0", "small_spin_limit=0"):
        monkeypatch.setenv("DHQR_TUNE", tune)
        monkeypatch.setenv("DHQR_SMALL", "1")
        ctx = pkg.Context(0)
        A = np.asfortranarray(A0.copy())
        al = np.zeros(n)
        pkg._lib.check(L.dhqr_qr_f64(ctx.handle, A.ctypes.data_as(ctypes.c_void_p), m, n, m, al.ctypes.data_as(ctypes.c_void_p), 0))
        out[tune] = (A, al)
        ctx.close()
    Ho, ao = orc.householder(A0)
    tol = 8.0 * max(n, 8) * np.finfo(np.float64).eps * np.abs(Ho).max()
    assert np.abs(out["small_flags=1"][0] - Ho).max() <= tol and np.abs(out["small_flags=1"][1] - ao).max() <= tol
    for tune in ("small_flags=0", "small_spin_limit=0"):
        assert np.array_equal(out[tune][0], out["small_flags=1"][0]) and np.array_equal(out[tune][1], out["small_flags=1"][1]), tune


@pytest.mark.parametrize("m,n", SMALL_SHAPES)
def test_small_route_vs_oracle(pkg, orc, torch_cuda, small_route, m, n):
    """qr! and `\\` of a matrix that fits one compute unit's registers: ONE single-workgroup launch each (k_small_qr_d,
    k_small_ldiv), on device tensors and on host arrays (the kernels then work on the pinned staging buffer across PCIe),
    every instantiation and its edges, against the oracle (src:122-148,198-213,215-294)"""
    torch = torch_cuda
    A0 = orc.rand_matrix(m, n, 41)
    Ho, ao = orc.householder(A0)
    b = orc.rand_vector(m, 42)
    xo = orc.solve(Ho, ao, b)
    scale = np.abs(Ho).max()
    tol = 8.0 * max(n, 8) * np.finfo(np.float64).eps
    got = []
    for nb in (0, 128):
        A = pkg.rand_colmajor(m, n, 41, "cuda:0")
        H = pkg.qr_(A, nb=nb)
        torch.cuda.synchronize()
        assert np.abs(H.A.cpu().numpy() - Ho).max() <= tol * scale
        assert np.abs(H.α.cpu().numpy() - ao).max() <= tol * scale
        got.append(H.A.cpu().numpy())
    assert np.array_equal(got[0], got[1])  # nb is ignored on this route: same launch
    bd = torch.tensor(b, device="cuda:0")
    x = pkg.ldiv(H, bd)
    assert np.array_equal(bd.cpu().numpy(), b), "H \\ b must not modify b (src:318)"
    assert np.abs(x.cpu().numpy() - xo).max() <= 1e-9 * np.abs(xo).max()
    assert pkg.residual(H, pkg.rand_colmajor(m, n, 41, "cuda:0")) < 1e-12
    # host arrays: the drop-in qr!(A::Matrix) / H \ b
    Ah = np.asfortranarray(A0.copy())
    Hh = pkg.qr_(Ah)
    assert np.array_equal(Hh.A, got[0]) and np.array_equal(Hh.α, H.α.cpu().numpy())
    xh = pkg.ldiv(Hh, b)
    assert np.array_equal(xh, x.cpu().numpy())
    # a strided host matrix (leading dimension > m)
    big = np.zeros((m + 5, n), order="F")
    big[:m] = A0
    Hv = pkg.qr_(big[:m])
    assert np.array_equal(np.asarray(Hv.A), got[0]) and np.all(big[m:] == 0.0)


def test_small_route_is_one_launch_group(pkg, orc, torch_cuda, small_route):
    """the route is taken (one reflector-apply group, no panel group) and leaves it when switched off or out of range"""
    ctx = small_route

    def groups(m, n, nb):
        ctx.reset_stats()
        ctx.set_profiling(True)
        A = pkg.rand_colmajor(m, n, 1, "cuda:0")
        pkg.qr_(A, nb=nb)
        st = ctx.stats()
        ctx.set_profiling(False)
        return st["n_rank1"], st["n_panel"]

    assert groups(220, 200, 128) == (1, 0)
    assert groups(110, 100, 0) == (1, 0)
    r1, pn = groups(300, 225, 128)
    assert r1 == 0 and pn >= 2
    ctx.set_small_route(False)
    r1, pn = groups(220, 200, 128)
    assert r1 == 0 and pn >= 2
    ctx.set_small_route(True)


def test_small_route_reference_inequality(pkg, orc, torch_cuda, small_route):
    """test/runtests.jl:61-63 at the two reference shapes the route covers: ||A'(Ax - b)|| < 8 x LAPACK's, several draws"""
    import scipy.linalg as sl
    for (m, n) in ((110, 100), (220, 200)):
        for seed in range(5):
            A0 = np.asfortranarray(orc.rand_matrix(m, n, 100 + seed))
            b = orc.rand_vector(m, 200 + seed)
            H = pkg.qr_(A0.copy(order="F"))
            x = pkg.ldiv(H, b)
            q, r = sl.qr(A0, mode="economic")
            x1 = sl.solve_triangular(r, q.T @ b)
            ne = np.linalg.norm(A0.T @ (A0 @ x - b))
            ne1 = np.linalg.norm(A0.T @ (A0 @ x1 - b))
            assert ne < 8 * ne1, (m, n, seed, ne, ne1)
