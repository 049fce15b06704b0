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
