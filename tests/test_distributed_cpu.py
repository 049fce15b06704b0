"""world_size-2 gloo tests (CPU):
 1. the reference's DArray structure restated (oracle/dist_oracle.py, BASELINE config 1:
    512 x 512, nprocs = 2) equals the single-process oracle;
 2. the product's ColumnCyclicQR orchestration (block-cyclic split, one panel broadcast per
    block, look-ahead, α replication, residual and solve pipelines) with an oracle-backed CPU
    backend injected by the test equals the single-process oracle;
 3. the same orchestration with the product's HipBackend marshalling bound to the EMULATED library
    (tests/simt: csrc/ compiled for the CPU), i.e. the real panel / apply / residual / solve entry
    points and kernels at world size 2 and 3 -- everything of the multi-GPU data path except RCCL
    and HIP streams."""
import numpy as np
import pytest

from dist_helpers import OracleBackend, run_ranks


def _ref_darray(rank, P, m, n):
    import torch.distributed as dist
    from oracle import dhqr_oracle as orc
    from oracle import dist_oracle as do
    A = orc.rand_matrix(m, n, 0)
    b = orc.rand_vector(m, 1)
    lo, hi = do.column_blocks(n, P)[rank]
    Al = np.array(A[:, lo:hi], order="F", copy=True)
    alpha = np.zeros(n)
    do.householder_darray(Al, m, n, alpha)
    x = do.solve_darray(Al, m, n, alpha, b)
    Ho, ao = orc.householder(A)
    xo = orc.solve(Ho, ao, b)
    scale = np.abs(Ho).max()
    assert np.abs(Al - Ho[:, lo:hi]).max() <= 1e-12 * scale
    assert np.abs(alpha - ao).max() <= 1e-12 * scale
    assert np.abs(x - xo).max() <= 1e-9 * np.abs(xo).max()
    return float(np.abs(Al - Ho[:, lo:hi]).max())


@pytest.mark.parametrize("m,n,P", [(512, 512, 2), (110, 100, 2), (67, 33, 3)])
def test_reference_darray_structure(m, n, P):
    run_ranks(_ref_darray, P, m, n)


def _cyclic(rank, P, m, n, lookahead):
    import torch
    import __graft_entry__ as g
    from oracle import dhqr_oracle as orc
    pkg = g.import_package()
    q = pkg.ColumnCyclicQR(m, n, backend=OracleBackend(), lookahead=lookahead)
    q.fill(5)
    q.factor()
    H, alpha = q.gather_full()
    A = orc.rand_matrix(m, n, 5)
    Ho, ao = orc.householder(A)
    scale = np.abs(Ho).max()
    assert np.abs(H - Ho).max() <= 1e-11 * scale, np.abs(H - Ho).max()
    assert np.abs(alpha - ao).max() <= 1e-11 * scale
    res = q.residual(5)
    assert res < 1e-13, res
    b = orc.rand_vector(m, 6)
    x = q.solve(torch.from_numpy(b.copy())).numpy()
    xo = orc.solve(Ho, ao, b)
    assert np.abs(x - xo).max() <= 1e-9 * np.abs(xo).max()
    return res


@pytest.mark.parametrize("lookahead", [True, False])
@pytest.mark.parametrize("m,n,P", [(300, 260, 2), (700, 520, 2), (400, 385, 3), (200, 100, 2), (640, 640, 2)])
def test_column_cyclic_orchestration(m, n, P, lookahead):
    run_ranks(_cyclic, P, m, n, lookahead)


def _darray(rank, P, m, n):
    """qr!(A::DArray) front-end: every rank passes its CONTIGUOUS column block (the reference's
    DistributedArrays layout, test/runtests.jl:71) and gets it back factored"""
    import importlib
    import torch
    import __graft_entry__ as g
    from oracle import dhqr_oracle as orc
    pkg = g.import_package()
    part = importlib.import_module("dhqr_amd.partition")
    A = orc.rand_matrix(m, n, 51)
    cols = part.contiguous_column_blocks(n, P)[rank]
    local = torch.from_numpy(np.array(A[:, cols.start: cols.stop], order="F"))
    # layout round trip first: scatter to block-cyclic and gather back is the identity
    q0 = pkg.ColumnCyclicQR(m, n, backend=OracleBackend())
    q0.load_contiguous_blocks(local)
    back = q0.store_contiguous_blocks()
    assert tuple(back.shape) == (m, len(cols)) and torch.equal(back, local)
    H0, _ = q0.gather_full()
    assert np.array_equal(H0, A)
    # the front-end proper
    q, alpha = pkg.qr_darray_(local, n, backend=OracleBackend())
    Ho, ao = orc.householder(A)
    scale = np.abs(Ho).max()
    if len(cols):  # (a rank may own no column at all)
        assert np.abs(local.numpy() - Ho[:, cols.start: cols.stop]).max() <= 1e-11 * scale
    assert np.abs(alpha.numpy() - ao).max() <= 1e-11 * scale
    b = orc.rand_vector(m, 52)
    x = q.solve(torch.from_numpy(b.copy())).numpy()
    assert np.abs(x - orc.solve(Ho, ao, b)).max() <= 1e-9 * np.abs(orc.solve(Ho, ao, b)).max()
    return True


@pytest.mark.parametrize("m,n,P", [(300, 260, 2), (400, 385, 3), (130, 5, 3), (200, 2, 3)])
def test_darray_layout_front_end(m, n, P):
    """contiguous blocks in, contiguous blocks out (incl. ranks that own few or no columns)"""
    run_ranks(_darray, P, m, n)


def test_darray_front_end_single_rank():
    import torch
    import __graft_entry__ as g
    from oracle import dhqr_oracle as orc
    pkg = g.import_package()
    A = orc.rand_matrix(200, 150, 53)
    local = torch.from_numpy(A.copy(order="F"))
    q, alpha = pkg.qr_darray_(local, 150, backend=OracleBackend())
    Ho, ao = orc.householder(A)
    assert np.abs(local.numpy() - Ho).max() <= 1e-11 * np.abs(Ho).max()
    assert np.abs(alpha.numpy() - ao).max() <= 1e-11 * np.abs(Ho).max()


def _cyclic_emulated(rank, P, m, n, so):
    """ColumnCyclicQR with the product's HipBackend marshalling bound to the EMULATED library: the real
    dhqr_panel_factor / _pack / _apply / form_r0 / diff_norms / backsub_block / fill entry points and kernels
    under the real orchestration, world size P, gloo"""
    import torch
    import __graft_entry__ as g
    from oracle import dhqr_oracle as orc
    from dist_helpers import make_emu_backend
    pkg = g.import_package()
    q = pkg.ColumnCyclicQR(m, n, backend=make_emu_backend(so))
    q.fill(71)
    H0, _ = q.gather_full()
    A = orc.rand_matrix(m, n, 71)
    assert np.array_equal(H0, A)  # the device generator with the block-cyclic column map
    q.factor()
    H, alpha = q.gather_full()
    Ho, ao = orc.householder(A)
    scale = np.abs(Ho).max()
    assert np.abs(H - Ho).max() <= 1e-11 * scale
    assert np.abs(alpha - ao).max() <= 1e-11 * scale
    assert q.residual(71) < 1e-13
    b = orc.rand_vector(m, 72)
    x = q.solve(torch.from_numpy(b.copy())).numpy()
    xo = orc.solve(Ho, ao, b)
    assert np.abs(x - xo).max() <= 1e-9 * np.abs(xo).max()
    return True


@pytest.mark.parametrize("m,n,P", [(450, 300, 3)])
def test_column_cyclic_with_the_emulated_library(emulated_so, m, n, P):
    run_ranks(_cyclic_emulated, P, m, n, emulated_so)


def _rowsplit_emulated(rank, P, m, n, so):
    """RowSplitQR with the product's HipRowBackend marshalling and the dhqr_rs_* entry points of the emulated
    library: Gram / Cholesky / replay / commit / V'C kernels under the real all-reduce orchestration"""
    import importlib
    import torch
    import __graft_entry__ as g
    from oracle import dhqr_oracle as orc
    from dist_helpers import make_emu_row_backend
    g.import_package()
    rs = importlib.import_module("dhqr_amd.rowsplit")
    q = rs.RowSplitQR(m, n, backend=make_emu_row_backend(so))
    q.fill(81)
    q.factor()
    H, alpha = q.gather_full()
    Ho, ao = orc.householder(orc.rand_matrix(m, n, 81))
    scale = np.abs(Ho).max()
    assert np.abs(H - Ho).max() <= 1e-11 * scale, np.abs(H - Ho).max()
    assert np.abs(alpha - ao).max() <= 1e-11 * scale
    assert q.residual(81) < 1e-13
    b = orc.rand_vector(m, 82)
    x = q.solve(torch.from_numpy(b[q.row0: q.row0 + q.mloc].copy())).numpy()
    xo = orc.solve(Ho, ao, b)
    assert np.abs(x - xo).max() <= 1e-9 * np.abs(xo).max()
    return True


@pytest.mark.parametrize("m,n,P", [(1200, 256, 2)])
def test_row_split_with_the_emulated_library(emulated_so, m, n, P):
    run_ranks(_rowsplit_emulated, P, m, n, emulated_so)


def test_single_rank_without_process_group():
    import __graft_entry__ as g
    from oracle import dhqr_oracle as orc
    pkg = g.import_package()
    q = pkg.ColumnCyclicQR(300, 200, backend=OracleBackend())
    q.fill(2)
    q.factor()
    H, alpha = q.gather_full()
    Ho, ao = orc.householder(orc.rand_matrix(300, 200, 2))
    assert np.abs(H - Ho).max() <= 1e-11 * np.abs(Ho).max()
    assert q.residual(2) < 1e-13


def _rowsplit(rank, P, m, n):
    import importlib
    import __graft_entry__ as g
    from oracle import dhqr_oracle as orc
    from dist_helpers import NumpyRowBackend
    pkg = g.import_package()
    rs = importlib.import_module("dhqr_amd.rowsplit")
    q = rs.RowSplitQR(m, n, backend=NumpyRowBackend())
    q.fill(31)
    q.factor()
    H, alpha = q.gather_full()
    Ho, ao = orc.householder(orc.rand_matrix(m, n, 31))
    scale = np.abs(Ho).max()
    assert np.abs(H - Ho).max() <= 1e-11 * scale, np.abs(H - Ho).max()
    assert np.abs(alpha - ao).max() <= 1e-11 * scale
    res = q.residual(31)
    assert res < 1e-13, res
    import torch
    b = orc.rand_vector(m, 32)
    x = q.solve(torch.from_numpy(b[q.row0: q.row0 + q.mloc].copy())).numpy()
    xo = orc.solve(Ho, ao, b)
    assert np.abs(x - xo).max() <= 1e-9 * np.abs(xo).max()
    return res


@pytest.mark.parametrize("m,n,P", [(2048, 256, 2), (3000, 384, 3), (1024, 128, 2), (1536, 512, 1)])
def test_row_split_orchestration(m, n, P):
    """BASELINE configs[4] structure (rows split over ranks, all-reduce of Gram matrices and of the
    V'C partial dots) with the numpy backend under gloo, against the single-process oracle."""
    if P == 1:
        import torch.distributed as dist
        assert not dist.is_initialized()
        _rowsplit(0, 1, m, n)
    else:
        run_ranks(_rowsplit, P, m, n)
