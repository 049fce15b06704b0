"""Test-only helpers: gloo rank spawning, the EMULATED library (tests/simt: csrc/ host-compiled for the CPU),
gloo-backed communicator callbacks for it, and numpy stand-ins for the row-split backend.
The product never imports this file."""
import os
import sys
import traceback

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NB = 128


def run_ranks(fn, world_size, *args):
    """spawn `world_size` gloo ranks running fn(rank, world_size, *args); re-raise failures"""
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_entry, args=(fn, r, world_size, port, q, args)) for r in range(world_size)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    errs = [r for r in results if r[1] is not None]
    if errs:
        raise AssertionError("rank failures:\n" + "\n".join(f"[rank {r[0]}] {r[1]}" for r in errs))
    return {r[0]: r[2] for r in results}


def _entry(fn, rank, world_size, port, q, args):
    try:
        if ROOT not in sys.path:
            sys.path.insert(0, ROOT)
        import torch.distributed as dist
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        os.environ["OMP_NUM_THREADS"] = "2"
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("gloo", rank=rank, world_size=world_size)
        out = fn(rank, world_size, *args)
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, None, out))
    except BaseException:
        q.put((rank, traceback.format_exc(), None))


# ---------------------------------------------------------------------------------------------------
# The product's HipBackend driven by the EMULATED library (tests/simt: the unmodified csrc/ compiled for
# the CPU): the same marshalling code, the same C entry points, CPU tensors instead of device tensors.
# Lets the world_size-2/3 gloo tests run the real panel_factor / panel_apply / form_r0 / backsub kernels.
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def build_emulated_library(outdir):
    """host-compile csrc/dhqr_api.hip against tests/simt/fake (fiber mode) -> path of the .so"""
    import subprocess
    so = os.path.join(str(outdir), "libdhqr_emulated.so")
    subprocess.check_call([CLANG, "-x", "c++", "-std=c++20", "-O2", "-DSIMT_FIBERS", "-fPIC", "-shared",
                           "-Wno-unknown-attributes", "-Wno-psabi", "-Wno-unused-value",
                           "-I", os.path.join(ROOT, "tests", "simt", "fake"),
                           os.path.join(ROOT, "distributedhouseholderqr.jl_amd", "csrc", "dhqr_api.hip"), "-o", so])
    return so


def load_emulated_library(so):
    import ctypes
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "dhqr_lib_signatures", os.path.join(ROOT, "distributedhouseholderqr.jl_amd", "_lib.py"))
    sig = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sig)
    L = ctypes.CDLL(so)
    for name, (res, args) in sig.SIGNATURES.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    L.Stats = sig.Stats  # the dhqr_stats structure class the prototypes were declared with
    return L


def emulated_rank(so, nranks, rank):
    """(library, ctx handle, Communicator) of one gloo rank on the EMULATED library: the product's SPMD drivers
    (dhqr_cs_*) run unmodified; broadcast / all-reduce go through the CALLBACK transport into torch.distributed
    (gloo) on the numpy-backed "device" memory."""
    import ctypes
    import importlib
    import torch.distributed as dist
    import __graft_entry__ as g
    g.import_package()
    D = importlib.import_module("dhqr_amd.distributed")
    L = load_emulated_library(so)
    h = ctypes.c_void_p()
    assert L.dhqr_create(ctypes.byref(h), 0) == 0

    def _view(ptr, count):
        return torch.from_numpy(np.ctypeslib.as_array((ctypes.c_double * count).from_address(ptr)))

    def bcast(ptr, nbytes, root):
        dist.broadcast(_view(ptr, nbytes // 8), src=root)

    def allreduce(ptr, count):
        dist.all_reduce(_view(ptr, count))

    comm = D.Communicator.from_callbacks(h, L, nranks, rank, bcast, allreduce)
    return L, h, comm, D


def make_emu_row_backend(so):
    """the product's HipRowBackend (rowsplit.py, dhqr_rs_* entry points) bound to the emulated library"""
    import ctypes
    import importlib
    import __graft_entry__ as g
    g.import_package()
    HipRowBackend = importlib.import_module("dhqr_amd.rowsplit").HipRowBackend

    class _Ctx:
        def __init__(self, L):
            self.handle = ctypes.c_void_p()
            assert L.dhqr_create(ctypes.byref(self.handle), 0) == 0

        def use_torch_stream(self):
            pass

    class EmuRowBackend(HipRowBackend):
        def __init__(self):  # not calling HipRowBackend.__init__ (it needs a GPU)
            self.L = load_emulated_library(so)
            self.ctx = _Ctx(self.L)
            self.device = None
            self.tdev = torch.device("cpu")

    return EmuRowBackend()


class NumpyRowBackend:
    """Test-only numpy stand-in for the product's HipRowBackend (rowsplit.py): same call interface on
    CPU torch tensors, so RowSplitQR's orchestration (active-row bookkeeping, all-reduce / broadcast
    placement, commit, residual) runs under gloo without a GPU."""

    def __init__(self):
        from oracle import dhqr_oracle
        self.orc = dhqr_oracle

    def empty(self, m, n):
        return torch.zeros((max(n, 1), max(m, 1)), dtype=torch.float64).t()

    def zeros(self, n):
        return torch.zeros(n, dtype=torch.float64)

    def izeros(self, n):
        return torch.zeros(n, dtype=torch.int32)

    @staticmethod
    def _cm(t):  # flat buffer -> column-major 128 x 128 numpy view
        return t.numpy()[: NB * NB].reshape((NB, NB), order="F")

    def fill(self, A, rows, cols, seed, gm, row0):
        a = A.numpy()
        for j in range(cols):
            a[:rows, j] = self.orc.u01(seed, np.arange(rows, dtype=np.uint64) + np.uint64(row0 + j * gm))

    def gram(self, X, row, col, rows, G):
        x = X.numpy()[row: row + rows, col: col + NB] if rows > 0 else np.zeros((0, NB))
        self._cm(G)[:] = x.T @ x

    def chol(self, G, R, flag):
        try:
            self._cm(R)[:] = np.linalg.cholesky(self._cm(G)).T
        except np.linalg.LinAlgError:
            flag[0] = 1
            self._cm(R)[:] = np.eye(NB)

    def recon_top(self, A, row, col, R, alpha128, Rref, negMinv):
        top = A.numpy()[row: row + NB, col: col + NB].copy()
        Rm = self._cm(R)
        al, f = np.zeros(NB), np.zeros(NB)
        W = np.zeros((NB, NB))
        Rr = np.zeros((NB, NB))
        for j in range(NB):
            ajj, s = top[j, j], abs(Rm[j, j])
            al[j] = -np.sign(ajj) * s
            f[j] = 1.0 / np.sqrt(s * (s + abs(ajj)))
            vjj = (ajj - al[j]) * f[j]
            rrow = (al[j] / Rm[j, j]) * Rm[j, :]
            W[j, j + 1:] = (top[j, j + 1:] - rrow[j + 1:]) / vjj
            Rr[j, j + 1:] = rrow[j + 1:]
            v = f[j] * top[j + 1:, j]
            top[j + 1:, j + 1:] -= np.outer(v, W[j, j + 1:])
        M = np.triu(W, 1) + np.diag(1.0 / f)
        self._cm(negMinv)[:] = -np.linalg.inv(M)
        alpha128.numpy()[:] = al
        self._cm(Rref)[:] = Rr

    def mul(self, X, row, col, rows, negY, Out):
        o = Out.numpy()
        o[:] = 0.0
        if rows > 0:
            o[:rows, :NB] = X.numpy()[row: row + rows, col: col + NB] @ (-self._cm(negY))

    def fix_top(self, Vw, alpha128, negMinv):
        v = Vw.numpy()
        v[:NB, :NB] += alpha128.numpy()[:, None] * self._cm(negMinv)
        v[:NB, :NB] = np.tril(v[:NB, :NB])

    def commit(self, A, row, col, rows, Vw, diag_owner, Rref):
        if rows <= 0:
            return
        a, v = A.numpy(), Vw.numpy()
        if diag_owner:
            blk = a[row: row + rows, col: col + NB]
            low = np.tril(np.ones((rows, NB), dtype=bool))
            blk[low] = v[:rows, :NB][low]
            up = np.triu(np.ones((NB, NB), dtype=bool), 1)
            blk[:NB][up] = self._cm(Rref)[up]
        else:
            a[row: row + rows, col: col + NB] = v[:rows, :NB]

    def pack(self, A, row, col, rows, Vw, diag_owner):
        v = Vw.numpy()
        v[:] = 0.0
        if rows > 0:
            blk = A.numpy()[row: row + rows, col: col + NB]
            v[:rows, :NB] = np.tril(blk) if diag_owner else blk

    def build_t(self, S, T, Tt):
        Sm = self._cm(S)
        Tm = np.zeros((NB, NB))
        for j in range(NB):
            Tm[:j, j] = -Tm[:j, :j] @ Sm[:j, j]
            Tm[j, j] = 1.0
        self._cm(T)[:] = Tm
        self._cm(Tt)[:] = Tm.T

    def vtc(self, Vw, C, row, col, rows, ncols, W1):
        w = W1.numpy()[: NB * ncols].reshape((NB, ncols), order="F")
        w[:] = 0.0
        if rows > 0:
            w[:] = Vw.numpy()[:rows, :NB].T @ C.numpy()[row: row + rows, col: col + ncols]

    def tw(self, Top, W1, ncols, W2):
        w1 = W1.numpy()[: NB * ncols].reshape((NB, ncols), order="F")
        W2.numpy()[: NB * ncols].reshape((NB, ncols), order="F")[:] = self._cm(Top).T @ w1

    def vw(self, Vw, W2, C, row, col, rows, ncols):
        if rows > 0:
            w2 = W2.numpy()[: NB * ncols].reshape((NB, ncols), order="F")
            C.numpy()[row: row + rows, col: col + ncols] -= Vw.numpy()[:rows, :NB] @ w2

    def backsub(self, A, n, alpha, y):
        a, al, yy = A.numpy(), alpha.numpy(), y.numpy()
        for i in range(n - 1, -1, -1):
            yy[i] = (yy[i] - a[i, i + 1:n] @ yy[i + 1:n]) / al[i]

    def form_r0(self, A, m, n, alpha, W):
        a, w, al = A.numpy(), W.numpy(), alpha.numpy()
        w[:] = 0.0
        for j in range(n):
            w[:j, j] = a[:j, j]
            w[j, j] = al[j]

    def diff_norms(self, X, Y, m, n):
        x, y = X.numpy()[:m, :n], Y.numpy()[:m, :n]
        return float(((x - y) ** 2).sum()), float((x ** 2).sum())
