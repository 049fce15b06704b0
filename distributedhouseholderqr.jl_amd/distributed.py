"""Multi-GPU front-ends of the column-split Householder QR.  ALL orchestration (ownership, look-ahead, panel
broadcast, two-panel updates, solve, layout conversion) lives in libdhqr.so (csrc/dhqr_dist.h, dhqr_comm.h,
dhqr_mg.h); this module only creates communicators, owns device memory and forwards to the C ABI.

  reference (src/DistributedHouseholderQR.jl)                 here
  qr!(A::Matrix) on one process, many GPUs (no analogue)      MultiGpuQR(ndev) / qr_multi_(A, ndev)   -> dhqr_mg_*
  qr!(A::DArray) :115-120,311-315, one worker per block        ColumnCyclicQR(m, n, comm) / qr_darray_ -> dhqr_cs_*
  per-column `@spawnat` fan-out of Hj :141-143                 one broadcast per panel inside the library
  α::SharedArray :301-304                                      α replicated on every rank
  sum(fetch.(futures)) of partial dots :262-266                one all-reduce per block inside the library

Communicator(...) binds this process's rank to a transport: `Communicator.from_torch()` bootstraps RCCL
(ncclCommInitRank) by shipping the 128-byte unique id through torch.distributed -- the role Distributed.jl plays
for the Julia module; `Communicator.from_callbacks()` plugs any host-side broadcast / all-reduce in (used by the
CPU tests with gloo on the emulated library).  PyTorch is plumbing only: device memory and process bootstrap.
"""
from __future__ import annotations

import ctypes
import math
from typing import Optional

import numpy as np

from . import _lib
from ._lib import NB, check

try:
    import torch
except Exception:  # pragma: no cover
    torch = None

_P = ctypes.c_void_p


class _HostMem:
    """numpy-backed "device" memory for the emulated library (tests): hipMalloc is malloc there."""

    def empty(self, m, n):
        return np.zeros((m, max(n, 1)), order="F")

    def vec(self, n):
        return np.zeros(max(n, 1))

    @staticmethod
    def ptr(a, col=0):
        return _P(a.ctypes.data + 8 * col * (a.strides[1] // 8 if a.ndim == 2 else 1))

    @staticmethod
    def ld(a):
        return a.strides[1] // 8 if a.shape[1] > 1 else max(a.shape[0], 1)

    def to_host(self, a):
        return np.array(a, copy=True)

    def from_host(self, dst, src):
        dst[...] = src


class _TorchMem:
    def __init__(self, device):
        self.device = torch.device("cuda", device)

    def empty(self, m, n):
        return torch.empty((max(n, 1), m), dtype=torch.float64, device=self.device).t()

    def vec(self, n):
        return torch.zeros(max(n, 1), dtype=torch.float64, device=self.device)

    @staticmethod
    def ptr(a, col=0):
        return _P(a.data_ptr() + 8 * col * (a.stride(1) if a.dim() == 2 else 1))

    @staticmethod
    def ld(a):
        return a.stride(1) if a.shape[1] > 1 else max(a.shape[0], 1)

    def to_host(self, a):
        return a.cpu().numpy().copy()

    def from_host(self, dst, src):
        dst.copy_(torch.as_tensor(src, dtype=torch.float64).to(dst.device))


class Communicator:
    """One rank of a multi-GPU job (dhqr_comm).  `ctx_handle` is the dhqr_ctx of this rank's GPU."""

    def __init__(self, handle, lib, nranks, rank, keepalive=()):
        self.handle, self.L, self.nranks, self.rank = handle, lib, nranks, rank
        self._keep = keepalive
        self.ctx = None

    def order_with_torch(self, mem):
        """Device tensors (torch memory) are produced and consumed on torch's current stream: make it the library's
        caller stream for the call that follows (as api.py does for every entry point), so that e.g. the copy of b into
        the work vector is ordered before dhqr_cs_solve_f64 reads it.  No-op for host-memory stand-ins."""
        if self.ctx is not None and isinstance(mem, _TorchMem):
            self.ctx.use_torch_stream()

    @classmethod
    def from_torch(cls, ctx, group=None):
        """RCCL communicator over the ranks of a torch.distributed group (any backend carries the id)."""
        import torch.distributed as dist
        L = _lib.lib()
        if not dist.is_initialized():
            nranks, rank = 1, 0
        else:
            nranks, rank = dist.get_world_size(group), dist.get_rank(group)
        idbuf = (ctypes.c_char * _lib.UNIQUE_ID_BYTES)()
        if nranks > 1:
            box = [None]
            if rank == 0:
                check(L.dhqr_comm_unique_id(idbuf))
                box[0] = bytes(idbuf.raw)
            src = dist.get_global_rank(group, 0) if group is not None else 0
            dist.broadcast_object_list(box, src=src, group=group)
            idbuf.raw = box[0]
        h = _P()
        check(L.dhqr_comm_create_rank(ctypes.byref(h), ctx.handle, nranks, rank, idbuf))
        comm = cls(h, L, nranks, rank)
        comm.ctx = ctx  # device tensors of this rank are ordered with the library through torch's current stream
        return comm

    @classmethod
    def from_callbacks(cls, ctx_handle, lib, nranks, rank, bcast, allreduce):
        """bcast(ptr:int, nbytes:int, root:int) / allreduce(ptr:int, count:int) -> None (raise on error)."""
        def _b(user, dbuf, nbytes, root, stream):
            try:
                bcast(dbuf, nbytes, root)
                return 0
            except Exception:  # pragma: no cover - surfaced as DHQR_ECOMM
                import traceback
                traceback.print_exc()
                return 1

        def _a(user, dbuf, count, stream):
            try:
                allreduce(dbuf, count)
                return 0
            except Exception:  # pragma: no cover
                import traceback
                traceback.print_exc()
                return 1

        cb_b, cb_a = _lib.BCAST_FN(_b), _lib.ALLREDUCE_FN(_a)
        h = _P()
        rc = lib.dhqr_comm_create_callbacks(ctypes.byref(h), ctx_handle, nranks, rank, cb_b, cb_a, None)
        if rc != 0:
            raise _lib.DHQRError(rc, lib.dhqr_last_error().decode(errors="replace"))
        return cls(h, lib, nranks, rank, keepalive=(cb_b, cb_a))

    def bcast_tuning(self) -> dict:
        """which algorithm the panel broadcasts use and what the trial at creation measured (RCCL transport)"""
        a, r, g = ctypes.c_int32(), ctypes.c_double(), ctypes.c_double()
        check(self.L.dhqr_comm_get_bcast_tuning(self.handle, ctypes.byref(a), ctypes.byref(r), ctypes.byref(g)))
        return {"algorithm": {0: "ncclBroadcast", 1: "scatter + all-gather"}[a.value], "trial_ms_16MiB_ncclBroadcast": r.value,
                "trial_ms_16MiB_scatter_allgather": g.value}

    def counters(self) -> dict:
        """collectives carried so far (dhqr_comm_counters): all-reduces are counted as issued, also at one rank"""
        o = (ctypes.c_int64 * 4)()
        check(self.L.dhqr_comm_counters(self.handle, o))
        return {"n_bcast": o[0], "bytes_bcast": o[1], "n_allreduce": o[2], "bytes_allreduce": o[3]}

    def timing(self, on: int = -1) -> dict:
        """device time of this rank's collectives since the last call (dhqr_comm_timing): hipEvent pairs around every
        broadcast / all-reduce, the wait for the peers included; on = 1 / 0 starts / stops collecting"""
        o = (ctypes.c_double * 4)()
        check(self.L.dhqr_comm_timing(self.handle, on, o))
        return {"bcast_ms": o[0], "n_bcast": int(o[1]), "allreduce_ms": o[2], "n_allreduce": int(o[3])}

    def rccl_nranks(self) -> dict:
        """rank counts RCCL itself reports (ncclCommCount) for the main channel and the row-split lane's channel"""
        a, b = ctypes.c_int32(), ctypes.c_int32()
        check(self.L.dhqr_comm_rccl_nranks(self.handle, ctypes.byref(a), ctypes.byref(b)))
        return {"main": a.value, "lane": b.value}

    def close(self):
        if self.handle:
            self.L.dhqr_comm_destroy(self.handle)
            self.handle = _P()


class ColumnCyclicQR:
    """This rank's part of an m x n matrix split block-cyclically by columns + the replicated α
    (householder!(A::DArray, α), src:115-148).  Every rank makes the same calls.

        q = ColumnCyclicQR(m, n, comm); q.fill(seed); q.factor(); r = q.residual(seed); x = q.solve(b)
    """

    def __init__(self, m: int, n: int, comm: Optional[Communicator] = None, nb: int = NB, mem=None, ctx=None):
        if m < n:
            raise ValueError("m >= n required")
        if nb != NB:
            raise ValueError(f"panel width is fixed at {NB}")
        if comm is None:  # product default: RCCL over the torch.distributed world (or a single rank)
            from .api import get_context
            ctx = ctx or get_context(torch.cuda.current_device())
            comm = Communicator.from_torch(ctx)
        self.comm, self.L = comm, comm.L
        self.m, self.n, self.nb = m, n, nb
        self.P, self.rank = comm.nranks, comm.rank
        self.ncl = self.L.dhqr_cs_local_cols(n, self.P, self.rank)
        self.mem = mem if mem is not None else _TorchMem(torch.cuda.current_device())
        self.A = self.mem.empty(m, self.ncl)
        self.alpha = self.mem.vec(n)
        self._scratch = None

    def _check(self, rc):
        if rc != 0:
            raise _lib.DHQRError(rc, self.L.dhqr_last_error().decode(errors="replace"))

    def _ptrA(self):
        return self.mem.ptr(self.A), self.mem.ld(self.A)

    def fill(self, seed: int):
        """synthetic input: the local columns of A[i,j] = u01(seed, i + j*m)"""
        self.comm.order_with_torch(self.mem)
        p, ld = self._ptrA()
        self._check(self.L.dhqr_cs_fill_uniform_f64(self.comm.handle, p, self.m, self.n, ld, seed))
        return self

    def factor(self):
        self.comm.order_with_torch(self.mem)
        p, ld = self._ptrA()
        self._check(self.L.dhqr_cs_factor_f64(self.comm.handle, p, self.m, self.n, ld, self.mem.ptr(self.alpha)))
        return self

    def residual(self, seed: int) -> float:
        """||A - QR||_F / ||A||_F with A regenerated from `seed`"""
        self.comm.order_with_torch(self.mem)
        p, ld = self._ptrA()
        W, A0 = self.mem.empty(self.m, self.ncl), self.mem.empty(self.m, self.ncl)
        out = ctypes.c_double()
        self._check(self.L.dhqr_cs_residual_f64(self.comm.handle, p, self.m, self.n, ld, self.mem.ptr(self.alpha), seed,
                                                self.mem.ptr(W), self.mem.ptr(A0), ctypes.byref(out)))
        return out.value

    def solve(self, b):
        """`H \\ b` on the column split (src:317-321, 226-282): b (length m, the same on every rank) is not
        modified; returns x (length n) on every rank."""
        self.comm.order_with_torch(self.mem)
        p, ld = self._ptrA()
        y = self.mem.vec(self.m)
        if isinstance(y, np.ndarray):
            y[: self.m] = np.asarray(b)
        else:
            y[: self.m].copy_(b if torch.is_tensor(b) else torch.as_tensor(np.asarray(b)))
        work = self.mem.vec(self.m + NB + 16)
        self._check(self.L.dhqr_cs_solve_f64(self.comm.handle, p, self.m, self.n, ld, self.mem.ptr(self.alpha),
                                             self.mem.ptr(y), self.mem.ptr(work)))
        return y[: self.n].copy() if isinstance(y, np.ndarray) else y[: self.n].clone()

    # -- the reference's DArray layout: ONE contiguous column block per process (test/runtests.jl:71)
    def contiguous_range(self, rank=None):
        lo, hi = ctypes.c_int64(), ctypes.c_int64()
        self.L.dhqr_cs_contiguous_range(self.n, self.P, self.rank if rank is None else rank, ctypes.byref(lo), ctypes.byref(hi))
        return range(lo.value, hi.value)

    def _stage(self):
        if self._scratch is None:
            self._scratch = self.mem.empty(self.m, max(self.n // self.P + 1, NB))
        return self._scratch

    def load_contiguous_blocks(self, local_block):
        """local_block: this rank's m x len(contiguous_range()) column-major block (device memory)"""
        self.comm.order_with_torch(self.mem)
        p, ld = self._ptrA()
        w = len(self.contiguous_range())
        if tuple(local_block.shape) != (self.m, w):
            raise ValueError(f"rank {self.rank} must pass a {self.m} x {w} block, got {tuple(local_block.shape)}")
        bp = self.mem.ptr(local_block) if w else _P()
        self._check(self.L.dhqr_cs_load_contiguous_f64(self.comm.handle, p, self.m, self.n, ld, bp,
                                                       self.mem.ld(local_block) if w else self.m, self.mem.ptr(self._stage())))
        return self

    def store_contiguous_blocks(self, local_block):
        self.comm.order_with_torch(self.mem)
        p, ld = self._ptrA()
        w = len(self.contiguous_range())
        bp = self.mem.ptr(local_block) if w else _P()
        self._check(self.L.dhqr_cs_store_contiguous_f64(self.comm.handle, p, self.m, self.n, ld, bp,
                                                        self.mem.ld(local_block) if w else self.m, self.mem.ptr(self._stage())))
        return local_block

    def local_numpy(self):
        """(local block, α) as host arrays"""
        return self.mem.to_host(self.A)[:, : self.ncl], self.mem.to_host(self.alpha)[: self.n]


def qr_darray_(local_block, n: int, comm: Optional[Communicator] = None, mem=None):
    """qr!(A::DArray) (src:115-120, 311-315) for callers that hold the reference's layout: every rank passes ITS
    contiguous column block (device memory, column-major) of the m x n matrix; the block is overwritten with the
    factored columns and (factor object, replicated α) is returned; q.solve(b) is `H \\ b`."""
    m = local_block.shape[0]
    q = ColumnCyclicQR(m, n, comm=comm, mem=mem)
    q.load_contiguous_blocks(local_block)
    q.factor()
    q.store_contiguous_blocks(local_block)
    return q, q.alpha


def qr_darray_c64_(local_block, m: int, n: int, comm: Communicator):
    """qr!(A::DArray{ComplexF64}) (src:115-120, 311-315) for one process: `local_block` is this rank's CONTIGUOUS column
    block (complex128 numpy, column-major, m x w_r -- the reference's DistributedArrays layout) and is overwritten with the
    factored columns; returns the replicated α (dhqr_cs_qr_darray_c64: cyclic 64-column blocks inside)."""
    if not isinstance(local_block, np.ndarray) or local_block.dtype != np.complex128 or local_block.ndim != 2:
        raise TypeError("complex128 numpy block expected")
    if local_block.shape[1] and not local_block.flags.f_contiguous:
        raise ValueError("column-major block expected")
    if local_block.shape[0] != m:
        raise ValueError(f"the block must have {m} rows, got {local_block.shape[0]}")
    lo, hi = ctypes.c_int64(), ctypes.c_int64()
    comm.L.dhqr_cs_contiguous_range(n, comm.nranks, comm.rank, ctypes.byref(lo), ctypes.byref(hi))
    if local_block.shape[1] != hi.value - lo.value:
        raise ValueError(f"rank {comm.rank} must pass columns [{lo.value}, {hi.value}), got {local_block.shape[1]} columns")
    al = np.zeros(n, dtype=np.complex128)
    ptr = local_block.ctypes.data_as(_P) if local_block.shape[1] else None
    rc = comm.L.dhqr_cs_qr_darray_c64(comm.handle, ptr, m, n, m, al.ctypes.data_as(_P))
    if rc != 0:
        raise _lib.DHQRError(rc, comm.L.dhqr_last_error().decode(errors="replace"))
    return al


def _host_block(local_block, m, n, comm, dtype):
    """validate one process's contiguous HOST column block of an m x n DArray (the reference's DistributedArrays layout)"""
    if not isinstance(local_block, np.ndarray) or local_block.dtype != dtype or local_block.ndim != 2:
        raise TypeError(f"{np.dtype(dtype).name} numpy block expected")
    if local_block.shape[1] and not local_block.flags.f_contiguous:
        raise ValueError("column-major block expected")
    if local_block.shape[0] != m:
        raise ValueError(f"the block must have {m} rows, got {local_block.shape[0]}")
    lo, hi = ctypes.c_int64(), ctypes.c_int64()
    comm.L.dhqr_cs_contiguous_range(n, comm.nranks, comm.rank, ctypes.byref(lo), ctypes.byref(hi))
    if local_block.shape[1] != hi.value - lo.value:
        raise ValueError(f"rank {comm.rank} must pass columns [{lo.value}, {hi.value}), got {local_block.shape[1]} columns")
    return local_block.ctypes.data_as(_P) if local_block.shape[1] else None


def ldiv_darray_(local_block, m: int, n: int, alpha, b, comm: Communicator):
    """`qrA \\ b` for qrA = qr!(A::DArray) (src:317-321 with src:226-230, 256-270; test/runtests.jl:77-78) for one process:
    `local_block` is this rank's contiguous FACTORED column block (float64 or complex128 numpy, column-major, m x w_r), `alpha`
    the replicated diagonal of R, `b` the right-hand side (m, the same on every rank).  Nothing is modified; returns x (n) on
    every rank (dhqr_cs_ldiv_darray_f64 / _c64: one collective call, what a Julia worker binds)."""
    dtype = np.complex128 if np.iscomplexobj(local_block) else np.float64
    ptr = _host_block(local_block, m, n, comm, dtype)
    al = np.ascontiguousarray(alpha, dtype=dtype).reshape(-1)
    bb = np.ascontiguousarray(b, dtype=dtype).reshape(-1)
    if al.size != n or bb.size != m:
        raise ValueError(f"alpha must have {n} and b {m} elements, got {al.size} and {bb.size}")
    x = np.zeros(n, dtype=dtype)
    fn = comm.L.dhqr_cs_ldiv_darray_c64 if dtype == np.complex128 else comm.L.dhqr_cs_ldiv_darray_f64
    rc = fn(comm.handle, ptr, m, n, m, al.ctypes.data_as(_P), bb.ctypes.data_as(_P), x.ctypes.data_as(_P))
    if rc != 0:
        raise _lib.DHQRError(rc, comm.L.dhqr_last_error().decode(errors="replace"))
    return x


class MultiGpuQR:
    """Single-process multi-GPU handle (dhqr_mg_*): one host thread per device inside the library.
    `qr!(A; ndev)` of the Julia module binds the same entry points."""

    def __init__(self, ndev: Optional[int] = None, devices=None, lib=None):
        self.L = lib if lib is not None else _lib.lib()
        if devices is None:
            if ndev is None:
                cnt = ctypes.c_int32()
                self._check(self.L.dhqr_device_count(ctypes.byref(cnt)))
                ndev = cnt.value
            devices = list(range(ndev))
        self.devices = list(devices)
        arr = (ctypes.c_int32 * len(self.devices))(*self.devices)
        self._h = _P()
        self._check(self.L.dhqr_mg_create(ctypes.byref(self._h), arr, len(self.devices)))
        self.m = self.n = 0

    def _check(self, rc):
        if rc != 0:
            raise _lib.DHQRError(rc, self.L.dhqr_last_error().decode(errors="replace"))

    @property
    def ndev(self):
        return len(self.devices)

    @property
    def transport(self) -> str:
        t = ctypes.c_int32()
        self._check(self.L.dhqr_mg_info(self._h, None, ctypes.byref(t), None, None))
        return {0: "self", 1: "rccl", 2: "local-peer-copy", 3: "callback"}[t.value]

    def bcast_tuning(self) -> dict:
        """which algorithm the panel broadcasts use and what the trial at creation measured (RCCL transport)"""
        a, r, g = ctypes.c_int32(), ctypes.c_double(), ctypes.c_double()
        self._check(self.L.dhqr_mg_get_bcast_tuning(self._h, ctypes.byref(a), ctypes.byref(r), ctypes.byref(g)))
        return {"algorithm": {0: "ncclBroadcast", 1: "scatter + all-gather"}[a.value], "trial_ms_16MiB_ncclBroadcast": r.value,
                "trial_ms_16MiB_scatter_allgather": g.value}

    def comm_counters(self, rank: int = 0) -> dict:
        """collectives one rank's communicator has carried (dhqr_mg_comm_counters)"""
        o = (ctypes.c_int64 * 4)()
        self._check(self.L.dhqr_mg_comm_counters(self._h, rank, o))
        return {"n_bcast": o[0], "bytes_bcast": o[1], "n_allreduce": o[2], "bytes_allreduce": o[3]}

    def comm_timing(self, rank: int = 0, on: int = -1) -> dict:
        """device time of one rank's collectives since the last call (dhqr_mg_comm_timing)"""
        o = (ctypes.c_double * 4)()
        self._check(self.L.dhqr_mg_comm_timing(self._h, rank, on, o))
        return {"bcast_ms": o[0], "n_bcast": int(o[1]), "allreduce_ms": o[2], "n_allreduce": int(o[3])}

    def rccl_nranks(self) -> dict:
        a, b = ctypes.c_int32(), ctypes.c_int32()
        self._check(self.L.dhqr_mg_rccl_nranks(self._h, ctypes.byref(a), ctypes.byref(b)))
        return {"main": a.value, "lane": b.value}

    # device-resident path (what bench.py times)
    def alloc(self, m, n):
        self._check(self.L.dhqr_mg_alloc_f64(self._h, m, n))
        self.m, self.n = m, n
        return self

    def fill(self, seed):
        self._check(self.L.dhqr_mg_fill_uniform_f64(self._h, seed))
        return self

    def factor(self):
        self._check(self.L.dhqr_mg_factor_f64(self._h))
        return self

    def residual(self, seed) -> float:
        out = ctypes.c_double()
        self._check(self.L.dhqr_mg_residual_f64(self._h, seed, ctypes.byref(out)))
        return out.value

    def download(self):
        H = np.zeros((self.m, self.n), order="F")
        al = np.zeros(self.n)
        self._check(self.L.dhqr_mg_download_f64(self._h, H.ctypes.data_as(_P), self.m, al.ctypes.data_as(_P)))
        return H, al

    def _host_vec(self, b, length, what):
        """the C entry points take bare pointers: a short vector would be a host out-of-bounds read"""
        b = np.ascontiguousarray(b, dtype=np.float64).reshape(-1)
        if b.size != length:
            raise ValueError(f"{what} must have {length} elements, got {b.size}")
        return b

    def solve(self, b):
        b = self._host_vec(b, self.m, "b")
        x = np.zeros(self.n)
        self._check(self.L.dhqr_mg_solve_f64(self._h, b.ctypes.data_as(_P), x.ctypes.data_as(_P)))
        return x

    # host-in / host-out drop-ins
    def qr_(self, A):
        """qr!(A) over all devices: A (float64 or complex128 numpy, m x n) is factored in place; returns (A, α).
        complex128: cyclic blocks of 64 columns over the devices (dhqr_mg_qr_c64); `ldiv` of the single-GPU API
        (dhqr_ldiv_c64) solves with the result."""
        if not isinstance(A, np.ndarray) or A.dtype not in (np.float64, np.complex128) or A.ndim != 2:
            raise TypeError("float64 or complex128 numpy matrix expected")
        m, n = A.shape
        if m < n:
            raise ValueError("m >= n required")
        if A.dtype == np.complex128:
            F = A if A.flags.f_contiguous else np.asfortranarray(A)
            al = np.zeros(n, dtype=np.complex128)
            lda = F.strides[1] // 16 if n > 1 else max(m, 1)
            self._check(self.L.dhqr_mg_qr_c64(self._h, F.ctypes.data_as(_P), m, n, lda, al.ctypes.data_as(_P)))
            if F is not A:
                A[...] = F
            return A, al
        F = A if A.flags.f_contiguous else np.asfortranarray(A)
        al = np.zeros(n)
        lda = F.strides[1] // 8 if n > 1 else max(m, 1)
        self._check(self.L.dhqr_mg_qr_f64(self._h, F.ctypes.data_as(_P), m, n, lda, al.ctypes.data_as(_P)))
        if F is not A:
            A[...] = F
        self.m, self.n = m, n
        return A, al

    def ldiv(self, A, alpha, b):
        """`H \\ b` (src:317-321) for a factored HOST matrix over the handle's devices.  complex128: cyclic blocks of 64
        columns, Q'b and back substitution distributed like the reference's (dhqr_mg_ldiv_c64 -> zcs_solve)."""
        m, n = A.shape
        if isinstance(A, np.ndarray) and A.dtype == np.complex128:
            F = A if A.flags.f_contiguous else np.asfortranarray(A)
            al = np.ascontiguousarray(alpha, dtype=np.complex128).reshape(-1)
            bb = np.ascontiguousarray(b, dtype=np.complex128).reshape(-1)
            if al.size != n or bb.size != m:
                raise ValueError(f"alpha must have {n} and b {m} elements, got {al.size} and {bb.size}")
            x = np.zeros(n, dtype=np.complex128)
            lda = F.strides[1] // 16 if n > 1 else max(m, 1)
            self._check(self.L.dhqr_mg_ldiv_c64(self._h, F.ctypes.data_as(_P), m, n, lda, al.ctypes.data_as(_P),
                                                bb.ctypes.data_as(_P), x.ctypes.data_as(_P)))
            return x
        F = A if A.flags.f_contiguous else np.asfortranarray(A)
        x = np.zeros(n)
        bb = self._host_vec(b, m, "b")
        alpha = self._host_vec(alpha, n, "alpha")
        lda = F.strides[1] // 8 if n > 1 else max(m, 1)
        self._check(self.L.dhqr_mg_ldiv_f64(self._h, F.ctypes.data_as(_P), m, n, lda,
                                            alpha.ctypes.data_as(_P), bb.ctypes.data_as(_P),
                                            x.ctypes.data_as(_P)))
        self.m, self.n = m, n
        return x

    # row split (BASELINE configs[4]): device-resident tall-skinny matrix, 128-row aligned slabs over the devices
    def rs_alloc(self, m, n):
        self._check(self.L.dhqr_mg_rs_alloc_f64(self._h, m, n))
        self.m, self.n = m, n
        return self

    def rs_fill(self, seed):
        self._check(self.L.dhqr_mg_rs_fill_uniform_f64(self._h, seed))
        return self

    def rs_factor(self):
        self._check(self.L.dhqr_mg_rs_factor_f64(self._h))
        return self

    def rs_residual(self, seed) -> float:
        out = ctypes.c_double()
        self._check(self.L.dhqr_mg_rs_residual_f64(self._h, seed, ctypes.byref(out)))
        return out.value

    def rs_upload(self, A):
        F = np.asfortranarray(A, dtype=np.float64)
        if F.ndim != 2 or F.shape != (self.m, self.n):
            raise ValueError(f"rs_upload: matrix must be {self.m} x {self.n} (rs_alloc), got {F.shape}")
        self._check(self.L.dhqr_mg_rs_transfer_f64(self._h, F.ctypes.data_as(_P), F.shape[0], None, 1))
        return self

    def rs_download(self):
        H = np.zeros((self.m, self.n), order="F")
        al = np.zeros(self.n)
        self._check(self.L.dhqr_mg_rs_transfer_f64(self._h, H.ctypes.data_as(_P), self.m, al.ctypes.data_as(_P), 0))
        return H, al

    def rs_solve(self, b):
        b = self._host_vec(b, self.m, "b")
        x = np.zeros(self.n)
        self._check(self.L.dhqr_mg_rs_solve_f64(self._h, b.ctypes.data_as(_P), x.ctypes.data_as(_P)))
        return x

    # statistics
    def set_profiling(self, on):
        self._check(self.L.dhqr_mg_set_profiling(self._h, 1 if on else 0))

    def reset_stats(self):
        self._check(self.L.dhqr_mg_reset_stats(self._h))

    def stats(self, rank=0):
        st = _lib.Stats()
        a, b, c = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        self._check(self.L.dhqr_mg_get_stats(self._h, rank, ctypes.byref(st), ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        d = st.asdict()
        d.update(panels_fast=a.value, panels_fallback=b.value, bytes_bcast=c.value)
        return d

    def close(self):
        if self._h:
            self.L.dhqr_mg_destroy(self._h)
            self._h = _P()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


def qr_multi_(A, ndev: Optional[int] = None):
    """qr!(A; ndev): host matrix factored in place on `ndev` GPUs of this process; returns the reference's struct"""
    from .api import DistributedHouseholderQRStruct
    mg = MultiGpuQR(ndev)
    try:
        A, al = mg.qr_(A)
    finally:
        mg.close()
    return DistributedHouseholderQRStruct(A, al)
