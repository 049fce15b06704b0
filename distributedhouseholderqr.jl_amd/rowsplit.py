"""Row-split tall-skinny Householder QR (BASELINE configs[4]: 262144 x 4096 Float64 over 8 GPUs): front-end of the
C driver csrc/dhqr_rowsplit.h (dhqr_rs_*).  All orchestration -- all-reduced Gram matrices, the top-block replay on
the rank that holds the diagonal rows, device-side panel verification, the column-by-column fallback across ranks,
residual and solve -- is in libdhqr.so; this module owns the device memory and forwards.  The reference cannot
split rows (`@assert rowrange == 1:size(A,1)`, src:33).  The single-process flavour is MultiGpuQR.rs_*."""
from __future__ import annotations

import ctypes
from typing import Optional

import numpy as np

from . import _lib
from ._lib import NB
from .distributed import Communicator, _P, _TorchMem

try:
    import torch
except Exception:  # pragma: no cover
    torch = None


class RowSplitQR:
    """Usage (every rank):  q = RowSplitQR(m, n, comm); q.fill(seed); q.factor(); r = q.residual(seed)"""

    def __init__(self, m: int, n: int, comm: Optional[Communicator] = None, mem=None, ctx=None):
        if m < n:
            raise ValueError("m >= n required")
        if comm is None:
            from .api import get_context
            ctx = ctx or get_context(torch.cuda.current_device())
            comm = Communicator.from_torch(ctx)
        self.comm, self.L = comm, comm.L
        self.m, self.n = m, n
        self.P, self.rank = comm.nranks, comm.rank
        r0, ml = ctypes.c_int64(), ctypes.c_int64()
        self.L.dhqr_rs_row_range(m, self.P, self.rank, ctypes.byref(r0), ctypes.byref(ml))
        self.row0, self.mloc = r0.value, ml.value
        self.mem = mem if mem is not None else _TorchMem(torch.cuda.current_device())
        self.A = self.mem.empty(max(self.mloc, 2), n)   # mloc x n column-major (at least 2 rows: even leading dimension)
        self.alpha = self.mem.vec(n)

    def _check(self, rc):
        if rc != 0:
            raise _lib.DHQRError(rc, self.L.dhqr_last_error().decode(errors="replace"))

    def _ptrA(self):
        return self.mem.ptr(self.A), self.mem.ld(self.A)

    def fill(self, seed: int):
        self.comm.order_with_torch(self.mem)
        p, ld = self._ptrA()
        self._check(self.L.dhqr_rs_fill_uniform_f64(self.comm.handle, p, self.m, self.n, ld, seed))
        return self

    def factor(self):
        self.comm.order_with_torch(self.mem)
        p, ld = self._ptrA()
        self._check(self.L.dhqr_rs_factor_f64(self.comm.handle, p, self.m, self.n, ld, self.mem.ptr(self.alpha)))
        return self

    def residual(self, seed: int) -> float:
        self.comm.order_with_torch(self.mem)
        p, ld = self._ptrA()
        B, A0 = self.mem.empty(max(self.mloc, 1), self.n), self.mem.empty(max(self.mloc, 1), self.n)
        out = ctypes.c_double()
        self._check(self.L.dhqr_rs_residual_f64(self.comm.handle, p, self.m, self.n, ld, self.mem.ptr(self.alpha), seed,
                                                self.mem.ptr(B), self.mem.ptr(A0), ctypes.byref(out)))
        return out.value

    def solve(self, b_loc):
        """`H \\ b`: b_loc = this rank's rows of b (length mloc, not modified); returns x (length n) on every rank"""
        self.comm.order_with_torch(self.mem)
        p, ld = self._ptrA()
        y = self.mem.vec(max(self.mloc, 1))
        x = self.mem.vec(self.n)
        if self.mloc:
            if isinstance(y, np.ndarray):
                y[: self.mloc] = np.asarray(b_loc)
            else:
                y[: self.mloc].copy_(b_loc if torch.is_tensor(b_loc) else torch.as_tensor(np.asarray(b_loc)))
        self._check(self.L.dhqr_rs_solve_f64(self.comm.handle, p, self.m, self.n, ld, self.mem.ptr(self.alpha),
                                             self.mem.ptr(y), self.mem.ptr(x)))
        return x[: self.n].copy() if isinstance(x, np.ndarray) else x[: self.n].clone()

    def local_numpy(self):
        return self.mem.to_host(self.A)[: self.mloc, :], self.mem.to_host(self.alpha)[: self.n]
