"""Row-split tall-skinny Householder QR over torch.distributed (BASELINE configs[4]:
262144 x 4096 over 8 MI355X, "TSQR-style row split, RCCL all-reduce of partial dots").

The reference cannot split rows at all (`@assert rowrange == 1:size(A,1)`,
src/DistributedHouseholderQR.jl:33): every partial dot of src:208 would need a cross-process
reduction per column.  The R-first panel algorithm (csrc/dhqr_recon.h) needs those reductions only
per 128-column PANEL, as sums of small matrices:

    per panel:  all-reduce  G = sum_r P_r' P_r         (128 x 128)     -> R = chol(G) on every rank
                broadcast   (M^{-1}, alpha) from the owner of the diagonal rows (128 x 128 + 128)
                all-reduce  S = sum_r V_r' V_r         (128 x 128)     -> check ||v_j||^2 = 2, T
                all-reduce  W = sum_r V_r' C_r         (128 x ncols)   -> the cross-partition partial dots
    local:      V_r = P_r M^{-1},   C_r -= V_r (T' W)

Result: exactly the reference's factor format, distributed by rows -- V rows live where the matrix
rows live, R (strict upper) sits in the top n rows of rank 0, alpha is replicated.
Restrictions of this first version: n % 128 == 0, contiguous row slabs, rank 0 owns rows 0..n-1 (all
diagonal blocks), no column-by-column fallback (an ill-conditioned panel is retried with the Gram
matrix of the once-orthogonalised panel, CholeskyQR2, and otherwise raises).
All compute goes through a backend; tests inject a numpy backend to run this orchestration under gloo.
"""
from __future__ import annotations

import ctypes
import math

import torch
import torch.distributed as dist

from . import _lib
from ._lib import NB, check
from .api import empty_colmajor, get_context

_NN = NB * NB


class HipRowBackend:
    """device tensors + libdhqr.so (dhqr_rs_* entry points); everything on torch's current stream"""

    def __init__(self, device: int):
        self.device = device
        self.ctx = get_context(device)
        self.L = _lib.lib()
        self.tdev = torch.device("cuda", device)

    def empty(self, m, n):
        return empty_colmajor(max(m, 1), max(n, 1), self.tdev)

    def zeros(self, n):
        return torch.zeros(n, dtype=torch.float64, device=self.tdev)

    def izeros(self, n):
        return torch.zeros(n, dtype=torch.int32, device=self.tdev)

    def _p(self, A, row, col):
        return ctypes.c_void_p(A.data_ptr() + 8 * (row + col * A.stride(1)))

    def _v(self, t):
        return ctypes.c_void_p(t.data_ptr())

    def _go(self):
        self.ctx.use_torch_stream()
        return self.ctx.handle

    def fill(self, A, rows, cols, seed, gm, row0):
        if rows > 0:
            check(self.L.dhqr_fill_uniform_f64(self._go(), self._p(A, 0, 0), rows, cols, A.stride(1), seed, gm, row0,
                                               NB, 1, 0))

    def gram(self, X, row, col, rows, G):
        check(self.L.dhqr_rs_gram_f64(self._go(), self._p(X, row, col), X.stride(1), rows, self._v(G)))

    def chol(self, G, R, flag):
        check(self.L.dhqr_rs_chol_f64(self._go(), self._v(G), self._v(R), self._v(flag)))

    def recon_top(self, A, row, col, R, alpha128, Rref, negMinv):
        check(self.L.dhqr_rs_recon_top_f64(self._go(), self._p(A, row, col), A.stride(1), self._v(R),
                                           self._v(alpha128), self._v(Rref), self._v(negMinv)))

    def mul(self, X, row, col, rows, negY, Out):
        check(self.L.dhqr_rs_mul_f64(self._go(), self._p(X, row, col), X.stride(1), rows, self._v(negY),
                                     self._p(Out, 0, 0), Out.stride(1)))

    def fix_top(self, Vw, alpha128, negMinv):
        check(self.L.dhqr_rs_fix_top_f64(self._go(), self._p(Vw, 0, 0), Vw.stride(1), self._v(alpha128),
                                         self._v(negMinv)))

    def commit(self, A, row, col, rows, Vw, diag_owner, Rref):
        check(self.L.dhqr_rs_commit_f64(self._go(), self._p(A, row, col), A.stride(1), rows, self._p(Vw, 0, 0),
                                        Vw.stride(1), 1 if diag_owner else 0, self._v(Rref)))

    def pack(self, A, row, col, rows, Vw, diag_owner):
        check(self.L.dhqr_rs_pack_f64(self._go(), self._p(A, row, col), A.stride(1), rows, self._p(Vw, 0, 0),
                                      Vw.stride(1), 1 if diag_owner else 0))

    def build_t(self, S, T, Tt):
        check(self.L.dhqr_rs_build_t_f64(self._go(), self._v(S), NB, self._v(T), self._v(Tt)))

    def vtc(self, Vw, C, row, col, rows, ncols, W1):
        check(self.L.dhqr_rs_vtc_f64(self._go(), self._p(Vw, 0, 0), Vw.stride(1), self._p(C, row, col), C.stride(1),
                                     rows, ncols, self._v(W1)))

    def tw(self, Top, W1, ncols, W2):
        check(self.L.dhqr_rs_tw_f64(self._go(), self._v(Top), self._v(W1), ncols, self._v(W2)))

    def vw(self, Vw, W2, C, row, col, rows, ncols):
        check(self.L.dhqr_rs_vw_f64(self._go(), self._p(Vw, 0, 0), Vw.stride(1), self._v(W2), self._p(C, row, col),
                                    C.stride(1), rows, ncols))

    def backsub(self, A, n, alpha, y):
        """in place: y[0:n] <- R^{-1} y[0:n] with R = strict upper of A[0:n, 0:n] and diag alpha (src:244-254)"""
        h = self._go()
        for hi in range(n, 0, -NB):
            lo = max(0, hi - NB)
            check(self.L.dhqr_backsub_block_f64(h, self._p(A, 0, 0), A.stride(1), self._v(alpha), self._v(y), lo, hi, 1, 1))

    def form_r0(self, A, m, n, alpha, W):
        check(self.L.dhqr_form_r0_f64(self._go(), self._p(A, 0, 0), m, n, A.stride(1), self._v(alpha),
                                      self._p(W, 0, 0), W.stride(1), NB, 1, 0))

    def diff_norms(self, X, Y, m, n):
        out = (ctypes.c_double * 2)()
        if m > 0:
            check(self.L.dhqr_diff_norms_f64(self._go(), self._p(X, 0, 0), X.stride(1), self._p(Y, 0, 0), Y.stride(1),
                                             m, n, out))
        return out[0], out[1]


class RowSplitQR:
    """Usage (every rank):  q = RowSplitQR(m, n); q.fill(seed); q.factor(); r = q.residual(seed)"""

    def __init__(self, m: int, n: int, group=None, backend=None, tol: float = 2e-12):
        if n % NB:
            raise ValueError(f"row-split path needs n % {NB} == 0")
        self.m, self.n, self.group, self.tol = m, n, group, tol
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.P = dist.get_world_size(group) if dist.is_initialized() else 1
        q, r = divmod(m, self.P)
        sizes = [q + (1 if i < r else 0) for i in range(self.P)]
        self.row0 = sum(sizes[: self.rank])
        self.mloc = sizes[self.rank]
        if sizes[0] < n:
            raise ValueError("rank 0 must own the first n rows (all diagonal blocks): m / P >= n required")
        be = self.be = backend if backend is not None else HipRowBackend(torch.cuda.current_device())
        self.A = be.empty(self.mloc, n)
        self.alpha = be.zeros(n)
        ldv = (self.mloc + 15) // 16 * 16
        self.Vw = be.empty(ldv, NB)
        self.G, self.S, self.R = be.zeros(_NN), be.zeros(_NN), be.zeros(_NN)
        self.T, self.Tt, self.Rref = be.zeros(_NN), be.zeros(_NN), be.zeros(_NN)
        self.bc = be.zeros(_NN + NB)  # broadcast unit: [-M^{-1} | alpha of the panel]
        self.W1, self.W2 = be.zeros(NB * n), be.zeros(NB * n)
        self.flag = be.izeros(4)
        self.stats = {"panels": 0, "cholqr2_retries": 0}
        self.Ts = {}  # compact-WY T of every panel (128 x 128 each), kept for the solve

    # ------------------------------------------------------------------ helpers
    def _allreduce(self, t):
        if self.P > 1:
            dist.all_reduce(t, group=self.group)

    def _bcast0(self, t):
        if self.P > 1:
            src = dist.get_global_rank(self.group, 0) if self.group is not None else 0
            dist.broadcast(t, src=src, group=self.group)

    def _active(self, c0):
        """(first local row, number of local rows) that belong to the panel whose diagonal starts at c0"""
        off = c0 if self.rank == 0 else 0
        return off, self.mloc - off

    def fill(self, seed: int):
        self.be.fill(self.A, self.mloc, self.n, seed, self.m, self.row0)

    # ------------------------------------------------------------------ factorisation
    def _panel_r(self, c0):
        """R of the panel (all-reduced Gram matrix + Cholesky), with one CholeskyQR2 retry"""
        be = self.be
        off, rows = self._active(c0)
        be.gram(self.A, off, c0, rows, self.G)
        self._allreduce(self.G)
        self.flag.zero_()
        be.chol(self.G, self.R, self.flag)

    def _reflectors(self, c0):
        """V (local rows) from R: top-block replay on rank 0, broadcast, one local GEMM; returns max | ||v||^2 - 2 |"""
        be = self.be
        off, rows = self._active(c0)
        negMinv, alpha128 = self.bc[:_NN], self.bc[_NN:]
        if self.rank == 0:
            be.recon_top(self.A, c0, c0, self.R, alpha128, self.Rref, negMinv)
        self._bcast0(self.bc)
        be.mul(self.A, off, c0, rows, negMinv, self.Vw)
        if self.rank == 0:
            be.fix_top(self.Vw, alpha128, negMinv)
        be.gram(self.Vw, 0, 0, rows, self.S)
        self._allreduce(self.S)
        d = self.S.view(NB, NB).diagonal()
        dev = (d - 2.0).abs().max().item()           # host sync: the verification of the fast path
        bad = int(self.flag[0].item()) != 0 or not math.isfinite(dev)
        return float("inf") if bad else dev

    def factor(self):
        be, n = self.be, self.n
        for c0 in range(0, n, NB):
            off, rows = self._active(c0)
            self._panel_r(c0)
            dev = self._reflectors(c0)
            if dev > self.tol:
                # CholeskyQR2: orthogonalise the panel once (Q1 = P R1^{-1} is what Vw-like storage
                # cannot hold here), so refine R through the Gram matrix of P R^{-1} instead:
                # R2 = chol((P R^{-1})'(P R^{-1})), R <- R2 R.  Needs R^{-1}; reuse the device kernels via
                # torch for the two 128 x 128 products (plumbing-size work).
                self.stats["cholqr2_retries"] += 1
                R1 = self.R.view(NB, NB).t().clone()              # column-major buffer -> math matrix
                R1inv = torch.linalg.inv(torch.triu(R1))
                self.bc[:_NN].copy_((-R1inv).t().reshape(-1))     # -R1^{-1} in column-major
                be.mul(self.A, off, c0, rows, self.bc[:_NN], self.Vw)
                be.gram(self.Vw, 0, 0, rows, self.G)
                self._allreduce(self.G)
                self.flag.zero_()
                be.chol(self.G, self.R, self.flag)
                R2 = self.R.view(NB, NB).t().clone()
                self.R.copy_((torch.triu(R2) @ torch.triu(R1)).t().reshape(-1))
                dev = self._reflectors(c0)
                if dev > 1e3 * self.tol:
                    raise RuntimeError(f"row-split panel at column {c0} is numerically rank deficient "
                                       f"(| ||v||^2 - 2 | = {dev:.2e}); no column-by-column fallback in this path")
            be.build_t(self.S, self.T, self.Tt)
            self.Ts[c0] = self.T.clone()
            be.commit(self.A, off, c0, rows, self.Vw, self.rank == 0, self.Rref)
            self.alpha[c0: c0 + NB].copy_(self.bc[_NN:])
            ncols = n - c0 - NB
            if ncols > 0:
                w1 = self.W1[: NB * ncols]
                be.vtc(self.Vw, self.A, off, c0 + NB, rows, ncols, w1)
                self._allreduce(w1)                                # cross-partition partial dots
                be.tw(self.T, w1, ncols, self.W2)
                be.vw(self.Vw, self.W2, self.A, off, c0 + NB, rows, ncols)
            self.stats["panels"] += 1
        return self

    # ------------------------------------------------------------------ metric
    def residual(self, seed: int) -> float:
        """||A - QR||_F / ||A||_F: every rank forms ITS ROWS of Q*R by re-applying the panels in
        reverse order to [R; 0] (one all-reduce of V'B per panel), A regenerated from `seed`."""
        be, n = self.be, self.n
        B = be.empty(self.mloc, n)
        if self.rank == 0:
            be.form_r0(self.A, self.mloc, n, self.alpha, B)
        else:
            B.zero_()
        for c0 in reversed(range(0, n, NB)):
            off, rows = self._active(c0)
            be.pack(self.A, off, c0, rows, self.Vw, self.rank == 0)
            be.gram(self.Vw, 0, 0, rows, self.S)
            self._allreduce(self.S)
            be.build_t(self.S, self.T, self.Tt)
            ncols = n - c0
            w1 = self.W1[: NB * ncols]
            be.vtc(self.Vw, B, off, c0, rows, ncols, w1)
            self._allreduce(w1)
            be.tw(self.Tt, w1, ncols, self.W2)     # Q (not Q'): op(T) = T
            be.vw(self.Vw, self.W2, B, off, c0, rows, ncols)
        A0 = be.empty(self.mloc, n)
        be.fill(A0, self.mloc, n, seed, self.m, self.row0)
        d2, x2 = be.diff_norms(A0, B, self.mloc, n)
        if self.P > 1:
            t = torch.tensor([d2, x2], dtype=torch.float64, device=self.alpha.device)
            dist.all_reduce(t, group=self.group)
            d2, x2 = t.tolist()
        return math.sqrt(d2 / x2)

    # ------------------------------------------------------------------ solve
    def solve(self, b_loc):
        """`H \\ b` (src:317-321) for the row split.  `b_loc` = this rank's rows of b (length mloc, not
        modified).  Q'b (src:215-242): per panel the partial dots V_r' b_r are all-reduced (a 128-vector),
        then b_r -= V_r (T' w) locally.  Back substitution (src:244-254) is local to rank 0, which owns R;
        x (length n) is broadcast to every rank."""
        be, n = self.be, self.n
        y = be.empty(self.mloc, 1)
        y[: self.mloc, 0].copy_(b_loc)
        w1, w2 = self.W1[:NB], self.W2[:NB]
        for c0 in range(0, n, NB):
            off, rows = self._active(c0)
            be.pack(self.A, off, c0, rows, self.Vw, self.rank == 0)
            be.vtc(self.Vw, y, off, 0, rows, 1, w1)
            self._allreduce(w1)
            be.tw(self.Ts[c0], w1, 1, w2)
            be.vw(self.Vw, w2, y, off, 0, rows, 1)
        x = be.zeros(n)
        if self.rank == 0:
            yv = y[: self.mloc, 0]
            be.backsub(self.A, n, self.alpha, yv)
            x.copy_(yv[:n])
        self._bcast0(x)
        return x

    def gather_full(self):
        """(H, alpha) as host numpy arrays on every rank (small problems / tests)"""
        import numpy as np
        loc = self.A[: self.mloc, :].cpu().numpy()
        parts = [None] * self.P
        if self.P > 1:
            dist.all_gather_object(parts, loc, group=self.group)
        else:
            parts = [loc]
        return np.asfortranarray(np.vstack(parts)), self.alpha.cpu().numpy().copy()
