"""1-D column-split Householder QR over torch.distributed (one process per GPU; backend "nccl" is
RCCL over xGMI on ROCm).  Replaces the reference's DArray path:

  reference (src/DistributedHouseholderQR.jl)                 here
  householder!(A::DArray, α)  :115-120  owners visited        ColumnCyclicQR.factor(): owners visited
      sequentially, contiguous column blocks                   per 128-column block, block-CYCLIC layout
  per-column `@spawnat` fan-out of Hj :141-143 (n*np RPCs,     ONE broadcast per panel of the packed
      "this is most expensive")                                (V, T, α) buffer, posted asynchronously so
                                                               it overlaps the trailing update (look-ahead)
  α::SharedArray :301-304                                      α replicated on every rank (rides in the
                                                               broadcast buffer)
  _solve_householder1!/2! over procs :226-282, per-row         solve(): per-block hand-off of b, and ONE
      sum(fetch.(futures)) of partial dots :262-266            all-reduce of the partial dots per block

The factorisation needs no all-reduce under a column split (dots run over rows, which are local);
the only exchange is the panel broadcast.  All compute goes through a backend object; the product
backend is HipBackend (libdhqr.so).  tests/ inject an oracle-backed CPU backend to exercise this
orchestration under gloo with world_size 2 -- the product never does.
"""
from __future__ import annotations

import ctypes
import math


import torch
import torch.distributed as dist

from . import _lib
from ._lib import NB, check
from .api import Context, empty_colmajor, get_context
from .partition import BlockCyclicColumns, contiguous_column_blocks


class HipBackend:
    """Device tensors + libdhqr.so.  Column-major local block with lda == m."""

    def __init__(self, device: int):
        self.device = device
        self.ctx = get_context(device)
        self.L = _lib.lib()
        self.torch_device = torch.device("cuda", device)
        # look-ahead lane: its own high-priority torch stream AND its own dhqr context (= its own
        # workspaces), so narrow update + panel factorisation + broadcast of panel k+1 run
        # underneath the wide trailing update of panel k on the main stream
        self.hi_stream = torch.cuda.Stream(device=self.torch_device, priority=-1)
        self.ctx_hi = Context(device)
        self._lane = self.ctx

    def lane(self, hi: bool):
        """context manager: run the enclosed backend calls on the look-ahead lane (hi) or main"""
        be = self

        class _Lane:
            def __enter__(self_inner):
                self_inner.prev = be._lane
                be._lane = be.ctx_hi if hi else be.ctx
                self_inner.cm = torch.cuda.stream(be.hi_stream) if hi else None
                if self_inner.cm is not None:
                    self_inner.cm.__enter__()

            def __exit__(self_inner, *exc):
                if self_inner.cm is not None:
                    self_inner.cm.__exit__(*exc)
                be._lane = self_inner.prev
                return False

        return _Lane()

    def record_main(self):
        """event on the current (main) stream the look-ahead lane can wait on"""
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        return ev

    def hi_wait(self, ev):
        if ev is not None:
            self.hi_stream.wait_event(ev)

    def main_wait_hi(self):
        torch.cuda.current_stream(self.device).wait_stream(self.hi_stream)

    def record_current(self):
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        return ev

    def wait_event(self, ev):
        torch.cuda.current_stream(self.device).wait_event(ev)

    # -- memory
    def empty(self, m, n):
        return empty_colmajor(m, max(n, 1), self.torch_device)

    def zeros_vec(self, n):
        return torch.zeros(n, dtype=torch.float64, device=self.torch_device)

    def panel_buffer(self, rows):
        return torch.empty(self.L.dhqr_panel_buffer_elems(rows), dtype=torch.float64, device=self.torch_device)

    def panel_elems(self, rows):
        return self.L.dhqr_panel_buffer_elems(rows)

    def alpha_of(self, vt, rows):
        off = self.L.dhqr_panel_ldv(rows) * NB + 2 * NB * NB
        return vt[off: off + NB]

    # -- compute (all on torch's current stream)
    def _p(self, A, row, col):
        return ctypes.c_void_p(A.data_ptr() + 8 * (row + col * A.stride(1)))

    def fill(self, A, ncols, seed, gm, nb, nranks, rank):
        if ncols == 0:
            return
        self._lane.use_torch_stream()
        check(self.L.dhqr_fill_uniform_f64(self._lane.handle, self._p(A, 0, 0), A.shape[0], ncols, A.stride(1),
                                           seed, gm, 0, nb, nranks, rank))

    def panel_factor(self, A, c0, lc0, w, vt):
        self._lane.use_torch_stream()
        rows = A.shape[0] - c0
        check(self.L.dhqr_panel_factor_f64(self._lane.handle, self._p(A, c0, lc0), rows, w, A.stride(1),
                                           ctypes.c_void_p(vt.data_ptr())))

    def panel_pack(self, A, c0, lc0, w, vt):
        self._lane.use_torch_stream()
        rows = A.shape[0] - c0
        check(self.L.dhqr_panel_pack_f64(self._lane.handle, self._p(A, c0, lc0), rows, w, A.stride(1),
                                         ctypes.c_void_p(vt.data_ptr())))

    def panel_apply(self, vt, C, c0, lo, cnt, trans):
        if cnt <= 0:
            return
        self._lane.use_torch_stream()
        rows = C.shape[0] - c0
        ldc = C.stride(1) if C.dim() == 2 and C.shape[1] > 1 else C.shape[0]
        ptr = ctypes.c_void_p(C.data_ptr() + 8 * (c0 + lo * ldc))
        check(self.L.dhqr_panel_apply_f64(self._lane.handle, ctypes.c_void_p(vt.data_ptr()), rows, ptr, cnt, ldc,
                                          1 if trans else 0))

    def form_r0(self, A, ncols, alpha, W, nb, nranks, rank):
        if ncols == 0:
            return
        self._lane.use_torch_stream()
        check(self.L.dhqr_form_r0_f64(self._lane.handle, self._p(A, 0, 0), A.shape[0], ncols, A.stride(1),
                                      ctypes.c_void_p(alpha.data_ptr()), self._p(W, 0, 0), W.stride(1), nb,
                                      nranks, rank))

    def diff_norms(self, X, Y, ncols):
        out = (ctypes.c_double * 2)()
        if ncols > 0:
            self._lane.use_torch_stream()
            check(self.L.dhqr_diff_norms_f64(self._lane.handle, self._p(X, 0, 0), X.stride(1), self._p(Y, 0, 0),
                                             Y.stride(1), X.shape[0], ncols, out))
        return out[0], out[1]

    def backsub_block(self, A, lc0, alpha, b, lo, hi, diag, update):
        """one block step of the back substitution with this rank's columns [lc0, lc0+hi-lo)"""
        self._lane.use_torch_stream()
        base = A.data_ptr() + 8 * (lc0 - lo) * A.stride(1)  # so that global column j sits at base + j*lda
        check(self.L.dhqr_backsub_block_f64(self._lane.handle, ctypes.c_void_p(base), A.stride(1),
                                            ctypes.c_void_p(alpha.data_ptr()), ctypes.c_void_p(b.data_ptr()),
                                            lo, hi, 1 if diag else 0, 1 if update else 0))

    def synchronize(self):
        torch.cuda.synchronize(self.device)


class ColumnCyclicQR:
    """Distributed factor object: the local block of a block-cyclic column split + replicated α.

    Usage (every rank):  q = ColumnCyclicQR(m, n); q.fill(seed); q.factor(); r = q.residual(seed)
    """

    def __init__(self, m: int, n: int, nb: int = NB, group=None, backend=None, lookahead: bool = True):
        if m < n:
            raise ValueError("m >= n required")
        if nb != NB:
            raise ValueError(f"panel width is fixed at {NB}")
        self.m, self.n, self.nb = m, n, nb
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.P = dist.get_world_size(group) if dist.is_initialized() else 1
        self.layout = BlockCyclicColumns(n, nb, self.P)
        self.ncl = self.layout.local_ncols(self.rank)
        self.be = backend if backend is not None else HipBackend(torch.cuda.current_device())
        self.lookahead = lookahead
        self.A = self.be.empty(m, self.ncl)
        self.alpha = self.be.zeros_vec(n)
        self.vt = [self.be.panel_buffer(m), self.be.panel_buffer(m)]
        self._work = {}
        self._ev_panel = {}

    # ------------------------------------------------------------------ helpers
    def _src(self, k):  # global rank of the owner of block k
        o = self.layout.owner(k)
        return dist.get_global_rank(self.group, o) if (self.group is not None and dist.is_initialized()) else o

    def _rows(self, k):
        return self.m - k * self.nb

    def _bcast(self, k, buf, async_op):
        if self.P == 1:
            return None
        t = buf[: self.be.panel_elems(self._rows(k))]
        return dist.broadcast(t, src=self._src(k), group=self.group, async_op=async_op)

    def fill(self, seed: int):
        """synthetic input: the local columns of A[i,j] = u01(seed, i + j*m)"""
        self.be.fill(self.A, self.ncl, seed, self.m, self.nb, self.P, self.rank)

    # ------------------------------------------------------------------ factorisation
    def _factor_and_post(self, k, pre=None):
        """owner factors block k into vt[k%2] (after `pre()`, e.g. the narrow update that brings
        the block up to date); every rank posts the (async) broadcast."""
        lay = self.layout
        buf = self.vt[k % 2]
        if lay.owner(k) == self.rank:
            if pre is not None:
                pre()
            self.be.panel_factor(self.A, k * self.nb, lay.local_col_start(k), len(lay.block_cols(k)), buf)
            if hasattr(self.be, "record_current"):
                # the panel may have been produced on the look-ahead lane: consumers on the main
                # stream wait for this event (the broadcast alone orders them only when P > 1)
                self._ev_panel[k] = self.be.record_current()
        self._work[k] = self._bcast(k, buf, async_op=True)

    def _wait(self, k):
        ev = self._ev_panel.pop(k, None)
        if ev is not None:
            self.be.wait_event(ev)
        w = self._work.pop(k, None)
        if w is not None:
            w.wait()

    def factor(self):
        """householder!(A::DArray, α) (src:115-148) on the block-cyclic column split."""
        lay, r, nb, be = self.layout, self.rank, self.nb, self.be
        K = lay.nblocks
        # Two lanes (panel k+1 on a second stream underneath the wide update k) only pay at world
        # size 1.  With P > 1 the block owner's own wide update is off the critical path (the next
        # P-1 panels belong to other ranks) while its panel IS the critical path, and the panel's
        # single-workgroup kernels run 4-20x slower when they share the GPU with the GEMMs
        # (profiles/r01c_*), so the owner factors first and updates afterwards.
        two_lanes = self.lookahead and hasattr(be, "lane") and self.P == 1
        self._factor_and_post(0)
        ev_wide = None  # main-stream event after the previous wide update
        for k in range(K):
            buf = self.vt[k % 2]
            rows, c0 = self._rows(k), k * nb
            w = len(lay.block_cols(k))
            self._wait(k)
            self.alpha[c0: c0 + w].copy_(be.alpha_of(buf, rows)[:w])
            lo, cnt = lay.trailing_local_cols(r, k)
            if k + 1 >= K:
                break
            if self.lookahead and lay.owner(k + 1) == r:
                # look-ahead: bring block k+1 up to date, factor it and ship it while the rest of
                # trailing update k is still running
                w1 = len(lay.block_cols(k + 1))
                if two_lanes:
                    # the panel path synchronises its own stream on the host once (verification
                    # flag of the fast path), so the wide update is queued on the main stream
                    # FIRST; the look-ahead lane only waits for what was queued before it
                    ev_now = be.record_main()      # alpha copy / receipt of panel k / wide update k-1
                    be.panel_apply(buf, self.A, c0, lo + w1, cnt - w1, True)
                    be.hi_wait(ev_now)
                    with be.lane(True):
                        self._factor_and_post(k + 1, pre=lambda: be.panel_apply(buf, self.A, c0, lo, w1, True))
                else:
                    self._factor_and_post(k + 1, pre=lambda: be.panel_apply(buf, self.A, c0, lo, w1, True))
                    be.panel_apply(buf, self.A, c0, lo + w1, cnt - w1, True)
            elif self.lookahead:
                self._factor_and_post(k + 1)  # non-owner: only posts the receive
                be.panel_apply(buf, self.A, c0, lo, cnt, True)
            else:
                be.panel_apply(buf, self.A, c0, lo, cnt, True)
                self._factor_and_post(k + 1)
        if two_lanes:
            be.main_wait_hi()
        return self

    # ------------------------------------------------------------------ metric
    def residual(self, seed: int) -> float:
        """||A - QR||_F / ||A||_F with A regenerated from `seed`: every rank forms its columns of
        Q*R by re-applying the panels in reverse order (one broadcast per panel again)."""
        lay, r, nb = self.layout, self.rank, self.nb
        be = self.be
        W = be.empty(self.m, self.ncl)
        be.form_r0(self.A, self.ncl, self.alpha, W, nb, self.P, r)
        buf = self.vt[0]
        for k in reversed(range(lay.nblocks)):
            c0 = k * nb
            if lay.owner(k) == r:
                be.panel_pack(self.A, c0, lay.local_col_start(k), len(lay.block_cols(k)), buf)
            self._bcast(k, buf, async_op=False)
            kb = lay.first_local_block_after(r, k - 1)  # first local block >= k
            if kb < lay.nblocks:
                lo = lay.local_col_start(kb)
                be.panel_apply(buf, W, c0, lo, self.ncl - lo, False)
        A0 = be.empty(self.m, self.ncl)
        be.fill(A0, self.ncl, seed, self.m, nb, self.P, r)
        d2, x2 = be.diff_norms(A0, W, self.ncl)
        if self.P > 1:
            t = torch.tensor([d2, x2], dtype=torch.float64, device=self.alpha.device)
            dist.all_reduce(t, group=self.group)
            d2, x2 = t.tolist()
        return math.sqrt(d2 / x2)

    # ------------------------------------------------------------------ solve
    def solve(self, b):
        """`H \\ b` on the column split (src:317-321, src:226-282).  `b` (length m, replicated on
        every rank) is not modified; returns x (length n) on every rank.
        Q'b: the owner of each block applies it and hands the updated tail of b on (the reference
        walks owners sequentially with b in shared memory, src:226-230).  Back substitution: every
        rank accumulates the contribution of ITS columns to the rows above; per block one
        all-reduce sums those partial dots (the reference's sum(fetch.(futures)), src:262-266),
        the owner solves the diagonal block and broadcasts x."""
        lay, r, nb, be = self.layout, self.rank, self.nb, self.be
        m, n = self.m, self.n
        y = b.clone()
        buf = self.vt[0]
        for k in range(lay.nblocks):
            c0 = k * nb
            if lay.owner(k) == r:
                be.panel_pack(self.A, c0, lay.local_col_start(k), len(lay.block_cols(k)), buf)
                be.panel_apply(buf, y, c0, 0, 1, True)
            if self.P > 1:
                dist.broadcast(y[c0:], src=self._src(k), group=self.group)
        u = torch.zeros_like(y)  # u = -(sum over MY solved blocks of R[:, blk] x_blk)
        x = y  # solved in place block by block
        for k in reversed(range(lay.nblocks)):
            c0 = k * nb
            w = len(lay.block_cols(k))
            s = u[c0: c0 + w].clone()
            if self.P > 1:
                dist.all_reduce(s, group=self.group)  # cross-partition partial dots
            if lay.owner(k) == r:
                x[c0: c0 + w] += s
                be.backsub_block(self.A, lay.local_col_start(k), self.alpha, x, c0, c0 + w, True, False)
            if self.P > 1:
                dist.broadcast(x[c0: c0 + w], src=self._src(k), group=self.group)
            if lay.owner(k) == r and c0 > 0:
                u[c0: c0 + w].copy_(x[c0: c0 + w])
                be.backsub_block(self.A, lay.local_col_start(k), self.alpha, u, c0, c0 + w, False, True)
                u[c0: c0 + w].zero_()
        return x[:n].clone()

    # ------------------------------------------------------------------ the reference's DArray layout
    # qr!(A::DArray) (src:115-120, test/runtests.jl:71) receives CONTIGUOUS column blocks, one per
    # process (DistributedArrays' default split).  The factorisation here runs on a block-cyclic layout
    # (contiguous blocks leave the owners of the early columns idle: <= 5.4x on 8 GPUs), so
    # a caller holding the reference's layout converts on the way in and out.  One broadcast / reduce of
    # each rank's block per direction: a one-off O(mn) exchange next to the O(mn^2) factorisation, built
    # only from collectives every backend has (NCCL/RCCL and gloo).
    def _stage(self, width):
        return torch.empty((width, self.m), dtype=self.A.dtype, device=self.A.device)  # == m x width column-major

    def _my_runs(self, cols):
        """pieces (first global col, count, local col) of the global column range `cols` that this rank
        owns in the block-cyclic layout (a run never crosses a cyclic block)"""
        lay, out = self.layout, []
        if len(cols) == 0:
            return out
        for k in range(cols.start // self.nb, (cols.stop - 1) // self.nb + 1):
            if lay.owner(k) != self.rank:
                continue
            lo, hi = max(k * self.nb, cols.start), min((k + 1) * self.nb, cols.stop, self.n)
            if hi > lo:
                out.append((lo, hi - lo, lay.local_col_start(k) + lo - k * self.nb))
        return out

    def load_contiguous_blocks(self, local_block):
        """Scatter-in: `local_block` (m x w_r, any strides) holds this rank's columns
        contiguous_column_blocks(n, P)[rank] of the global matrix -- the reference's DArray layout."""
        blocks = contiguous_column_blocks(self.n, self.P)
        mine = blocks[self.rank]
        if tuple(local_block.shape) != (self.m, len(mine)):
            raise ValueError(f"rank {self.rank} must pass a {self.m} x {len(mine)} block, got {tuple(local_block.shape)}")
        for s, cols in enumerate(blocks):
            if len(cols) == 0:
                continue
            buf = self._stage(len(cols))
            if s == self.rank:
                buf.copy_(local_block.t())
            if self.P > 1:
                src = dist.get_global_rank(self.group, s) if self.group is not None else s
                dist.broadcast(buf, src=src, group=self.group)
            for g0, cnt, l0 in self._my_runs(cols):
                self.A[:, l0: l0 + cnt].copy_(buf[g0 - cols.start: g0 - cols.start + cnt].t())
        return self

    def store_contiguous_blocks(self):
        """Gather-out: this rank's contiguous column block (m x w_r, column-major) of the FACTORED matrix
        in the reference's DArray layout; alpha is replicated already (self.alpha, the SharedArray of
        src:301-304)."""
        blocks = contiguous_column_blocks(self.n, self.P)
        out = None
        for s, cols in enumerate(blocks):
            if len(cols) == 0:
                if s == self.rank:
                    out = self._stage(0).t()
                continue
            buf = self._stage(len(cols))
            buf.zero_()
            for g0, cnt, l0 in self._my_runs(cols):
                buf[g0 - cols.start: g0 - cols.start + cnt].copy_(self.A[:, l0: l0 + cnt].t())
            if self.P > 1:
                dst = dist.get_global_rank(self.group, s) if self.group is not None else s
                dist.reduce(buf, dst=dst, group=self.group)  # every column has exactly one non-zero contributor
            if s == self.rank:
                out = buf.t()
        return out



    # ------------------------------------------------------------------ gather (tests / small n)
    def gather_full(self):
        """(H, alpha) as host numpy arrays on every rank -- Array(A::DArray) for small problems."""
        import numpy as np
        loc = self.A[:, : self.ncl].cpu().numpy() if self.ncl else np.zeros((self.m, 0))
        parts = [None] * self.P
        if self.P > 1:
            dist.all_gather_object(parts, loc, group=self.group)
        else:
            parts = [loc]
        H = np.zeros((self.m, self.n), order="F")
        for rk, blk in enumerate(parts):
            for jl in range(blk.shape[1]):
                H[:, self.layout.global_col(rk, jl)] = blk[:, jl]
        return H, self.alpha.cpu().numpy().copy()


def qr_darray_(local_block, n: int, group=None, backend=None, lookahead: bool = True):
    """qr!(A::DArray) (src:115-120, 311-315) for callers that hold the reference's layout: every rank
    passes ITS contiguous column block of the m x n matrix (contiguous_column_blocks(n, P)[rank]); the
    block is overwritten with the factored columns (V on/below the diagonal, R above) and the replicated
    alpha (the reference's SharedArray) is returned together with the distributed factor object, whose
    .solve(b) is `H \\ b`.  Internally: scatter to block-cyclic, factor, gather back."""
    m = local_block.shape[0]
    q = ColumnCyclicQR(m, n, group=group, backend=backend, lookahead=lookahead)
    q.load_contiguous_blocks(local_block)
    q.factor()
    fac = q.store_contiguous_blocks()
    if fac.numel():
        local_block.copy_(fac)
    return q, q.alpha
