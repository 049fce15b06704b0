"""Multi-process restatement of the reference's DArray path (TEST INFRASTRUCTURE ONLY).

Mirrors householder!(A::DArray, α) (src/DistributedHouseholderQR.jl:115-120) and the distributed
solve (src:226-230, 256-282) with the reference's own structure: CONTIGUOUS column blocks
(DistributedArrays default, test/runtests.jl:71), owners visited sequentially, one "broadcast" of
the dense column Hj per column (src:138-143), α shared by all (SharedArray, src:301-304 -> here a
replicated vector updated by broadcast), per-row sum-reduction of partial dots in the back
substitution (src:262-266).  Transport: torch.distributed (gloo) instead of Distributed.jl.
BASELINE config 1 (512 x 512, nprocs = 2, CPU) runs through this file.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

from . import dhqr_oracle as orc


def column_blocks(n: int, nparts: int):
    q, r = divmod(n, nparts)
    out, lo = [], 0
    for p in range(nparts):
        w = q + (1 if p < r else 0)
        out.append((lo, lo + w))
        lo += w
    return out


def householder_darray(Al: np.ndarray, m: int, n: int, alpha: np.ndarray, group=None):
    """Al: this process's contiguous column block (m x nloc, column-major), factored in place."""
    rank, P = dist.get_rank(group), dist.get_world_size(group)
    blocks = column_blocks(n, P)
    lo, hi = blocks[rank]
    L = orc.lib()
    ldh = Al.strides[1] // 8 if Al.shape[1] > 1 else m
    Hj = np.zeros(m)
    tHj = torch.from_numpy(Hj)
    ta = torch.zeros(1, dtype=torch.float64)
    for p in range(P):                      # src:116  for p in procs(A)
        plo, phi = blocks[p]
        for j in range(plo, phi):           # src:127  for j in Hl.colrange
            if p == rank:
                L.dhqr_oracle_reflector_step(orc._ptr(Al), m, ldh, j, lo, orc._ptr(alpha), orc._ptr(Hj))
                ta[0] = alpha[j]
            dist.broadcast(tHj, src=p, group=group)   # src:141-143 @spawnat fan-out of Hj
            dist.broadcast(ta, src=p, group=group)    # src:130 α[j] through the SharedArray
            alpha[j] = ta[0].item()
            if hi > lo:
                L.dhqr_oracle_householder_inner(orc._ptr(Al), m, ldh, j, orc._ptr(Hj), lo, hi, n)
    return Al, alpha


def solve_darray(Al: np.ndarray, m: int, n: int, alpha: np.ndarray, b: np.ndarray, group=None):
    rank, P = dist.get_rank(group), dist.get_world_size(group)
    blocks = column_blocks(n, P)
    lo, hi = blocks[rank]
    L = orc.lib()
    ldh = Al.strides[1] // 8 if Al.shape[1] > 1 else m
    b = np.array(b, dtype=np.float64, copy=True)   # src:318 SharedArray(b)
    tb = torch.from_numpy(b)
    for p in range(P):                              # src:227-229 sequential over owners
        if p == rank and hi > lo:
            L.dhqr_oracle_solve1_inner(orc._ptr(b), orc._ptr(Al), m, ldh, lo, hi, n)
        dist.broadcast(tb, src=p, group=group)      # b lives in shared memory in the reference
    part = torch.zeros(1, dtype=torch.float64)
    for i in range(n - 1, -1, -1):                  # src:260
        part[0] = L.dhqr_oracle_solve2_inner(orc._ptr(b), orc._ptr(Al), ldh, i, lo, hi, n) if hi > lo else 0.0
        dist.all_reduce(part, group=group)          # src:266 sum(fetch.(futures))
        b[i] = (b[i] - part[0].item()) / alpha[i]   # src:267
    return b[:n].copy()
