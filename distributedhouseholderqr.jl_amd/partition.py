"""Partition / locality layer (host logic, no device code).

Mirrors src/DistributedHouseholderQR.jl:11-40 (`localcols`, `localindexes`, `columnblocks`,
`LocalColumnBlock`) and adds the 1-D *block-cyclic* column layout the multi-GPU driver uses
(contiguous blocks cap 8-GPU speed-up at ~5.4x, so nb-wide column blocks are
dealt round-robin, ScaLAPACK 1 x P).  All indices here are 0-based.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Tuple


def contiguous_column_blocks(n: int, nparts: int) -> List[range]:
    """DistributedArrays' default even split used by the reference's tests (test/runtests.jl:71,
    dist=(1, nworkers())): the first `n % nparts` chunks get one extra column."""
    q, r = divmod(n, nparts)
    out, lo = [], 0
    for p in range(nparts):
        w = q + (1 if p < r else 0)
        out.append(range(lo, lo + w))
        lo += w
    return out


@dataclass(frozen=True)
class LocalColumnBlock:
    """src:26-36: a view of the local column block addressed by GLOBAL column index.
    `colrange` is the contiguous global range owned (the reference's layout); `dj` == Δj."""
    colrange: range

    @property
    def dj(self) -> int:  # src:34  Δj = colrange[1] - 1
        return self.colrange.start

    def local(self, j: int) -> int:  # src:37-40  j - Δj
        if j not in self.colrange:
            raise IndexError(f"global column {j} not in {self.colrange}")
        return j - self.dj

    def trailing(self, j: int, n: int) -> range:  # src:201  intersect(j+1:n, colrange)
        lo = max(j + 1, self.colrange.start)
        hi = min(n, self.colrange.stop)
        return range(lo, max(lo, hi))


@dataclass(frozen=True)
class BlockCyclicColumns:
    """1-D block-cyclic column layout: global column j lives in block j // nb, owned by rank
    (j // nb) % nranks, at local column ((j // nb) // nranks) * nb + j % nb."""
    n: int
    nb: int
    nranks: int

    @property
    def nblocks(self) -> int:
        return (self.n + self.nb - 1) // self.nb

    def owner(self, k: int) -> int:
        return k % self.nranks

    def block_cols(self, k: int) -> range:
        return range(k * self.nb, min(self.n, (k + 1) * self.nb))

    def local_blocks(self, rank: int) -> List[int]:
        return list(range(rank, self.nblocks, self.nranks))

    def local_ncols(self, rank: int) -> int:
        return sum(len(self.block_cols(k)) for k in self.local_blocks(rank))

    def local_col_start(self, k: int) -> int:
        """first local column of global block k on its owner (all earlier local blocks are full)"""
        return (k // self.nranks) * self.nb

    def global_col(self, rank: int, jl: int) -> int:
        return ((jl // self.nb) * self.nranks + rank) * self.nb + jl % self.nb

    def local_col(self, j: int) -> Tuple[int, int]:
        k = j // self.nb
        return self.owner(k), self.local_col_start(k) + j % self.nb

    def first_local_block_after(self, rank: int, k: int) -> int:
        """smallest global block index > k owned by `rank` (may be >= nblocks)."""
        k1 = k + 1
        return k1 + ((rank - k1) % self.nranks)

    def trailing_local_cols(self, rank: int, k: int) -> Tuple[int, int]:
        """(first local column, count) of this rank's columns in global blocks > k: always a
        SUFFIX of the local storage, so the trailing update is one GEMM pair per rank."""
        kb = self.first_local_block_after(rank, k)
        total = self.local_ncols(rank)
        if kb >= self.nblocks:
            return total, 0
        lo = self.local_col_start(kb)
        return lo, total - lo
