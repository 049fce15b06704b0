"""Host-side partition logic (src:11-40 mirror + block-cyclic layout)."""
import pytest


def test_contiguous_blocks_match_distributedarrays_default(pkg):
    # 512 columns over 2 procs (BASELINE config 1): 256 each
    assert pkg.contiguous_column_blocks(512, 2) == [range(0, 256), range(256, 512)]
    assert pkg.contiguous_column_blocks(10, 3) == [range(0, 4), range(4, 7), range(7, 10)]


def test_local_column_block(pkg):
    lcb = pkg.LocalColumnBlock(range(256, 512))
    assert lcb.dj == 256 and lcb.local(300) == 44
    assert list(lcb.trailing(10, 512)) == list(range(256, 512))
    assert list(lcb.trailing(400, 512)) == list(range(401, 512))
    assert len(lcb.trailing(511, 512)) == 0
    with pytest.raises(IndexError):
        lcb.local(5)


@pytest.mark.parametrize("n,nb,p", [(1024, 128, 8), (1000, 128, 3), (130, 128, 2), (100, 128, 4), (4096, 128, 8)])
def test_block_cyclic_maps(pkg, n, nb, p):
    lay = pkg.BlockCyclicColumns(n, nb, p)
    seen = {}
    for r in range(p):
        ncl = lay.local_ncols(r)
        for jl in range(ncl):
            g = lay.global_col(r, jl)
            assert 0 <= g < n and g not in seen
            seen[g] = (r, jl)
            assert lay.local_col(g) == (r, jl)
    assert len(seen) == n
    for r in range(p):
        for k in range(-1, lay.nblocks):
            lo, cnt = lay.trailing_local_cols(r, k)
            want = [jl for jl in range(lay.local_ncols(r)) if lay.global_col(r, jl) // nb > k]
            assert cnt == len(want)
            if cnt:
                assert want == list(range(lo, lo + cnt))  # always a suffix of local storage
