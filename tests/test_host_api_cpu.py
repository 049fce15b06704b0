"""Host-side mirror of the Julia API (api.py): the parts that do not need a GPU -- element-type
dispatch (Julia's multiple dispatch on eltype), the struct constructor (src:301-309) and argument
validation that must fire before any device work."""
import numpy as np
import pytest


def test_struct_alpha_follows_eltype(pkg):
    # src:306-309  α = zeros(eltype(A), size(A, 2))
    H = pkg.DistributedHouseholderQRStruct(np.zeros((5, 3), order="F"))
    assert H.α.dtype == np.float64 and H.α.shape == (3,) and H.alpha is H.α
    Hc = pkg.DistributedHouseholderQRStruct(np.zeros((5, 3), dtype=complex, order="F"))
    assert Hc.α.dtype == np.complex128 and Hc.α.shape == (3,)


def test_default_path_per_element_type(pkg):
    api = pkg.api
    assert api._resolve_nb(np.zeros((4000, 40)), None) == pkg.NB    # Float64: blocked MFMA path ...
    assert api._resolve_nb(np.zeros((api.DEFAULT_UNBLOCKED_MAX_ROWS + 1, 4)), None) == pkg.NB
    assert api._resolve_nb(np.zeros((api.DEFAULT_UNBLOCKED_MAX_ROWS, 4)), None) == 0   # ... the reference's order for short matrices
    assert api._resolve_nb(np.zeros((4, 4)), None) == 0
    assert api._resolve_nb(np.zeros((4, 4)), pkg.NB) == pkg.NB      # explicit blocked
    assert api._resolve_nb(np.zeros((4, 4)), 0) == 0                # explicit unblocked
    assert api._resolve_nb(np.zeros((4, 4), dtype=complex), None) == 0   # ComplexF64: unblocked
    assert api._resolve_nb(np.zeros((4, 4), dtype=complex), 0) == 0
    with pytest.raises(ValueError):
        api._resolve_nb(np.zeros((4, 4), dtype=complex), 128)       # no blocked complex path yet


def test_bad_host_arguments_raise_before_device_work(pkg):
    with pytest.raises(TypeError):
        pkg.householder_(np.zeros((4, 3), dtype=np.float32), np.zeros(3))
    with pytest.raises(TypeError):
        pkg.householder_(np.zeros((4, 3), order="F"), np.zeros(2))       # α too short
    with pytest.raises(TypeError):
        pkg.householder_(np.zeros((4, 3), dtype=complex, order="F"), np.zeros(3))  # α must be complex
