"""CPU oracle for the DistributedHouseholderQR hot path -- numpy twin + ctypes loader.

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; the product package never imports this module.

PARITY STATUS: "parity unpinned" at the element level (no Julia here, no golden vectors in the
reference).  Pinned against the reference's own test assertions and LAPACK -- see the header of
oracle/dhqr_oracle.c and tests/test_oracle.py.

All `src:` citations are /root/reference/src/DistributedHouseholderQR.jl.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libdhqr_oracle.so")
_SRC = os.path.join(_HERE, "dhqr_oracle.c")

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def build(force: bool = False) -> str:
    """Compile oracle/dhqr_oracle.c -> oracle/libdhqr_oracle.so (gcc, OpenMP). Generic x86-64
    code generation: the .so travels to the GPU box whose host CPU may differ."""
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(_SRC):
        subprocess.check_call(
            ["gcc", "-O3", "-fopenmp", "-fPIC", "-shared", "-std=c11", "-o", _SO, _SRC, "-lm"]
        )
    return _SO


_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        i64, u64, f64 = ctypes.c_int64, ctypes.c_uint64, ctypes.c_double
        p = ctypes.c_void_p
        L.dhqr_oracle_u01.restype = f64
        L.dhqr_oracle_u01.argtypes = [u64, u64]
        L.dhqr_oracle_fill.argtypes = [p, i64, i64, i64, u64]
        L.dhqr_oracle_alphafactor.restype = f64
        L.dhqr_oracle_alphafactor.argtypes = [f64]
        L.dhqr_oracle_partialdot.restype = f64
        L.dhqr_oracle_partialdot.argtypes = [p, p, i64, i64]
        L.dhqr_oracle_nrm2.restype = f64
        L.dhqr_oracle_nrm2.argtypes = [p, i64]
        L.dhqr_oracle_householder_inner.argtypes = [p, i64, i64, i64, p, i64, i64, i64]
        L.dhqr_oracle_reflector_step.argtypes = [p, i64, i64, i64, i64, p, p]
        L.dhqr_oracle_householder.argtypes = [p, i64, i64, i64, p]
        L.dhqr_oracle_householder_prefix.argtypes = [p, i64, i64, i64, p, i64]
        L.dhqr_oracle_solve1_inner.argtypes = [p, p, i64, i64, i64, i64, i64]
        L.dhqr_oracle_solve2_inner.restype = f64
        L.dhqr_oracle_solve2_inner.argtypes = [p, p, i64, i64, i64, i64, i64]
        L.dhqr_oracle_solve.argtypes = [p, p, i64, i64, i64, p]
        L.dhqr_oracle_form_qr.argtypes = [p, i64, i64, i64, p, p, i64]
        L.dhqr_oracle_num_threads.restype = ctypes.c_int
        _lib = L
    return _lib


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(ctypes.c_void_p)


def _fcheck(a: np.ndarray) -> np.ndarray:
    assert a.dtype == np.float64 and a.flags["F_CONTIGUOUS"], "oracle wants column-major float64"
    return a


# ----------------------------------------------------------------------------- generator
def u01(seed: int, idx: np.ndarray) -> np.ndarray:
    """uniform [0,1) from (seed, linear index); bit-identical to dhqr_oracle_u01 / the device fill."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + (idx.astype(np.uint64) + np.uint64(1)) * _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def rand_matrix(m: int, n: int, seed: int) -> np.ndarray:
    """A[i,j] = u01(seed, i + j*m), column-major  (stands in for rand(T,m,n), test/runtests.jl:45)."""
    idx = np.arange(m * n, dtype=np.uint64)
    return np.asfortranarray(u01(seed, idx).reshape((m, n), order="F"))


def rand_vector(m: int, seed: int) -> np.ndarray:
    """b[i] = u01(seed, i)  (stands in for rand(T,m), test/runtests.jl:46)."""
    return u01(seed, np.arange(m, dtype=np.uint64))


# ----------------------------------------------------------------------------- C oracle wrappers
def householder(A: np.ndarray):
    """qr!(A) for a plain matrix (src:311-315 -> src:113 -> src:122-148). Returns (H, alpha) with
    H a factored COPY of A (lower-incl-diag = V, strict upper = R) and alpha = diag(R)."""
    H = np.array(A, dtype=np.float64, order="F", copy=True)
    m, n = H.shape
    alpha = np.zeros(n)
    lib().dhqr_oracle_householder(_ptr(H), m, n, H.strides[1] // 8, _ptr(alpha))
    return H, alpha


def householder_prefix(H: np.ndarray, alpha: np.ndarray, ncols: int) -> None:
    """In place: process only the first `ncols` reflectors (bench cpu_baseline bounded sample)."""
    _fcheck(H)
    m, n = H.shape
    lib().dhqr_oracle_householder_prefix(_ptr(H), m, n, H.strides[1] // 8, _ptr(alpha), ncols)


def solve(H: np.ndarray, alpha: np.ndarray, b: np.ndarray) -> np.ndarray:
    """H \\ b  (src:317-321 -> src:284-294): copies b, returns x = b[1:n]."""
    _fcheck(H)
    m, n = H.shape
    bb = np.array(b, dtype=np.float64, copy=True)
    lib().dhqr_oracle_solve(_ptr(bb), _ptr(H), m, n, H.strides[1] // 8, _ptr(alpha))
    return bb[:n].copy()


def partialdot(a: np.ndarray, b: np.ndarray, lo: int, hi: int) -> float:
    """src:42-49 with is = lo:hi-1 (0-based, hi exclusive)."""
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    return lib().dhqr_oracle_partialdot(_ptr(a), _ptr(b), lo, hi)


def form_qr(H: np.ndarray, alpha: np.ndarray) -> np.ndarray:
    """Q*R rebuilt from the factor format (test helper for ||A-QR||/||A||)."""
    _fcheck(H)
    m, n = H.shape
    B = np.zeros((m, n), order="F")
    lib().dhqr_oracle_form_qr(_ptr(H), m, n, H.strides[1] // 8, _ptr(alpha), _ptr(B), m)
    return B


def num_threads() -> int:
    return lib().dhqr_oracle_num_threads()


# ----------------------------------------------------------------------------- numpy twin
def alphafactor_np(x: float) -> float:
    """src:8"""
    return -np.sign(x)


def householder_np(A: np.ndarray):
    """Pure-numpy restatement of src:122-148 + src:198-213 (small cases only; O(m n^2) in Python
    column loops). Independent of the C oracle: tests require the two to agree."""
    H = np.array(A, dtype=np.float64, order="F", copy=True)
    m, n = H.shape
    alpha = np.zeros(n)
    for j in range(n):
        s = np.sqrt(np.sum(H[j:, j].astype(np.longdouble) ** 2)).astype(np.float64)  # src:129
        alpha[j] = s * alphafactor_np(H[j, j])          # src:130
        f = 1.0 / np.sqrt(s * (s + abs(H[j, j])))       # src:131
        H[j, j] -= alpha[j]                             # src:132
        H[j:, j] *= f                                   # src:133-135
        Hj = H[:, j].copy()                             # src:138-140
        if j + 1 < n:                                   # src:198-213
            sdot = Hj[j:] @ H[j:, j + 1:]               # partialdot per trailing column
            H[j:, j + 1:] -= np.outer(Hj[j:], sdot)     # hotloop!
    return H, alpha


def solve_np(H: np.ndarray, alpha: np.ndarray, b: np.ndarray) -> np.ndarray:
    """Pure-numpy restatement of src:215-224 + src:244-254 + src:284-294."""
    m, n = H.shape
    b = np.array(b, dtype=np.float64, copy=True)
    for j in range(n):
        s = H[j:, j] @ b[j:]
        b[j:] -= H[j:, j] * s
    for i in range(n - 1, -1, -1):
        b[i] = (b[i] - H[i, i + 1:n] @ b[i + 1:n]) / alpha[i]
    return b[:n].copy()
