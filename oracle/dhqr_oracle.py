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
_SRC_C64 = os.path.join(_HERE, "dhqr_oracle_c64.c")

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def build(force: bool = False) -> str:
    """Compile oracle/dhqr_oracle.c -> oracle/libdhqr_oracle.so (gcc, OpenMP). Generic x86-64
    code generation: the .so travels to the GPU box whose host CPU may differ."""
    newest = max(os.path.getmtime(_SRC), os.path.getmtime(_SRC_C64))
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < newest:
        subprocess.check_call(
            ["gcc", "-O3", "-fopenmp", "-fPIC", "-shared", "-std=c11", "-o", _SO, _SRC, _SRC_C64, "-lm"]
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
        # ComplexF64 methods (oracle/dhqr_oracle_c64.c)
        L.dhqr_oracle_alphafactor_c64.argtypes = [p, p]
        L.dhqr_oracle_partialdot_c64.argtypes = [p, p, i64, i64, p]
        L.dhqr_oracle_householder_inner_c64.argtypes = [p, i64, i64, i64, p, i64, i64, i64]
        L.dhqr_oracle_householder_c64.argtypes = [p, i64, i64, i64, p]
        L.dhqr_oracle_solve_c64.argtypes = [p, p, i64, i64, i64, p]
        L.dhqr_oracle_form_qr_c64.argtypes = [p, i64, i64, i64, p, p, i64]
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


# ============================================================================= ComplexF64 methods
def rand_matrix_c(m: int, n: int, seed: int) -> np.ndarray:
    """rand(ComplexF64, m, n) stand-in (test/runtests.jl:45): re and im parts are consecutive
    draws, A[i,j] = u01(seed, 2(i + j m)) + im*u01(seed, 2(i + j m) + 1) -- i.e. the REAL fill of
    the interleaved 2m x n view, which is how the device generates it."""
    r = u01(seed, np.arange(2 * m * n, dtype=np.uint64))
    return np.asfortranarray((r[0::2] + 1j * r[1::2]).reshape((m, n), order="F"))


def rand_vector_c(m: int, seed: int) -> np.ndarray:
    r = u01(seed, np.arange(2 * m, dtype=np.uint64))
    return r[0::2] + 1j * r[1::2]


def _zcheck(a: np.ndarray) -> np.ndarray:
    assert a.dtype == np.complex128 and a.flags["F_CONTIGUOUS"], "oracle wants column-major complex128"
    return a


def alphafactor_c(x: complex) -> complex:
    """src:9"""
    xin = np.array([x.real, x.imag])
    out = np.zeros(2)
    lib().dhqr_oracle_alphafactor_c64(_ptr(xin), _ptr(out))
    return complex(out[0], out[1])


def partialdot_c(a: np.ndarray, b: np.ndarray, lo: int, hi: int) -> complex:
    """src:51-59 with is = lo:hi-1 (0-based, hi exclusive): sum conj(a[i]) * b[i]."""
    a = np.ascontiguousarray(a, dtype=np.complex128)
    b = np.ascontiguousarray(b, dtype=np.complex128)
    out = np.zeros(2)
    lib().dhqr_oracle_partialdot_c64(_ptr(a), _ptr(b), lo, hi, _ptr(out))
    return complex(out[0], out[1])


def householder_c(A: np.ndarray):
    """qr!(A) for a ComplexF64 matrix; returns (H, alpha) with alpha complex (diag(R), not
    phase-normalised: alpha_j = -exp(i arg a_jj) * ||a_j||)."""
    H = np.array(A, dtype=np.complex128, order="F", copy=True)
    m, n = H.shape
    alpha = np.zeros(n, dtype=np.complex128)
    lib().dhqr_oracle_householder_c64(_ptr(H), m, n, H.strides[1] // 16, _ptr(alpha))
    return H, alpha


def solve_c(H: np.ndarray, alpha: np.ndarray, b: np.ndarray) -> np.ndarray:
    """H \\ b for ComplexF64 (src:317-321)."""
    _zcheck(H)
    m, n = H.shape
    bb = np.array(b, dtype=np.complex128, copy=True)
    al = np.ascontiguousarray(alpha, dtype=np.complex128)
    lib().dhqr_oracle_solve_c64(_ptr(bb), _ptr(H), m, n, H.strides[1] // 16, _ptr(al))
    return bb[:n].copy()


def form_qr_c(H: np.ndarray, alpha: np.ndarray) -> np.ndarray:
    _zcheck(H)
    m, n = H.shape
    B = np.zeros((m, n), dtype=np.complex128, order="F")
    al = np.ascontiguousarray(alpha, dtype=np.complex128)
    lib().dhqr_oracle_form_qr_c64(_ptr(H), m, n, H.strides[1] // 16, _ptr(al), _ptr(B), m)
    return B


def householder_c_np(A: np.ndarray):
    """Pure-numpy restatement of the ComplexF64 path (src:9, 51-59, 122-148, 171-213); small
    cases only.  Independent of the C oracle: tests require the two to agree."""
    H = np.array(A, dtype=np.complex128, order="F", copy=True)
    m, n = H.shape
    alpha = np.zeros(n, dtype=np.complex128)
    for j in range(n):
        col = H[j:, j]
        s = float(np.sqrt(np.sum(col.real.astype(np.longdouble) ** 2 + col.imag.astype(np.longdouble) ** 2)))
        alpha[j] = s * (-np.exp(1j * np.angle(H[j, j])))   # src:9, src:130
        f = 1.0 / np.sqrt(s * (s + abs(H[j, j])))          # src:131
        H[j, j] -= alpha[j]
        H[j:, j] *= f
        Hj = H[:, j].copy()
        if j + 1 < n:
            sdot = np.conj(Hj[j:]) @ H[j:, j + 1:]          # src:51-59
            H[j:, j + 1:] -= np.outer(Hj[j:], sdot)         # src:150-154
    return H, alpha


def solve_c_np(H: np.ndarray, alpha: np.ndarray, b: np.ndarray) -> np.ndarray:
    m, n = H.shape
    b = np.array(b, dtype=np.complex128, copy=True)
    for j in range(n):
        s = np.conj(H[j:, j]) @ b[j:]
        b[j:] -= H[j:, j] * s
    for i in range(n - 1, -1, -1):
        b[i] = (b[i] - H[i, i + 1:n] @ b[i + 1:n]) / alpha[i]
    return b[:n].copy()
