"""Host-side mirror of the reference's Julia API on top of libdhqr.so.

Reference (src/DistributedHouseholderQR.jl)        here
  qr!(A)                         :311-315          qr_(A, nb=...)             ('!' -> trailing '_')
  DistributedHouseholderQRStruct :296-309          DistributedHouseholderQRStruct(A, α)
  H \\ b                          :317-321          ldiv(H, b)  /  H.solve(b)
  householder!(A, α)             :113-120          householder_(A, α, nb=...)
  solve_householder!(b, H, α)    :284-294          solve_householder_(b, H, α)
  partialdot(a, b, is, T)        :42-49, :51-59    partialdot(a, b, lo, hi)

The reference's functions are generic over the element type (its tests run Float64 and ComplexF64,
test/runtests.jl:43); here the dtype of the argument selects the method the same way: float64 ->
the *_f64 entry points (blocked MFMA path by default), complex128 -> the *_c64 entry points
(unblocked path; nb must be None or 0).

Inputs are either host numpy arrays (column-major float64; goes through the host-in/host-out
C entry points, like qr!(::Matrix)) or CUDA/HIP torch tensors in column-major layout
(stride(0) == 1; device-resident entry points).  PyTorch is only plumbing here (device memory and
streams); every FLOP runs in the HIP library.  No CPU fallback exists.
"""
from __future__ import annotations

import ctypes
import os
from typing import Dict, Optional

import numpy as np

from . import _lib
from ._lib import NB, DHQRError, Stats, check

try:  # torch is only needed for device-resident tensors
    import torch
except Exception:  # pragma: no cover
    torch = None


class Context:
    """One dhqr_ctx per (process, GPU)."""

    def __init__(self, device: int = 0):
        self._h = ctypes.c_void_p()
        self.device = device
        check(_lib.lib().dhqr_create(ctypes.byref(self._h), device))

    @property
    def handle(self):
        return self._h

    def use_torch_stream(self):
        """run on torch's current stream of this device (so torch ops and dhqr kernels order).  DHQR_OWN_STREAM=1
        (experiments): keep the context's own stream; the caller then orders with torch by synchronising."""
        if os.environ.get("DHQR_OWN_STREAM") == "1":
            check(_lib.lib().dhqr_use_own_stream(self._h))
            return
        s = torch.cuda.current_stream(self.device).cuda_stream
        check(_lib.lib().dhqr_set_stream(self._h, ctypes.c_void_p(s)))

    def synchronize(self):
        check(_lib.lib().dhqr_synchronize(self._h))

    def trim(self):
        """release the device copy / staging buffers / solve workspaces the context keeps between calls (dhqr_trim)"""
        check(_lib.lib().dhqr_trim(self._h))

    def set_profiling(self, on: bool):
        check(_lib.lib().dhqr_set_profiling(self._h, 1 if on else 0))

    def reset_stats(self):
        check(_lib.lib().dhqr_reset_stats(self._h))

    def stats(self) -> dict:
        st = Stats()
        check(_lib.lib().dhqr_get_stats(self._h, ctypes.byref(st)))
        return st.asdict()

    def panel_counters(self):
        """(panels done by the R-first fast path, panels that fell back to the step kernels)"""
        a, b = ctypes.c_int64(), ctypes.c_int64()
        check(_lib.lib().dhqr_get_panel_counters(self._h, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def set_r_source(self, source: int):
        """1 Gram/Cholesky (default), 2 CholeskyQR2, 3 TSQR-HR (csrc/dhqr_tsqr.h) for the R-first panel path"""
        check(_lib.lib().dhqr_set_r_source(self._h, int(source)))

    def set_tsqr_rung(self, on: bool):
        """False: rejected panels skip the TSQR-HR rung (straight to the column-by-column kernels)"""
        check(_lib.lib().dhqr_set_tsqr_rung(self._h, 1 if on else 0))

    def set_small_route(self, on: bool):
        """False: matrices that fit one compute unit's registers go through the general drivers too (csrc/dhqr_small.h)"""
        check(_lib.lib().dhqr_set_small_route(self._h, 1 if on else 0))

    def solve_retries(self) -> int:
        """solves repeated with the per-step kernels after a wait of the persistent Q'b kernel expired (dhqr.h)"""
        a = ctypes.c_int64()
        check(_lib.lib().dhqr_get_solve_retries(self._h, ctypes.byref(a)))
        return a.value

    def tsqr_count(self) -> int:
        a = ctypes.c_int64()
        check(_lib.lib().dhqr_get_tsqr_count(self._h, ctypes.byref(a)))
        return a.value

    def close(self):
        if self._h:
            _lib.lib().dhqr_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


_contexts: Dict[int, Context] = {}


def get_context(device: Optional[int] = None) -> Context:
    if device is None:
        device = torch.cuda.current_device() if (torch is not None and torch.cuda.is_available()) else 0
    if device not in _contexts:
        _contexts[device] = Context(device)
    return _contexts[device]


# ------------------------------------------------------------------------------- helpers
def _is_tensor(x) -> bool:
    return torch is not None and isinstance(x, torch.Tensor)


def _is_complex(x) -> bool:
    if _is_tensor(x):
        return x.dtype == torch.complex128
    return isinstance(x, np.ndarray) and x.dtype == np.complex128


def _host_ld(F) -> int:
    """leading dimension (in elements) of a column-major host matrix; numpy's relaxed strides report an
    arbitrary stride for a single column, where the reference's Matrix simply has ld = m"""
    m, n = F.shape
    return F.strides[1] // F.itemsize if n > 1 else max(m, 1)


DEFAULT_UNBLOCKED_MAX_ROWS = 480  # Float64: at most this many rows -> nb = 0 by default (tools/mid_bench.py)


def _resolve_nb(A, nb):
    """nb=None: the default of the element type (Float64: 128 blocked, but the reference's unblocked order up to 480 rows,
    where the K-reflector passes finish before the blocked driver's panel chains do: 440 x 400 1.46 ms against 1.53,
    300 x 260 0.97 against 1.00, 520 x 512 1.86 against 1.49 the other way, profiles/r06_mid_sizes.txt; ComplexF64: 64
    blocked -- the trailing update runs on the FP64 MFMA kernels through the real embedding -- for n >= 256, unblocked
    below)"""
    if _is_complex(A):
        if nb not in (None, 0, _lib.ZNB):
            raise ValueError(f"ComplexF64: nb must be None, 0 (unblocked) or {_lib.ZNB} (blocked)")
        if nb is None:
            return _lib.ZNB if A.shape[1] >= 256 else 0
        return nb
    if nb is None:
        return 0 if A.shape[0] <= DEFAULT_UNBLOCKED_MAX_ROWS else NB
    return nb


def _dev_matrix(A, dtype=None):
    """(ptr, m, n, lda, device) of a column-major CUDA tensor (float64 unless `dtype` is given);
    lda counts elements."""
    dtype = torch.float64 if dtype is None else dtype
    if A.dtype != dtype or not A.is_cuda:
        raise TypeError(f"device path needs a {dtype} CUDA tensor")
    if A.dim() != 2:
        raise ValueError("matrix expected")
    m, n = A.shape
    if m > 1 and A.stride(0) != 1:
        raise ValueError("column-major layout required (stride(0) == 1); build it with "
                         "empty_colmajor(m, n) or X.t() of a contiguous (n, m) tensor")
    lda = A.stride(1) if n > 1 else max(m, 1)
    return ctypes.c_void_p(A.data_ptr()), m, n, lda, A.device.index


def _dev_vector(v, length=None, dtype=None):
    dtype = torch.float64 if dtype is None else dtype
    if v.dtype != dtype or not v.is_cuda or v.dim() != 1 or (v.numel() > 1 and v.stride(0) != 1):
        raise TypeError(f"contiguous {dtype} CUDA vector expected")
    if length is not None and v.numel() < length:
        raise ValueError(f"vector shorter than {length}")
    return ctypes.c_void_p(v.data_ptr())


def empty_colmajor(m: int, n: int, device="cuda"):
    """uninitialised m x n float64 device matrix in column-major layout (lda = m)."""
    return torch.empty((n, m), dtype=torch.float64, device=device).t()


def rand_colmajor(m: int, n: int, seed: int, device="cuda", *, global_m=None, row0=0, colblock=NB,
                  nranks=1, rank=0):
    """Device-side synthetic input: A[i,j] = u01(seed, gi + gj*global_m), the generator shared with
    oracle/ (stands in for rand(T,m,n), test/runtests.jl:45).  With nranks > 1 fills the LOCAL
    block of a block-cyclic column layout."""
    A = empty_colmajor(m, n, device)
    ctx = get_context(A.device.index)
    ctx.use_torch_stream()
    ptr, m_, n_, lda, _ = _dev_matrix(A)
    check(_lib.lib().dhqr_fill_uniform_f64(ctx.handle, ptr, m_, n_, lda, seed,
                                           global_m if global_m is not None else m, row0, colblock,
                                           nranks, rank))
    return A


def rand_vector_device(m: int, seed: int, device="cuda"):
    return rand_colmajor(m, 1, seed, device).reshape(-1)


def rand_colmajor_c(m: int, n: int, seed: int, device="cuda"):
    """rand(ComplexF64, m, n) stand-in on the device: the Float64 generator run over the
    interleaved 2m x n real view, A[i,j] = u01(seed, 2(i + j m)) + im*u01(seed, 2(i + j m) + 1)
    (identical to oracle rand_matrix_c)."""
    A = torch.empty((n, m), dtype=torch.complex128, device=device).t()
    ctx = get_context(A.device.index)
    ctx.use_torch_stream()
    check(_lib.lib().dhqr_fill_uniform_f64(ctx.handle, ctypes.c_void_p(A.data_ptr()), 2 * m, n, 2 * m, seed,
                                           2 * m, 0, NB, 1, 0))
    return A


# ------------------------------------------------------------------------------- API mirror
class DistributedHouseholderQRStruct:
    """src:296-309: the factored matrix `A` (V on/below the diagonal, R strictly above) and
    `α` = diag(R).  `alpha` is an ASCII alias of `α`."""

    def __init__(self, A, α=None):
        self.A = A
        if α is None:  # src:306-309  α = zeros(eltype(A), size(A, 2))
            n = A.shape[1]
            α = torch.zeros(n, dtype=A.dtype, device=A.device) if _is_tensor(A) else np.zeros(n, dtype=A.dtype)
        self.α = α

    @property
    def alpha(self):
        return self.α

    def solve(self, b):
        return ldiv(self, b)

    def __repr__(self):
        return f"DistributedHouseholderQRStruct(A={tuple(self.A.shape)}, α={tuple(self.α.shape)})"


def _householder_c64(A, α, nb=0):
    """ComplexF64 method of householder! (src:9, 51-59, 122-148, 171-213): nb = 0 unblocked, 64 blocked."""
    L = _lib.lib()
    if _is_tensor(A):
        ptr, m, n, lda, dev = _dev_matrix(A, torch.complex128)
        ctx = get_context(dev)
        ctx.use_torch_stream()
        check(L.dhqr_factor_c64_nb(ctx.handle, ptr, m, n, lda, _dev_vector(α, n, torch.complex128), nb))
        ctx.synchronize()  # qr! is synchronous in the reference; this also collects a pipeline hand-over error (dhqr.h)
        return A, α
    if not isinstance(α, np.ndarray) or α.dtype != np.complex128 or α.size < A.shape[1] or not α.flags.c_contiguous:
        raise TypeError("α must be a contiguous complex128 numpy vector of length n")
    m, n = A.shape
    F = A if A.flags.f_contiguous else np.asfortranarray(A)
    check(L.dhqr_qr_c64_nb(get_context().handle, F.ctypes.data_as(ctypes.c_void_p), m, n,
                           _host_ld(F), α.ctypes.data_as(ctypes.c_void_p), nb))
    if F is not A:
        A[...] = F
    return A, α


def householder_(A, α, nb: Optional[int] = None):
    """householder!(A, α) (src:113): factor A in place, fill α. nb=0 -> unblocked rank-1 path
    (the reference's algorithm verbatim), nb=128 -> blocked MFMA path (the Float64 default).
    complex128 input selects the ComplexF64 method (unblocked). Returns (A, α)."""
    L = _lib.lib()
    nb = _resolve_nb(A, nb)
    if _is_complex(A):
        return _householder_c64(A, α, nb)
    if _is_tensor(A):
        ptr, m, n, lda, dev = _dev_matrix(A)
        ctx = get_context(dev)
        ctx.use_torch_stream()
        check(L.dhqr_factor_f64(ctx.handle, ptr, m, n, lda, _dev_vector(α, n), nb))
        if nb == 0:  # the unblocked path is fully asynchronous and uses the inter-workgroup lead pipeline: its error word
            ctx.synchronize()  # is only reported by a synchronising entry point (the blocked driver reads it itself)
        return A, α
    if not isinstance(A, np.ndarray) or A.dtype != np.float64 or A.ndim != 2:
        raise TypeError("float64 numpy matrix or CUDA tensor expected")
    if not isinstance(α, np.ndarray) or α.dtype != np.float64 or α.size < A.shape[1] or not α.flags.c_contiguous:
        raise TypeError("α must be a contiguous float64 numpy vector of length n")
    m, n = A.shape
    ctx = get_context()
    F = A if A.flags.f_contiguous else np.asfortranarray(A)
    check(L.dhqr_qr_f64(ctx.handle, F.ctypes.data_as(ctypes.c_void_p), m, n, _host_ld(F),
                        α.ctypes.data_as(ctypes.c_void_p), nb))
    if F is not A:
        A[...] = F  # in-place semantics of qr! for row-major callers
    return A, α


def qr_(A, nb: Optional[int] = None) -> DistributedHouseholderQRStruct:
    """qr!(A) (src:311-315): mutates A, returns the struct."""
    H = DistributedHouseholderQRStruct(A)
    householder_(H.A, H.α, nb=nb)
    return H


def solve_householder_(b, H, α):
    """solve_householder!(b, H, α) (src:284-294): returns x = b[1:n] (a copy, like Julia's b[1:n]).  A device
    tensor b is overwritten like the reference's b (b <- Q'b, then back substitution in place); a HOST b is
    uploaded and left untouched (the solve happens in device memory)."""
    L = _lib.lib()
    if _is_complex(H):
        if _is_tensor(H):
            ptr, m, n, lda, dev = _dev_matrix(H, torch.complex128)
            ctx = get_context(dev)
            ctx.use_torch_stream()
            check(L.dhqr_solve_c64(ctx.handle, ptr, m, n, lda, _dev_vector(α, n, torch.complex128),
                                   _dev_vector(b, m, torch.complex128)))
            x = b[:n].clone()
            ctx.synchronize()
            return x
        m, n = H.shape
        F = H if H.flags.f_contiguous else np.asfortranarray(H)
        x = np.empty(n, dtype=np.complex128)
        bb = np.ascontiguousarray(b, dtype=np.complex128)
        check(L.dhqr_ldiv_c64(get_context().handle, F.ctypes.data_as(ctypes.c_void_p), m, n,
                              _host_ld(F),
                              np.ascontiguousarray(α, dtype=np.complex128).ctypes.data_as(ctypes.c_void_p),
                              bb.ctypes.data_as(ctypes.c_void_p), x.ctypes.data_as(ctypes.c_void_p)))
        return x
    if _is_tensor(H):
        ptr, m, n, lda, dev = _dev_matrix(H)
        ctx = get_context(dev)
        ctx.use_torch_stream()
        check(L.dhqr_solve_f64(ctx.handle, ptr, m, n, lda, _dev_vector(α, n), _dev_vector(b, m)))
        ctx.synchronize()  # an expired hand-over wait is reported here -- or the solve repeated without the persistent kernel (dhqr.h)
        return b[:n].clone()
    m, n = H.shape
    F = H if H.flags.f_contiguous else np.asfortranarray(H)
    x = np.empty(n)
    bb = np.ascontiguousarray(b, dtype=np.float64)
    ctx = get_context()
    check(L.dhqr_ldiv_f64(ctx.handle, F.ctypes.data_as(ctypes.c_void_p), m, n, _host_ld(F),
                          np.ascontiguousarray(α).ctypes.data_as(ctypes.c_void_p),
                          bb.ctypes.data_as(ctypes.c_void_p), x.ctypes.data_as(ctypes.c_void_p)))
    return x


def ldiv(H: DistributedHouseholderQRStruct, b):
    """`H \\ b` (src:317-321): least-squares solution of length n; the caller's b is NOT modified
    (the reference copies it into a SharedArray first, src:318)."""
    if _is_tensor(H.A):
        return solve_householder_(b.clone(), H.A, H.α)
    return solve_householder_(b, H.A, H.α)


def _partialdot_c64(a, b, lo: int, hi: int) -> complex:
    """partialdot(a, b, lo:hi-1, ComplexF64) (src:51-59): sum conj(a[i]) b[i]"""
    L = _lib.lib()
    out = (ctypes.c_double * 2)()
    if not _is_tensor(a):
        a = np.ascontiguousarray(a, dtype=np.complex128)
        b = np.ascontiguousarray(b, dtype=np.complex128)
        check(L.dhqr_partialdot_host_c64(get_context().handle, a.ctypes.data_as(ctypes.c_void_p),
                                         b.ctypes.data_as(ctypes.c_void_p), lo, hi, out))
        return complex(out[0], out[1])
    ctx = get_context(a.device.index)
    ctx.use_torch_stream()
    check(L.dhqr_partialdot_c64(ctx.handle, _dev_vector(a, hi, torch.complex128),
                                _dev_vector(b, hi, torch.complex128), lo, hi, out))
    return complex(out[0], out[1])


def partialdot(a, b, lo: int, hi: int):
    """partialdot(a, b, lo:hi-1, T) (src:42-49 Float64, src:51-59 ComplexF64 = conj(a).b), 0-based
    with hi exclusive, reduced on the GPU."""
    L = _lib.lib()
    if _is_complex(a) or _is_complex(b):
        return _partialdot_c64(a, b, lo, hi)
    if not _is_tensor(a):  # host vectors: the entry point the Julia module binds
        a = np.ascontiguousarray(a, dtype=np.float64)
        b = np.ascontiguousarray(b, dtype=np.float64)
        out = ctypes.c_double()
        check(L.dhqr_partialdot_host_f64(get_context().handle, a.ctypes.data_as(ctypes.c_void_p),
                                         b.ctypes.data_as(ctypes.c_void_p), lo, hi, ctypes.byref(out)))
        return out.value
    ctx = get_context(a.device.index)
    ctx.use_torch_stream()
    out = ctypes.c_double()
    check(L.dhqr_partialdot_f64(ctx.handle, _dev_vector(a, hi), _dev_vector(b, hi), lo, hi, ctypes.byref(out)))
    return out.value


# ------------------------------------------------------------------------------- metric helpers
def apply_q_(H: DistributedHouseholderQRStruct, B, trans: bool):
    """B <- Q' B (trans) or Q B, in place, device tensors only."""
    ptr, m, n, lda, dev = _dev_matrix(H.A)
    bptr, mb, nrhs, ldb, _ = _dev_matrix(B)
    if mb != m:
        raise ValueError("row mismatch")
    ctx = get_context(dev)
    ctx.use_torch_stream()
    check(_lib.lib().dhqr_apply_q_f64(ctx.handle, ptr, m, n, lda, bptr, nrhs, ldb, 1 if trans else 0))
    return B


def get_r(H: DistributedHouseholderQRStruct):
    r"""n x n upper-triangular R of a Float64 device factorisation (strict upper part of H.A + α on the
    diagonal, src:296-309), as a new column-major device tensor.  The reference exposes R only
    implicitly through `\`; SURVEY.md section 8f rank 2 asks for the explicit extraction."""
    ptr, m, n, lda, dev = _dev_matrix(H.A)
    W = empty_colmajor(m, n, H.A.device)
    wptr, _, _, ldw, _ = _dev_matrix(W)
    ctx = get_context(dev)
    ctx.use_torch_stream()
    check(_lib.lib().dhqr_form_r0_f64(ctx.handle, ptr, m, n, lda, _dev_vector(H.α, n), wptr, ldw, NB, 1, 0))
    return W[:n, :]


def get_q(H: DistributedHouseholderQRStruct):
    """explicit thin Q (m x n, column-major device tensor): Q = H_1 ... H_n applied to [I; 0]."""
    m, n = H.A.shape
    Q = empty_colmajor(m, n, H.A.device)
    Q.zero_()
    Q.diagonal().fill_(1.0)
    return apply_q_(H, Q, trans=False)


def residual(H: DistributedHouseholderQRStruct, Aorig, work=None) -> float:
    """||Aorig - Q R||_F / ||Aorig||_F on the device (north-star metric)."""
    ptr, m, n, lda, dev = _dev_matrix(H.A)
    optr, mo, no, ldo, _ = _dev_matrix(Aorig)
    if (mo, no) != (m, n):
        raise ValueError("shape mismatch")
    if work is None:
        work = empty_colmajor(m, n, H.A.device)
    wptr, _, _, ldw, _ = _dev_matrix(work)
    if ldw != m:
        raise ValueError("work must have leading dimension m")
    ctx = get_context(dev)
    ctx.use_torch_stream()
    out = ctypes.c_double()
    check(_lib.lib().dhqr_residual_f64(ctx.handle, ptr, m, n, lda, _dev_vector(H.α, n), optr, ldo, wptr,
                                       ctypes.byref(out)))
    return out.value


_bench_ctx = {}


def bench_context(device: Optional[int] = None):
    """(libdhqr_bench.so, context handle created by it) for the micro-benchmarks of include/dhqr_bench.h -- they are not
    in the product library, and contexts are per library."""
    import torch as _t
    dev = _t.cuda.current_device() if device is None else int(device)
    if dev not in _bench_ctx:
        B = _lib.lib_bench()
        h = ctypes.c_void_p()
        rc = B.dhqr_create(ctypes.byref(h), dev)
        if rc != 0:
            raise DHQRError(rc, B.dhqr_last_error().decode(errors="replace"))
        _bench_ctx[dev] = (B, h)
    return _bench_ctx[dev]


def bench_check(B, rc):
    if rc != 0:
        raise DHQRError(rc, B.dhqr_last_error().decode(errors="replace"))


def bench_mfma_tflops(device: Optional[int] = None) -> float:
    B, h = bench_context(device)
    out = ctypes.c_double()
    bench_check(B, B.dhqr_bench_mfma_f64(h, ctypes.byref(out)))
    return out.value


def bench_stream_gbps(nbytes: int = 1 << 30, device: Optional[int] = None) -> float:
    B, h = bench_context(device)
    out = ctypes.c_double()
    bench_check(B, B.dhqr_bench_stream_f64(h, nbytes, ctypes.byref(out)))
    return out.value
