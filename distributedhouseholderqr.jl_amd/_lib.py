"""ctypes binding of libdhqr.so (include/dhqr.h).  The HIP library is the product: there is no CPU
fallback here -- a missing or unloadable library raises immediately."""
from __future__ import annotations

import ctypes
import os
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_PKG, "libdhqr.so")
BENCH_SO_PATH = os.path.join(_PKG, "libdhqr_bench.so")  # superset with the micro-benchmarks (include/dhqr_bench.h)
CSRC = os.path.join(_PKG, "csrc")
NB = 128  # DHQR_NB
CS_BLOCK = 256  # DHQR_CS_BLOCK: cyclic block of the multi-GPU column split (a pair of panels)
ZNB = 64  # DHQR_ZNB: complex reflectors per panel of the blocked ComplexF64 path

OK, EINVAL, EHIP, ENOMEM, ENODEVICE, ECOMM = 0, -1, -2, -3, -4, -5
COMM_SELF, COMM_RCCL, COMM_LOCAL, COMM_CALLBACK = 0, 1, 2, 3
UNIQUE_ID_BYTES = 128


class DHQRError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libdhqr error {code}: {msg}")
        self.code = code


class Stats(ctypes.Structure):
    _fields_ = [(n, ctypes.c_double) for n in
                ("ms_panel", "ms_tbuild", "ms_gemm_vta", "ms_gemm_tw", "ms_gemm_avw", "ms_rank1",
                 "ms_solve")] + \
               [(n, ctypes.c_int64) for n in
                ("n_panel", "n_tbuild", "n_gemm_vta", "n_gemm_tw", "n_gemm_avw", "n_rank1",
                 "n_solve")] + \
               [(n, ctypes.c_double) for n in ("flops_gemm_vta", "flops_gemm_avw", "bytes_rank1", "bytes_panel")]

    def asdict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


def build(force: bool = False) -> str:
    """hipcc --offload-arch=gfx950 build of csrc/ -> libdhqr.so (cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    srcs.append(os.path.join(_PKG, "..", "include", "dhqr.h"))
    newest = max(os.path.getmtime(s) for s in srcs)
    srcs.append(os.path.join(_PKG, "..", "include", "dhqr_bench.h"))
    newest = max(newest, os.path.getmtime(srcs[-1]))
    if force or any(not os.path.exists(so) or os.path.getmtime(so) < newest for so in (SO_PATH, BENCH_SO_PATH)):
        subprocess.check_call(["bash", os.path.join(CSRC, "build.sh")])
    return SO_PATH


_i32, _i64, _u64, _f64 = ctypes.c_int32, ctypes.c_int64, ctypes.c_uint64, ctypes.c_double
_p = ctypes.c_void_p
_pp = ctypes.POINTER(ctypes.c_void_p)
_pd = ctypes.POINTER(ctypes.c_double)
_pi32, _pi64 = ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int64)
# dhqr_bcast_fn / dhqr_allreduce_fn (include/dhqr.h): the CALLBACK transport
BCAST_FN = ctypes.CFUNCTYPE(ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p)
ALLREDUCE_FN = ctypes.CFUNCTYPE(ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p)

# name -> (restype, argtypes); mirrors include/dhqr.h one to one
SIGNATURES = {
    "dhqr_version": (_i32, []),
    "dhqr_last_error": (ctypes.c_char_p, []),
    "dhqr_device_count": (_i32, [ctypes.POINTER(_i32)]),
    "dhqr_create": (_i32, [_pp, _i32]),
    "dhqr_destroy": (_i32, [_p]),
    "dhqr_set_stream": (_i32, [_p, _p]),
    "dhqr_use_own_stream": (_i32, [_p]),
    "dhqr_synchronize": (_i32, [_p]),
    "dhqr_trim": (_i32, [_p]),
    "dhqr_set_profiling": (_i32, [_p, _i32]),
    "dhqr_reset_stats": (_i32, [_p]),
    "dhqr_get_stats": (_i32, [_p, ctypes.POINTER(Stats)]),
    "dhqr_get_panel_counters": (_i32, [_p, ctypes.POINTER(_i64), ctypes.POINTER(_i64)]),
    "dhqr_set_r_source": (_i32, [_p, _i32]),
    "dhqr_set_tsqr_rung": (_i32, [_p, _i32]),
    "dhqr_set_small_route": (_i32, [_p, _i32]),
    "dhqr_get_solve_retries": (_i32, [_p, _pi64]),
    "dhqr_tsqr_r_f64": (_i32, [_p, _p, _i64, _i64, _p]),
    "dhqr_get_tsqr_count": (_i32, [_p, ctypes.POINTER(_i64)]),
    "dhqr_fill_uniform_f64": (_i32, [_p, _p, _i64, _i64, _i64, _u64, _i64, _i64, _i64, _i32, _i32]),
    "dhqr_factor_f64": (_i32, [_p, _p, _i64, _i64, _i64, _p, _i32]),
    "dhqr_qr_f64": (_i32, [_p, _p, _i64, _i64, _i64, _p, _i32]),
    "dhqr_solve_f64": (_i32, [_p, _p, _i64, _i64, _i64, _p, _p]),
    "dhqr_backsub_block_f64": (_i32, [_p, _p, _i64, _p, _p, _i64, _i64, _i32, _i32]),
    "dhqr_ldiv_f64": (_i32, [_p, _p, _i64, _i64, _i64, _p, _p, _p]),
    "dhqr_partialdot_f64": (_i32, [_p, _p, _p, _i64, _i64, _pd]),
    "dhqr_partialdot_host_f64": (_i32, [_p, _p, _p, _i64, _i64, _pd]),
    "dhqr_factor_c64": (_i32, [_p, _p, _i64, _i64, _i64, _p]),
    "dhqr_qr_c64": (_i32, [_p, _p, _i64, _i64, _i64, _p]),
    "dhqr_factor_c64_nb": (_i32, [_p, _p, _i64, _i64, _i64, _p, _i32]),
    "dhqr_qr_c64_nb": (_i32, [_p, _p, _i64, _i64, _i64, _p, _i32]),
    "dhqr_solve_c64": (_i32, [_p, _p, _i64, _i64, _i64, _p, _p]),
    "dhqr_ldiv_c64": (_i32, [_p, _p, _i64, _i64, _i64, _p, _p, _p]),
    "dhqr_partialdot_c64": (_i32, [_p, _p, _p, _i64, _i64, _pd]),
    "dhqr_partialdot_host_c64": (_i32, [_p, _p, _p, _i64, _i64, _pd]),
    "dhqr_apply_q_f64": (_i32, [_p, _p, _i64, _i64, _i64, _p, _i64, _i64, _i32]),
    "dhqr_residual_f64": (_i32, [_p, _p, _i64, _i64, _i64, _p, _p, _i64, _p, _pd]),
    "dhqr_panel_ldv": (_i64, [_i64]),
    "dhqr_panel_buffer_elems": (_i64, [_i64]),
    "dhqr_panel_factor_f64": (_i32, [_p, _p, _i64, _i64, _i64, _p]),
    "dhqr_panel_pack_f64": (_i32, [_p, _p, _i64, _i64, _i64, _p]),
    "dhqr_form_r0_f64": (_i32, [_p, _p, _i64, _i64, _i64, _p, _p, _i64, _i64, _i32, _i32]),
    "dhqr_diff_norms_f64": (_i32, [_p, _p, _i64, _p, _i64, _i64, _i64, _pd]),
    "dhqr_panel_apply_f64": (_i32, [_p, _p, _i64, _p, _i64, _i64, _i32]),
    "dhqr_comm_unique_id": (_i32, [_p]),
    "dhqr_comm_create_rank": (_i32, [_pp, _p, _i32, _i32, _p]),
    "dhqr_comm_create_callbacks": (_i32, [_pp, _p, _i32, _i32, BCAST_FN, ALLREDUCE_FN, _p]),
    "dhqr_comm_destroy": (_i32, [_p]),
    "dhqr_comm_info": (_i32, [_p, _pi32, _pi32, _pi32, _pi64]),
    "dhqr_comm_counters": (_i32, [_p, _pi64]),
    "dhqr_comm_get_bcast_tuning": (_i32, [_p, _pi32, _pd, _pd]),
    "dhqr_comm_rccl_nranks": (_i32, [_p, _pi32, _pi32]),
    "dhqr_cs_local_cols": (_i64, [_i64, _i32, _i32]),
    "dhqr_cs_contiguous_range": (None, [_i64, _i32, _i32, _pi64, _pi64]),
    "dhqr_cs_fill_uniform_f64": (_i32, [_p, _p, _i64, _i64, _i64, _u64]),
    "dhqr_cs_factor_f64": (_i32, [_p, _p, _i64, _i64, _i64, _p]),
    "dhqr_cs_residual_f64": (_i32, [_p, _p, _i64, _i64, _i64, _p, _u64, _p, _p, _pd]),
    "dhqr_cs_solve_f64": (_i32, [_p, _p, _i64, _i64, _i64, _p, _p, _p]),
    "dhqr_cs_load_contiguous_f64": (_i32, [_p, _p, _i64, _i64, _i64, _p, _i64, _p]),
    "dhqr_cs_store_contiguous_f64": (_i32, [_p, _p, _i64, _i64, _i64, _p, _i64, _p]),
    "dhqr_cs_qr_darray_f64": (_i32, [_p, _p, _i64, _i64, _i64, _p]),
    "dhqr_mg_create": (_i32, [_pp, _pi32, _i32]),
    "dhqr_mg_destroy": (_i32, [_p]),
    "dhqr_mg_info": (_i32, [_p, _pi32, _pi32, _pi64, _pi64]),
    "dhqr_mg_get_bcast_tuning": (_i32, [_p, _pi32, _pd, _pd]),
    "dhqr_mg_rccl_nranks": (_i32, [_p, _pi32, _pi32]),
    "dhqr_mg_comm_counters": (_i32, [_p, _i32, _pi64]),
    "dhqr_mg_alloc_f64": (_i32, [_p, _i64, _i64]),
    "dhqr_mg_fill_uniform_f64": (_i32, [_p, _u64]),
    "dhqr_mg_factor_f64": (_i32, [_p]),
    "dhqr_mg_residual_f64": (_i32, [_p, _u64, _pd]),
    "dhqr_mg_upload_f64": (_i32, [_p, _p, _i64, _p]),
    "dhqr_mg_download_f64": (_i32, [_p, _p, _i64, _p]),
    "dhqr_mg_qr_f64": (_i32, [_p, _p, _i64, _i64, _i64, _p]),
    "dhqr_mg_qr_c64": (_i32, [_p, _p, _i64, _i64, _i64, _p]),
    "dhqr_cs_local_cols_c64": (_i64, [_i64, _i32, _i32]),
    "dhqr_cs_factor_c64": (_i32, [_p, _p, _i64, _i64, _i64, _p]),
    "dhqr_cs_qr_darray_c64": (_i32, [_p, _p, _i64, _i64, _i64, _p]),
    "dhqr_comm_timing": (_i32, [_p, _i32, _p]),
    "dhqr_mg_comm_timing": (_i32, [_p, _i32, _i32, _p]),
    "dhqr_cs_ldiv_darray_f64": (_i32, [_p, _p, _i64, _i64, _i64, _p, _p, _p]),
    "dhqr_cs_ldiv_darray_c64": (_i32, [_p, _p, _i64, _i64, _i64, _p, _p, _p]),
    "dhqr_cs_solve_work_c64": (_i64, [_i64, _i32]),
    "dhqr_cs_solve_c64": (_i32, [_p, _p, _i64, _i64, _i64, _p, _p, _p]),
    "dhqr_mg_ldiv_c64": (_i32, [_p, _p, _i64, _i64, _i64, _p, _p, _p]),
    "dhqr_mg_solve_f64": (_i32, [_p, _p, _p]),
    "dhqr_mg_ldiv_f64": (_i32, [_p, _p, _i64, _i64, _i64, _p, _p, _p]),
    "dhqr_mg_set_profiling": (_i32, [_p, _i32]),
    "dhqr_mg_reset_stats": (_i32, [_p]),
    "dhqr_mg_get_stats": (_i32, [_p, _i32, ctypes.POINTER(Stats), _pi64, _pi64, _pi64]),
    "dhqr_rs_row_range": (None, [_i64, _i32, _i32, _pi64, _pi64]),
    "dhqr_rs_fill_uniform_f64": (_i32, [_p, _p, _i64, _i64, _i64, _u64]),
    "dhqr_rs_factor_f64": (_i32, [_p, _p, _i64, _i64, _i64, _p]),
    "dhqr_rs_residual_f64": (_i32, [_p, _p, _i64, _i64, _i64, _p, _u64, _p, _p, _pd]),
    "dhqr_rs_solve_f64": (_i32, [_p, _p, _i64, _i64, _i64, _p, _p, _p]),
    "dhqr_mg_rs_alloc_f64": (_i32, [_p, _i64, _i64]),
    "dhqr_mg_rs_fill_uniform_f64": (_i32, [_p, _u64]),
    "dhqr_mg_rs_factor_f64": (_i32, [_p]),
    "dhqr_mg_rs_residual_f64": (_i32, [_p, _u64, _pd]),
    "dhqr_mg_rs_transfer_f64": (_i32, [_p, _p, _i64, _p, _i32]),
    "dhqr_mg_rs_solve_f64": (_i32, [_p, _p, _p]),
}

# include/dhqr_bench.h: exported by libdhqr_bench.so only
BENCH_SIGNATURES = {
    "dhqr_bench_mfma_f64": (_i32, [_p, _pd]),
    "dhqr_bench_issue_f64": (_i32, [_p, _i32, _i32, _pd, _pd]),
    "dhqr_bench_issue2_f64": (_i32, [_p, _i32, _i32, _i32, _pd]),
    "dhqr_bench_stream_f64": (_i32, [_p, _i64, _pd]),
    "dhqr_debug_mfma_probe": (_i32, [_p, _p, _p, _p]),
    "dhqr_bench_gemm_f64": (_i32, [_p, _i32, _i64, _i64, _i32, _pd]),
    "dhqr_bench_mma_probe_f64": (_i32, [_p, _i32, _i32, _pd]),
    "dhqr_bench_lane_probe_f64": (_i32, [_p, _i64, _i32, _i32, _i32, _pd]),
    "dhqr_debug_hold_cus": (_i32, [_p, _i32, _i32, _i32]),
    "dhqr_debug_smq_phases": (_i32, [_p, _pd]),
}

_lib = None


def lib() -> ctypes.CDLL:
    """Load libdhqr.so (built in-tree by build()).  Raises if it is missing: no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise ImportError(
                f"{SO_PATH} not found: build it with __graft_entry__.build() "
                "(hipcc --offload-arch=gfx950); this package has no CPU fallback")
        L = ctypes.CDLL(SO_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the library does not export it
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


_bench = None


def lib_bench() -> ctypes.CDLL:
    """libdhqr_bench.so: the product source + the micro-benchmarks / the MFMA layout probe (include/dhqr_bench.h).
    Its contexts are its own: pass handles created by ITS dhqr_create."""
    global _bench
    if _bench is None:
        if not os.path.exists(BENCH_SO_PATH):
            raise ImportError(f"{BENCH_SO_PATH} not found: build it with __graft_entry__.build()")
        L = ctypes.CDLL(BENCH_SO_PATH)
        for name, (res, args) in list(SIGNATURES.items()) + list(BENCH_SIGNATURES.items()):
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _bench = L
    return _bench


def check(rc: int) -> None:
    if rc != OK:
        raise DHQRError(rc, lib().dhqr_last_error().decode(errors="replace"))
