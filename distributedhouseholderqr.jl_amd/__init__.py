"""distributedhouseholderqr.jl_amd -- MI355X-native hot path of jwscook/DistributedHouseholderQR.jl.

The directory name contains a dot, so it is loaded by path: `__graft_entry__.import_package()`
registers it in sys.modules as `dhqr_amd`.  Contents: csrc/ (HIP kernels + the C ABI of
include/dhqr.h), _lib.py (ctypes binding), api.py (host mirror of the reference's Julia API),
partition.py (index maps), distributed.py (front-ends of the multi-GPU C drivers), rowsplit.py, julia/ (ccall wrapper).
"""
import os as _os

# Kernel arguments written straight into device memory (a few microseconds less per launch; the blocked driver issues
# ~40 launches per panel: profiles/r05_ab_dev_kernarg.txt).  Only effective when the HIP runtime has not initialised yet
# (importing torch does not initialise it); a caller's own setting wins.  C / Julia callers: INTEGRATION.md section 5.
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

from . import _lib
from ._lib import NB, DHQRError, build
from .api import (Context, DistributedHouseholderQRStruct, apply_q_, bench_check, bench_context, bench_mfma_tflops,
                  bench_stream_gbps, empty_colmajor, get_context, get_q, get_r, householder_, ldiv, partialdot,
                  qr_, rand_colmajor, rand_colmajor_c, rand_vector_device, residual, solve_householder_)
from .distributed import ColumnCyclicQR, Communicator, MultiGpuQR, ldiv_darray_, qr_darray_, qr_darray_c64_, qr_multi_
from .rowsplit import RowSplitQR
from .partition import BlockCyclicColumns, LocalColumnBlock, contiguous_column_blocks

__all__ = [
    "NB", "DHQRError", "build", "Context", "DistributedHouseholderQRStruct", "apply_q_",
    "bench_check", "bench_context", "bench_mfma_tflops", "bench_stream_gbps", "empty_colmajor", "get_context", "get_q", "get_r", "householder_",
    "ldiv", "partialdot", "qr_", "rand_colmajor", "rand_colmajor_c", "rand_vector_device", "residual",
    "solve_householder_", "ColumnCyclicQR", "Communicator", "MultiGpuQR", "ldiv_darray_", "qr_darray_", "qr_darray_c64_", "qr_multi_", "RowSplitQR", "BlockCyclicColumns", "LocalColumnBlock", "contiguous_column_blocks",
]
