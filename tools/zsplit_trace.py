"""one qr!(A; ndev = 2 logical ranks) of a 4096^2 ComplexF64 matrix for rocprofv3 --kernel-trace --stats (after one warm-up call)"""
import sys
import numpy as np
sys.path.insert(0, ".")
import __graft_entry__
pkg = __graft_entry__.import_package()
ranks = int(sys.argv[1]) if len(sys.argv) > 1 else 2
mg = pkg.MultiGpuQR(devices=[0] * ranks)
n = 4096
rng = np.random.default_rng(1)
A0 = np.asfortranarray((rng.random((n, n)) - 0.5) + 1j * (rng.random((n, n)) - 0.5))
for _ in range(2):
    A = A0.copy(order="F")
    H, al = mg.qr_(A)
print("bcast", mg.comm_counters(0))
mg.close()
