import ctypes, os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from dist_helpers import load_emulated_library
from oracle import dhqr_oracle as orc
L = load_emulated_library("/tmp/asan/libdhqr_emulated.so")
P = ctypes.c_void_p
def ptr(a): return a.ctypes.data_as(P)
for (m, n, vec) in [(300, 200, None), (777, 130, None), (260, 257, 2), (1100, 1000, 2), (70, 50, None), (129, 129, None), (550, 500, None), (600, 160, None), (600, 160, 2), (500, 289, 2)]:
    if vec: os.environ["DHQR_TUNE"] = f"qtb_vec={vec}"
    else: os.environ.pop("DHQR_TUNE", None)
    h = P(); assert L.dhqr_create(ctypes.byref(h), 0) == 0
    A0 = orc.rand_matrix(m, n, 31); Ho, ao = orc.householder(A0); b = orc.rand_vector(m, 32)
    H = np.asfortranarray(Ho).copy(order="F"); bb = b.copy(); al = ao.copy()
    assert L.dhqr_solve_f64(h, ptr(H), m, n, m, ptr(al), ptr(bb)) == 0
    xo = orc.solve(Ho, ao, b)
    print(m, n, vec, np.abs(bb[:n] - xo).max() / np.abs(xo).max(), flush=True)
    L.dhqr_destroy(h)
