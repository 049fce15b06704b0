"""phase clock of k_small_qr_d (libdhqr_bench.so): where a column step of the small route goes, per wave.  usage: smq_phases.py"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

pkg = g.import_package()
BL, h = pkg.bench_context(0)
P = ctypes.c_void_p
names = ["all before tq1", "build (barrier form)", "barrier wait", "FLAGS: wait v_j", "FLAGS: build block"]
for m, n in ((110, 100), (220, 200)):
    A = pkg.rand_colmajor(m, n, 3, "cuda:0")
    import torch
    al = torch.zeros(n, dtype=torch.float64, device="cuda:0")
    out = (ctypes.c_double * 54)()
    pkg.bench_check(BL, BL.dhqr_debug_smq_phases(h, out))  # clear
    for _ in range(3):
        pkg.bench_check(BL, BL.dhqr_factor_f64(h, P(A.data_ptr()), m, n, m, P(al.data_ptr()), 0))
    pkg.bench_check(BL, BL.dhqr_debug_smq_phases(h, out))
    v = np.array(out[:]).reshape(9, 6)
    print(f"== {m} x {n}: cycles per step")
    for w in range(9):
        if v[w, 5] > 0:
            print(f"  wave {w}: " + "  ".join(f"{names[q]} {v[w, q] / v[w, 5]:7.0f}" for q in range(5)) + f"   total {v[w, :3].sum() / v[w, 5]:7.0f}")
