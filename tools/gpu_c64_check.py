"""Torch-free first-light check of the ComplexF64 entry points (host in / host out) against the
oracle: loads libdhqr.so with ctypes only, so it starts in seconds on a fresh GPU box.
Usage (GPU box):  python tools/gpu_c64_check.py  -> prints one JSON line, exit code 0 on success."""
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import dhqr_oracle as orc  # noqa: E402  (this script is a checker, not product code)

L = ctypes.CDLL(os.path.join(ROOT, "distributedhouseholderqr.jl_amd", "libdhqr.so"))
L.dhqr_last_error.restype = ctypes.c_char_p
i64, p = ctypes.c_int64, ctypes.c_void_p
L.dhqr_create.argtypes = [ctypes.POINTER(p), ctypes.c_int32]
L.dhqr_qr_c64.argtypes = [p, p, i64, i64, i64, p]
L.dhqr_ldiv_c64.argtypes = [p, p, i64, i64, i64, p, p, p]
L.dhqr_partialdot_host_c64.argtypes = [p, p, p, i64, i64, p]


def ck(rc):
    if rc != 0:
        raise RuntimeError(f"rc={rc}: {L.dhqr_last_error().decode()}")


def P(a):
    return a.ctypes.data_as(p)


def main():
    h = p()
    ck(L.dhqr_create(ctypes.byref(h), 0))
    out = {"cases": []}
    ok = True
    rng = np.random.default_rng(0)
    worst = 0.0
    for N in range(1, 21):
        a = rng.random(N) + 1j * rng.random(N)
        b = rng.random(N) + 1j * rng.random(N)
        for i in range(N):
            r = np.zeros(2)
            ck(L.dhqr_partialdot_host_c64(h, P(a), P(b), i, N, P(r)))
            worst = max(worst, abs(complex(r[0], r[1]) - np.vdot(a[i:], b[i:])))
    out["partialdot_kat_max_abs_err"] = worst
    ok &= worst < 1e-13
    for (m, n) in [(5, 3), (64, 64), (110, 100), (1500, 40), (5000, 24), (440, 400), (2200, 2000)]:
        A0 = orc.rand_matrix_c(m, n, 3)
        A = A0.copy(order="F")
        al = np.zeros(n, dtype=complex)
        t0 = time.time()
        ck(L.dhqr_qr_c64(h, P(A), m, n, m, P(al)))
        dt = time.time() - t0
        Ho, ao = orc.householder_c(A0)
        scale = np.abs(Ho).max()
        dH, da = np.abs(A - Ho).max() / scale, np.abs(al - ao).max() / scale
        b = orc.rand_vector_c(m, 4)
        x = np.zeros(n, dtype=complex)
        ck(L.dhqr_ldiv_c64(h, P(A), m, n, m, P(al), P(b), P(x)))
        xo = orc.solve_c(Ho, ao, b)
        dx = np.abs(x - xo).max() / np.abs(xo).max()
        res = np.linalg.norm(A0 - orc.form_qr_c(A, al)) / np.linalg.norm(A0) if n <= 400 else None
        out["cases"].append({"m": m, "n": n, "dH": dH, "dalpha": da, "dx": dx, "residual": res, "host_s": dt})
        ok &= dH < 1e-11 and da < 1e-11 and (res is None or res < 1e-12)
    out["ok"] = bool(ok)
    print(json.dumps(out))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
