"""Randomised shapes through the multi-rank drivers with R rank threads on ONE GPU (column split dhqr_mg_* and row split
dhqr_mg_rs_*; ComplexF64 column split dhqr_mg_qr_c64 against the single-GPU factorisation, relative to kappa(A)):
||A - QR|| / ||A|| < 1e-12 for every case.  Development aid (tools/README.md)."""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

pkg = g.import_package()
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ncases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
for case in range(ncases):
    R = rng.choice([1, 2, 3, 4, 5, 8])
    n = rng.choice([rng.randrange(1, 300), rng.randrange(128, 1800), 128 * rng.randrange(1, 14), 256 * rng.randrange(1, 8) + rng.choice([0, 1, 127, 128, 129])])
    m = n + rng.choice([0, 1, 37, 128, 129, rng.randrange(0, 3000)])
    split = rng.choice(["cols", "rows", "zcols"])
    if split == "rows":
        m = max(m, n + 0)
    mg = pkg.MultiGpuQR(devices=[0] * R)
    try:
        if split == "zcols":  # ComplexF64 column split, host in / host out: residual of the factorisation formed on one GPU
            import numpy as np
            import torch
            if m > 1500:
                m = n + (m - n) % 700
            r0 = np.random.default_rng(case)
            A0 = np.asfortranarray((r0.random((m, n)) - 0.5) + 1j * (r0.random((m, n)) - 0.5))
            H, al = mg.qr_(A0.copy(order="F"))
            x0 = r0.random(n) + 1j * r0.random(n)
            x = np.asarray(mg.ldiv(H, al, A0 @ x0))
            H1 = pkg.qr_(torch.from_numpy(A0.T.copy()).cuda().t(), nb=64 if n >= 256 else 0)
            torch.cuda.synchronize()
            d = np.abs(H1.A.cpu().numpy() - H).max() / np.abs(H).max()  # against the single-GPU factorisation
            kap = np.linalg.cond(A0)
            r = max(d / max(kap, 1.0), np.abs(x - x0).max() / max(kap, 1.0) * 1e-3)
        elif split == "cols":
            mg.alloc(m, n); mg.fill(case); mg.factor(); r = mg.residual(case)
        else:
            mg.rs_alloc(m, n); mg.rs_fill(case); mg.rs_factor(); r = mg.rs_residual(case)
        ok = r < 1e-12
    except Exception as e:  # report and go on
        r, ok = repr(e)[:200], False
    finally:
        mg.close()
    bad += 0 if ok else 1
    print(("ok  " if ok else "BAD ") + f"{split} R={R} m={m} n={n} resid={r}", flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
