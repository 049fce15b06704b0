"""Randomised shapes through the multi-rank drivers with R rank threads on ONE GPU (column split dhqr_mg_* and row split
dhqr_mg_rs_*): ||A - QR|| / ||A|| < 1e-12 for every case.  Development aid (tools/README.md)."""
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
    split = rng.choice(["cols", "rows"])
    if split == "rows":
        m = max(m, n + 0)
    mg = pkg.MultiGpuQR(devices=[0] * R)
    try:
        if split == "cols":
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
