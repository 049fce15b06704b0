"""Randomised MID-SIZE shapes through the single-GPU blocked driver (edge tiles beside the direct-load interior tiles, partial
K-tiles of the direct-load k_gemm_tn2, odd m -> scalar paths, quad steps forced on): ||A - QR|| / ||A|| < 1e-12 for every case.
  python tools/gpu_fuzz_blocked.py [seed] [cases]"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("DHQR_QUAD_MIN_COLS", "1024")  # quad steps on these sizes too
import __graft_entry__ as g  # noqa: E402

pkg = g.import_package()
import torch  # noqa: E402

rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ncases = int(sys.argv[2]) if len(sys.argv) > 2 else 12
bad = 0
for case in range(ncases):
    n = rng.choice([rng.randrange(1500, 7000), 128 * rng.randrange(12, 56), 256 * rng.randrange(6, 24) + rng.choice([1, 2, 127, 129])])
    m = n + rng.choice([0, 1, 2, 130, rng.randrange(0, 2500)])
    A = pkg.rand_colmajor(m, n, case + 1, "cuda:0")
    H = pkg.qr_(A, nb=128)
    torch.cuda.synchronize()
    r = pkg.residual(H, pkg.rand_colmajor(m, n, case + 1, "cuda:0"))
    ok = r < 1e-12
    bad += not ok
    print(("ok " if ok else "BAD"), f"m={m} n={n} resid={r:.3e}", flush=True)
    del A, H
print("failures:", bad)
sys.exit(1 if bad else 0)
