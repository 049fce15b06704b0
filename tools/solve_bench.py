"""Timing of the device-resident solve (dhqr_solve_f64: b <- Q'b + back substitution) on factored matrices
that are already in HBM, with torch events on the stream the library runs on.  Algorithmic bytes: one pass over V
(8 (mn - n^2/2) bytes) + one over R (4 n^2 bytes); printed beside the achieved GB/s and the fraction of the 8 TB/s roof.
Usage: python tools/solve_bench.py [m,n ...]   (DHQR_SOLVE_PIPE=0 selects the round-1 path for the A/B)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import __graft_entry__ as g  # noqa: E402

pkg = g.import_package()


def run(m, n, reps=5):
    A = pkg.rand_colmajor(m, n, 0, "cuda:0")
    A0 = A.clone()
    H = pkg.qr_(A, nb=128)
    b = pkg.rand_vector_device(m, 1, "cuda:0")
    x = pkg.ldiv(H, b)  # warm-up (workspaces)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        bb = b.clone()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        pkg.solve_householder_(bb, H.A, H.α)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    # quality: normal-equation residual ||A'(Ax - b)|| relative to ||A'b||
    r = A0 @ x - b
    ne = (A0.T @ r).norm().item() / (A0.T @ b).norm().item()
    ms = min(ts)
    byts = 8.0 * (m * n - n * n / 2.0) + 4.0 * n * n
    out = {"m": m, "n": n, "solve_ms_min": round(ms, 4), "solve_ms_all": [round(t, 4) for t in ts],
           "algorithmic_GB": round(byts / 1e9, 4), "GBps": round(byts / ms / 1e6, 1), "frac_of_8TBps": round(byts / ms / 1e6 / 8000.0, 4),
           "normal_eq_rel": ne, "pipe": os.environ.get("DHQR_SOLVE_PIPE", "1")}
    print(json.dumps(out), flush=True)
    del A, A0, H
    torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    shapes = [(4400, 4000), (8192, 8192), (16384, 16384), (32768, 32768)]
    if len(sys.argv) > 1:
        shapes = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]]
    for m, n in shapes:
        run(m, n)
