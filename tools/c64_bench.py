"""Timing of the ComplexF64 paths (argv: n [nb = 0 | 64]) on one GPU: n x n device-resident factorisation,
algorithmic HBM rate = 32 B per trailing element per reflector.  Prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import __graft_entry__ as g  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    nb = int(sys.argv[2]) if len(sys.argv) > 2 else 0   # 0 unblocked, 64 blocked (real-embedding MFMA update)
    pkg = g.import_package()
    out = {"n": n, "nb": nb}
    times = []
    for it in range(2):
        A = pkg.rand_colmajor_c(n, n, 0, "cuda:0")
        alpha = torch.zeros(n, dtype=torch.complex128, device="cuda:0")
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        pkg.householder_(A, alpha, nb=nb)
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1))
    ms = min(times)
    by = 32.0 * sum((n - j) * (n - j - 1) for j in range(n))
    fl = 16.0 * sum((n - j) * (n - j - 1) for j in range(n))
    v2 = (torch.tril(A).abs() ** 2).sum(dim=0)
    out.update(ms=ms, ms_all=times, algorithmic_GBps=by / ms / 1e6, real_GFLOPs=fl / ms / 1e6,
               max_abs_v2_minus_2=float((v2 - 2).abs().max().item()))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
