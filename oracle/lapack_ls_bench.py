"""TEST / BASELINE INFRASTRUCTURE (bench.py's `also` solve entry only): what the reference's test file times beside its own
`qr!(A) \\ b` -- `LinearAlgebra.qr!(A) \\ b` = LAPACK geqrf + ormqr + trtrs (test/runtests.jl:55-56) -- from SciPy's OpenBLAS
on the host cores, at the reference's seven shapes (test/runtests.jl:42).  Few BLAS threads (this pool's hosts run OpenBLAS
faster with 8-16 threads than with their default); the best of two repetitions per shape (one above 1 s).  Prints a cumulative JSON line per shape."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SHAPES = ((110, 100), (220, 200), (440, 400), (880, 800), (1100, 1000), (2200, 2000), (4400, 4000))


def main():
    import numpy as np
    from scipy.linalg import lapack
    from threadpoolctl import threadpool_info, threadpool_limits
    from oracle import dhqr_oracle as orc
    have = min([d["num_threads"] for d in threadpool_info() if d.get("user_api") == "blas"] or [1])
    threads = min(16, have)  # never RAISE the thread count of a loaded OpenBLAS (observed to crash it); only lower it
    out = []
    with threadpool_limits(limits=threads):
        for m, n in SHAPES:
            A0 = orc.rand_matrix(m, n, 0)
            b0 = orc.rand_vector(m, 1)
            best = None
            for rep in range(2):
                if rep and best > 1.0:
                    break  # one repetition of the slow shapes: the whole script stays inside bench.py's budget
                A = A0.copy(order="F")
                b = b0.copy()
                t0 = time.perf_counter()
                qr, tau, _, info = lapack.dgeqrf(A, overwrite_a=True)
                c, _, info2 = lapack.dormqr("L", "T", qr, tau, b.reshape(m, 1), lwork=64 * n)
                x, info3 = lapack.dtrtrs(qr[:n, :n], c[:n], lower=0)
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            r = A0 @ x[:, 0] - b0
            out.append({"m": m, "n": n, "seconds": best, "normal_eq": float(np.linalg.norm(A0.T @ r)),
                        "info": [int(info), int(info2), int(info3)]})
            print(json.dumps({"threads": threads, "shapes": out}), flush=True)  # cumulative: the last line is the result


if __name__ == "__main__":
    main()
