"""TEST / BASELINE INFRASTRUCTURE (bench.py's cpu_baseline leg only): LAPACK dgeqrf from SciPy's OpenBLAS on the host
cores, n x n of the oracle's generator -- the comparator of the reference's own tests (test/runtests.jl:49-53).  The
pool's hosts run this OpenBLAS FASTER with few threads (4096^2: 16 s with its default 64 threads, 2.1 s with 16), so the
best of two small thread counts is reported together with that count.  Prints one JSON line."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    from scipy.linalg import lapack
    from threadpoolctl import threadpool_info, threadpool_limits
    from oracle import dhqr_oracle as orc
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    # never RAISE the thread count of a loaded OpenBLAS (observed to crash it); only lower it
    have = min([d["num_threads"] for d in threadpool_info() if d.get("user_api") == "blas"] or [1])
    best = None
    for threads in sorted({min(16, have), min(8, have)}, reverse=True):
        A = np.empty((n, n), order="F")
        orc.lib().dhqr_oracle_fill(orc._ptr(A), n, n, n, 0)
        with threadpool_limits(limits=threads):
            t0 = time.perf_counter()
            _, _, _, info = lapack.dgeqrf(A, overwrite_a=True)
            dt = time.perf_counter() - t0
        cur = {"n": n, "threads": threads, "seconds": dt, "gflops": (4.0 / 3.0) * n ** 3 / dt / 1e9, "info": int(info)}
        if best is None or cur["gflops"] > best["gflops"]:
            best = cur
    print(json.dumps(best), flush=True)


if __name__ == "__main__":
    main()
