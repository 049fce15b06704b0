"""Analytic schedule model of the 1-D block-cyclic column-split factorisation (distributed.py) at
P = 1, 2, 4, 8 GPUs -- a MODEL, not a measurement (multi-GPU runs are the driver's).

Inputs are the single-GPU rates measured in round 1 (profiles/r01c_*):
  trailing GEMMs in situ                         54 TFLOP/s per GPU
  panel chain at 32768 rows, uncontended         0.30 ms single-workgroup kernels (fixed)
                                               + 0.25 ms Gram/product GEMMs, pack, unpack (scale with rows)
  narrow update of the next panel block          0.03 ms + 0.10 ms * rows/32768
  broadcast of the packed (V,T,alpha) buffer     latency + bytes / bandwidth  (both assumed: xGMI ring)
Per rank the model keeps one compute timeline; the owner of block k+1 runs narrow update -> panel ->
broadcast before its own wide update (the "owner first" order of ColumnCyclicQR.factor at P > 1), every
other rank applies panel k to its trailing columns as soon as it has arrived.

  python tools/scaling_model.py [--small-ms 0.30] [--bw-gbps 100] [--lat-us 40]
"""
import argparse


def simulate(P, n=32768, nb=128, gemm_tflops=54.0, small_ms=0.30, var_ms=0.25, bw_gbps=100.0, lat_us=40.0):
    K = n // nb
    free = [0.0] * P          # time each rank's compute stream becomes free
    arrive = [0.0] * K        # time panel k is available on every rank

    def rows(k):
        return n - k * nb

    def t_panel(k):
        return (small_ms + var_ms * rows(k) / n) * 1e-3

    def t_narrow(k):
        return (0.03 + 0.10 * rows(k) / n) * 1e-3

    def t_bcast(k):
        if P == 1:
            return 0.0
        return lat_us * 1e-6 + rows(k) * nb * 8 / (bw_gbps * 1e9)

    def t_wide(k, ncols):
        return 4.0 * nb * rows(k) * ncols / (gemm_tflops * 1e12)

    def local_trailing_cols(r, k, exclude_next):
        # columns of blocks j > k owned by r (block-cyclic), optionally without block k+1
        cnt = 0
        for j in range(k + 1, K):
            if j % P == r and not (exclude_next and j == k + 1):
                cnt += nb
        return cnt

    # panel 0
    free[0] = t_panel(0)
    arrive[0] = free[0] + t_bcast(0)
    for k in range(K - 1):
        o = (k + 1) % P
        # owner of k+1: narrow update with panel k, factor, broadcast, then the rest of its update
        start = max(free[o], arrive[k])
        done_panel = start + t_narrow(k) + t_panel(k + 1)
        arrive[k + 1] = done_panel + t_bcast(k + 1)
        free[o] = done_panel + t_wide(k, local_trailing_cols(o, k, True))
        for r in range(P):
            if r != o:
                free[r] = max(free[r], arrive[k]) + t_wide(k, local_trailing_cols(r, k, False))
    return max(max(free), arrive[K - 1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--small-ms", type=float, default=0.30)
    ap.add_argument("--bw-gbps", type=float, default=100.0)
    ap.add_argument("--lat-us", type=float, default=40.0)
    a = ap.parse_args()
    t1 = simulate(1, small_ms=a.small_ms)
    print(f"single-workgroup part of the panel chain {a.small_ms:.2f} ms, broadcast {a.bw_gbps:.0f} GB/s + {a.lat_us:.0f} us")
    print("  P   time [ms]   speed-up vs model P=1   (model P=1 has no look-ahead overlap: measured 0.92 s)")
    for P in (1, 2, 4, 8):
        t = simulate(P, small_ms=a.small_ms, bw_gbps=a.bw_gbps, lat_us=a.lat_us)
        print(f"  {P}   {t * 1e3:8.1f}   {t1 / t:5.2f}x    vs measured 0.92 s: {0.92 / t:5.2f}x")


if __name__ == "__main__":
    main()
