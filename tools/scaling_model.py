"""Analytic schedule model of the block-cyclic column-split factorisation (csrc/dhqr_dist.h) at P = 1, 2, 4, 8
GPUs -- a MODEL, not a measurement (no multi-GPU node is reachable from the build environment).

Mirrors the driver's structure: panels in pairs; per rank a LANE timeline (narrow updates of its own next panels,
their factorisation, the pair's cross term) and a WIDE timeline (apply pair g to the local blocks beyond pair g+1,
the blocks of pair g+2 first); a panel is available to the other ranks one broadcast after it is factored.  Cyclic block
= a pair of panels (--own 2, the shipped layout: the second panel of a pair never waits for a broadcast) or one panel
(--own 1, the layout until mid round 2).

Inputs measured on one MI355X (profiles/r02_*; round 4 re-checked the chain against the per-launch trace
profiles/r04_blocked32768_per_launch.csv.gz: 290 us per 32768-row panel incl. ~40 us of event bubbles, unchanged kernels):
  trailing GEMMs in situ                          61 TFLOP/s per GPU (r5, direct-load kernels: K = 256 pair update, both passes; 64 with quad steps at one GPU; r4: 54 / 60)
  panel chain, uncontended                        0.215 ms single-workgroup kernels + small GEMMs + launch gaps (fixed)
                                                + 0.035 ms * rows/32768 (Gram / product GEMMs, commit)
                                                  [fit to the per-panel time of the no-look-ahead driver at 8192^2 /
                                                  16384^2 / 24576^2 minus its host synchronisation, profiles/r02_panel_chain.txt;
                                                  before this round's panel-kernel work: 0.32 + 0.05]
  narrow update of one 128-column block           0.03 ms + 0.10 ms * rows/32768   (pair: x 1.6)
  pair cross term V_b' V_a                        0.02 ms + 0.04 ms * rows/32768
Assumed (NOT measured): broadcast of one panel = latency + bytes / bandwidth.

  python tools/scaling_model.py [--small-ms 0.215] [--var-ms 0.035] [--bw-gbps 100] [--lat-us 40] [--own 2]
"""
import argparse


def simulate(P, n=32768, nb=128, gemm_tflops=61.0, small_ms=0.215, var_ms=0.035, bw_gbps=100.0, lat_us=40.0, own=2, side=False):
    # own = panels per cyclic block: 2 (DHQR_CS_BLOCK = 256: the shipped layout) or 1 (the earlier 128-column blocks)
    K = n // nb
    G = K // 2
    rows = lambda k: n - k * nb
    t_panel = lambda k: (small_ms + var_ms * rows(k) / n) * 1e-3
    t_narrow = lambda k, pair: (0.03 + 0.10 * rows(k) / n) * 1e-3 * (1.6 if pair else 1.0)
    t_cross = lambda k: (0.02 + 0.04 * rows(k) / n) * 1e-3
    # side stream (round 4, DHQR_LANE_SIDE): Y = V_a' C_b and the pair's cross term run beside panel a's / b's verification
    # and commit -- the split-K product + reduction of the narrow update (45 of 87 us at 32768 rows) and the cross term leave
    # the chain (per-launch trace profiles/r04_blocked32768_per_launch.csv.gz); the events cost ~10 us
    if side:
        t_cross = lambda k: 0.010e-3
    t_bcast = lambda k: 0.0 if P == 1 else lat_us * 1e-6 + rows(k) * nb * 8 / (bw_gbps * 1e9)
    t_wide = lambda g, ncols: 2 * 4.0 * nb * (rows(2 * g) - nb / 2) * ncols / (gemm_tflops * 1e12)
    lane = [0.0] * P
    wide = [0.0] * P
    avail = [0.0] * K                   # time panel k has arrived everywhere
    group_ready = [0.0] * G             # per group: max over ranks of "pair assembled" (same everywhere up to the cross term)
    head_done = [[0.0] * P for _ in range(G)]
    wide_done = [[0.0] * P for _ in range(G)]

    def local_cols(r, lo_blk, hi_blk):
        return sum(nb for j in range(lo_blk, min(hi_blk, K)) if (j // own) % P == r)

    def produce(h):
        a, b = 2 * h, 2 * h + 1
        for idx, x in enumerate((a, b)):
            o = (x // own) % P
            t = lane[o]
            if h >= 1:
                t = max(t, group_ready[h - 1])                                   # pair h-1 assembled
                if h >= 2:
                    t = max(t, (head_done if P > 1 else wide_done)[h - 2][o])   # block x carries pair h-2
                # own = 2: both panels are local and adjacent: ONE merged 256-column update (charged to the first)
                t += t_narrow(2 * (h - 1), True) if (own == 1 or idx == 0) else 0.0
            if idx == 1:
                # panel a -> block b: local when the pair lives on one rank, else after a's broadcast has arrived
                t = (t if own == 2 else max(t, avail[a])) + t_narrow(a, False) * (0.5 if (side and own == 2) else 1.0)
            t += t_panel(x)
            lane[o] = t
            avail[x] = t + t_bcast(x)
        ready = max(avail[a], avail[b]) + t_cross(a)
        for r in range(P):
            lane[r] = max(lane[r], ready)
        group_ready[h] = ready

    produce(0)
    for g in range(G):
        if 2 * g + 2 >= K:
            break
        for r in range(P):
            t = max(wide[r], group_ready[g])
            t += t_wide(g, local_cols(r, 2 * g + 4, 2 * g + 6))                  # head: blocks of pair g+2
            head_done[g][r] = t
            t += t_wide(g, local_cols(r, 2 * g + 6, K))
            wide[r] = wide_done[g][r] = t
        produce(g + 1)
    return max(max(wide), max(lane))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--small-ms", type=float, default=0.215)
    ap.add_argument("--var-ms", type=float, default=0.035)
    ap.add_argument("--bw-gbps", type=float, default=100.0)
    ap.add_argument("--lat-us", type=float, default=40.0)
    ap.add_argument("--own", type=int, default=2, help="panels per cyclic block: 2 (shipped) or 1 (128-column blocks)")
    ap.add_argument("--t1", type=float, default=0.774, help="measured 1-GPU time in seconds (round 4: 0.774)")
    a = ap.parse_args()
    t1 = simulate(1, small_ms=a.small_ms, var_ms=a.var_ms)
    print(f"fixed part of the panel chain {a.small_ms:.2f} ms, broadcast {a.bw_gbps:.0f} GB/s + {a.lat_us:.0f} us (assumed)")
    print(f"  P   model time [ms]   vs model P=1   (measured 1 GPU: {a.t1:.3f} s; the model's P=1 has no contention between lane and wide)")
    for P in (1, 2, 4, 8):
        t = simulate(P, small_ms=a.small_ms, var_ms=a.var_ms, bw_gbps=a.bw_gbps, lat_us=a.lat_us, own=a.own)
        ts = simulate(P, small_ms=a.small_ms, var_ms=a.var_ms, bw_gbps=a.bw_gbps, lat_us=a.lat_us, own=a.own, side=True)
        print(f"  {P}   {t * 1e3:8.1f}        {t1 / t:5.2f}x        vs measured {a.t1:.3f} s: {a.t1 / t:5.2f}x"
              f"     with a lane side stream at P > 1 (not built: a fifth busy stream shares a hardware queue): {ts * 1e3:7.1f} ms = {a.t1 / ts:5.2f}x")


if __name__ == "__main__":
    main()
