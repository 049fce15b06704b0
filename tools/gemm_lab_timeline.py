"""Timeline of tools/gemm_lab <variant>_tl: per workgroup [hw_id, xcc_id, t_start, t_prologue_done, t_after_ktile[0..NKT), t_stores_issued,
t_stores_drained] (s_memtime of wave 0).  Prints phase statistics and, per CU, how much of the time 0 / 1 / 2 workgroups were
inside their K loops.  usage: gemm_lab_timeline.py <file>"""
import sys
import numpy as np

raw = np.fromfile(sys.argv[1], dtype=np.uint64)
grid, stride, nkt, _ = (int(x) for x in raw[:4])
d = raw[4:].reshape(grid, stride).astype(np.int64)
d = d[d[:, 2] != 0]
hw, xcc = d[:, 0], d[:, 1] & 0xF
cu = (hw >> 8) & 0xF
sh = (hw >> 12) & 0x1
se = (hw >> 13) & 0x7
simd = (hw >> 4) & 0x3
cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
t0, tp = d[:, 2], d[:, 3]
tk = d[:, 4:4 + nkt]
ts, te = d[:, 4 + nkt], d[:, 5 + nkt]
print("workgroups", len(d), "CUs seen", len(np.unique(cuid)), "K-tiles", nkt)
pro = tp - t0
kloop = tk[:, -1] - tp
st = ts - tk[:, -1]
dr = te - ts
tot = te - t0
for name, x in (("prologue", pro), ("K loop", kloop), ("store issue", st), ("store drain", dr), ("total", tot)):
    print(f"{name:12s} mean {x.mean():9.0f}  p10 {np.percentile(x,10):9.0f}  median {np.median(x):9.0f}  p90 {np.percentile(x,90):9.0f}  max {x.max():9.0f}")
kt = np.diff(np.concatenate([tp[:, None], tk], axis=1), axis=1)
print("K-tile duration (cycles): mean %.0f median %.0f p10 %.0f p90 %.0f p99 %.0f" % (kt.mean(), np.median(kt), np.percentile(kt, 10), np.percentile(kt, 90), np.percentile(kt, 99)))
print("per K-tile index mean:", " ".join(f"{x:.0f}" for x in kt.mean(axis=0)))
# per CU occupancy
acc = np.zeros(4)
span = 0
gaps = []
for c in np.unique(cuid):
    m = cuid == c
    ev = []
    for a, b in zip(tp[m], tk[m, -1]):
        ev.append((a, 1))
        ev.append((b, -1))
    ev.sort()
    lo, hi = t0[m].min(), te[m].max()
    span += hi - lo
    cur, last = 0, lo
    for tt, dv in ev:
        acc[min(cur, 3)] += tt - last
        last = tt
        cur += dv
    acc[0] += hi - last
    # gap between a workgroup's end and the start of the next one on the same CU slot (approx.: sorted starts vs ends)
    s_ = np.sort(t0[m]); e_ = np.sort(te[m])
    if len(s_) > 2:
        gaps.append(np.median(s_[2:] - e_[:-2]))
print("CU time with k workgroups inside the K loop: 0: %.3f  1: %.3f  2: %.3f  (3+: %.3f)" % tuple(acc / span))
print("median refill gap (end of a workgroup -> start of its successor on the CU): %.0f cycles" % np.median(gaps))
print("tiles per CU: mean %.1f min %d max %d" % (np.mean([np.sum(cuid == c) for c in np.unique(cuid)]), min(np.sum(cuid == c) for c in np.unique(cuid)), max(np.sum(cuid == c) for c in np.unique(cuid))))
mf = nkt * 64 * 64
print("MFMA pipe cycles per tile per SIMD: %d; CU-level: sum(total)/span/2 = %.3f; pipe busy lower bound = tiles*%d/span = %.3f" % (mf, tot.sum() / span / 2, mf, len(d) * mf / span))
if stride >= 40 + 6 * nkt and d[:, 40:].any():
    x = d[:, 40:40 + 6 * nkt].reshape(len(d), nkt, 6)
    kstart = np.concatenate([tp[:, None], tk[:, :-1]], axis=1)  # wave 0: K-tile start (after the previous barrier)
    mf = x[:, :, 0] - kstart            # wave 0: top of K-tile -> last MFMA issued
    sk = x[:, :, :4].max(axis=2) - x[:, :, :4].min(axis=2)   # skew of the four waves at "last MFMA issued"
    vm = x[:, :-1, 4] - x[:, :-1, 0]    # wave 0: last MFMA issued -> operands landed (C add + vmcnt wait)
    br = tk[:, :-1] - x[:, :-1, 4]      # wave 0: barrier wait (+ stamp)
    for name, a in (("issue+MFMA block (wave 0)", mf), ("wave skew at block end", sk), ("C add + vmcnt wait (wave 0)", vm), ("barrier wait (wave 0)", br)):
        print(f"{name:30s} mean {a.mean():8.0f} median {np.median(a):8.0f} p10 {np.percentile(a,10):8.0f} p90 {np.percentile(a,90):8.0f} p99 {np.percentile(a,99):8.0f}")
