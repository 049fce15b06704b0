"""Where the wide stream waits for the look-ahead lane: reads the per-launch CSV of tools/prof_summary.py --per-launch
(Name,Stream,StartNs,DurationNs,Workgroups,Threads,GapToPrevOnStreamNs), takes the LAST factorisation on the stream that
carries the wide k_gemm_nn_quad launches, prints its kernel / gap totals, every gap above 30 us, and (--around K) both
streams interleaved around the K-th wide subtraction launch.  usage: python tools/lane_gaps.py per_launch.csv[.gz] [--around K]"""
import collections
import csv
import gzip
import sys

import numpy as np

path = sys.argv[1]
around = int(sys.argv[sys.argv.index("--around") + 1]) if "--around" in sys.argv else None
op = gzip.open if path.endswith(".gz") else open
rows = [r for r in csv.reader(l for l in op(path, "rt") if not l.startswith("#"))][1:]
allk = [(r[1], r[0].replace("void ", "").split("(")[0][:46], int(r[2]), int(r[3]), int(r[4])) for r in rows
        if "k_" in r[0] and "at::" not in r[0]]
allk.sort(key=lambda x: x[2])
cnt = collections.Counter(x[0] for x in allk if x[1].startswith("k_gemm_nn_quad<2, 128>"))
wide = cnt.most_common(1)[0][0]
w = [x for x in allk if x[0] == wide]
st = np.array([x[2] for x in w])
en = st + np.array([x[3] for x in w])
gap = st[1:] - en[:-1]
i0 = int(np.argmax(gap)) + 1  # the pause between the factorisations
w, st, en = w[i0:], st[i0:], en[i0:]
gap = st[1:] - en[:-1]
print(f"# wide stream = {wide}: {len(w)} launches, span {(en[-1] - st[0]) / 1e6:.2f} ms, kernels {sum(x[3] for x in w) / 1e6:.2f} ms, gaps {gap.sum() / 1e6:.2f} ms")
tot = collections.defaultdict(lambda: [0, 0, 0])
for k, x in enumerate(w):
    t = tot[x[1]]
    t[0] += 1
    t[1] += x[3]
    t[2] += gap[k - 1] if k else 0
for n, t in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"{n:48s} n={t[0]:5d} dur={t[1] / 1e6:8.2f} ms  gap before={t[2] / 1e6:7.2f} ms")
big = [(k, gap[k - 1]) for k in range(1, len(w)) if gap[k - 1] > 30e3]
print(f"# gaps > 30 us: {len(big)}, {sum(g for _, g in big) / 1e6:.2f} ms")
for k, g in big:
    print(f"t={(st[k] - st[0]) / 1e6:7.1f} ms gap {g / 1e3:7.1f} us before {w[k][1][:28]} after {w[k - 1][1][:30]}")
if around is not None:
    q = [x for x in w if x[1].startswith("k_gemm_nn_quad<2, 128>")]
    x = q[around]
    t_end = x[2] + x[3]
    print(f"# around wide subtraction launch {around}: duration {x[3] / 1e3:.1f} us, times relative to its end")
    for y in allk:
        if y[2] + y[3] > x[2] - 300e3 and y[2] < t_end + 1.5e6:
            print(f"{y[0][:9]:9s} {y[1]:46s} start {(y[2] - t_end) / 1e3:9.1f} dur {y[3] / 1e3:8.1f} wgs {y[4]}")
