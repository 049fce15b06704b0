"""per-launch durations of the unblocked path's k_rankk_fused launches (rocprofv3 rocpd db), in launch order, against
the algorithmic bytes of each launch: where a factorisation loses time (wide early launches: bulk bandwidth; late ones:
the lead workgroup's chain).  usage: python tools/rankk_profile.py <results.db> <n> <K> [nfact]"""
import sqlite3
import sys

db, n, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
nfact = int(sys.argv[4]) if len(sys.argv) > 4 else 1
cur = sqlite3.connect(db).cursor()
rows = cur.execute("select start, end - start from kernels where name like '%k_rankk_%' order by start").fetchall()
per = len(rows) // nfact
rows = rows[-per:]  # last factorisation
print(f"# {len(rows)} k_rankk_fused launches of the last factorisation, n = {n}, K = {K}")
print("# launch  jlo  us  ideal_us_at_5TB/s  GB/s")
tot = tot_ideal = 0.0
bins = {}
for i, (_, d) in enumerate(rows):
    jlo = (i - 1) * K if i > 0 else 0
    by = 16.0 * (n - jlo) * max(0, n - jlo - K) if i > 0 else 16.0 * n * K
    us = d / 1e3
    tot += us
    tot_ideal += by / 5e12 * 1e6
    b = bins.setdefault(jlo // 1024, [0.0, 0.0, 0])
    b[0] += us
    b[1] += by
    b[2] += 1
    if i % max(1, per // 24) == 0:
        print(f"{i:6d} {jlo:6d} {us:9.1f} {by / 5e12 * 1e6:9.1f} {by / d:9.0f}")
print("# by 1024 columns: jlo range, launches, ms, GB/s, avg us per launch")
for k in sorted(bins):
    us, by, cnt = bins[k]
    print(f"{k * 1024:6d}+ {cnt:6d} {us / 1e3:9.2f} {by / us / 1e3:9.0f} {us / cnt:9.1f}")
print(f"# total {tot / 1e3:.1f} ms, at 5 TB/s {tot_ideal / 1e3:.1f} ms")
