"""rocprofv3 (rocpd sqlite output) -> per-kernel summary CSV, the same columns as --stats.
usage: python tools/prof_summary.py <results.db> <out.csv> ["command line that was profiled"]"""
import csv
import sqlite3
import sys


def main(db, out, cmd=""):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute(
        "select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
        "from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    with open(out, "w", newline="") as f:
        if cmd:
            f.write(f"# rocprofv3 --kernel-trace --stats -- {cmd}\n")
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            w.writerow([r[0], r[1], int(r[2]), f"{r[3]:.1f}", f"{100.0 * r[2] / tot:.2f}", int(r[4]), int(r[5])])
    print("wrote", out, "kernels:", len(rows), "total ms:", tot / 1e6)


if __name__ == "__main__":
    main(*sys.argv[1:4])
