"""rocprofv3 (rocpd sqlite output) -> per-kernel summary CSV, the same columns as --stats.
usage: python tools/prof_summary.py <results.db> <out.csv> ["command line that was profiled"]
       python tools/prof_summary.py --by-stream <results.db> <out.csv> ["command line"]
--by-stream splits every kernel by HIP stream and by launch size (>= 400 workgroups = "wide"): the
blocked driver runs the wide trailing update on the caller's stream and the panel / narrow-update lane
on a second one, and bench.py's hipEvent groups time the caller's stream only."""
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


def by_stream(db, out, cmd=""):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute(
        "select name, stream, case when grid_x*grid_y*grid_z/(workgroup_x*workgroup_y*workgroup_z) >= 400 "
        "then 'wide' else 'small' end as cls, count(*), sum(duration), avg(duration), min(duration), max(duration) "
        "from kernels group by name, stream, cls order by 5 desc").fetchall()
    with open(out, "w", newline="") as f:
        if cmd:
            f.write(f"# rocprofv3 --kernel-trace -- {cmd}  (split by stream and launch size)\n")
        w = csv.writer(f)
        w.writerow(["Name", "Stream", "LaunchClass", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs"])
        for r in rows:
            w.writerow([r[0], r[1], r[2], r[3], int(r[4]), f"{r[5]:.1f}", int(r[6]), int(r[7])])
    print("wrote", out, "rows:", len(rows))


def per_launch(db, out, cmd=""):
    """every launch of the wide stream's kernels in time order: start, duration, workgroups, gap to the previous kernel
    of the same stream -- to see where a step's time goes launch by launch (tail effects, gaps between kernels)"""
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute(
        "select name, stream, start, end, grid_x*grid_y*grid_z/(workgroup_x*workgroup_y*workgroup_z), workgroup_x "
        "from kernels order by start").fetchall()
    last_end = {}
    with open(out, "w", newline="") as f:
        if cmd:
            f.write(f"# rocprofv3 --kernel-trace -- {cmd}  (every launch)\n")
        w = csv.writer(f)
        w.writerow(["Name", "Stream", "StartNs", "DurationNs", "Workgroups", "Threads", "GapToPrevOnStreamNs"])
        t0 = rows[0][2] if rows else 0
        for name, stream, st, en, wgs, thr in rows:
            gap = st - last_end[stream] if stream in last_end else 0
            last_end[stream] = en
            w.writerow([name.split("(")[0][:60], stream, st - t0, en - st, wgs, thr, gap])
    print("wrote", out, "launches:", len(rows))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--per-launch":
        per_launch(*sys.argv[2:5])
    elif len(sys.argv) > 1 and sys.argv[1] == "--by-stream":
        by_stream(*sys.argv[2:5])
    else:
        main(*sys.argv[1:4])
