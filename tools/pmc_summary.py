"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, csv) -> per-kernel HBM bytes.
gfx950 correction (MI355X_MICROARCH.md, HBM section; re-calibrated here on k_stream_bench, which reads and writes
exactly 256 MiB per launch: FETCH_SIZE reports 131075 KiB, WRITE_SIZE 262144 KiB): bytes read = 2 * FETCH_SIZE KiB,
bytes written = WRITE_SIZE KiB.   usage: pmc_summary.py <dir with <cfg>_FETCH_SIZE/ <cfg>_WRITE_SIZE/> [out.json]"""
import collections
import csv
import glob
import json
import os
import sys


def load(path):
    f = glob.glob(os.path.join(path, "*counter_collection.csv"))
    return list(csv.DictReader(open(f[0]))) if f else []


def main():
    root = sys.argv[1]
    res = {}
    for fdir in sorted(glob.glob(os.path.join(root, "*_FETCH_SIZE"))):
        cfg = os.path.basename(fdir)[: -len("_FETCH_SIZE")]
        agg = collections.OrderedDict()
        for r in load(fdir):
            k = r["Kernel_Name"].split("(")[0]
            a = agg.setdefault(k, {"launches": 0, "read_bytes": 0.0, "write_bytes": 0.0, "max_grid": 0})
            a["launches"] += 1
            a["read_bytes"] += float(r["Counter_Value"]) * 2.0 * 1024.0
            a["max_grid"] = max(a["max_grid"], int(r["Grid_Size"]))
        for r in load(os.path.join(root, cfg + "_WRITE_SIZE")):
            k = r["Kernel_Name"].split("(")[0]
            a = agg.setdefault(k, {"launches": 0, "read_bytes": 0.0, "write_bytes": 0.0, "max_grid": 0})
            a["write_bytes"] += float(r["Counter_Value"]) * 1024.0
        res[cfg] = agg
        print("==", cfg)
        for k, a in sorted(agg.items(), key=lambda kv: -(kv[1]["read_bytes"] + kv[1]["write_bytes"]))[:14]:
            print(f"{k[:48]:50s} launches {a['launches']:6d}  read {a['read_bytes'] / 1e9:10.3f} GB  write {a['write_bytes'] / 1e9:10.3f} GB")
    if len(sys.argv) > 2:
        json.dump(res, open(sys.argv[2], "w"), indent=1)


if __name__ == "__main__":
    main()
