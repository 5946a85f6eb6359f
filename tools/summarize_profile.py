"""Condenses a tools/profile.sh output directory into a short text summary:
per-kernel time from the kernel trace and per-dispatch FETCH_SIZE / WRITE_SIZE."""

import csv
import glob
import os
import sys
from collections import defaultdict


def find(root, suffix):
    return sorted(glob.glob(os.path.join(root, "**", "*" + suffix), recursive=True))


def main(out):
    for path in find(os.path.join(out, "trace"), "kernel_stats.csv"):
        print("== kernel stats:", os.path.relpath(path, out))
        for i, row in enumerate(csv.reader(open(path))):
            if i < 12:
                print(", ".join(c[:70] for c in row))
    for name in ("pmc_fetch", "pmc_write"):
        for path in find(os.path.join(out, name), "counter_collection.csv"):
            agg = defaultdict(lambda: [0, 0.0])
            for row in csv.DictReader(open(path)):
                key = (row["Kernel_Name"][:60], row["Counter_Name"])
                agg[key][0] += 1
                agg[key][1] += float(row["Counter_Value"])
            print("== counters:", os.path.relpath(path, out))
            for (k, c), (n, v) in sorted(agg.items()):
                print(f"{k:60s} {c:12s} dispatches={n:4d} mean={v / n:.6g}")


if __name__ == "__main__":
    main(sys.argv[1])
