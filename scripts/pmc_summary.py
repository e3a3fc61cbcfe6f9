#!/usr/bin/env python
"""Per-kernel averages of rocprofv3 --pmc counter_collection.csv files (one row per dispatch x counter)."""
import csv
import collections
import glob
import os
import sys


def main(root):
    out = collections.OrderedDict()
    for path in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
        with open(path) as f:
            for row in csv.DictReader(f):
                k = (row["Kernel_Name"].split("(")[0], row["Counter_Name"])
                d = out.setdefault(k, [0.0, 0])
                d[0] += float(row["Counter_Value"])
                d[1] += 1
    print("kernel,counter,dispatches,avg_per_dispatch,total")
    for (kern, ctr), (tot, n) in out.items():
        print("%s,%s,%d,%.6g,%.6g" % (kern, ctr, n, tot / n, tot))


if __name__ == "__main__":
    main(sys.argv[1])
