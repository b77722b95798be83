#!/usr/bin/env python
"""Average rocprofv3 counter-collection CSVs per kernel and counter.

    python tools/pmc_summarize.py OUT.csv DIR [DIR ...]

Every DIR is the output directory of one `rocprofv3 --pmc ...` pass; rows of the
library's own kernels (pg::*) are averaged over dispatches.
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    acc = defaultdict(list)
    for d in dirs:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(path, newline="") as f:
                for row in csv.DictReader(f):
                    name = row["Kernel_Name"]
                    if not name.startswith("pg::"):
                        continue
                    acc[(name.split("(")[0], row["Counter_Name"])].append(float(row["Counter_Value"]))
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "counter", "mean_per_dispatch", "dispatches"])
        for (k, c), v in sorted(acc.items()):
            w.writerow([k, c, "%.1f" % (sum(v) / len(v)), len(v)])


if __name__ == "__main__":
    main()
