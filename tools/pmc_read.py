"""Sums rocprofv3 --pmc csv output per kernel and counter."""
import csv
import glob
import sys
from collections import defaultdict

tot = defaultdict(float)
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"]
            if "render_kernel" not in name:
                continue
            tot[(name.split("(")[0], row["Counter_Name"])] += float(row["Counter_Value"])
for k in sorted(tot):
    print(f"{k[0]:60s} {k[1]:28s} {tot[k]:.6g}")
