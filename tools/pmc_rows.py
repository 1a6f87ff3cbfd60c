"""Aggregate a rocprofv3 --pmc counter_collection.csv: per kernel-name prefix, mean of each counter per dispatch."""
import collections
import csv
import glob
import sys

rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"][:60]
        rows[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, cs in rows.items():
    if "gemm" in name.lower() or "Cijk" in name:
        print(name, {k: round(sum(v) / len(v)) for k, v in sorted(cs.items())}, "n=%d" % len(next(iter(cs.values()))))
