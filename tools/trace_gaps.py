"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace csv: total per step and the largest gaps with the
kernels on either side.   python tools/trace_gaps.py <dir>"""
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]))
rows.sort()
ad = [i for i, r in enumerate(rows) if "adamw_kernel" in r[2]]
if len(ad) > 2:            # timed region of bench.py: after the 2nd optimizer step (warmup) up to the last one
    rows = rows[ad[1] + 1: ad[-1] + 1]
    print("steps in window:", len(ad) - 2)
gaps = []
busy = 0
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    gaps.append((s1 - e0, n0, n1))
    busy += e0 - s0
span = rows[-1][1] - rows[0][0]
print("kernels", len(rows), "span_ms", span / 1e6, "busy_ms", busy / 1e6, "idle_ms", (span - busy) / 1e6)
pos = [g for g in gaps if g[0] > 0]
print("gaps>0:", len(pos), "sum_ms", sum(g[0] for g in pos) / 1e6, " >20us:", sum(1 for g in pos if g[0] > 20000),
      "sum_ms", sum(g[0] for g in pos if g[0] > 20000) / 1e6)
import collections
by = collections.Counter()
for g in pos:
    by[(g[1], g[2])] += g[0]
for (a, b), v in by.most_common(14):
    print(round(v / 1e6, 2), "ms  after", a, "-> before", b)
