"""Per-launch timeline of ONE optimizer step from a rocprofv3 kernel trace (the last step between two adamw launches): start, duration, gap to the
previous launch, grid, kernel.   rocprofv3 --kernel-trace --output-format csv -d /tmp/ktr -- python bench.py --steps 2 --warmup 1 ... ;
python tools/step_timeline.py /tmp/ktr [--top N]"""
import csv
import glob
import os
import re
import sys


def short(n):
    n = n.replace("void ", "")
    n = re.sub(r"\((anonymous namespace)\)::", "", n)
    m = re.match(r"([\w:]+(<[^(]*>)?)", n)
    return (m.group(1) if m else n)[:70]


def main():
    root = sys.argv[1]
    f = root if root.endswith(".csv") else sorted(glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True))[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "adamw" in r["Kernel_Name"]]
    a, b = idx[-2] + 1, idx[-1] + 1
    t0, prev, tot, gaps = int(rows[a]["Start_Timestamp"]), None, 0, 0
    by = {}
    for r in rows[a:b]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = (s - prev) / 1e3 if prev else 0.0
        prev = max(e, prev or 0)
        tot += e - s
        gaps += max(gap, 0.0)
        k = short(r["Kernel_Name"])
        by.setdefault(k, [0, 0.0])
        by[k][0] += 1
        by[k][1] += (e - s) / 1e3
        print(f"{(s - t0) / 1e6:9.3f} ms {(e - s) / 1e3:10.1f} us  gap {gap:6.1f}  grid {r['Grid_Size_X']:>8}  {k}")
    span = (int(rows[b - 1]["End_Timestamp"]) - t0) / 1e6
    print(f"launches {b - a}  kernel time {tot / 1e6:.3f} ms  span {span:.3f} ms  gaps {gaps / 1e3:.3f} ms")
    for k, (n, us) in sorted(by.items(), key=lambda kv: -kv[1][1]):
        print(f"  {us / 1e3:9.3f} ms  {n:4d} x {us / n:9.1f} us  {k}")


if __name__ == "__main__":
    main()
