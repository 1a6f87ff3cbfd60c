import csv, sys, glob, re
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"void ", "", n)
    return n[:70]
print("total_ms", tot/1e6)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:28]:
    print(f'{float(r["TotalDurationNs"])/tot*100:5.1f}%  calls={r["Calls"]:>6}  avg_us={float(r["AverageNs"])/1e3:9.1f}  {short(r["Name"])}')
