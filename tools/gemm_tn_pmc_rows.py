import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_tn" in r["Kernel_Name"]:
            rows.append((int(r["Dispatch_Id"]), r["Counter_Name"], float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
rows.sort()
labels = ["lm_head sync0", "lm_head sync0", "lm_head sync128", "lm_head sync128", "gate|up sync0", "gate|up sync0", "gate|up sync128", "gate|up sync128"]
for (d, c, v, ns), lab in zip(rows, labels):
    print(f"{lab:18s} {c} = {v * 1024 * 2 / 1e9:7.2f} GB (x2 corrected)   {ns / 1e6:6.2f} ms")
