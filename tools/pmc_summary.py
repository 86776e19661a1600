"""Median per-dispatch value of each rocprofv3 counter per kernel, from the counter_collection.csv files of one or
more `rocprofv3 --pmc ...` passes.  Usage: pmc_summary.py OUT.csv DIR [DIR ...] (directories are searched)."""
import csv, glob, os, statistics, sys
out, dirs = sys.argv[1], sys.argv[2:]
vals = {}
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per = {}
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0]
            key = (name, r["Counter_Name"], r["Dispatch_Id"])
            per[key] = per.get(key, 0.0) + float(r["Counter_Value"])   # sum over XCC / instance rows
        for (name, ctr, _), v in per.items():
            vals.setdefault((name, ctr), []).append(v)
with open(out, "w", newline="") as fo:
    w = csv.writer(fo)
    w.writerow(["kernel", "counter", "dispatches", "median_per_dispatch"])
    for (name, ctr), v in sorted(vals.items()):
        if name.startswith("void aslr::") or "aslr" in name:
            w.writerow([name, ctr, len(v), "%.1f" % statistics.median(v)])
print("wrote", out, len(vals), "rows")
