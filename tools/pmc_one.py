"""Average PMC counters per kernel-name substring. Usage: pmc_one.py <dir> <substr>"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if sys.argv[2] in r["Kernel_Name"]:
        acc[(r["Kernel_Name"][:60], r.get("Grid_Size", ""), r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, g, c), v in sorted(acc.items()):
    print(k, g, c.ljust(28), f"{sum(v) / len(v):14.0f}", len(v))
