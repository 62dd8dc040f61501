"""Average rocprofv3 --pmc counters per kernel (name filter) from one or more counter_collection.csv passes.
Usage: pmc_kernels.py <filter substring> <dir> [<dir> ...]; kernels are keyed by a short form of their template arguments."""
import csv, glob, sys, collections, re
flt = sys.argv[1]
tab = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[2:]:
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            kn = r["Kernel_Name"]
            if flt not in kn:
                continue
            m = re.search(r"(k_\w+)(<[^>]*>)?", kn)
            key = (m.group(1) + (m.group(2) or "")) if m else kn[:40]
            tab[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
ctrs = sorted({c for k in tab for c in tab[k]})
print("kernel".ljust(52) + "".join(c[-20:].rjust(22) for c in ctrs))
for k in sorted(tab):
    print(k[:50].ljust(52) + "".join(f"{sum(tab[k][c]) / max(len(tab[k][c]), 1):22.0f}" if c in tab[k] else " " * 22 for c in ctrs))
