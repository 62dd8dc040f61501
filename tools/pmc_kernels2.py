"""Per-launch counter table of ONE steady-state scan from several rocprofv3 --pmc passes (merged by launch position).
Usage: pmc_kernels2.py <pass dir> ...   prints one line per launch: position, kernel, counter = value ..."""
import csv, glob, sys, collections
table = collections.OrderedDict()
for d in sys.argv[1:]:
    fs = glob.glob(d + "/*/*counter_collection.csv")
    if not fs:
        continue
    rows = [r for r in csv.DictReader(open(fs[0])) if "(anonymous namespace)::k_" in r["Kernel_Name"]]
    by = collections.OrderedDict()
    for r in rows:
        by.setdefault(int(r["Dispatch_Id"]), []).append(r)
    ids = sorted(by)
    starts = [i for i, did in enumerate(ids) if "k_points_to_blocks" in by[did][0]["Kernel_Name"]]
    a, b = starts[-2], starts[-1]
    for pos, did in enumerate(ids[a:b]):
        rs = by[did]
        name = rs[0]["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:46]
        ent = table.setdefault(pos, {"name": name, "c": collections.OrderedDict()})
        for r in rs:
            ent["c"][r["Counter_Name"]] = ent["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
for pos, ent in table.items():
    print(f"{pos:2d} {ent['name']:46s} " + " ".join(f"{k}={v:.4g}" for k, v in ent["c"].items()))
