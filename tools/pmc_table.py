"""Aggregate rocprofv3 --pmc counter_collection.csv per conv layer (by launch order within a scan).
Usage: pmc_table.py <dir> [<dir> ...]   (several passes are merged by layer)"""
import csv, glob, sys, collections
sys.path.insert(0, ".")
names = ["conv1p1s2", "block1.conv1", "block1.conv2", "conv2p2s2", "block2.conv1", "block2.conv2",
         "conv3p4s2", "block3.conv1", "block3.conv2", "conv4p8s2", "block4.conv1", "block4.conv2",
         "block5.conv1", "block5.conv2", "block6.conv1", "block6.conv2",
         "block7.conv1", "block7.conv2", "block8.conv1", "block8.conv2"]       # k_conv launches in forward order
up_names = ["convtr4", "convtr5", "convtr6", "convtr7"]                        # k_upconv launches
table = collections.defaultdict(lambda: collections.defaultdict(list))
other = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    rows = list(csv.DictReader(open(f)))
    by_disp = collections.OrderedDict()
    for r in rows:
        by_disp.setdefault(int(r["Dispatch_Id"]), []).append(r)
    conv_i = up_i = -1
    for did in sorted(by_disp):
        rs = by_disp[did]
        kn = rs[0]["Kernel_Name"]
        if "k_points_to_blocks" in kn:
            conv_i = -1
            up_i = -1
        if "k_upconv<" in kn:
            up_i += 1
            key = up_names[up_i] if 0 <= up_i < len(up_names) else f"up{up_i}"
            for r in rs:
                table[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            continue
        if "k_conv<" in kn or "k_conv_px<" in kn:
            conv_i += 1
            key = names[conv_i] if 0 <= conv_i < len(names) else f"conv{conv_i}"
            for r in rs:
                table[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        else:
            short = kn.split("(")[0].split("::")[-1][:28]
            for r in rs:
                other[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
def show(tab, order):
    ctrs = sorted({c for k in tab for c in tab[k]})
    print("layer".ljust(20) + "".join(c[-22:].rjust(24) for c in ctrs))
    for k in order:
        if k not in tab: continue
        print(k.ljust(20) + "".join(f"{sum(tab[k][c]) / max(len(tab[k][c]), 1):24.0f}" if c in tab[k] else " " * 24 for c in ctrs))
show(table, names + up_names)
print()
show(other, sorted(other))
