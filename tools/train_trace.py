"""Per-dispatch view of ONE training step from a rocprofv3 --kernel-trace csv: the dispatches between the last two
k_permute_weights launches, in stream order, with duration and the gap to the previous dispatch's end.
usage: python tools/train_trace.py <kernel_trace.csv> [--only k_wgrad]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
only = sys.argv[3] if len(sys.argv) > 3 and sys.argv[2] == "--only" else None
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "k_permute_weights" in r["Kernel_Name"]]
a, b = marks[-2], marks[-1]
prev_end = None
tot = {}
for r in rows[a:b]:
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    name = re.sub(r"^void ", "", name).split("(")[0][:60]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = 0 if prev_end is None else (s - prev_end) / 1e3
    prev_end = e
    g = (r.get("Grid_Size_X") or r.get("Grid_Size"), r.get("Grid_Size_Y"), r.get("Grid_Size_Z"))
    tot.setdefault(name, [0, 0.0])
    tot[name][0] += 1
    tot[name][1] += (e - s) / 1e3
    if only is None or only in name:
        print(f"{name:<62} {(e - s) / 1e3:8.2f} us  gap {gap:7.2f}  grid {g}")
print("---- totals of the step")
for k, (n, t) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:<62} {n:4d} {t:9.1f} us")
print(f"span {(int(rows[b]['Start_Timestamp']) - int(rows[a]['Start_Timestamp'])) / 1e3:.1f} us, kernels {sum(t for _, t in tot.values()):.1f} us")
