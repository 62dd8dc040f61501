"""Per-dispatch view of ONE forward from a rocprofv3 --kernel-trace csv of bench.py: the dispatches between the last two
launches of a marker kernel (default k_points_to_blocks), in start order, with duration and the gap to the previous end.
usage: python tools/step_trace.py <kernel_trace.csv> [marker]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2] if len(sys.argv) > 2 else "k_points_to_blocks"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
a, b = marks[-2], marks[-1]
prev_end = None
tot = 0.0
for r in rows[a:b]:
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    name = re.sub(r"^void ", "", name).split("(")[0][:64]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = 0 if prev_end is None else (s - prev_end) / 1e3
    prev_end = e
    tot += (e - s) / 1e3
    print(f"{name:<66} {(e - s) / 1e3:8.2f} us  gap {gap:7.2f}  grid {r.get('Grid_Size_X') or r.get('Grid_Size')} wg {r.get('Workgroup_Size_X') or r.get('Workgroup_Size')}")
print(f"span {(int(rows[b]['Start_Timestamp']) - int(rows[a]['Start_Timestamp'])) / 1e3:.1f} us, kernels {tot:.1f} us, {b - a} dispatches")
