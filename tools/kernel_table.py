"""Print per-kernel (name, calls/scan, avg us) from a rocprofv3 kernel_stats.csv and the timeline of one
steady-state scan from kernel_trace.csv.  Usage: kernel_table.py <dir> [--timeline]"""
import csv, glob, sys
d = sys.argv[1]
f = glob.glob(d + "/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
for r in rows:
    print(r["Name"][:64].ljust(64), r["Calls"].rjust(6), f"{float(r['AverageNs'])/1000:9.2f} us", r["Percentage"])
if "--timeline" in sys.argv:
    t = glob.glob(d + "/*/*kernel_trace.csv")[0]
    rows = list(csv.DictReader(open(t)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "k_points_to_blocks" in r["Kernel_Name"]]
    a, b = idx[len(idx) // 2], idx[len(idx) // 2 + 1]
    t0 = int(rows[a]["Start_Timestamp"])
    for r in rows[a:b]:
        st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"{(st - t0) / 1000:9.1f} us  dur {(en - st) / 1000:7.2f}  grid {r['Grid_Size_X']:>8}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']}  {r['Kernel_Name'][:70]}")
