"""From a rocprofv3 kernel trace of a multi-stream bench run: wall time, summed kernel time, union busy time."""
import csv, glob, sys
t = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(t)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_points_to_blocks" in r["Kernel_Name"]]
a, b = idx[len(idx) // 4], idx[3 * len(idx) // 4]
seg = rows[a:b]
n_scans = len([r for r in seg if "k_points_to_blocks" in r["Kernel_Name"]])
t0, t1 = int(seg[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in seg)
tot = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in seg)
busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"scans {n_scans}  wall/scan {(t1 - t0) / n_scans / 1000:.1f} us  sum(kernel)/scan {tot / n_scans / 1000:.1f} us  union-busy/scan {busy / n_scans / 1000:.1f} us  streams {len(set(r['Stream_Id'] for r in seg))} queues {len(set(r['Queue_Id'] for r in seg))}")
