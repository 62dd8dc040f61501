"""Concurrency analysis of a rocprofv3 --kernel-trace of the pipelined bench: per kernel name the mean duration under
contention (23 scans in flight) against its duration when it has the GPU alone (a --streams 1 trace), and the time-
averaged number of kernels in flight.  usage: overlap.py <pipelined_kernel_trace.csv> [<serial_kernel_trace.csv>]"""
import csv, sys, collections

def load(path):
    rows = []
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
    rows.sort()
    return rows

def stats(rows, skip=0.5):
    rows = [r for r in rows if r[2].startswith("k_")]          # the library's kernels
    sel = rows[int(len(rows) * skip):]                          # steady-state part (launch order)
    d = collections.defaultdict(list)
    for s, e, n in sel:
        d[n].append(e - s)
    wall = max(r[1] for r in sel) - min(r[0] for r in sel)
    busy = sum(e - s for s, e, _ in sel)
    return {n: (len(v), sum(v) / len(v)) for n, v in d.items()}, wall, busy

pip = load(sys.argv[1])
ps, wall, busy = stats(pip)
print(f"pipelined: wall {wall/1e3:.0f} us, sum of kernel durations {busy/1e3:.0f} us -> {busy/wall:.2f} kernels in flight on average")
ser = None
if len(sys.argv) > 2:
    ser, swall, sbusy = stats(load(sys.argv[2]))
    print(f"serial   : wall {swall/1e3:.0f} us, sum of kernel durations {sbusy/1e3:.0f} us -> {sbusy/swall:.2f}")
print(f"{'kernel':58s} {'calls':>6s} {'pipelined us':>12s} {'alone us':>9s} {'stretch':>8s} {'share of busy':>13s}")
for n, (c, m) in sorted(ps.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
    alone = ser.get(n, (0, float('nan')))[1] if ser else float('nan')
    print(f"{n[:58]:58s} {c:6d} {m/1e3:12.1f} {alone/1e3:9.1f} {m/alone if alone == alone else float('nan'):8.2f} {c*m/busy*100:12.1f}%")
