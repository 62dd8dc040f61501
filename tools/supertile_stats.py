"""Distinct input rows referenced by a 64-row supertile of the 3^4 neighbour table (GPU dump via
tools/microbench/dump_nbr.py): how much would an LDS feature cache per workgroup save?"""
import sys, numpy as np
f = open(sys.argv[1] if len(sys.argv) > 1 else "/tmp/nbr.bin", "rb")
V, ldn, ntiles = np.fromfile(f, np.int64, 3)
nbr = np.fromfile(f, np.int32, 81 * ldn).reshape(81, ldn)
for rows_per in (64, 128):
    nst = (V + rows_per - 1) // rows_per
    distinct, gathers, span = [], [], []
    for s in range(nst):
        blk = nbr[:, s * rows_per: min(V, (s + 1) * rows_per)]
        v = blk[blk >= 0]
        u = np.unique(v)
        distinct.append(len(u)); gathers.append(len(v)); span.append(int(u.max() - u.min() + 1) if len(u) else 0)
    d, g, sp = np.array(distinct), np.array(gathers), np.array(span)
    print(f"V={V}: supertile of {rows_per} rows: gathers/supertile mean {g.mean():.0f}, distinct rows mean {d.mean():.0f} p90 {np.percentile(d,90):.0f} "
          f"p99 {np.percentile(d,99):.0f} max {d.max()}, reuse {g.sum()/d.sum():.2f}x, row-index span p50 {np.percentile(sp,50):.0f} p90 {np.percentile(sp,90):.0f}")
