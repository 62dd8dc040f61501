"""HBM traffic of one scan from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE), per the recipe of
MI355X_MICROARCH.md section HBM: separate passes; FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE reports 1/2 of the bytes of wide coalesced streaming reads -> the read side is DOUBLED
(upper estimate: our gathers are narrower than 16 B/lane streams, for which the factor is uncalibrated).
Usage: traffic_pmc.py <fetch_dir> <write_dir> <out.json>"""
import csv, glob, hashlib, json, os, sys, collections


def csrc_sha():
    """Same hash as bench.py: the traffic figure is only reported for the build of the kernels it was measured with."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = os.path.join(root, "sps_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".inc.h")):
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def per_scan(d, counter):
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    starts = [i for i, r in enumerate(rows) if "k_points_to_blocks" in r["Kernel_Name"]]
    a, b = starts[-2], starts[-1]                    # one steady-state scan
    tot = collections.defaultdict(float)
    for r in rows[a:b]:
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        tot[name.split("(")[0][:48]] += float(r["Counter_Value"])
    return tot

fetch = per_scan(sys.argv[1], "FETCH_SIZE")
write = per_scan(sys.argv[2], "WRITE_SIZE")
kib_f, kib_w = sum(fetch.values()), sum(write.values())
out = {"csrc_sha": csrc_sha(), "fetch_size_kib_raw": kib_f, "write_size_kib": kib_w,
       "hbm_bytes_per_scan": int((2 * kib_f + kib_w) * 1024),
       "hbm_bytes_per_scan_uncorrected": int((kib_f + kib_w) * 1024),
       "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over bench.py --streams 1; "
                 "sum over the dispatches of one steady-state scan; read side doubled (gfx950 FETCH_SIZE = 1/2 for wide "
                 "streams, MI355X_MICROARCH.md HBM section); Infinity-Cache hits are counted, working set < 256 MiB",
       "per_kernel_kib": {k: [round(fetch.get(k, 0), 1), round(write.get(k, 0), 1)] for k in sorted(set(fetch) | set(write))}}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "per_kernel_kib"}))
