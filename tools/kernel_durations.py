"""Per-launch kernel durations of one forward from a rocprofv3 kernel trace of the SERIAL bench run
(`bench.py --streams 1`): the launches between two consecutive k_points_to_blocks are one scan; position by position
the durations are averaged over the steady-state scans of the trace.  bench.py maps the positions onto its stages (same
launch order) and reports `roofline.dominant_kernel_rocprof_us` / `serial_kernel_rocprof_us_per_step` when the file
belongs to the running build (csrc_sha).  Usage: kernel_durations.py <rocprof dir> <out.json>"""
import csv, glob, hashlib, json, os, sys


def csrc_sha():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = os.path.join(root, "sps_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".inc.h")):
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def main():
    t = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
    rows = [r for r in csv.DictReader(open(t)) if "(anonymous namespace)::k_" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if "k_points_to_blocks" in r["Kernel_Name"]]
    scans = [rows[a:b] for a, b in zip(starts[:-1], starts[1:])]
    # steady state: the most common launch count, second half of the trace
    scans = scans[len(scans) // 2:]
    n = max(set(len(s) for s in scans), key=[len(s) for s in scans].count)
    scans = [s for s in scans if len(s) == n]
    names = [r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0] for r in scans[0]]
    us = [sum((int(s[i]["End_Timestamp"]) - int(s[i]["Start_Timestamp"])) for s in scans) / len(scans) / 1000.0 for i in range(n)]
    # the workload and device of the traced run (bench.py attaches the durations only to a line of the same workload): the
    # collect script runs `bench.py --streams 1` with its default --azimuth / --scenes
    workload = {"azimuth": int(os.environ.get("SPS_KD_AZIMUTH", "1750")), "scenes": int(os.environ.get("SPS_KD_SCENES", "4"))}
    device = None
    agents = glob.glob(sys.argv[1] + "/*/*agent_info.csv")
    if agents:
        gpus = [r for r in csv.DictReader(open(agents[0])) if r.get("Agent_Type") == "GPU"]
        device = (gpus[0].get("Product_Name") or gpus[0].get("Name")) if gpus else None
        if device and device.strip().lower() in ("unknown", "n/a", ""):
            device = None
    out = {"csrc_sha": csrc_sha(), "workload": workload, **({"device": device} if device else {}), "scans_averaged": len(scans), "launches": [[nm, round(u, 3)] for nm, u in zip(names, us)],
           "sum_us": round(sum(us), 2),
           "method": "rocprofv3 --kernel-trace over bench.py --streams 1; per launch position the mean duration over the "
                     "steady-state scans of the trace"}
    json.dump(out, open(sys.argv[2], "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "launches"}))


if __name__ == "__main__":
    main()
