"""The real-data branch of scripts/predict.py at scale: a synthetic $DATA tree of N LiDAR-like scans of
~100k points (poses along a track, map = union of scans taken along it, one point per 0.1 m voxel), then
`scripts/predict.py -seq ... --timing` with the items assembled on the device, and (few scans) on the host for comparison.
usage: predict_items_timing.py [n_scans=64] [host_scans=4] [map_voxel=0.1] [tree_dir (kept; with a 5th argument: only write the tree)]"""
import os, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sps_amd import synthetic

def main():
    n_scans = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    host_scans = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    map_voxel = float(sys.argv[3]) if len(sys.argv) > 3 else 0.1
    root = sys.argv[4] if len(sys.argv) > 4 else tempfile.mkdtemp(prefix="sps_data_")
    seq = "20220629"
    os.makedirs(root, exist_ok=True)
    os.makedirs(os.path.join(root, "maps")); os.makedirs(os.path.join(root, "sequence", seq, "scans")); os.makedirs(os.path.join(root, "sequence", seq, "poses"))
    step = 0.5
    # map: scans every 2 m along the track, merged, one point per 0.1 m voxel (a map built by accumulating registered scans
    # and voxel-grid filtering at the network's resolution): ~3 map points within r = 0.1 m of a scan point on a surface
    t0 = time.time()
    parts = [synthetic.lidar_scan(seed=1000 + i, x_offset=2.0 * i)[:, :3] for i in range(int(n_scans * step / 2.0) + 3)]
    m = np.concatenate(parts).astype(np.float64)
    key = np.floor(m / map_voxel).astype(np.int64)
    _, first = np.unique(key, axis=0, return_index=True)
    m = m[np.sort(first)]
    np.save(os.path.join(root, "maps", "base_map.asc.npy"), np.c_[m, np.ones(len(m))])
    np.savetxt(os.path.join(root, "sequence", seq, "map_transform"), np.eye(4), delimiter=",")
    npts = []
    for i in range(n_scans):
        s = synthetic.lidar_scan(seed=1 + i, x_offset=step * i)                    # [n,4] = x,y,z,label in the map frame
        pose = np.eye(4); pose[0, 3] = step * i
        sensor = s.astype(np.float64).copy(); sensor[:, 0] -= step * i             # sensor frame: pose brings it back
        stamp = f"{1656500000.0 + i:.6f}"
        np.save(os.path.join(root, "sequence", seq, "scans", stamp + ".npy"), sensor)
        np.savetxt(os.path.join(root, "sequence", seq, "poses", stamp + ".txt"), pose, delimiter=",")
        npts.append(len(s))
    print(f"data tree: {n_scans} scans of {int(np.mean(npts))} points, map {len(m)} points at one per {map_voxel} m voxel ({time.time() - t0:.1f} s to write)", flush=True)
    if len(sys.argv) > 5:
        return
    env = dict(os.environ, DATA=root, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cfg = os.path.join(ROOT, "config", "config.yaml")
    for name, extra in (("device items, batch 1", ["-b", "1"]), ("device items, batch 4", ["-b", "4"])):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "predict.py"), "-seq", seq, "-c", cfg, "--timing"] + extra,
                           capture_output=True, text=True, env=env, cwd=ROOT)
        print(name, "|", [l for l in r.stdout.splitlines() if l.startswith("timing")] or r.stderr[-1500:], flush=True)
    if host_scans:
        # the reference's path on a few scans: DataLoader + scipy KD-trees (the first `host_scans` scans of the sequence)
        for f in sorted(os.listdir(os.path.join(root, "sequence", seq, "scans")))[host_scans:]:
            os.remove(os.path.join(root, "sequence", seq, "scans", f)); os.remove(os.path.join(root, "sequence", seq, "poses", f[:-4] + ".txt"))
        t = time.time()
        r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "predict.py"), "-seq", seq, "-c", cfg, "--timing", "--host-items", "-b", "1"],
                           capture_output=True, text=True, env=env, cwd=ROOT)
        print(f"host items (scipy), {host_scans} scans, whole run {time.time() - t:.1f} s |", [l for l in r.stdout.splitlines() if l.startswith("timing")] or r.stderr[-1500:])

if __name__ == "__main__":
    main()
