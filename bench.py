#!/usr/bin/env python3
"""bench.py -- scans/sec of the SPS per-scan hot path on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one pass of the whole hot path over one batch of synthetic input that is already
resident in HBM: quantise + voxel hash + stride pyramid + kernel maps + the 33 sparse convs of
CustomMinkUNet14 (fp32) + slice + sigmoid + the per-scan metric sums (written to a device row).
Workload = BASELINE config 2: one ~100k-point LiDAR-like scan + its variant-B submap at 0.1 m voxels.
Scans are sharded data-parallel (each rank owns its scans; weak scaling); the only exchange is one
RCCL all-gather of the per-scan metric rows at the end of the sequence, inside the timed region.

Prints ONE JSON line on rank 0, including
  "roofline":     whole-path algorithmic bytes / GPU time per scan (hipEvents on the
                  launch stream over the timed region) against the 8 TB/s HBM peak, plus the per-stage
                  breakdown and the dominant kernel stage;
  "cpu_baseline": the C restatement of the MinkowskiEngine algorithm (oracle/, kind "port") timed on
                  this box's host cores on a bounded sample of the same workload (rank 0, N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CFG = {
    "DATA": {"SPLIT": {"TEST": ["synthetic"]}},
    "MODEL": {"VOXEL_SIZE": 0.1},
    "FILTER": {"THRESHOLD": 0.84},
}


def synthetic_weights(net, seed=0):
    """Random-init weights of the reference architecture (resnet.py:87-94 scheme under
    torch.manual_seed) with randomised BN statistics so that BN is not the identity."""
    g = torch.Generator().manual_seed(seed)
    sd = net.state_dict()
    for k, v in sd.items():
        if k.endswith("bn.weight") or k.endswith("running_var"):
            v.copy_(torch.empty_like(v).uniform_(0.5, 1.5, generator=g))
        elif k.endswith("bn.bias") or k.endswith("running_mean"):
            v.copy_(torch.randn(v.shape, generator=g) * 0.1)
        elif k.endswith("final.bias"):
            v.zero_()
    net.model.mark_weights_dirty()
    return net


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--azimuth", type=int, default=1750, help="azimuth steps of the synthetic LiDAR (1750 -> ~100k pts)")
    ap.add_argument("--threads", type=int, default=1,
                    help="host threads issuing the launches (each owns streams s = t mod threads)")
    ap.add_argument("--streams", type=int, default=23,
                    help="independent scans in flight per GPU (one HIP stream + native context each); 1 = strictly serial. "
                         "The HIP runtime multiplexes streams onto 4 hardware queues: counts of the form 4k+3 measure "
                         "5-15 %% above their neighbours (DESIGN.md section 4), three streams already reach 94 %% of the best")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1 (nccl = RCCL; gloo only for "
                    "exercising the multi-rank control flow on a single GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--unfused-metrics", action="store_true",
                    help="forward and metric sums as two native calls (sps_forward + sps_metrics_dev) instead of sps_forward_metrics")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target wall time of the CPU baseline sample")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("for --gpus N > 1 launch with: python -m torch.distributed.run --nnodes=1 "
                             f"--nproc-per-node {args.gpus} --master-addr 127.0.0.1 --master-port P bench.py --gpus {args.gpus}")
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")
    if args.backend != "nccl":
        local = local % max(torch.cuda.device_count(), 1)      # debug: several ranks may share one GPU
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # nccl == RCCL on ROCm
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    from sps_amd import roofline, synthetic
    from sps_amd.models.models import SPSNet, get_context, metrics_from_sums

    # ---- workload: config 2, one scan (+ submap) per rank, seeds differ per rank --------------
    torch.manual_seed(0)
    net = synthetic_weights(SPSNet(CFG), seed=0).to(dev).eval().freeze()
    scene = synthetic.make_scene(scan_seed=1 + rank, n_azimuth=args.azimuth)
    batch_np = scene["batch"]
    batch = torch.from_numpy(batch_np).to(dev)
    n_points, n_scan = len(batch_np), scene["n_scan"]
    S = max(1, args.streams)
    main_stream = torch.cuda.current_stream()
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)] if S > 1 else [main_stream]
    ctxs = [get_context(local, st.cuda_stream) for st in streams]
    for cx in ctxs:
        cx.reserve(n_points)
    ctx = ctxs[0]
    K, W = args.steps, args.warmup
    rows = torch.zeros((max(K, 1), 8), dtype=torch.float64, device=dev)      # per-scan metric rows
    eps = float(CFG["FILTER"]["THRESHOLD"])
    torch.cuda.synchronize()

    def step(i):
        """One scan: SPSNet.forward (HIP path) + metric sums into this scan's device row, on stream i mod S."""
        st = streams[i % S]
        with torch.cuda.stream(st):
            if args.unfused_metrics:
                scores = net(batch)
                ctxs[i % S].metrics_dev(scores.data_ptr(), batch.data_ptr(), batch.stride(0), n_points, eps, 1,
                                        rows[i % max(K, 1)].data_ptr(), st.cuda_stream)
            else:
                scores, _ = net.forward_metrics(batch, 1, rows[i % max(K, 1)])
        step_scores[0] = scores
        return scores

    step_scores = [None]

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(W):
        step(i)
    if dist is not None:                      # warm the collective of the timed region (communicator, buffers)
        dist.all_gather([torch.empty_like(rows) for _ in range(world)], rows)
    barrier()
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in streams]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in streams]
    t0 = time.perf_counter()
    for st, e in zip(streams, ev0):
        e.record(st)
    if args.threads > 1:
        # several host threads issue the launches (ctypes releases the GIL inside libsps_hip.so): thread t owns
        # the streams s = t mod T, so a stream is only ever touched by one thread
        import threading
        T = args.threads

        def worker(t):
            torch.cuda.set_device(dev)
            for i in range(K):
                if (i % S) % T == t:
                    step(i)
        th = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
        for x in th:
            x.start()
        for x in th:
            x.join()
        scores = step_scores[0]
    else:
        for i in range(K):
            scores = step(i)
    for st, e in zip(streams, ev1):
        e.record(st)
    gathered = None
    if dist is not None:                      # the path's one exchange step: per-scan metric rows
        for st in streams:
            main_stream.wait_stream(st)
        gathered = [torch.empty_like(rows) for _ in range(world)]
        dist.all_gather(gathered, rows)
    barrier()
    elapsed = time.perf_counter() - t0
    # GPU time of the timed region: hipEvents on every launch stream, first start -> last end
    gpu_ms = max(ev0[0].elapsed_time(e1) for e1 in ev1)
    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    for cx, st in zip(ctxs, streams):
        cx.check_errors(st.cuda_stream)
    stream = streams[0]

    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- whole-job numbers ------------------------------------------------------------------
    total_scans = K * world
    value = total_scans / elapsed
    all_rows = torch.cat(gathered, 0) if gathered is not None else rows
    per_scan = [metrics_from_sums(r) for r in all_rows.cpu().numpy()]
    mean_metrics = {k: float(np.mean([m[k] for m in per_scan])) for k in ("loss", "r2", "dIoU", "precision", "recall", "f1")}

    # ---- roofline: algorithmic bytes of THIS run / GPU time per scan --------------------------
    V = ctx.level_counts()
    pairs3 = [sum(ctx.map_pairs(l)) for l in range(5)]
    pairs5 = sum(ctx.map_pairs(5))
    work = roofline.algorithmic_work(n_points, V, pairs3, pairs5)
    t_scan = gpu_ms * 1e-3 / K
    achieved = work["bytes"] / t_scan / 1e9
    # per-stage breakdown (separate short pass with stage events; not part of the timed region)
    ctx.profile_enable(True)
    acc, reps = {}, 10
    order = []
    for _ in range(reps):
        with torch.cuda.stream(streams[0]):
            net(batch)
        for name, ms in ctx.profile_read():
            if name not in acc:
                order.append(name)
            acc[name] = acc.get(name, 0.0) + ms / reps
    ctx.profile_enable(False)
    stages = []
    for name in order:
        pl = work["per_layer"].get(name)
        entry = {"stage": name, "ms": round(acc[name], 5)}
        if pl:
            entry["alg_gbs"] = round(pl["bytes"] / (acc[name] * 1e-3) / 1e9, 1) if acc[name] > 0 else None
            entry["alg_tflops"] = round(pl["flops"] / (acc[name] * 1e-3) / 1e12, 2) if acc[name] > 0 else None
        stages.append(entry)
    dom = max((s for s in stages if s["stage"] in work["per_layer"]), key=lambda s: s["ms"])
    # HBM bytes per scan from the committed PMC passes of this build (tools/traffic_pmc.py), if present
    traffic, traffic_note = None, "not measured for this build"
    tp = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tp):
        tj = json.load(open(tp))
        traffic, traffic_note = tj["hbm_bytes_per_scan"], tj["method"]
    roof = {
        "bound": "hbm", "achieved": round(achieved, 2), "peak": roofline.HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / roofline.HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_note": traffic_note,
        "kernel": "whole per-scan path (all launches of one forward + metric sums); "
                  f"{S} independent scans in flight on {S} HIP streams" if S > 1 else
                  "whole per-scan path (all launches of one forward + metric sums), strictly serial",
        "alg_bytes_per_scan": work["bytes"], "alg_flops_per_scan": work["flops"],
        "gpu_ms_per_scan": round(t_scan * 1e3, 4),
        "frac_vs_measured_copy_6290": round(achieved / roofline.HBM_COPY_GBS, 5),
        "mfma_f32_frac": round(work["flops"] / t_scan / 1e12 / roofline.MFMA_F32_PEAK_TFLOPS, 5),
        "dominant_stage": dom, "stage_ms_sum": round(sum(s["ms"] for s in stages), 4), "stages": stages,
    }

    # ---- CPU baseline: the oracle's C restatement on this box's host cores (checker only) ----
    cpu = None
    parity = None
    if world == 1 and not args.no_cpu_baseline:
        from oracle import c_oracle
        sd = {k.replace("model.MinkUNet.", ""): v.detach().cpu().numpy() for k, v in net.state_dict().items()
              if "num_batches_tracked" not in k}
        blob = c_oracle.pack_blob(sd)
        host_cores = os.cpu_count() or 1
        coords = np.ascontiguousarray(batch_np[:, :5])
        vs = CFG["MODEL"]["VOXEL_SIZE"]
        # the port's OpenMP loops are fine-grained: on a many-core host more threads is slower, so take
        # the best of a few thread counts (one scan each) and report the count actually used as `cores`
        best = None
        for th in sorted({t for t in (1, 4, 8, 16, 32, 64) if t <= host_cores}):
            c_oracle.forward(blob, coords, vs, nthreads=th, want_details=False)
            t = time.perf_counter()
            ref, info = c_oracle.forward(blob, coords, vs, nthreads=th, want_details=False)
            dt = time.perf_counter() - t
            if th == 1:
                single = dt
            if best is None or dt < best[1]:
                best = (th, dt)
        cores, first = best
        nrep = max(2, min(400, int(args.cpu_seconds / max(first, 1e-3))))
        t = time.perf_counter()
        for _ in range(nrep):
            c_oracle.forward(blob, coords, vs, nthreads=cores, want_details=False)
        per = (time.perf_counter() - t) / nrep
        cpu = {"value": round(1.0 / per, 3), "unit": "scans/s", "cores": cores, "kind": "port",
               "sample": f"{nrep} repeats of the same config-2 scan ({n_points} rows) through the C restatement of the "
                         f"MinkowskiEngine algorithm (ME itself unavailable), OpenMP with {cores} threads = the fastest of "
                         f"1/4/8/16/32/64 on this {host_cores}-core host",
               "single_thread_scans_per_s": round(1.0 / single, 3)}
        s = scores.cpu().numpy()
        e = np.float32(eps)
        band = np.abs(ref - e) > 1e-5
        parity = {"max_abs_score_err_vs_oracle": float(np.max(np.abs(s - ref))),
                  "label_mismatches_outside_1e-5_band": int(np.sum((s < e)[band] != (ref < e)[band]))}

    out = {
        "metric": "scans/sec @100k pts, 0.1 m voxel", "value": round(value, 2), "unit": "scans/s",
        "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(elapsed / K * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE config 2: single ~100k-pt LiDAR-like scan + variant-B submap, 0.1 m voxel, "
                               "CustomMinkUNet14 fp32, 1 scan per step per GPU",
                   "scan_points": n_scan, "rows": n_points, "voxels_per_level": V,
                   "pairs_3x3x3x3_per_level": pairs3, "pairs_5x5x5x1": pairs5, "sharding": f"dp{world}",
                   "streams_per_gpu": S},
        "roofline": roof, "cpu_baseline": cpu, "parity": parity, "mean_metrics": mean_metrics,
        "host_cores": os.cpu_count(),
    }
    print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
