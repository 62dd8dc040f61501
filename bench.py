#!/usr/bin/env python3
"""bench.py -- scans/sec of the SPS per-scan hot path on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W [--config 2|3|4]
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...; without a launcher in
   the environment `python bench.py --gpus N` starts that command itself as a child process and relays rank 0's line)

A "step" is one pass of the whole hot path over one batch of synthetic input that is already resident in HBM:
quantise + voxel hash + stride pyramid + kernel maps + the 33 sparse convs of CustomMinkUNet14 (fp32) + slice +
sigmoid + the per-scan metric sums (written to a row of a device table).  It runs through sps_amd.engine.ScanEngine
-- the same loop scripts/predict.py runs -- with `--streams` independent scans in flight.

  --config 2 (default, the configuration BASELINE.json's metric is quoted on): one ~100k-point LiDAR-like scan +
             its variant-B submap at 0.1 m voxels per step;
  --config 3: streamed sequence, batch = 4 consecutive scans per step (collate_fn layout, batch column 0..3);
  --config 4: NCLT-like, >= 300k active level-0 voxels, 5x map, one scan per step.
`value` is always scans/s (config 3 processes 4 scans per step).

Everything that is not steady state (arena, weight upload, one forward per context) happens in ScanEngine.prepare()
before the timed region, whatever --warmup is; short runs use fewer streams (about three timed steps per stream).
`value` is measured with every step's [N,6] batch copied from a pinned HOST buffer to the device inside the timed region
("H2D of the input included"); the same steps with the inputs already resident in HBM are timed first and
reported next to it (`resident_inputs`).

Scans are sharded data-parallel (each rank owns its scans; weak scaling); the only exchange is one RCCL all-gather of
the per-scan metric rows at the end of the sequence, inside the timed region.

Prints ONE JSON line on rank 0, including
  "roofline":     whole-path algorithmic bytes / GPU time per step (hipEvents on the launch streams
                  over the timed region) against the 8 TB/s HBM peak, the per-layer breakdown (hipEvents around every
                  kernel stage, serial pass) and the dominant kernel with its own roofline fractions;
  "cpu_baseline": the C restatement of the MinkowskiEngine algorithm (oracle/, kind "port") timed on this box's host
                  cores on a bounded sample of the same workload (rank 0, N = 1 only);
  "parity":       scores / labels / dIoU of the measured path against that oracle on the same input and weights.
"""
from __future__ import annotations

import time

T_IMPORT0 = time.perf_counter()               # (before numpy / torch: `wall_s.imports_before_main` of the line)

import argparse  # noqa: E402
import hashlib  # noqa: E402
import json  # noqa: E402
import math  # noqa: E402
import os  # noqa: E402
import sys  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CFG = {
    "DATA": {"SPLIT": {"TEST": ["synthetic"]}},
    "MODEL": {"VOXEL_SIZE": 0.1},
    "FILTER": {"THRESHOLD": 0.84},
}
KEY_W, KEY_B = "model.MinkUNet.final.kernel", "model.MinkUNet.final.bias"


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


def calibrate_final(net, batch, eps, frac=0.3, gain=8.0):
    """Rescale the `final` 1x1 conv (kernel x gain, bias shifted) so that ~frac of the scan scores are >= eps.  With
    plain random-init weights no score reaches 0.84, every prediction is "stable" and TP = FP = 0: the uIoU of the
    metric line would be vacuous.  Synthetic weights are an input of the benchmark, not part of the path."""
    scores = net(batch)
    s = scores[batch[:, 4] == 1].double().clamp(1e-9, 1 - 1e-9)
    sd = net.state_dict()
    b0 = float(sd[KEY_B].reshape(-1)[0])
    logits = torch.log(s / (1 - s)) - b0
    k = max(1, min(len(logits), int(round((1.0 - frac) * len(logits)))))
    q = float(torch.kthvalue(gain * logits, k).values)
    sd[KEY_W].mul_(gain)
    sd[KEY_B].fill_(math.log(eps / (1.0 - eps)) - q)
    net.model.mark_weights_dirty()
    return float(sd[KEY_B].reshape(-1)[0])


def morton_sorted(batch, vs):
    """Rows of a [N,6] batch sorted by (batch index, Z-order of the 0.4 m block of the voxel, time index)."""
    q = np.floor(batch[:, 1:4] / np.float32(vs)).astype(np.int64) + (1 << 17)
    blk = q >> 2
    key = np.zeros(len(batch), np.int64)
    for bit in range(16):
        for ax in range(3):
            key |= ((blk[:, ax] >> bit) & 1) << (3 * bit + ax)
    key = (batch[:, 0].astype(np.int64) << 56) | (key << 1) | batch[:, 4].astype(np.int64)
    return np.ascontiguousarray(batch[np.argsort(key, kind="stable")])


def cpu_quota():
    """CPUs this job may use: the smallest of the host's cores, the process's affinity mask and the cgroup CPU quota
    (cgroup v2 cpu.max, v1 cfs_quota_us / cfs_period_us); the CPU baseline's thread sweep stops there."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(round(int(q) / int(p)))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, int(round(q / p))))
        except (OSError, ValueError):
            pass
    return max(1, n)


def same_device(a, b):
    """Product names as rocprofv3's agent table and torch spell them ("AMD Instinct MI355X" either way, give or take a
    prefix); an absent name does not veto."""
    if not a or not b:
        return True
    a, b = a.lower().strip(), b.lower().strip()
    return a in b or b in a


def csrc_sha():
    h = hashlib.sha256()
    d = os.path.join(ROOT, "sps_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".inc.h")):
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def build_workload(config, rank, azimuth, seq_scans, vs, scenes=4):
    """Returns (list of [N,6] float32 numpy batches that the steps cycle through, n_batches per step, description)."""
    from sps_amd import synthetic
    if config == 2:
        # `scenes` distinct scans (different noise / labels against the same map), cycled: no step re-reads the previous
        # step's input out of the caches
        map_points = synthetic.build_map(n_azimuth=azimuth)
        scans = [synthetic.make_scene(scan_seed=1 + rank + 100 * i, n_azimuth=azimuth, voxel_size=vs, map_points=map_points)["batch"]
                 for i in range(max(1, scenes))]
        return scans, 1, (f"BASELINE config 2: single ~100k-pt LiDAR-like scan + variant-B submap, 0.1 m voxel, "
                          f"CustomMinkUNet14 fp32, 1 scan per step per GPU ({len(scans)} distinct scans cycled)")
    if config == 3:
        n = max(4, (seq_scans // 4) * 4)
        scans = list(synthetic.make_sequence(n, first_seed=100 + 1000 * rank, voxel_size=vs, n_azimuth=azimuth))
        return [synthetic.collate(scans[i: i + 4]) for i in range(0, n, 4)], 4, (
            f"BASELINE config 3: streamed sequence ({n} distinct consecutive scans, sensor advancing 0.5 m/scan, cycled), "
            "batch = 4 scans per step (collate_fn layout), CustomMinkUNet14 fp32")
    if config == 4:
        sc = synthetic.make_nclt_scene(seed=5 + rank, voxel_size=vs)
        return [sc["batch"]], 1, ("BASELINE config 4: NCLT-like scan (3 merged 128-beam scans, 100 m range) + variant-B "
                                  "submap of a 25-position map, >= 300k active voxels, 1 scan per step per GPU")
    raise SystemExit("--config must be 2, 3 or 4")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--config", type=int, default=2, help="BASELINE config: 2 (default, the headline), 3 (batch = 4 sequence), 4 (NCLT-like)")
    ap.add_argument("--azimuth", type=int, default=1750, help="azimuth steps of the synthetic LiDAR (1750 -> ~100k pts)")
    ap.add_argument("--seq-scans", type=int, default=32, help="config 3: distinct scans of the sequence (cycled)")
    ap.add_argument("--scenes", type=int, default=4, help="config 2: distinct scans cycled through the steps")
    ap.add_argument("--streams", type=int, default=8,
                    help="independent steps in flight per GPU (one HIP stream + native context each, the current stream being "
                         "one of them); 1 = strictly serial.  The HIP runtime deals a process's streams to 4 hardware queues: "
                         "pipeline counts that are multiples of 4 load the queues evenly (DESIGN.md section 3.2).  Default 8; "
                         "runs of fewer than 40 steps use 4 so that every pipeline executes several timed steps")
    ap.add_argument("--side-streams-only", action="store_true", help="DIAGNOSTIC: the current stream is NOT one of the --streams pipelines (round 4's engine)")
    ap.add_argument("--pipelined-geometry", action="store_true",
                    help="DIAGNOSTIC: with --streams 1 keep the launch geometry of the several-pipelines engine (sps_ctx_set_pipelined): "
                         "what the rocprofv3 / counter passes of tools/collect_evidence.sh run, so that their per-kernel numbers "
                         "describe the kernels of the headline run")
    ap.add_argument("--exact-streams", action="store_true", help="DIAGNOSTIC: use exactly --streams streams even for short runs")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1 (nccl = RCCL; gloo only for "
                    "exercising the multi-rank control flow on a single GPU)")
    ap.add_argument("--force-dist", action="store_true", help="initialise the process group and run the metric all-gather even at "
                    "world size 1 (under torchrun --nproc-per-node 1): executes the RCCL path on a single-GPU box")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-h2d", action="store_true", help="skip the host-buffer-fed timed region: `value` is then the "
                    "resident-input rate (diagnostic runs; the headline includes the copy, SURVEY 8(d))")
    ap.add_argument("--no-stages", action="store_true", help="skip the per-layer hipEvent pass")
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="target wall time of the CPU baseline sample")
    ap.add_argument("--order", default="scan", choices=["scan", "morton"],
                    help="DIAGNOSTIC: row order of the synthetic batches -- 'scan' = as a LiDAR delivers them (ring by ring; "
                         "the default and the headline), 'morton' = rows pre-sorted by the Z-order of their voxel (probes how "
                         "much the voxel-row order, which follows the first occurrence of each block, costs in cache locality)")
    args = ap.parse_args()

    t_main = time.perf_counter()
    wall = {}                                   # where the run's wall time goes (seconds; the driver's clock holds the imports too)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves.  A CHILD process (this one has not
        # touched the GPU: nothing above calls into HIP), never an exec; rank 0's JSON line and the exit code are relayed.
        from sps_amd import parallel
        raise SystemExit(parallel.spawn_ranks(args.gpus, os.path.abspath(__file__), sys.argv[1:]))
    if world != args.gpus:
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")
    if args.backend != "nccl":
        local = local % max(torch.cuda.device_count(), 1)      # debug: several ranks may share one GPU
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    from sps_amd import hostplace
    placement = hostplace.bind_to_gpu_numa(local)      # before the first pinned allocation
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # nccl == RCCL on ROCm
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    from sps_amd import roofline
    from sps_amd.engine import ScanEngine
    from sps_amd.models.models import SPSNet, metrics_from_sums

    K, W = max(1, args.steps), max(0, args.warmup)
    eps = float(CFG["FILTER"]["THRESHOLD"])
    vs = CFG["MODEL"]["VOXEL_SIZE"]

    # ---- workload (host generation is not timed) ----------------------------------------------------------------
    batches_np, nb, workload = build_workload(args.config, rank, args.azimuth, args.seq_scans, vs, args.scenes)
    if args.order == "morton":
        batches_np = [morton_sorted(b, vs) for b in batches_np]
        workload += " [rows pre-sorted in Morton order: diagnostic]"
    batches = [torch.from_numpy(b).to(dev) for b in batches_np]
    pinned = [torch.from_numpy(b).pin_memory() for b in batches_np]
    max_rows = max(len(b) for b in batches_np)
    n_points = len(batches_np[0])
    n_scan = int((batches_np[0][:, 4] == 1).sum())

    # ---- weights: random init of the reference architecture, `final` calibrated so both labels occur -------------
    torch.manual_seed(0)
    net = synthetic_weights(SPSNet(CFG), seed=0).to(dev).eval().freeze()
    bias = torch.tensor([calibrate_final(net, batches[0], eps) if rank == 0 else 0.0], dtype=torch.float64, device=dev)
    if dist is not None:                      # replicated weights: every rank takes rank 0's calibration
        dist.broadcast(bias, 0)
        if rank != 0:
            sd = net.state_dict()
            sd[KEY_W].mul_(8.0)
            sd[KEY_B].fill_(float(bias.item()))
            net.model.mark_weights_dirty()

    # ---- engine: arena + shared weights + one forward per context, all before the timed region -------------------
    # every pipeline should execute several timed steps (a run of K steps on K streams measures the fill + drain of K
    # one-step pipelines): K < 5 x the default 8 pipelines -> 4 (one per hardware queue), K < 8 -> 2
    S = max(1, min(args.streams, K))
    if S > 2 and K < 5 * S and not args.exact_streams:
        from sps_amd.engine import SHORT_RUN_STREAMS
        S = SHORT_RUN_STREAMS if (args.streams >= SHORT_RUN_STREAMS and K >= 2 * SHORT_RUN_STREAMS) else 2
    eng = ScanEngine(net, dev, streams=S, max_rows=max_rows, table_rows=K * nb, stage_cols=0 if args.no_h2d else 6,
                     include_main=not args.side_streams_only, pipelined=True if args.pipelined_geometry else None)
    streams = eng.streams
    main_stream = eng.main

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    issue_s = [0.0]
    wall["setup_workload_weights_engine"] = round(time.perf_counter() - t_main, 2)

    def timed_region(feed):
        """W untimed + exactly K timed steps; returns (wall seconds incl. the metric all-gather, GPU ms first start ->
        last end over all launch streams, gathered rows or None, scores of the last step)."""
        eng.reset_table(K * nb)
        for i in range(W):
            eng.submit(feed[i % len(feed)], nb, row=(i % K) * nb)
        if dist is not None:                  # warm the collective of the timed region (communicator, buffers)
            dist.all_gather([torch.empty_like(eng.table) for _ in range(world)], eng.table)
        barrier()
        ev0 = [torch.cuda.Event(enable_timing=True) for _ in streams]
        ev1 = [torch.cuda.Event(enable_timing=True) for _ in streams]
        t0 = time.perf_counter()
        for st, e in zip(streams, ev0):
            e.record(st)
        scores = None
        for i in range(K):
            scores = eng.submit(feed[i % len(feed)], nb, row=i * nb)
        issue_s[0] = time.perf_counter() - t0          # host time to issue the K steps (no synchronisation inside)
        for st, e in zip(streams, ev1):
            e.record(st)
        gathered = None
        if dist is not None:                  # the path's one exchange step: per-scan metric rows
            for st in streams:
                main_stream.wait_stream(st)
            gathered = [torch.empty_like(eng.table) for _ in range(world)]
            dist.all_gather(gathered, eng.table)
        barrier()
        elapsed = time.perf_counter() - t0
        gpu_ms = max(ev0[0].elapsed_time(e1) for e1 in ev1)     # first start -> last end over the launch streams
        el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        if dist is not None:
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
        eng.finish()                          # sticky device errors (coordinate range) surface here
        return float(el.item()), gpu_ms, gathered, scores

    # Headline ("H2D of the input included"): every step's [N,6] batch is copied from a pinned host buffer to the
    # device on the step's stream INSIDE the timed region.  The same K steps with the inputs already resident in HBM
    # are measured next to it (`resident_inputs`).
    res_el, res_gpu, res_gathered, res_scores = timed_region(batches)
    res_issue_ms = issue_s[0] / K * 1e3
    resident = {"value": round(K * nb * world / res_el, 2), "unit": "scans/s", "ms_per_step": round(res_el / K * 1e3, 4),
                "gpu_ms_per_step": round(res_gpu / K, 4), "host_issue_ms_per_step": round(res_issue_ms, 4),
                "note": "same K steps, inputs already resident in HBM when the timed region starts"}
    if args.no_h2d:
        elapsed, gpu_ms, gathered, scores, inputs = res_el, res_gpu, res_gathered, res_scores, "resident in HBM (--no-h2d: diagnostic)"
    else:
        elapsed, gpu_ms, gathered, scores = timed_region(pinned)
        inputs = "pinned host buffer -> device copy of every step's [N,6] batch inside the timed region"
    host_issue_ms = issue_s[0] / K * 1e3
    rows_resident = (torch.cat(gathered, 0) if gathered is not None else eng.table[: K * nb]).cpu().numpy().copy()
    h2d = {"value": round(K * nb * world / elapsed, 2), "unit": "scans/s", "ms_per_step": round(elapsed / K * 1e3, 4),
           "gpu_ms_per_step": round(gpu_ms / K, 4), "bytes_per_step": int(batches_np[0].nbytes),
           "note": "= the headline `value`"} if not args.no_h2d else None

    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    wall["timed_regions_with_warmup"] = round(time.perf_counter() - t_main - wall["setup_workload_weights_engine"], 2)
    t_phase = time.perf_counter()
    # ---- whole-job numbers ------------------------------------------------------------------
    total_scans = K * nb * world
    value = total_scans / elapsed
    per_scan = [metrics_from_sums(r) for r in rows_resident]
    mean_metrics = {k: float(np.mean([m[k] for m in per_scan])) for k in ("loss", "r2", "dIoU", "precision", "recall", "f1")}
    confusion = {k: float(np.mean([m[k] for m in per_scan])) for k in ("tp", "fp", "fn", "tn")}

    # ---- roofline: algorithmic bytes of THIS run / GPU time per step --------------------------
    # (one of the engine's own pipelines: the same compact, inference-only context and the same native call -- forward + metric
    #  sums -- as the timed steps)
    ctx = eng.ctxs[-1]
    probe_out = torch.empty((nb, 8), dtype=torch.float64, device=dev)

    def probe():
        with torch.cuda.stream(streams[-1]):
            net.forward_metrics(batches[0], nb, probe_out, ctx=ctx)

    probe()
    torch.cuda.synchronize()
    V = ctx.level_counts()
    pairs3 = [sum(ctx.map_pairs(l)) for l in range(5)]
    pairs5 = sum(ctx.map_pairs(5))
    work = roofline.algorithmic_work(n_points, V, pairs3, pairs5)
    t_step = gpu_ms * 1e-3 / K
    achieved = work["bytes"] / t_step / 1e9
    stages, dom, dom_bytes, classes = [], None, None, {}
    if not args.no_stages:
        # per-stage breakdown (separate serial pass with one hipEvent after every kernel stage, on the launch stream)
        ctx.profile_enable(True)
        acc, reps, order = {}, 10, []
        for _ in range(reps):
            probe()
            for name, ms in ctx.profile_read():
                if name not in acc:
                    order.append(name)
                acc[name] = acc.get(name, 0.0) + ms / reps
        kernel_of = dict(zip([n for n, _ in ctx.profile_read()], ctx.profile_kernels()))
        ctx.profile_enable(False)
        def stage_work(name):
            """algorithmic work of a stage; a fused launch "a+b" (a transposed convolution in its producer's epilogue) carries both"""
            parts = [work["per_layer"].get(p) for p in name.split("+")]
            if any(p is None for p in parts):
                return None
            return {k: sum(p[k] for p in parts) for k in ("bytes", "flops")}

        for name in order:
            pl = stage_work(name)
            entry = {"stage": name, "ms": round(acc[name], 5)}
            if pl and acc[name] > 0:
                gbs = pl["bytes"] / (acc[name] * 1e-3) / 1e9
                tf = pl["flops"] / (acc[name] * 1e-3) / 1e12
                hf, mf = gbs / roofline.HBM_PEAK_GBS, tf / roofline.MFMA_F32_PEAK_TFLOPS
                # the roof that binds this layer = the longer of its two floors (bytes / 8 TB/s, flops / 157.3 TF)
                entry.update(alg_bytes=pl["bytes"], alg_gbs=round(gbs, 1), hbm_frac=round(hf, 4),
                             alg_tflops=round(tf, 2), mfma_f32_frac=round(mf, 4),
                             roof_bound="hbm" if hf >= mf else "mfma", roof_frac=round(max(hf, mf), 4))
            # the stages grouped by the kernel they launch -- what `rocprofv3 --stats` aggregates by name
            kern = kernel_of.get(name) or "(memset)"
            entry["kernel"] = kern
            cl = classes.setdefault(kern, {"us": 0.0, "launches": 0, "bytes": 0, "flops": 0})
            cl["us"] += acc[name] * 1e3
            cl["launches"] += len(kern.split("+")) if kern != "(memset)" else 0
            if pl:
                cl["bytes"] += pl["bytes"]
                cl["flops"] += pl["flops"]
            stages.append(entry)
        layer_stages = [s for s in stages if "alg_bytes" in s]
        dom = max(layer_stages, key=lambda s: s["ms"])                      # longest kernel stage
        dom_bytes = max(layer_stages, key=lambda s: s.get("alg_bytes", 0))   # most algorithmic bytes (HBM-shaped)
    # HBM bytes per scan from PMC passes (tools/traffic_pmc.py): only if they were taken with THIS build of the kernels
    traffic, traffic_note = None, "not measured for this build"
    tp = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tp) and args.config == 2:
        tj = json.load(open(tp))
        if tj.get("csrc_sha") == csrc_sha():
            traffic, traffic_note = tj["hbm_bytes_per_scan"], tj["method"]
        else:
            traffic_note = "profiles/traffic.json is stale: it was measured with an earlier build of the kernels"
    # the binding budget (DESIGN: MFMA + VALU time per SIMD and scan, tools/pmc_stalls.sh + vector_pipe_budget.py): attached only
    # when it was measured on THIS build of the kernels
    vpipe = {}
    vp = os.path.join(ROOT, "profiles", "vector_pipe_budget.json")
    if os.path.exists(vp) and args.config == 2:
        vj = json.load(open(vp))
        if vj.get("csrc_sha") == csrc_sha():
            vpipe = {"vector_pipe_us_per_scan": vj["vector_pipe_us_per_scan"], "vector_pipe_mfma_us": vj["mfma_us"],
                     "vector_pipe_valu_us": vj["valu_us"]}
    roof = {
        "bound": "hbm", "achieved": round(achieved, 2), "peak": roofline.HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / roofline.HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_note": traffic_note,
        "kernel": (f"whole per-step path (all launches of one forward + metric sums); {S} independent steps in flight on {S} "
                   "HIP streams") if S > 1 else "whole per-step path (all launches of one forward + metric sums), strictly serial",
        "alg_bytes_per_step": work["bytes"], "alg_flops_per_step": work["flops"],
        "gpu_ms_per_step": round(t_step * 1e3, 4),
        "frac_vs_measured_copy_6290": round(achieved / roofline.HBM_COPY_GBS, 5),
        "mfma_f32_frac": round(work["flops"] / t_step / 1e12 / roofline.MFMA_F32_PEAK_TFLOPS, 5),
        "dominant_kernel": dom, "largest_traffic_kernel": dom_bytes if stages else None,
        "stage_ms_sum": round(sum(s["ms"] for s in stages), 4), "stages": stages,
        "vector_pipe_us_per_scan": vpipe.get("vector_pipe_us_per_scan"), "vector_pipe_mfma_us": vpipe.get("vector_pipe_mfma_us"),
        "vector_pipe_valu_us": vpipe.get("vector_pipe_valu_us"),
    }
    # rocprofv3 kernel durations of the same build (tools/kernel_durations.py, written by tools/collect_evidence.sh): the
    # launches of one forward in order, mapped onto the stages -- kernel time without the launch gaps the hipEvent stages hold
    rocprof_us = {}
    kp = os.path.join(ROOT, "profiles", "kernel_durations.json")
    if classes and os.path.exists(kp) and args.config == 2:
        kj = json.load(open(kp))
        seq = [s for s in stages if s["kernel"] != "(memset)"]
        n_launch = sum(len(s["kernel"].split("+")) for s in seq)
        same_run = (kj.get("workload", {"azimuth": args.azimuth, "scenes": args.scenes}) == {"azimuth": args.azimuth, "scenes": args.scenes}
                    and same_device(kj.get("device"), torch.cuda.get_device_name(dev)))
        if kj.get("csrc_sha") == csrc_sha() and len(kj["launches"]) == n_launch and same_run:
            i = 0
            for s in seq:
                k = len(s["kernel"].split("+"))
                rocprof_us[s["kernel"]] = rocprof_us.get(s["kernel"], 0.0) + sum(u for _, u in kj["launches"][i:i + k])
                i += k
    if classes:
        # flat scalars (nested values do not survive the driver's parser): the kernel CLASS with the most time per step in the
        # serial pass (hipEvents around every launch), with its own roofline fractions
        dk, dv = max(classes.items(), key=lambda kv: kv[1]["us"])
        t = dv["us"] * 1e-6
        roof.update(
            dominant_kernel_name=dk, dominant_kernel_us=round(dv["us"], 2),
            dominant_kernel_launches_per_step=dv["launches"],
            dominant_kernel_alg_bytes=dv["bytes"], dominant_kernel_alg_flops=dv["flops"],
            dominant_kernel_hbm_frac=round(dv["bytes"] / t / 1e9 / roofline.HBM_PEAK_GBS, 5) if dv["bytes"] else None,
            dominant_kernel_mfma_frac=round(dv["flops"] / t / 1e12 / roofline.MFMA_F32_PEAK_TFLOPS, 5) if dv["flops"] else None,
            # the same class / the whole forward as rocprofv3 sees them (kernel durations only; null unless
            # profiles/kernel_durations.json was taken with this build): dominant_kernel_us above holds the launch gaps too
            dominant_kernel_rocprof_us=round(rocprof_us[dk], 2) if dk in rocprof_us else None,
            dominant_kernel_rocprof_mfma_frac=(round(dv["flops"] / (rocprof_us[dk] * 1e-6) / 1e12 / roofline.MFMA_F32_PEAK_TFLOPS, 5)
                                               if dk in rocprof_us and dv["flops"] else None),
            serial_kernel_rocprof_us_per_step=round(sum(rocprof_us.values()), 2) if rocprof_us else None,
            serial_kernel_us_per_step=round(sum(v["us"] for v in classes.values()), 2),
            launches_per_step=sum(v["launches"] for v in classes.values()),
            kernel_classes="; ".join(f"{k}: {v['us']:.1f} us / {v['launches']} launches"
                                     for k, v in sorted(classes.items(), key=lambda kv: -kv[1]["us"])),
        )

    wall["stage_pass_and_roofline"] = round(time.perf_counter() - t_phase, 2)
    t_phase = time.perf_counter()
    # ---- CPU baseline + parity: the oracle's C restatement on this box's host cores (checker only) ----
    cpu = None
    parity = None
    if world == 1 and not args.no_cpu_baseline:
        from oracle import c_oracle, sps_oracle
        sd = {k.replace("model.MinkUNet.", ""): v.detach().cpu().numpy() for k, v in net.state_dict().items()
              if "num_batches_tracked" not in k}
        blob = c_oracle.pack_blob(sd)
        host_cores = os.cpu_count() or 1
        quota = cpu_quota()                                       # what this JOB may use (cgroup quota / affinity), not the host's cores
        batch_np = batches_np[(K - 1) % len(batches_np)]          # the batch of the last timed step
        coords = np.ascontiguousarray(batch_np[:, :5])
        # the port runs tile by tile of output rows (round 4; the per-offset loops of rounds 1-3 stopped scaling at 16
        # threads); the serial voxel hash (first-occurrence order) bounds it: take the best of a few thread counts (one pass
        # each) and report the count actually used as `cores`
        best, single = None, None
        sweep = sorted({t for t in (1, 4, 8, 16, 32, 64, 128, 256) if t < quota} | {quota})
        for th in sweep:
            if args.config != 2 and th == 1:
                continue
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
        cpu = {"value": round(nb / per, 3), "unit": "scans/s", "cores": cores, "kind": "port",
               "sample": f"{nrep} repeats of one step's batch ({len(batch_np)} rows, {nb} scan(s)) through the C restatement of "
                         f"the MinkowskiEngine algorithm (ME itself unavailable), OpenMP over tiles of output rows with {cores} "
                         f"threads = the fastest of {'/'.join(map(str, sweep))} (the job's CPU quota: {quota} of the host's {host_cores} cores)",
               "cpu_quota": quota,
               "single_thread_scans_per_s": round(nb / single, 3) if single else None}
        s = scores.cpu().numpy()
        e = np.float32(eps)
        band = np.abs(ref - e) > 1e-5
        dio_gpu, dio_ref, conf_gpu, conf_ref = [], [], [], []
        for b in range(nb):
            r = batch_np[:, 0] == b
            mo = sps_oracle.predict_metrics(ref[r], batch_np[r], eps)
            mg = metrics_from_sums(rows_resident[(K - 1) * nb + b])
            dio_gpu.append(mg["dIoU"])
            dio_ref.append(mo["dIoU"])
            scan = batch_np[r][:, 4] == 1
            pr, gt = ref[r][scan] >= e, batch_np[r][scan, 5].astype(np.float32) >= e
            conf_ref.append([int((gt & pr).sum()), int((~gt & pr).sum()), int((gt & ~pr).sum()), int((~gt & ~pr).sum())])
            conf_gpu.append([int(mg["tp"]), int(mg["fp"]), int(mg["fn"]), int(mg["tn"])])
        parity = {"max_abs_score_err_vs_oracle": float(np.max(np.abs(s - ref))),
                  "label_mismatches_outside_1e-5_band": int(np.sum((s < e)[band] != (ref < e)[band])),
                  "rows_inside_band": int((~band).sum()),
                  "unstable_fraction_oracle": float((ref[batch_np[:, 4] == 1] >= e).mean()),
                  "dIoU_gpu": dio_gpu, "dIoU_oracle": dio_ref,
                  "confusion_TP_FP_FN_TN_gpu": conf_gpu, "confusion_TP_FP_FN_TN_oracle": conf_ref}

    wall["cpu_baseline_and_parity"] = round(time.perf_counter() - t_phase, 2)
    wall["main_total"] = round(time.perf_counter() - t_main, 2)
    wall["imports_before_main"] = round(t_main - T_IMPORT0, 2)
    out = {
        "metric": "scans/sec @100k pts, 0.1 m voxel", "value": round(value, 2), "unit": "scans/s",
        "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(elapsed / K * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": workload, "baseline_config": args.config, "scans_per_step": nb,
                   "scan_points": n_scan, "rows": n_points, "voxels_per_level": V,
                   "pairs_3x3x3x3_per_level": pairs3, "pairs_5x5x5x1": pairs5, "sharding": f"dp{world}",
                   "streams_per_gpu": S, "final_bias_calibrated": float(bias.item()),
                   "arena_mb_per_context": round(eng.ctxs[-1].arena_bytes() / 2**20, 1),
                   "arena_mb_all_contexts": round(sum(cx.arena_bytes() for cx in eng.ctxs) / 2**20, 1),
                   "compact_arenas": eng.compact},
        "roofline": roof, "cpu_baseline": cpu, "parity": parity, "mean_metrics": mean_metrics,
        "mean_confusion": confusion, "inputs": inputs, "resident_inputs": resident, "h2d_inclusive": h2d,
        "resident_value": resident["value"],
        "dist_backend": (dist.get_backend() if dist is not None else None), "dist_world_size": (dist.get_world_size() if dist is not None else 1),
        "host_cores": os.cpu_count(), "host_placement": placement,
        "host_issue_ms_per_step": round(host_issue_ms, 4), "wall_s": wall,
    }
    print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
