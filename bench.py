#!/usr/bin/env python3
"""bench.py -- frames/sec of the CDV-SLAM per-frame update hot path on MI355X.

One "step" = one steady-state SLAM.update() hot path (cdvslam/slam.py:480-526) on the synthetic
512x384 TartanAir-shaped patch graph of BASELINE.md section 2 (default_cdvo.yaml: 96 patches, window 10,
E = 47,712 edges):  ingest of the new frame's features into the channels-last ring -> reproject ->
patch-graph index build -> 2-level correlation -> neighbors -> fastba.BA(iterations=2).
The Update network is replaced by fixed delta / weight tensors.  All inputs are resident in HBM before
the timed region.

    python bench.py --gpus N --steps K --warmup W
(N > 1: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`;
 every rank runs an independent sequence -- the path does not shard inside a sequence, SURVEY.md 8(e) --
 and the per-rank [fps, pose checksum] pairs are gathered with one RCCL all_gather.)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def corr_algorithmic_bytes(st):
    """out E*882*2 + coords E*72 + idx E*16 + gmap tiles U*C*9*2 + feature maps once."""
    import numpy as np
    E = st.E
    U = len(np.unique(st.kk))
    C = st.cfg.C
    h, w = st.cfg.ht // st.cfg.res, st.cfg.wd // st.cfg.res
    maps = st.cfg.mem * C * (h * w + (h // 4) * (w // 4)) * 2
    return E * 882 * 2 + E * 72 + E * 16 + U * C * 9 * 2 + maps


def cpu_baseline(st, max_seconds=30.0):
    """The CPU oracle (oracle/, a port of the reference algorithm) timed on this host on a bounded
    sample: one full update of the same workload (all E edges) -- reproject, 2-level correlation with
    the reference's half arithmetic, neighbors, 2 BA iterations."""
    import numpy as np
    from oracle import oracle as O
    t0 = time.perf_counter()
    coords = O.transform(st.poses, st.patches, st.intrinsics, st.ii, st.jj, st.kk)
    coords = np.ascontiguousarray(coords.transpose(0, 3, 1, 2))
    t1 = time.perf_counter()
    # bounded sample of the correlation: a contiguous slice of edges, scaled to E
    n = st.E
    probe = min(n, 2048)
    tp = time.perf_counter()
    O.slam_corr(st.gmap, st.fmap1, st.fmap2, coords[:probe], st.ii1[:probe], st.jj1[:probe], 3, "ref")
    per_edge = (time.perf_counter() - tp) / probe
    sample = int(min(n, max(probe, (max_seconds * 0.8) / max(per_edge, 1e-9))))
    tc = time.perf_counter()
    O.slam_corr(st.gmap, st.fmap1, st.fmap2, coords[:sample], st.ii1[:sample], st.jj1[:sample], 3, "ref")
    t_corr = (time.perf_counter() - tc) * (n / sample)
    t2 = time.perf_counter()
    O.neighbors(st.kk, st.jj)
    O.fastba(st.poses, st.patches, st.intrinsics[0], st.target, st.weight, st.lmbda, st.ii, st.jj, st.kk, st.t0, st.n, 2,
             np.float32)
    t3 = time.perf_counter()
    total = (t1 - t0) + t_corr + (t3 - t2)
    return {
        "value": 1.0 / total, "unit": "frames/s", "cores": O.num_threads(), "kind": "port",
        "sample": "1 update of the bench workload (E=%d): reproject + neighbors + BA(2 it) on all edges, "
                  "correlation timed on the first %d edges and scaled to E; oracle/cdv_oracle.c, %d thread(s)"
                  % (n, sample, O.num_threads()),
        "seconds_per_update": total,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="default", help="synthetic workload (default | stress | init | pr1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch the ~20 kernels of a step one by one instead of "
                    "replaying a captured hipGraph")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    args = ap.parse_args()

    import numpy as np
    import torch

    from cdv_slam_amd import synth
    from cdv_slam_amd.update import UpdatePath

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    dist_on = world > 1
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend="nccl", device_id=dev)  # RCCL over xGMI

    # every rank tracks its own sequence: same config, its own seed
    st = synth.make_state(args.config, buffer_size=64, seed=1234 + rank)
    up = UpdatePath(st, dev)

    def barrier():
        if dist_on:
            dist.barrier(device_ids=[local_rank])
        torch.cuda.synchronize()

    use_graph = not args.no_graph
    if use_graph:
        up.capture()
    run_step = up.step_graph if use_graph else up.step
    for _ in range(args.warmup):
        run_step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run_step()
    barrier()
    elapsed = time.perf_counter() - t0

    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if dist_on:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed_max = float(el.item())

    # ---- dominant kernel (fused correlation): HIP events around its launch inside full steps ----------
    corr_ms = []
    coords = None
    for _ in range(max(10, min(args.steps, 50))):
        up.step()
        coords = up.last_coords
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        up.corr_only(coords)
        e1.record()
        e1.synchronize()
        corr_ms.append(e0.elapsed_time(e1))
    corr_ms = float(np.median(corr_ms))
    corr_bytes = corr_algorithmic_bytes(st)
    achieved = corr_bytes / (corr_ms * 1e-3) / 1e9

    # per-stage breakdown (informative)
    stages = up.stage_times(reps=20)

    # ---- gather per-rank metrics: [fps, pose checksum] (trajectory-metric gather of SURVEY.md 8(e)) ---
    mine = torch.tensor([args.steps / elapsed, float(up.poses.double().abs().sum().item())], dtype=torch.float64,
                        device=dev)
    if dist_on:
        allm = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allm, mine)
        per_rank = [[float(x[0]), float(x[1])] for x in allm]
    else:
        per_rank = [[float(mine[0]), float(mine[1])]]

    if rank == 0:
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "corr_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(args.config, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        res = {
            "metric": "frames/sec per GPU (CDVO update, 96 patches, win=10); ATE vs ref",
            "value": world * args.steps / elapsed_max,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed_max / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f16 features / f32 accumulate+geometry",
            "data": "synthetic",
            "config": {
                "workload": "%s: SLAM.update hot path on a synthetic 512x384 stream, M=%d patches/frame, "
                            "OPTIMIZATION_WINDOW=%d, E=%d edges, %d free poses; ingest+reproject+graph index+"
                            "2-level corr+neighbors+BA(2 it); one independent sequence per GPU"
                            % (st.cfg.name, st.cfg.M, st.cfg.opt_window, st.E, st.n - st.t0),
                "edges": st.E, "patches_per_frame": st.cfg.M, "window": st.cfg.opt_window,
                "launch": "hipGraph replay" if use_graph else "eager launches",
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "kernel": "corr_fused_kernel<1>", "avg_launch_ms": corr_ms, "algorithmic_bytes": corr_bytes,
            },
            "stages_us": stages,
            "per_rank": per_rank,
        }
        if not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(st, args.cpu_seconds)
        else:
            res["cpu_baseline"] = None
        print(json.dumps(res))
    if dist_on:
        dist.barrier(device_ids=[local_rank])
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
