#!/usr/bin/env python3
"""bench.py -- frames/sec of the CDV-SLAM per-frame update hot path on MI355X.

One "step" = one steady-state SLAM.update() hot path (cdvslam/slam.py:480-526) on the synthetic
512x384 TartanAir-shaped patch graph of BASELINE.md section 2 (default_cdvo.yaml: 96 patches, window 10,
E = 47,712 edges):  ingest of the new frame's features into the channels-last ring -> reproject ->
patch-graph index build -> 2-level correlation -> neighbors -> fastba.BA(iterations=2).
The Update network is replaced by fixed delta / weight tensors.  All inputs are resident in HBM before
the timed region.

    python bench.py --gpus N --steps K --warmup W
N > 1: either launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (the ranks
read RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment) or run plainly -- then this process starts the N
rank processes itself, before it touches any GPU, and exits with their status.  Every rank runs an independent
sequence -- the path does not shard inside a sequence, SURVEY.md 8(e) -- and the per-rank [pose checksum, fps] pairs
are gathered with one RCCL all_gather: cdv_slam_amd/replicas.py.  The line is only printed if the gathered world
really has N ranks.
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
VALU_ISSUE_PEAK_GINST = 1024 * 2.4 / 4.0   # 256 CUs x 4 SIMDs, one VALU wave-instruction per 4 cycles, 2.4 GHz: 614.4 G/s


def corr_algorithmic_bytes(st):
    """out E*882*2 + coords E*72 + idx E*16 + gmap tiles U*C*9*2 + feature maps once."""
    import numpy as np
    E = st.E
    U = len(np.unique(st.kk))
    C = st.cfg.C
    h, w = st.cfg.ht // st.cfg.res, st.cfg.wd // st.cfg.res
    maps = st.cfg.mem * C * (h * w + (h // 4) * (w // 4)) * 2
    return E * 882 * 2 + E * 72 + E * 16 + U * C * 9 * 2 + maps


CORR_LAUNCHES_PER_PAIR = 4


def corr_event_ms(up, reps, launches=CORR_LAUNCHES_PER_PAIR):
    """average launch duration of the fused correlation: HIP event pairs on the launching stream around
    CORR_LAUNCHES_PER_PAIR back-to-back launches, recorded inside a stream of full steps (two queued in front of every
    pair, so that the device, not the host, sets the pace) and read after ONE synchronisation at the end; median over the
    pairs, divided by the launches per pair.  (One launch per pair carries the pair's own packet handling -- 2.5 us on most
    boxes of the pool, 6 us on some: 35.0 against 38.3 us for the same kernel, whose own duration rocprofv3 puts at 32.3 --
    32.9 us on either; four launches per pair leave a quarter of that in the figure.)"""
    import numpy as np
    import torch
    pairs = []
    for _ in range(reps):
        up.step()
        up.step()
        coords = up.last_coords
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(launches):
            up.corr_only(coords)
        e1.record()
        pairs.append((e0, e1))
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in pairs])) / launches


def closed_loop_ate(dev, frames=126, progress=None):
    """The metric's "ATE vs ref" half over a STREAM (BASELINE.json: within 1e-4; evaluate_tartan.py:63-70): the GPU stream runner
    and the oracle-driven runner (oracle/stream_py.py: the reference's arithmetic -- half-precision correlation, float32
    fastba -- and the same operator stub) side by side from the same frames, keyframes dropped by the reference's own flow
    test on each side's own state; Sim(3)-aligned ATE-RMSE between the two final trajectories.  Reduced frame size (24
    patches per frame, 256 x 192 images) so that the CPU side takes seconds, not minutes."""
    from cdv_slam_amd import metrics
    from cdv_slam_amd.stream import DeviceStreamRunner
    from oracle.stream_py import StreamOracle, closed_loop
    cfg = dict(M=24, ht=192, wd=256, C=24, buffer_size=256, keyframe_thresh=2.5)
    t0 = time.perf_counter()
    res = closed_loop(DeviceStreamRunner(dev, **cfg), StreamOracle(**cfg), frames=frames, drop="flow", progress=progress)
    ate = metrics.ate_rmse(res["poses_oracle"], res["poses_gpu"])
    return {"value": ate, "unit": "scene units (RMSE of the camera centres, Sim(3)-aligned)", "frames": res["frames"],
            "keyframes": res["keyframes"], "dropped_by_the_keyframe_test": res["dropped"], "edges": res["edges"],
            "edge_lists_bit_identical_every_frame": bool(res["edges_identical"]),
            "keyframe_decisions_differing": len(res["decisions_differ"]), "target": 1e-4,
            "path_length": float(np_path_length(res["poses_oracle"])),
            "against": "oracle/stream_py.py: orc_transform + orc_corr (c10::Half arithmetic) + orc_fastba float32 + edges_py, "
                       "closed loop over %d frames, 24 patches/frame, 256x192" % res["frames"],
            "seconds": time.perf_counter() - t0}


def np_path_length(poses):
    import numpy as np
    from cdv_slam_amd import metrics
    c = metrics.camera_centres(poses)
    return np.linalg.norm(np.diff(c, axis=0), axis=1).sum()


def cpu_baseline(st, max_seconds=30.0, gpu_poses=None):
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
    p_cpu, _, _ = O.fastba(st.poses, st.patches, st.intrinsics[0], st.target, st.weight, st.lmbda, st.ii, st.jj, st.kk,
                           st.t0, st.n, 2, np.float32)
    t3 = time.perf_counter()
    total = (t1 - t0) + t_corr + (t3 - t2)
    # BASELINE.json configs[0]: the reference's Python ba.py path (cdvslam/ba.py:86-185 as restated in oracle/ba_py.py,
    # numpy float32, ep = 1.0) on the 10-frame / 96-patch fully connected graph, two Gauss-Newton iterations
    from cdv_slam_amd import synth
    from oracle import ba_py
    pr = synth.make_state("pr1", features=False)
    hh, ww = pr.cfg.ht // pr.cfg.res, pr.cfg.wd // pr.cfg.res
    P, X = pr.poses, pr.patches
    tb = time.perf_counter()
    for _ in range(2):
        P, X, _ = ba_py.BA(P, X, pr.intrinsics, pr.target, pr.weight, pr.lmbda, pr.ii, pr.jj, pr.kk,
                           [-64, -64, ww + 64, hh + 64], ep=1.0, fixedp=1, dtype=np.float32)
    t_bapy = time.perf_counter() - tb
    try:      # the numpy restatement's matrix products run on the BLAS pool: say how many threads that was
        from threadpoolctl import threadpool_info
        blas_threads = max([int(i.get("num_threads", 1)) for i in threadpool_info()] or [1])
    except Exception:
        blas_threads = None
    ate = None
    if gpu_poses is not None:
        # "ATE vs ref" of the metric, in the form available here: the trajectory after ONE update on the GPU against the
        # CPU port's from the same state, Sim(3)-aligned RMSE of the camera centres as evaluate_tartan.py:63-70
        from cdv_slam_amd import metrics
        lo = max(st.t0 - 12, 0)
        ate = {"what": "ONE update from the benchmark's patch-graph state (the stream form is `closed_loop` below)",
               "value": metrics.ate_rmse(p_cpu[lo:st.n], gpu_poses[lo:st.n]), "unit": "scene units (RMSE, Sim(3)-aligned)",
               "frames": int(st.n - lo), "against": "oracle/cdv_oracle.c float32, same patch-graph state, 1 update (2 GN iterations)",
               "moved_by_update": metrics.ate_rmse(st.poses[lo:st.n], gpu_poses[lo:st.n])}
    return {
        "ate_vs_oracle": ate,
        "ba_py_pr1": {"value": 1.0 / t_bapy, "unit": "BA(2 it)/s", "edges": int(pr.E), "seconds": t_bapy, "threads": blas_threads,
                      "what": "oracle/ba_py.py (restated cdvslam/ba.py) on BASELINE.json configs[0], numpy f32"},
        "value": 1.0 / total, "unit": "frames/s", "cores": O.num_threads(), "kind": "port",
        "sample": "1 update of the bench workload (E=%d): reproject + neighbors + BA(2 it) on all edges; "
                  "2-level correlation (reference half arithmetic) %s; oracle/cdv_oracle.c, %d thread(s)"
                  % (n, "on all edges" if sample >= n else "timed on the first %d edges and scaled to E" % sample,
                     O.num_threads()),
        "seconds_per_update": total,
    }


def dry_run(args):
    """the N-rank plumbing of main() without a GPU: same group calls, a trivial step"""
    import numpy as np
    import torch
    from cdv_slam_amd.replicas import ReplicaGroup, aggregate_rate, summarise
    if os.environ.get("CDV_BENCH_TEST_FAIL_RANK") == os.environ.get("RANK", "0"):
        raise SystemExit(3)      # tests/test_replicas_gloo.py: a rank that dies before the rendezvous
    grp = ReplicaGroup(backend="gloo", device=torch.device("cpu"))
    if grp.world != args.gpus:
        raise SystemExit("process group has %d ranks, --gpus says %d" % (grp.world, args.gpus))
    x = np.random.default_rng(grp.sequence_seed()).standard_normal(1 << 16)
    for _ in range(args.warmup):
        x = np.roll(x, 1)
    grp.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        x = np.roll(x, 1)
    grp.barrier()
    elapsed = time.perf_counter() - t0
    elapsed_max = grp.max_over_ranks(elapsed)
    per_rank = grp.gather_metrics([float(np.abs(x).sum()), args.steps / elapsed])
    if len(per_rank) != args.gpus:
        raise SystemExit("gathered %d rows, --gpus says %d" % (len(per_rank), args.gpus))
    if grp.rank == 0:
        print(json.dumps({"metric": "dry run (no GPU, not a measurement)", "dry_run": True, "n_gpus": grp.world,
                          "value": aggregate_rate(args.steps, elapsed_max, grp.world), "unit": "steps/s",
                          "steps": args.steps, "warmup": args.warmup, "per_rank": per_rank,
                          "per_rank_summary": summarise(per_rank)}))
    grp.close()


def spawn_ranks(n, argv, timeout_s=900.0):
    """`bench.py --gpus N` run plainly: start the N rank processes (one per GPU, rendezvous on 127.0.0.1) from a parent
    that never touches a GPU, forward rank 0's JSON line, exit with the worst status.  All children are polled: when one
    exits non-zero (or the whole run exceeds `timeout_s`) the others are terminated instead of waiting in the rendezvous
    for torch's own timeout.  Every rank's stdout / stderr is kept on disk (gpurun_out/bench_rank<r>.log)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    logdir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(logdir, exist_ok=True)
    procs, logs = [], []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        out = open(os.path.join(logdir, "bench_rank%d.log" % r), "w+")
        err = open(os.path.join(logdir, "bench_rank%d.err" % r), "w")
        logs.append((out, err))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdout=out, stderr=err))
    t0 = time.perf_counter()
    failed = None
    while any(p.poll() is None for p in procs):
        bad = [r for r, p in enumerate(procs) if p.poll() not in (None, 0)]
        if bad or time.perf_counter() - t0 > timeout_s:
            failed = "rank %d exited with %d" % (bad[0], procs[bad[0]].returncode) if bad else "timed out after %.0f s" % timeout_s
            for p in procs:                    # the exact PIDs started above
                if p.poll() is None:
                    p.terminate()
            for p in procs:
                try:
                    p.wait(timeout=20)
                except subprocess.TimeoutExpired:
                    p.kill()
            break
        time.sleep(0.05)
    rcs = [p.wait() for p in procs]
    logs[0][0].seek(0)
    sys.stdout.write(logs[0][0].read())
    sys.stdout.flush()
    for out, err in logs:
        out.close()
        err.close()
    if failed:
        sys.stderr.write("bench.py --gpus %d: %s; per-rank logs in %s/bench_rank*.{log,err}\n" % (n, failed, logdir))
        return max(1, max(abs(rc) for rc in rcs))
    return max(abs(rc) for rc in rcs)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--windows", type=int, default=5, help="timed windows of K steps each; the median window is reported")
    ap.add_argument("--config", default="default", help="synthetic workload (default | stress | init | pr1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--settle-ms", type=float, default=300.0, help="untimed clock-settling load before the warm-up steps")
    ap.add_argument("--graph", action="store_true", help="replay the ~20 launches of a step as one captured hipGraph "
                    "(default: eager launches on one stream; measured to run at the same rate)")
    ap.add_argument("--no-graph", action="store_true", help="(default, kept for older command lines)")
    ap.add_argument("--overlap", action="store_true", help="build the patch-graph index on a side stream under the "
                    "correlation (measured: no gain, the dispatcher does not interleave the small kernels)")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--no-dropin", action="store_true", help="skip timing the reference-shaped drop-in call sequence")
    ap.add_argument("--no-extra", action="store_true", help="skip the stress-configuration and stream sub-records")
    ap.add_argument("--dry-run", action="store_true", help="no GPU: the launch / rendezvous / gather plumbing only, on "
                    "gloo with a trivial step (tests/test_replicas_gloo.py); the line says dry_run and is not a measurement")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))

    import numpy as np
    import torch

    from cdv_slam_amd import synth
    from cdv_slam_amd.update import UpdatePath

    from cdv_slam_amd.replicas import ReplicaGroup, aggregate_rate, summarise

    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world_env:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world_env))
    if args.dry_run:
        return dry_run(args)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    grp = ReplicaGroup(backend="nccl", device=dev)  # "nccl" is RCCL on ROCm: xGMI between the GPUs of the node
    rank, world = grp.rank, grp.world
    if world != args.gpus:
        raise SystemExit("process group has %d ranks, --gpus says %d" % (world, args.gpus))

    # every rank tracks its own sequence: same config, its own seed
    st = synth.make_state(args.config, buffer_size=64, seed=grp.sequence_seed())
    up = UpdatePath(st, dev, overlap=args.overlap)

    use_graph = args.graph
    if use_graph:
        up.capture()
    run_step = up.step_graph if use_graph else up.step
    # Clock settling, before (not instead of) the W warm-up steps: the chip's power management takes some tenths of a
    # second of sustained load to reach the clocks a stream of updates runs at; a 20-step run (2.5 ms of GPU work) measured
    # cold reads 6 % low.  Untimed, the same step as everything else, reported as `settle_ms`.
    if args.settle_ms > 0:
        t_s = time.perf_counter()
        while time.perf_counter() - t_s < args.settle_ms * 1e-3:
            for _ in range(50):
                run_step()
            torch.cuda.synchronize()
    dbg = os.environ.get("CDV_BENCH_DEBUG") == "1"

    def events_since(block, before):
        """failure events (ops.BA_EVENTS order) an EventBlock has counted since `before`: a timed section during which a
        bundle adjustment was skipped measured less work than it claims -- its rate is then reported as null"""
        torch.cuda.synchronize()
        return [int(a - b) for a, b in zip(block.counts(), before)]

    if dbg:
        print("debug: corr event ms after settle: %.5f" % corr_event_ms(up, 50), file=sys.stderr, flush=True)
    for _ in range(args.warmup):
        run_step()
    # `--windows` timed windows of EXACTLY K steps each, every one bracketed by barrier + synchronise on both sides and
    # reduced with MAX over the ranks; the line reports the MEDIAN window (SURVEY 8(d) asks for a median) and lists all of them
    windows = []
    ev_head0 = up.graph.events.counts()
    for _ in range(max(1, args.windows)):
        grp.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            run_step()
        grp.barrier()
        windows.append(grp.max_over_ranks(time.perf_counter() - t0))
    elapsed_max = float(np.median(windows))
    elapsed = elapsed_max
    ev_head = events_since(up.graph.events, ev_head0)

    # ---- dominant kernel (fused correlation): HIP events around its launch inside full steps ----------
    # (event pairs recorded back to back in a stream of full steps and read after ONE synchronisation at the end: a host
    # wait after every launch lets the chip idle between iterations, and the kernel then reads 2 us slower than in the
    # stream the metric is about.  The pair still includes ~2 us of packet handling around the kernel: rocprofv3's average
    # over the same command, profiles/r2_kernel_stats_default.txt, is the kernel's own duration)
    corr_ms = corr_event_ms(up, max(10, min(args.steps, 50)))
    # ... and ONE launch per pair inside full steps (its inputs as cold as a step leaves them, the pair's own packet handling
    # included): the figure rounds 1-3 quoted, kept next to the four-launch one so that the two can be compared
    corr_ms_single = corr_event_ms(up, max(10, min(args.steps, 50)), launches=1)
    if dbg:
        print("debug: corr event ms after the windows: %.5f, again %.5f" % (corr_ms, corr_event_ms(up, 50)), file=sys.stderr, flush=True)
    corr_bytes = corr_algorithmic_bytes(st)
    achieved = corr_bytes / (corr_ms * 1e-3) / 1e9

    # per-stage breakdown (informative)
    stages = up.stage_times(reps=20)

    # ---- the same update the way an UNCHANGED slam.py issues it: planar rings written with torch ops, two
    # cuda_corr.forward calls + torch.stack, cuda_ba.neighbors, cuda_ba.forward, all through the drop-in module names
    # (cdv_slam_amd.update.DropinPath; tests/test_gpu_parity.py::test_reference_call_sequence_through_the_dropin_names
    # shows it gives the same bits as UpdatePath) -- timed like the headline number, reported next to it
    dropin = None
    if st.fmap1 is not None and not args.no_dropin:
        from cdv_slam_amd.update import DropinPath
        dp = DropinPath(st, dev)
        nd = 100                      # its own fixed count: the sub-records do not depend on --steps
        for _ in range(5):
            dp.step()
        torch.cuda.synchronize()
        from cdv_slam_amd import ops as _ops
        dg = _ops._device_graph(dev)
        ev_d0 = dg.events.counts()
        td = time.perf_counter()
        for _ in range(nd):
            dp.step()
        torch.cuda.synchronize()
        td = time.perf_counter() - td
        ev_d = events_since(dg.events, ev_d0)
        dropin = {"value": (nd / td) if not any(ev_d) else None, "unit": "frames/s", "ms_per_step": 1e3 * td / nd, "steps": nd,
                  "ba_events": ev_d,
                  # the per-call bookkeeping ran compiled (cdv_slam_amd/_dropin_fast.so) if it was armed when the timed region ended
                  "compiled_bookkeeping": bool(_ops._armed_pair is not None and _ops._armed_graph is not None),
                  "what": "the reference's own call sequence through the install_dropin() names on the reference's state layouts, handed over "
                          "the way slam.py hands it over: gmap / poses / patches / intrinsics as FRESH views per access, edge tensors "
                          "re-created by torch.cat every step, the two per-level correlations through an autograd.Function under "
                          "autocast, torch.stack of the results; per step one index build, one tile conversion, one fused launch"}
        del dp

    # ---- the other configurations the metric's neighbourhood asks about, as sub-records measured after the timed region
    # (rank 0 of a single-GPU run only): BASELINE.json configs[4] (`stress`: 196 patches per frame, window 22) as updates/s
    # of the same UpdatePath, and SURVEY.md 8(d)(iii): end-to-end frames/s of a synthetic stream -- state write, edge
    # append, update, keyframe bookkeeping with a dropped frame every third frame, stub networks, Python glue included
    extra = {}
    if world == 1 and not args.no_extra and st.fmap1 is not None and args.config == "default":
        try:
            st2 = synth.make_state("stress", buffer_size=64, seed=grp.sequence_seed())
            up2 = UpdatePath(st2, dev)
            for _ in range(30):
                up2.step()
            torch.cuda.synchronize()
            n2 = 200
            ev_s0 = up2.graph.events.counts()
            t2 = time.perf_counter()
            for _ in range(n2):
                up2.step()
            torch.cuda.synchronize()
            t2 = time.perf_counter() - t2
            ev_s = events_since(up2.graph.events, ev_s0)
            ms2 = corr_event_ms(up2, 30)
            b2 = corr_algorithmic_bytes(st2)
            tr2, tr2_src = None, None
            try:      # counter traffic of the same kernel on this workload, collected in separate --pmc passes (profiles/)
                pm2 = json.load(open(os.path.join(ROOT, "profiles", "corr_traffic.json"))).get("stress", {})
                tr2 = pm2.get("hbm_bytes_per_launch")
                tr2_src = None if tr2 is None else ("NOT measured in this run: profiles/%s_corr_pmc_stress.txt, 2 x FETCH_SIZE + WRITE_SIZE "
                                                   "per launch (scripts/pmc_r2.sh)" % pm2.get("round", "r5"))
            except Exception:
                pass
            extra["stress"] = {"value": (n2 / t2) if not any(ev_s) else None, "unit": "frames/s", "ms_per_step": 1e3 * t2 / n2, "steps": n2,
                               "ba_events": ev_s,
                               "roofline": {"bound": "hbm", "achieved": b2 / (ms2 * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                            "frac": b2 / (ms2 * 1e-3) / 1e9 / HBM_PEAK_GBS, "kernel": "corr_fused2_kernel<24, 2, stream>",
                                            "avg_launch_ms": ms2, "algorithmic_bytes": b2, "traffic": tr2, "traffic_source": tr2_src},
                               "what": "BASELINE.json configs[4]: the same update path, M=%d patches/frame, window %d, E=%d edges, "
                                       "%d free poses" % (st2.cfg.M, st2.cfg.opt_window, st2.E, st2.n - st2.t0)}
            del up2, st2
        except Exception as ex:      # a sub-record must not take the headline down with it
            extra["stress"] = {"error": repr(ex)}
        try:
            from cdv_slam_amd.stream import DeviceStreamRunner
            run = DeviceStreamRunner(dev, buffer_size=512, keyframe_thresh=12.5, pose_step=0.1)    # KEYFRAME_THRESH of config.py:20
            for _ in range(45):       # reach the steady state (E = 47,712 at the default window)
                run.frame(drop=False)
            for f in range(30):       # ... and the reference's keyframe test deciding on the device
                run.frame(drop=None)
            torch.cuda.synchronize()
            n_a, _ = run.counts()
            nf = 600
            ev_f0 = run.events.counts()
            ts = time.perf_counter()
            for f in range(nf):
                run.frame(drop=None)
            t_enq = time.perf_counter() - ts
            torch.cuda.synchronize()
            ts = time.perf_counter() - ts
            n_kf, E_s = run.counts()
            ev_f = events_since(run.events, ev_f0)
            # the same stream as hipGraph replays of two captured frames each (every size is on the device, so nothing in the 24
            # launches depends on the host; launches sized for the edge capacity)
            graph = None
            try:
                if run.cur != 0:
                    run.frame(drop=None)
                replay = run.capture_pair()
                for _ in range(10):
                    replay()
                torch.cuda.synchronize()
                ev_g0 = run.events.counts()
                n_g0, _ = run.counts()
                tg = time.perf_counter()
                for _ in range(150):
                    replay()
                tg_enq = time.perf_counter() - tg
                torch.cuda.synchronize()
                tg = time.perf_counter() - tg
                n_g1, _ = run.counts()
                ev_g = events_since(run.events, ev_g0)
                graph = {"value": (300 / tg) if not any(ev_g) else None, "unit": "frames/s", "host_ms_per_frame": 1e3 * tg_enq / 300,
                         "ba_events": ev_g, "keyframes_kept_in_the_timed_frames": int(n_g1 - n_g0),
                         "what": "two frames (24 launches) captured once as a hipGraph, 150 replays; inputs staged once; the kept "
                                 "keyframes go round the patch table's 30 frames of ids during the replays"}
            except Exception as ex:
                graph = {"error": repr(ex)}
            extra["stream_fps"] = {"value": (nf / ts) if not any(ev_f) else None, "unit": "frames/s", "ms_per_frame": 1e3 * ts / nf, "frames": nf,
                                   "ba_events": ev_f, "hipgraph_replay": graph,
                                   "host_enqueue_ms_per_frame": 1e3 * t_enq / nf, "edges": int(E_s), "keyframes": int(n_kf),
                                   "keyframes_dropped_in_the_timed_frames": int(nf - (n_kf - n_a)),
                                   "what": "SURVEY 8(d)(iii): synthetic 512x384 stream end to end with every size on the device "
                                           "(cdv_slam_amd.stream.DeviceStreamRunner), 12 launches per frame and no read-back: state "
                                           "write + patch tiles + edge append, ring ingest + index + reprojection, two-level "
                                           "correlation, operator stub, BA(2), point cloud of the removal window (slam.py:524-526), "
                                           "keyframe test from flow_mag decided ON THE DEVICE (slam.py:399-413) with the removal, index "
                                           "shift, frame-buffer shift and removal-window pruning it triggers (slam.py:415-458); stub "
                                           "feature / update networks; device RNG drawn once; Python glue included"}
            del run
            # BASELINE.json configs[4] as a stream: 196 patches per frame, OPTIMIZATION_WINDOW 22 -- the bundle adjustment on the
            # 10 < N <= 32 path with its window on the device (cdv_ba_forward_dyn); same frame sequence, fewer frames
            try:
                run2 = DeviceStreamRunner(dev, M=196, opt_window=22, buffer_size=256, keyframe_thresh=12.5, pose_step=0.1)
                for _ in range(45):
                    run2.frame(drop=False)
                for _ in range(20):
                    run2.frame(drop=None)
                torch.cuda.synchronize()
                n_a2, _ = run2.counts()
                nf2 = 240
                ev_w0 = run2.events.counts()
                tw = time.perf_counter()
                for _ in range(nf2):
                    run2.frame(drop=None)
                torch.cuda.synchronize()
                tw = time.perf_counter() - tw
                n_k2, E_w = run2.counts()
                ev_w = events_since(run2.events, ev_w0)
                extra["stream_fps"]["stress"] = {"value": (nf2 / tw) if not any(ev_w) else None, "unit": "frames/s",
                                                 "ms_per_frame": 1e3 * tw / nf2, "frames": nf2, "ba_events": ev_w, "edges": int(E_w),
                                                 "keyframes": int(n_k2), "keyframes_dropped_in_the_timed_frames": int(nf2 - (n_k2 - n_a2)),
                                                 "what": "the same device-resident stream at BASELINE.json configs[4]: 196 patches per "
                                                         "frame, OPTIMIZATION_WINDOW 22 (bundle adjustment on the 10 < N <= 32 path)"}
                del run2
            except Exception as ex:
                extra["stream_fps"]["stress"] = {"error": repr(ex)}
            # BASELINE.json configs[2]'s mechanics (LOOP_CLOSURE): the host-sized StreamRunner -- the reference decides on the host
            # whether an update sees long-range edges (slam.py:507) and selects its loop edges on the CPU (patchgraph.py:71-97), and
            # so does this runner: one read-back per update, edges_loop every GLOBAL_OPT_FREQ frames, the GLOBAL bundle adjustment
            # over inactive + active edges when loop edges exist.  A camera on a closed circle so that loops really close.
            try:
                import math
                from cdv_slam_amd.stream import StreamRunner
                circle = lambda t: [-0.5 * math.cos(2 * math.pi * t / 60.0), -0.5 * math.sin(2 * math.pi * t / 60.0), 0.0, 0.0, 0.0, 0.0, 1.0]
                run3 = StreamRunner(dev, buffer_size=256, loop_closure=True, max_edge_age=1000, global_opt_freq=15, backend_thresh=64.0,
                                    pose_init=circle)
                for _ in range(70):
                    run3.frame(drop=False)
                torch.cuda.synchronize()
                g0, ev_l0 = run3.n_global, run3.graph.events.counts()
                nf3 = 90
                tl = time.perf_counter()
                for _ in range(nf3):
                    run3.frame(drop=False)
                torch.cuda.synchronize()
                tl = time.perf_counter() - tl
                ev_l = events_since(run3.graph.events, ev_l0)
                if run3.graph_full is not None:
                    ev_l = [a + b for a, b in zip(ev_l, run3.graph_full.events.counts())]
                extra["stream_fps"]["loop_closure"] = {
                    "value": (nf3 / tl) if not any(ev_l) else None, "unit": "frames/s", "ms_per_frame": 1e3 * tl / nf3, "frames": nf3,
                    "ba_events": ev_l, "global_bundle_adjustments_in_the_timed_frames": int(run3.n_global - g0),
                    "keyframes": int(run3.n), "edges": int(run3.edges.E), "inactive_edges": int(run3.edges.E_inac),
                    "what": "configs[2]'s mechanics on a synthetic closed path: patch ring of MAX_EDGE_AGE = 1000 frames, proximity loop "
                            "edges every GLOBAL_OPT_FREQ = 15 frames, global BA over inactive + active edges (hundreds of free poses) when "
                            "an update sees long-range edges; host-sized runner (cdv_slam_amd.stream.StreamRunner) with the reference's own "
                            "host decisions and read-backs, NOT the device-resident stream"}
                del run3
            except Exception as ex:
                extra["stream_fps"]["loop_closure"] = {"error": repr(ex)}
        except Exception as ex:
            extra["stream_fps"] = {"error": repr(ex)}

    # ---- gather per-rank metrics: [pose checksum, fps] (trajectory-metric gather of SURVEY.md 8(e)) ---
    per_rank = grp.gather_metrics([float(up.poses.double().abs().sum().item()), args.steps / elapsed])
    if len(per_rank) != args.gpus:
        raise SystemExit("gathered %d rows, --gpus says %d" % (len(per_rank), args.gpus))

    if rank == 0:
        traffic, pmc = None, {}
        tpath = os.path.join(ROOT, "profiles", "corr_traffic.json")
        if os.path.exists(tpath):
            try:
                pmc = json.load(open(tpath)).get(args.config, {})
                traffic = pmc.get("hbm_bytes_per_launch")
            except Exception:
                traffic, pmc = None, {}
        # second bound of the same kernel: vector-instruction issue.  Wave-instructions per launch from the PMC pass
        # (SQ_INSTS_VALU, profiles/corr_traffic.json) over the launch time measured above, against one VALU
        # wave-instruction per 4 cycles per SIMD (the issue cost of the guide's cycle table; 1,024 SIMDs at 2.4 GHz)
        valu = pmc.get("valu_insts_per_launch")
        roofline_valu = None
        if valu:
            a_v = valu / (corr_ms * 1e-3) / 1e9
            roofline_valu = {"bound": "valu_issue", "achieved": a_v, "peak": VALU_ISSUE_PEAK_GINST, "unit": "G wave-instr/s",
                             "frac": a_v / VALU_ISSUE_PEAK_GINST, "kernel": "corr_fused2_kernel<24, 2>",
                             "insts_per_launch": valu, "avg_launch_ms": corr_ms,
                             "source": "SQ_INSTS_VALU of profiles/%s_corr_pmc_%s.txt" % (pmc.get("round", "r2"), args.config)}
        res = {
            "metric": "frames/sec per GPU (CDVO update, 96 patches, win=10); ATE vs ref",
            # (a window in which a bundle adjustment was skipped or not applied measured less than an update: no number then)
            "value": aggregate_rate(args.steps, elapsed_max, world) if not any(ev_head) else None,
            "ba_events": ev_head,
            "ba_events_what": "failure events counted by the BA kernels during the timed windows, per path (ops.BA_EVENTS order: not "
                              "positive definite, U_max exceeded, hand-off lost, index in its error state); every sub-record carries "
                              "its own; a non-zero entry nulls that record's value",
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "settle_ms": args.settle_ms,
            "ms_per_step": 1e3 * elapsed_max / args.steps,
            "timed_windows": {"count": len(windows), "steps_each": args.steps, "reported": "median",
                              "ms_per_step": [1e3 * w / args.steps for w in windows]},
            "rccl_ranks": world if world > 1 else None,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "dtype_detail": "f32 arithmetic throughout (MFMA accumulate, geometry, BA); feature maps, patch tiles and the "
                            "correlation output are stored as f16 like the reference's",
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
                "traffic_source": (None if traffic is None else "NOT measured in this run: FETCH_SIZE (x2, gfx950) + WRITE_SIZE per "
                                   "launch from separate rocprofv3 --pmc passes of the same workload, profiles/corr_traffic.json (%s)"
                                   % pmc.get("round", "r2")),
                "kernel": "corr_fused2_kernel<24, 2, stream>", "avg_launch_ms": corr_ms, "algorithmic_bytes": corr_bytes,
                "launches_per_event_pair": CORR_LAUNCHES_PER_PAIR,
                "avg_launch_ms_single_launch_pair": corr_ms_single,
                "frac_single_launch_pair": corr_bytes / (corr_ms_single * 1e-3) / 1e9 / HBM_PEAK_GBS,
            },
            "roofline_valu": roofline_valu,
            "stages_us": stages,
            "dropin_fps": dropin,
            "stress": extra.get("stress"),
            "stream_fps": extra.get("stream_fps"),
            "per_rank": per_rank,
            "per_rank_summary": summarise(per_rank),
        }
        if not args.no_cpu_baseline and world == 1:   # the CPU port is timed next to the single-GPU number only
            up.reset()
            up.step()
            torch.cuda.synchronize()
            res["cpu_baseline"] = cpu_baseline(st, args.cpu_seconds, gpu_poses=up.poses.cpu().numpy())
            one = res["cpu_baseline"].pop("ate_vs_oracle")
            try:
                res["ate_vs_oracle"] = closed_loop_ate(dev, progress=lambda m: print(m, file=sys.stderr, flush=True))
                res["ate_vs_oracle"]["one_update_at_benchmark_size"] = one
            except Exception as ex:
                res["ate_vs_oracle"] = {"error": repr(ex), "one_update_at_benchmark_size": one}
        else:
            res["cpu_baseline"] = None
            res["ate_vs_oracle"] = None
        print(json.dumps(res))
    grp.close()


if __name__ == "__main__":
    main()
