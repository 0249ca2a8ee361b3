"""The loop-closure stream of bench.py (stream_fps.loop_closure) on its own, for rocprofv3 / cProfile:
    python scripts/run_loop_closure.py [frames] [--profile]"""
import cProfile, math, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cdv_slam_amd.stream import StreamRunner
dev = torch.device("cuda:0")
nf = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 90
circle = lambda t: [-0.5 * math.cos(2 * math.pi * t / 60.0), -0.5 * math.sin(2 * math.pi * t / 60.0), 0.0, 0.0, 0.0, 0.0, 1.0]
run = StreamRunner(dev, buffer_size=256, loop_closure=True, max_edge_age=1000, global_opt_freq=15, backend_thresh=64.0, pose_init=circle)
for _ in range(70):
    run.frame(drop=False)
torch.cuda.synchronize()
g0 = run.n_global
pr = cProfile.Profile() if "--profile" in sys.argv else None
t0 = time.perf_counter()
if pr: pr.enable()
for _ in range(nf):
    run.frame(drop=False)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
if pr: pr.disable()
t_all = time.perf_counter() - t0
print("%d frames: %.3f ms per frame (host issue %.3f), %d global bundle adjustments, n = %d, E = %d + %d inactive" % (
    nf, 1e3 * t_all / nf, 1e3 * t_host / nf, run.n_global - g0, run.n, run.edges.E, run.edges.E_inac))
if pr:
    ps = pstats.Stats(pr)
    rows = sorted(((tt, ct, nc, "%s:%d %s" % (os.path.basename(fn), ln, nm)) for (fn, ln, nm), (cc, nc, tt, ct, _) in ps.stats.items()), reverse=True)
    print("self us / cumul us / calls per frame")
    for tt, ct, nc, w in rows[:32]:
        print("%9.1f %9.1f %7.1f  %s" % (1e6 * tt / nf, 1e6 * ct / nf, nc / nf, w))
