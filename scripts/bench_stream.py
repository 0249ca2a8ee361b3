"""End-to-end frames/s of a synthetic stream through the device path: per frame the state write,
edge append, one update (prologue, correlation, BA x2) and keyframe bookkeeping, with stub feature / update networks.
    python scripts/bench_stream.py [frames] [drop_every]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cdv_slam_amd.stream import StreamRunner

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 300
drop_every = int(sys.argv[2]) if len(sys.argv) > 2 else 0
run = StreamRunner(torch.device("cuda:0"))
for f in range(40):   # reach the steady state (E = 47,712 at the default window)
    run.frame(drop=False)
torch.cuda.synchronize()
t = time.perf_counter()
for f in range(frames):
    n, E = run.frame(drop=bool(drop_every) and f % drop_every == drop_every - 1)
torch.cuda.synchronize()
dt = time.perf_counter() - t
print("stream: %d frames in %.3f s = %.0f frames/s end to end (n = %d keyframes, E = %d edges, %d inactive edges, "
      "drop every %d)" % (frames, dt, frames / dt, n, E, run.edges.E_inac, drop_every))
assert torch.isfinite(run.poses[:n]).all() and torch.isfinite(run.patches[:n * run.M]).all()
