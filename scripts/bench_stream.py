"""End-to-end frames/s of a synthetic stream with every size on the device (cdv_slam_amd.stream.
DeviceStreamRunner): per frame the state write, edge append, one update (prologue, correlation, operator stub, BA x2), the
point cloud and keyframe() with the reference's flow test decided on the device.
    python scripts/bench_stream.py [frames] [pose_step]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cdv_slam_amd.stream import DeviceStreamRunner

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 600
step = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1
run = DeviceStreamRunner(torch.device("cuda:0"), buffer_size=512, pose_step=step)
for f in range(45):   # reach the steady state (E = 47,712 at the default window)
    run.frame(drop=False)
for f in range(30):
    run.frame(drop=None)
torch.cuda.synchronize()
n0, _ = run.counts()
t = time.perf_counter()
for f in range(frames):
    run.frame(drop=None)
torch.cuda.synchronize()
dt = time.perf_counter() - t
n, E = run.counts()
print("stream: %d frames in %.3f s = %.0f frames/s end to end (n = %d keyframes, %d of the timed frames dropped by the keyframe "
      "test, E = %d edges, %d inactive edges)" % (frames, dt, frames / dt, n, frames - (n - n0), E, run.E_inac))
assert torch.isfinite(run.poses[:n]).all() and torch.isfinite(run.patches[:n * run.M]).all()
