"""N updates through the reference-shaped call sequence (DropinPath.step), for rocprofv3:  python scripts/run_dropin.py [config] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cdv_slam_amd import synth
from cdv_slam_amd.update import DropinPath
cfg = sys.argv[1] if len(sys.argv) > 1 else "default"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dp = DropinPath(synth.make_state(cfg), torch.device("cuda:0"))
for _ in range(20):
    dp.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    dp.step()
torch.cuda.synchronize()
print("%.1f us per update" % (1e6 * (time.perf_counter() - t0) / steps))
