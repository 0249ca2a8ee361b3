"""Workload for a kernel trace of the reference-shaped call sequence (update.DropinPath):
    rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -- python3 scripts/profile_dropin.py [config] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cdv_slam_amd import synth
from cdv_slam_amd.update import DropinPath

cfg = sys.argv[1] if len(sys.argv) > 1 else "default"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
dev = torch.device("cuda:0")
st = synth.make_state(cfg, buffer_size=64, seed=1234)
dp = DropinPath(st, dev)
for _ in range(5):
    dp.step()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(steps):
    dp.step()
torch.cuda.synchronize()
print("dropin: %.1f us per step" % (1e6 * (time.perf_counter() - t) / steps))
