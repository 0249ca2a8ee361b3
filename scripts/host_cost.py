"""Host-side cost of one update step: enqueue time of a tiny graph (the GPU work is shorter than the Python / ctypes
path, so the step time IS the host cost).  72 us per step measured: the default config (128 us of GPU work) is GPU-bound."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cdv_slam_amd import synth
from cdv_slam_amd.update import UpdatePath
dev = torch.device("cuda:0")
st = synth.make_state("tiny")
up = UpdatePath(st, dev)
for _ in range(20): up.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(500): up.step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("tiny: host enqueue %.1f us/step, total %.1f us/step" % ((t1 - t0) / 500 * 1e6, (t2 - t0) / 500 * 1e6))
