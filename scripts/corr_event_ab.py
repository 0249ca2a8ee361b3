"""in-stream event timing of the fused correlation (what bench.py reports as roofline.avg_launch_ms), for A/B runs"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from cdv_slam_amd import synth
from cdv_slam_amd.update import UpdatePath
st = synth.make_state("default", buffer_size=64, seed=1234)
up = UpdatePath(st, torch.device("cuda:0"))
t = time.perf_counter()
while time.perf_counter() - t < 0.5:
    for _ in range(50): up.step()
    torch.cuda.synchronize()

def pairs(nfront, label):
    ps = []
    for _ in range(50):
        for _ in range(nfront): up.step()
        c = up.last_coords
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); up.corr_only(c); e1.record()
        ps.append((e0, e1))
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ps)
    print("%-40s median %.2f min %.2f p90 %.2f us" % (label, 1e3*ms[25], 1e3*ms[0], 1e3*ms[45]))

pairs(1, "1 step in front")
pairs(2, "2 steps in front")
for w in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): up.step()
    torch.cuda.synchronize(); print("window %.4f ms/step" % (1e3 * (time.perf_counter() - t0) / 200))
pairs(1, "after 3 windows, 1 step in front")
pairs(2, "after 3 windows, 2 steps in front")
x = torch.tensor([1.0], dtype=torch.float64, device="cuda:0"); x.item()
pairs(1, "after a tensor round trip")
from cdv_slam_amd.replicas import ReplicaGroup
grp = ReplicaGroup(backend="nccl", device=torch.device("cuda:0"))
pairs(1, "with a ReplicaGroup alive")
grp.barrier(); grp.max_over_ranks(1.0)
pairs(1, "after barrier + max_over_ranks")
