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
for rep in range(3):
    pairs = []
    t0 = time.perf_counter()
    for _ in range(50):
        up.step()
        c = up.last_coords
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); up.corr_only(c); e1.record()
        pairs.append((e0, e1))
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    tt = time.perf_counter() - t0
    ms = sorted(a.elapsed_time(b) for a, b in pairs)
    print("corr event us: median %.2f min %.2f p90 %.2f | host enqueue %.1f us/iter, total %.1f us/iter" % (1e3*ms[25], 1e3*ms[0], 1e3*ms[45], 1e6*th/50, 1e6*tt/50))
