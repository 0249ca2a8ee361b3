"""time only the fused correlation kernel (GPU); CDV_CORR_EXP selects diagnostic variants"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cdv_slam_amd import synth
from cdv_slam_amd.update import UpdatePath
dev = torch.device("cuda:0")
st = synth.make_state(sys.argv[1] if len(sys.argv) > 1 else "default")
up = UpdatePath(st, dev)
coords = up.step()["coords"]
ts = []
for _ in range(40):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); up.corr_only(coords); e1.record(); e1.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
print("EXP", os.environ.get("CDV_CORR_EXP", "0"), "SORT", os.environ.get("CDV_SORT", "1"), "corr us median %.1f min %.1f" % (np.median(ts), np.min(ts)))
