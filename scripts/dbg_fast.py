import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cdv_slam_amd import synth, ops
from cdv_slam_amd.update import DropinPath
dev = torch.device("cuda:0")
st = synth.make_state("default")
dp = DropinPath(st, dev)
g = ops._device_graph(dev)
for it in range(3):
    dp.step()
    print(it, "armed graph", ops._armed_graph is not None, "armed pair", ops._armed_pair is not None, "builds", g.n_builds, "fused", ops._pairing.n_fused)
ag = ops._armed_graph
r = ops._fast.neighbors(ag[0], dp.kk, dp.jj, ops._stream())
print("neighbors direct:", None if r is None else (type(r), r[2] if not isinstance(r, int) else r))
print(dp.kk.dtype, dp.kk.is_contiguous(), dp.kk.numel(), g.E_cap, g.events.seen, g.events.counts())
dp.step(ingest=False)
print("after same-tensor step: builds", g.n_builds, "armed", ops._armed_graph is not None)
