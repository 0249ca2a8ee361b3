"""Time the global bundle adjustment (N > 32 free poses, slam.py:460-478) on a synthetic keyframe graph.
    python scripts/bench_global_ba.py [frames] [M]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cdv_slam_amd import synth, ops

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 300
M = int(sys.argv[2]) if len(sys.argv) > 2 else 96
dev = torch.device("cuda:0")
st = synth.make_state("global", features=False, frames=frames, M=M, buffer_size=frames + 16, ht=384, wd=512)
T = lambda a: torch.as_tensor(a, device=dev)
poses0, patches0 = T(st.poses).float(), T(st.patches).float()
args = (T(st.intrinsics).float(), T(st.target).float(), T(st.weight).float(), torch.tensor([st.lmbda], device=dev),
        T(st.ii), T(st.jj), T(st.kk), st.cfg.M, st.t0, st.n, 2, True)
g = ops.GraphIndex(dev, E_cap=st.E, k_range=(frames + 16) * M)
U = len(np.unique(st.kk))
print("global BA: frames %d, M %d, E %d, U %d, N %d free poses (6N = %d)" % (frames, M, st.E, U, st.n - st.t0, 6 * (st.n - st.t0)))
ts = []
for it in range(6):
    poses, patches = poses0.clone(), patches0.clone()
    torch.cuda.synchronize()
    t = time.perf_counter()
    ops.ba_forward(poses, patches, *args, U_max=U, graph=g)
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t)
print("fastba.BA(iterations=2, eff_impl=True): first call %.1f ms (workspace zeroing), then %.2f ms (median of %d)"
      % (ts[0] * 1e3, np.median(ts[1:]) * 1e3, len(ts) - 1))
print("pose update |dX| max %.3e" % float((poses - poses0).abs().max()))
