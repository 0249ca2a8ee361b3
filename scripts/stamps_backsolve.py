import ctypes, os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from cdv_slam_amd import synth, _lib, ops
lib = _lib.load(); dev = torch.device("cuda:0")
st = synth.make_state("global", features=False, frames=300, M=96, buffer_size=316, ht=384, wd=512)
T = lambda a: torch.as_tensor(a, device=dev)
poses0, patches0 = T(st.poses).float(), T(st.patches).float()
args = (T(st.intrinsics).float(), T(st.target).float(), T(st.weight).float(), torch.tensor([st.lmbda], device=dev), T(st.ii), T(st.jj), T(st.kk), st.cfg.M, st.t0, st.n, 1, True)
g = ops.GraphIndex(dev, E_cap=st.E, k_range=316 * 96); U = len(np.unique(st.kk))
for _ in range(3): ops.ba_forward(poses0.clone(), patches0.clone(), *args, U_max=U, graph=g)
torch.cuda.synchronize()
buf = torch.zeros((4096, 16), dtype=torch.int64, device=dev)
fn = lib.cdv_set_stamps_ba; fn.argtypes = [ctypes.c_void_p]; assert fn(ctypes.c_void_p(buf.data_ptr())) == 0
ops.ba_forward(poses0.clone(), patches0.clone(), *args, U_max=U, graph=g); torch.cuda.synchronize()
b = buf.cpu().numpy().astype(np.float64)
nb = 29
for wg in (0, 3, 7):
    rows = b[2000 + 64 * wg: 2000 + 64 * wg + nb]; rows = rows[rows[:, 5] > 0]
    if not len(rows): continue
    print("workgroup %d: %d steps" % (wg, len(rows)))
    for i0, i1, nme in [(0, 1, "row loads issued + barrier"), (1, 2, "solve (owner) / poll (others)"), (2, 3, "barrier"), (3, 4, "fold"), (4, 5, "barrier")]:
        x = rows[:, i1] - rows[:, i0]
        print("  %-32s median %7.0f min %7.0f max %7.0f" % (nme, np.median(x), x.min(), x.max()))
    o = np.sort(rows[:, 0]); print("  step to step median %.0f cycles" % np.median(np.diff(o)))
