"""Diagnostic: where the cycles go along the chain of diagonal blocks of the one-launch factorisation (ba_factor.hip; needs
`make -C cdv_slam_amd/csrc STAMPS=1`).  Run on the GPU box:
    CDV_LIB=cdv_slam_amd/libcdvslam_hip_stamps.so python scripts/stamps_baf.py [frames] [M]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cdv_slam_amd import synth, _lib, ops

lib = _lib.load()
dev = torch.device("cuda:0")
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 300
M = int(sys.argv[2]) if len(sys.argv) > 2 else 96
st = synth.make_state("global", features=False, frames=frames, M=M, buffer_size=frames + 16, ht=384, wd=512)
T = lambda a: torch.as_tensor(a, device=dev)
poses0, patches0 = T(st.poses).float(), T(st.patches).float()
args = (T(st.intrinsics).float(), T(st.target).float(), T(st.weight).float(), torch.tensor([st.lmbda], device=dev),
        T(st.ii), T(st.jj), T(st.kk), st.cfg.M, st.t0, st.n, 1, True)
g = ops.GraphIndex(dev, E_cap=st.E, k_range=(frames + 16) * M)
U = len(np.unique(st.kk))
for _ in range(3):
    ops.ba_forward(poses0.clone(), patches0.clone(), *args, U_max=U, graph=g)
torch.cuda.synchronize()
buf = torch.zeros((1024, 16), dtype=torch.int64, device=dev)
fn = lib.cdv_set_stamps_baf
fn.argtypes = [ctypes.c_void_p]
assert fn(ctypes.c_void_p(buf.data_ptr())) == 0
ops.ba_forward(poses0.clone(), patches0.clone(), *args, U_max=U, graph=g)
torch.cuda.synchronize()
b = buf.cpu().numpy().astype(np.float64)
nb = (6 * (st.n - st.t0) + 63) // 64
print("one-launch factorisation, N = %d free poses, %d block columns; cycles (medians over the stages)" % (st.n - st.t0, nb))
d = b[1:nb - 1]
print("wave 0 of the chain workgroup, stage c = 1 .. %d:" % (nb - 2))
for i0, i1, nme in [(0, 1, "factor the diagonal block (columns -> LDS as they come)"), (1, 2, "L(c, c) row-major -> LDS, post, pivot check")]:
    x = d[:, i1] - d[:, i0]
    print("  %-56s median %7.0f  min %7.0f  max %7.0f" % (nme, np.median(x), x.min(), x.max()))
nxt = b[2:nb, 0] - b[1:nb - 1, 1]
print("  %-56s median %7.0f  min %7.0f  max %7.0f" % ("end of the factorisation -> next diagonal block complete", np.median(nxt), nxt.min(), nxt.max()))
d3 = b[32 + 1:32 + nb - 1]
print("wave 1 (solves 32 rows of the neighbour behind the factorisation):")
for i0, i1, nme in [(0, 3, "wait for the neighbour (helpers)"), (3, 4, "solve, arrive")]:
    x = d3[:, i1] - d3[:, i0]
    print("  %-56s median %7.0f  min %7.0f  max %7.0f" % (nme, np.median(x), x.min(), x.max()))
lag = d3[:, 4] - d[:, 1]
print("  %-56s median %7.0f  min %7.0f  max %7.0f" % ("solve done after the factorisation's end", np.median(lag), lag.min(), lag.max()))
late = d3[:, 3] - d[:, 0]
print("  %-56s median %7.0f  min %7.0f  max %7.0f" % ("solve starts after the factorisation's start", np.median(late), late.min(), late.max()))
rt = b[0:nb, 14] / 100.0
step = np.diff(rt)
print("  stage to stage: median %.2f us (min %.2f, max %.2f); %.1f us from stage 0 to stage %d" % (np.median(step), step.min(), step.max(), rt[-1] - rt[0], nb - 1))
hh = b[64 + 1:64 + nb - 1]
print("helper 3 (wave 7), stage c = 1 .. %d:" % (nb - 2))
for i0, i1, nme in [(0, 1, "P flag + the pre-accumulated tiles"), (1, 2, "wait for L(c + 1, c - 1)"), (2, 3, "fetch its 16 rows -> LDS"),
                    (3, 4, "its product off the neighbour -> LDS"), (4, 5, "all rows in; its product off the diagonal block"),
                    (5, 6, "wait for L(c + 1, c)"), (6, 7, "its product off the diagonal block -> LDS")]:
    x = hh[:, i1] - hh[:, i0]
    print("  %-46s median %7.0f  min %7.0f  max %7.0f" % (nme, np.median(x), x.min(), x.max()))
o = b[192:192 + nb - 2]
o = o[(o[:, 0] > 0) & (o[:, 10] > 0)]
if len(o):
    print("item (c + 2, c): wait for L(c, c) %.0f, fetch %.0f, solve %.0f, out + drain + flag %.0f cycles (medians)" %
          (np.median(o[:, 3] - o[:, 2]), np.median(o[:, 4] - o[:, 3]), np.median(o[:, 6] - o[:, 5]), np.median(o[:, 10] - o[:, 6])))
