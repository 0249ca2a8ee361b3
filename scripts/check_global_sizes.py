"""The one-launch factorisation against the per-column launches of rounds 1-3 (CDV_BA_BLOCK_STEPS=1, a second process) at sizes the
test suite does not reach: same poses to rounding, and the time of both.   python scripts/check_global_sizes.py [frames] [M]"""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 600
M = int(sys.argv[2]) if len(sys.argv) > 2 else 32
out = sys.argv[3] if len(sys.argv) > 3 else None
from cdv_slam_amd import synth, ops
dev = torch.device("cuda:0")
st = synth.make_state("global", features=False, frames=frames, M=M, buffer_size=frames + 16, ht=384, wd=512)
T = lambda a: torch.as_tensor(a, device=dev)
poses0, patches0 = T(st.poses).float(), T(st.patches).float()
args = (T(st.intrinsics).float(), T(st.target).float(), T(st.weight).float(), torch.tensor([st.lmbda], device=dev),
        T(st.ii), T(st.jj), T(st.kk), st.cfg.M, st.t0, st.n, 2, True)
g = ops.GraphIndex(dev, E_cap=st.E, k_range=(frames + 16) * M)
U = len(np.unique(st.kk))
ts = []
for it in range(5):
    poses, patches = poses0.clone(), patches0.clone()
    torch.cuda.synchronize()
    t = time.perf_counter()
    ops.ba_forward(poses, patches, *args, U_max=U, graph=g)
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t)
assert ops.ba_status(dev) == (0, 0, 0, 0), ops.ba_status(dev)
p = poses.cpu().numpy()
assert np.isfinite(p).all()
mode = "per-column launches" if os.environ.get("CDV_BA_BLOCK_STEPS") == "1" else "one launch"
print("N = %d free poses (%d block columns), E = %d, %s: BA(2) %.2f ms" % (st.n - st.t0, (6 * (st.n - st.t0) + 63) // 64, st.E, mode, np.median(ts[1:]) * 1e3))
if out:
    np.save(out, p)
else:
    tmp = "/tmp/cdv_steps_%d.npy" % os.getpid()
    env = dict(os.environ, CDV_BA_BLOCK_STEPS="1")
    r = subprocess.run([sys.executable, os.path.abspath(__file__), str(frames), str(M), tmp], env=env, capture_output=True, text=True)
    print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-500:])
    q = np.load(tmp)
    os.remove(tmp)
    d = np.abs(p - q).max()
    print("largest difference between the two in any pose entry: %.3e (pose update itself up to %.3e)" % (d, np.abs(p - st.poses).max()))
    assert d < 2e-4, d
