"""Experiment: does an XCD-affine processing order (all edges of a target frame on one XCD's L2) change the fused
correlation's time?  Order built on the host (not timed).  Workgroup b is assumed to run on XCD b % 8."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cdv_slam_amd import synth, ops
from cdv_slam_amd.update import UpdatePath

dev = torch.device("cuda:0")
st = synth.make_state(sys.argv[1] if len(sys.argv) > 1 else "default", buffer_size=64)
up = UpdatePath(st, dev)
out = up.step()
coords = out["coords"]
E = st.E
by_j = np.argsort(st.jj % st.cfg.mem, kind="stable")            # edges grouped by target slot
nblk = (E + 3) // 4
order = np.full(nblk * 4, -1, np.int64)
# XCD x gets the x-th eighth of the grouped list, dealt to its blocks b = 8 i + x
parts = np.array_split(by_j, 8)
for x, part in enumerate(parts):
    blocks = np.arange(x, nblk, 8)
    pos = (blocks[:, None] * 4 + np.arange(4)[None, :]).reshape(-1)
    pos = pos[pos < E]
    n = min(len(pos), len(part))
    order[pos[:n]] = part[:n]
left = np.setdiff1d(np.arange(E), order[order >= 0])
free = np.nonzero(order[:E] < 0)[0]
order[free[:len(left)]] = left
order = order[:E]
assert np.array_equal(np.sort(order), np.arange(E))
variants = {"natural": None, "by target slot": by_j.astype(np.int32), "xcd-affine": order.astype(np.int32)}

def timed(optr, reps=50):
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.corr_fused(up.gmap_pm, up.fmap1, up.fmap2, coords, up.kk, up.jj, kmod=up.kmod, jmod=up.jmod, out=up.corr_out,
                       pixel_major=True, order_ptr=optr)
        e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return np.median(ts)

ref = ops.corr_fused(up.gmap_pm, up.fmap1, up.fmap2, coords, up.kk, up.jj, kmod=up.kmod, jmod=up.jmod, pixel_major=True).clone()
for name, o in variants.items():
    t = None if o is None else torch.as_tensor(o, device=dev)
    optr = None if t is None else ctypes.c_void_p(t.data_ptr())
    got = ops.corr_fused(up.gmap_pm, up.fmap1, up.fmap2, coords, up.kk, up.jj, kmod=up.kmod, jmod=up.jmod, pixel_major=True,
                         order_ptr=optr)
    assert torch.equal(got, ref)
    print("%-16s %.1f us" % (name, timed(optr)))
