"""corr_fused (separate coords / index tensors, as the drop-in's first call of a pair launches it) with and without the
table's processing order; and the level-1 check launch"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cdv_slam_amd import synth, ops
from cdv_slam_amd.update import UpdatePath
dev = torch.device("cuda:0")
st = synth.make_state(sys.argv[1] if len(sys.argv) > 1 else "default")
up = UpdatePath(st, dev)
coords = up.step()["coords"]
ii1 = torch.as_tensor(st.ii1, device=dev); jj1 = torch.as_tensor(st.jj1, device=dev)
buf = torch.empty((1, st.E, 7, 7, 3, 3, 2), dtype=torch.float16, device=dev)
def t(fn, n=200):
    for _ in range(20): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for name, op in (("no order", None), ("table order", up.graph.corr_order_ptr())):
    f = lambda: ops.corr_fused(up.gmap_pm, up.fmap1, up.fmap2, coords, ii1, jj1, scales=(1.0, 4.0), out=buf, pixel_major=True, order_ptr=op)
    f(); torch.cuda.synchronize()
    ref = buf.clone()
    print("%-12s %.1f us back to back" % (name, t(f)), "digest", float(ref.float().abs().sum()))
