"""Diagnostic: where the time goes inside the table build's sort launch (graph_tsort_kernel; needs `make STAMPS=1`).
    CDV_LIB=cdv_slam_amd/libcdvslam_hip_stamps.so python scripts/stamps_sort.py [config]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cdv_slam_amd import synth, _lib
from cdv_slam_amd.update import UpdatePath
lib = _lib.load()
dev = torch.device("cuda:0")
cfg = sys.argv[1] if len(sys.argv) > 1 else "default"
st = synth.make_state(cfg)
up = UpdatePath(st, dev)
for _ in range(5):
    up.step()
torch.cuda.synchronize()
buf = torch.zeros((8192, 16), dtype=torch.int64, device=dev)
fn = lib.cdv_set_stamps_graph; fn.argtypes = [ctypes.c_void_p]
assert fn(ctypes.c_void_p(buf.data_ptr())) == 0
up.step(); torch.cuda.synchronize()
buf.zero_()
up.step(); torch.cuda.synchronize()
b = buf.cpu().numpy().astype(np.float64)
npw = (up.graph.table_capacity + 7) // 8
live = b[:, 15] > 0
t0 = b[live][:, 0].min()
for name, sel in (("patch workgroups", np.arange(len(b)) < npw), ("edge workgroups", np.arange(len(b)) >= npw)):
    w = b[sel & live]
    if not len(w):
        continue
    print("%s: %d stamped" % (name, len(w)))
    print("  start  (us after the launch's first wave) median %.2f max %.2f" % (np.median(w[:, 0] - t0) / 100, (w[:, 0] - t0).max() / 100))
    print("  end                                       median %.2f max %.2f" % (np.median(w[:, 15] - t0) / 100, (w[:, 15] - t0).max() / 100))
    print("  cycles: start -> loads in hand / tables   median %.0f max %.0f" % (np.median(w[:, 2] - w[:, 1]), (w[:, 2] - w[:, 1]).max()))
    print("          -> end                            median %.0f max %.0f" % (np.median(w[:, 3] - w[:, 2]), (w[:, 3] - w[:, 2]).max()))
