"""Diagnostic: where the cycles go inside the window-BA kernels (needs `make -C cdv_slam_amd/csrc STAMPS=1`).
Run on the GPU box:  CDV_LIB=cdv_slam_amd/libcdvslam_hip_stamps.so python scripts/stamps_baw.py [config]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cdv_slam_amd import synth, _lib, ops
from cdv_slam_amd.update import UpdatePath

lib = _lib.load()
dev = torch.device("cuda:0")
cfg = sys.argv[1] if len(sys.argv) > 1 else "default"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 1     # 2: the stamps show the SECOND iteration (caches warm)
st = synth.make_state(cfg)
up = UpdatePath(st, dev)
for _ in range(5):
    up.step()
torch.cuda.synchronize()
buf = torch.zeros((8192, 16), dtype=torch.int64, device=dev)
fn = lib.cdv_set_stamps_baw
fn.argtypes = [ctypes.c_void_p]
assert fn(ctypes.c_void_p(buf.data_ptr())) == 0
# ONE iteration of BA alone, after a full step (so that caches look like the real sequence)
up.step(iterations=0)
torch.cuda.synchronize()
buf.zero_()
ops.ba_forward(up.poses, up.patches, up.intrinsics, up.target, up.weight, up.lmbda, up.ii, up.jj, up.kk, up.M, up.t0, up.n,
               iters, False, U_max=up.U_max, graph=up.graph)
torch.cuda.synchronize()
b = buf.cpu().numpy().astype(np.float64)
ck = b[:4000]
ck = ck[ck[:, 0] > 0]
print("iteration %d of %d after a full prologue + correlation" % (iters, iters))
print("chunk kernel: %d waves stamped" % len(ck))
names = ["loads L1-L3 issue + zero + barrier", "rounds (factor, E, gram)", "partials + finalize", "Edg store", "schur + slab"]
for i, nme in enumerate(names):
    d = ck[:, i + 1] - ck[:, i]
    print("  %-36s median %7.0f  p90 %7.0f  max %7.0f cycles" % (nme, np.median(d), np.percentile(d, 90), d.max()))
tot = ck[:, 6] - ck[:, 0]
print("  wave total median %.0f max %.0f cycles" % (np.median(tot), tot.max()))
for i, nme in zip(range(8, 14), ["factor (incl. load wait)", "E atomics", "X write", "operand read", "mfma", "emit"]):
    v = ck[:, i]
    print("    rounds: %-26s median %7.0f  p90 %7.0f max %7.0f" % (nme, np.median(v), np.percentile(v, 90), v.max()))
rt = ck[:, 14:16] / 100.0
t00 = rt[:, 0].min()
print("  realtime: last start %.2f us, first end %.2f, last end %.2f; median wave span %.2f us" %
      (rt[:, 0].max() - t00, rt[:, 1].min() - t00, rt[:, 1].max() - t00, np.median(rt[:, 1] - rt[:, 0])))
sol = b[4000]
print("finish kernel, solver wave:")
for i, nme in enumerate(["wait for the reduce", "row loads", "factor", "back substitution", "publish"]):
    print("  %-24s %8.0f cycles" % (nme, sol[i + 1] - sol[i]))
srt = sol[14:16] / 100.0
red = b[4100:4100 + 4 * 128]
red = red[red[:, 0] > 0]
print("finish kernel, %d reduce/retract waves:" % len(red))
for i, nme in enumerate(["reduce + publish", "preload", "wait for dX", "retract"]):
    d = red[:, i + 1] - red[:, i]
    print("  %-24s median %7.0f  max %7.0f cycles" % (nme, np.median(d), d.max()))
rrt = red[:, 14:16] / 100.0
t0 = min(srt[0], rrt[:, 0].min())
print("  realtime (us from the first wave's start): solver start %.2f end %.2f; reducers start %.2f..%.2f end %.2f..%.2f" %
      (srt[0] - t0, srt[1] - t0, rrt[:, 0].min() - t0, rrt[:, 0].max() - t0, rrt[:, 1].min() - t0, rrt[:, 1].max() - t0))
print("  chunk kernel end -> finish kernel start: %.2f us" % (t0 - rt[:, 1].max()))
