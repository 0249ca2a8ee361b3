"""Diagnostic: where the cycles go inside the 10 < N <= 32 BA kernels (needs `make -C cdv_slam_amd/csrc STAMPS=1`).
Run on the GPU box:  CDV_LIB=cdv_slam_amd/libcdvslam_hip_stamps.so python scripts/stamps_bam.py [config] [iterations]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cdv_slam_amd import synth, _lib, ops
from cdv_slam_amd.update import UpdatePath

lib = _lib.load()
dev = torch.device("cuda:0")
cfg = sys.argv[1] if len(sys.argv) > 1 else "stress"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 1     # 2: the stamps show the SECOND iteration (caches warm)
st = synth.make_state(cfg)
up = UpdatePath(st, dev)
for _ in range(5):
    up.step()
torch.cuda.synchronize()
buf = torch.zeros((8192, 16), dtype=torch.int64, device=dev)
fn = lib.cdv_set_stamps_bam
fn.argtypes = [ctypes.c_void_p]
assert fn(ctypes.c_void_p(buf.data_ptr())) == 0
up.step(iterations=0)
torch.cuda.synchronize()
buf.zero_()
ops.ba_forward(up.poses, up.patches, up.intrinsics, up.target, up.weight, up.lmbda, up.ii, up.jj, up.kk, up.M, up.t0, up.n,
               iters, False, U_max=up.U_max, graph=up.graph)
torch.cuda.synchronize()
b = buf.cpu().numpy().astype(np.float64)
ck = b[:4000]
ck = ck[ck[:, 0] > 0]
print("iteration %d of %d after a full prologue + correlation, N = %d free poses" % (iters, iters, up.n - up.t0))
print("chunk kernel: %d waves stamped" % len(ck))
for (i0, i1), nme in zip(((0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (5, 6)),
                         ["loads issue + zero + mask + barrier", "rounds (factor, E, products)", "partials + combine + fold",
                          "E store + schur tiles -> LDS", "slab copy-out", "tail (last barrier)"]):
    d = ck[:, i1] - ck[:, i0]
    print("  %-36s median %7.0f  p90 %7.0f  max %7.0f cycles" % (nme, np.median(d), np.percentile(d, 90), d.max()))
for (i0, i1), nme in zip(((0, 13), (13, 1), (1, 12), (12, 2), (2, 9), (9, 10), (10, 11), (11, 3), (3, 7), (7, 8), (8, 4)),
                         ["zero + records of both chunks + mask", "barrier", "chunk 0: loads + rounds", "chunk 1: loads + rounds", "partials -> barrier",
                          "copies summed + E_i, C, u", "barrier", "fold + barrier", "tile loop", "barrier", "B part scatter"]):
    d = ck[:, i1] - ck[:, i0]
    print("    %-34s median %7.0f  p90 %7.0f  max %7.0f cycles" % (nme, np.median(d), np.percentile(d, 90), d.max()))
tot = ck[:, 6] - ck[:, 0]
print("  wave total median %.0f max %.0f cycles" % (np.median(tot), tot.max()))
rt = ck[:, 14:16] / 100.0
t00 = rt[:, 0].min()
print("  realtime: last start %.2f us, first end %.2f, last end %.2f; median wave span %.2f us" %
      (rt[:, 0].max() - t00, rt[:, 1].min() - t00, rt[:, 1].max() - t00, np.median(rt[:, 1] - rt[:, 0])))
sol = b[4000]
print("finish kernel, solver workgroup (wave 0):")
for i, nme in zip((0, 2, 3, 4), ["load + unpack", "factor", "back substitution", "publish"]):
    print("  %-24s %8.0f cycles" % (nme, sol[i + 1 if i else 2] - sol[i]))
print("    factor: panel steps %.0f, trailing updates %.0f cycles" % (sol[8], sol[9]))
print("      panel (wave 0): row loads %.0f, columns %.0f, through the stores %.0f (from the step's start)" % (sol[10], sol[11], sol[12]))
for w in range(1, 8):
    sw = b[4000 + w]
    print("      wave %d: panel %.0f (loads %.0f, columns %.0f) trailing %.0f" % (w, sw[8], sw[10], sw[11], sw[9]))
srt = sol[14:16] / 100.0
red = b[4100:4100 + 8 * 80]
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
