"""Diagnostic: where the cycles go inside the big kernels (needs `make -C cdv_slam_amd/csrc STAMPS=1`).
Run on the GPU box:  CDV_LIB=cdv_slam_amd/libcdvslam_hip_stamps.so python scripts/stamps.py"""
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
nslot = max(st.E, 8192)
buf_c = torch.zeros((nslot, 16), dtype=torch.int64, device=dev)
buf_b = torch.zeros((8192, 16), dtype=torch.int64, device=dev)
for name, b in (("corr", buf_c), ("ba", buf_b)):
    fn = getattr(lib, "cdv_set_stamps_" + name)
    fn.argtypes = [ctypes.c_void_p]
    assert fn(ctypes.c_void_p(b.data_ptr())) == 0
up.step()
torch.cuda.synchronize()
c = buf_c.cpu().numpy()[: st.E].astype(np.float64)
d = np.diff(c[:, :9], axis=1)
names = ["idx+coords+box", "issue loads", "L0 wait+mfma+store", "blend0", "L1 mfma+store", "blend1", "stage out", "store"]
print("corr: per-wave cycles (median / mean) per phase, total %.0f" % np.median(c[:, 8] - c[:, 0]))
for i, nme in enumerate(names):
    print("  %-22s %8.0f %8.0f" % (nme, np.median(d[:, i]), d[:, i].mean()))
print("  kernel span (first start -> last end): %.0f cycles" % (c[:, 8].max() - c[:, 0].min()))
b = buf_b.cpu().numpy().astype(np.float64)
asm = b[:4096]
asm = asm[asm[:, 0] > 0]
print("assemble: waves %d" % len(asm))
for i, nme in enumerate(["(unused)", "slot"]):
    print("  %-22s %8.0f" % (nme, np.median(asm[:, i + 1] - asm[:, i])))
pass
print("  assemble wave total %.0f" % np.median(asm[:, 2] - asm[:, 0]))
rt = asm[:, 8:10] / 100.0  # s_memrealtime is 100 MHz -> microseconds
print("  assemble realtime: first start 0, last start %.1f us, first end %.1f, last end %.1f; median wave span %.1f us" % (rt[:, 0].max() - rt[:, 0].min(), rt[:, 1].min() - rt[:, 0].min(), rt[:, 1].max() - rt[:, 0].min(), np.median(rt[:, 1] - rt[:, 0])))
sol = b[4096:4097]   # wave 0 (the single-wave solver retires waves 1-3 after the load phase)
for i, nme in enumerate(["load replicas", "factor loop", "back-subst", "write+retr"]):
    print("solve %-22s %8.0f" % (nme, np.median(sol[:, i + 1] - sol[:, i])))
print("solve panel(sum) %8.0f trailing(sum) %8.0f" % (np.median(sol[:, 5]), np.median(sol[:, 6])))
span = rt[:, 1] - rt[:, 0]
order = np.argsort(-span)[:12]
ids = np.nonzero(b[:4096, 0] > 0)[0]
print("slowest assemble waves (slot -> chunk, sg, wave : span us):")
for o in order:
    sl = ids[o]
    print("   ", sl, "->", sl // 8 // 4, (sl // 8) % 4, sl % 8, ": %.1f" % span[o])
print("span percentiles", np.percentile(span, [50, 90, 99, 100]))
cyc = asm[:, 2] - asm[:, 0]
print("cycles of the slowest:", cyc[order][:6], " realtime start offsets:", (rt[order, 0] - rt[:, 0].min())[:6])
# histogram of spans by slot group
sgs = (ids // 8) % 4
for g in range(4):
    print("  sg", g, "median span %.1f max %.1f" % (np.median(span[sgs == g]), span[sgs == g].max()))
chs = ids // 32
slow = np.unique(chs[span > 10])
print("chunks with slow waves:", slow)

ph = np.stack([asm[:, 3] - asm[:, 0], asm[:, 4] - asm[:, 3], asm[:, 5] - asm[:, 4], asm[:, 2] - asm[:, 5]], 1)
print("slow waves phases [edge+E atomics, X write, gram+emit, tail] and passes:")
for o in order[:6]:
    print("   ", ph[o], asm[o, 6], " gram: read %d mfma %d emit %d" % (asm[o, 10], asm[o, 11], asm[o, 12]))
fast = np.argsort(span)[len(span) // 2]
print("median wave phases:", ph[fast], asm[fast, 6])
sch = b[5000:5000 + 4 * 64]
sch = sch[sch[:, 0] > 0]
srt = sch[:, 8:10] / 100.0
print("schur: waves %d; load+sync %0.f cycles, mfma+emit %.0f (median), max %.0f / %.0f" % (len(sch), np.median(sch[:, 1] - sch[:, 0]), np.median(sch[:, 2] - sch[:, 1]), (sch[:, 1] - sch[:, 0]).max(), (sch[:, 2] - sch[:, 1]).max()))
print("schur realtime: last start %.1f us, first end %.1f, last end %.1f" % (srt[:, 0].max() - srt[:, 0].min(), srt[:, 1].min() - srt[:, 0].min(), srt[:, 1].max() - srt[:, 0].min()))
print("assemble phase percentiles (50/90/99/max):")
for i, nme in enumerate(["edge+E atomics", "X write", "gram+emit", "tail"]):
    print("   %-16s" % nme, np.percentile(ph[:, i], [50, 90, 99, 100]))
