"""Host-side profile of the reference-shaped call sequence (DropinPath.step): where the Python time of one update goes.
    python scripts/profile_dropin_host.py [config] [steps]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cdv_slam_amd import synth
from cdv_slam_amd.update import DropinPath

cfg = sys.argv[1] if len(sys.argv) > 1 else "default"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
dev = torch.device("cuda:0")
st = synth.make_state(cfg)
dp = DropinPath(st, dev)
for _ in range(30):
    dp.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    dp.step()
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("per step: host issue %.1f us, with the device drained %.1f us" % (1e6 * t_host / steps, 1e6 * t_all / steps))
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    dp.step()
pr.disable()
torch.cuda.synchronize()
ps = pstats.Stats(pr)
ps.sort_stats("tottime")
rows = []
for (fn, line, name), (cc, nc, tt, ct, callers) in ps.stats.items():
    rows.append((tt, ct, nc, "%s:%d %s" % (os.path.basename(fn), line, name)))
rows.sort(reverse=True)
print("%-9s %-9s %-7s %s   (us per step; under the profiler everything is ~2x slower)" % ("self", "cumul", "calls", "where"))
for tt, ct, nc, w in rows[:45]:
    print("%-9.1f %-9.1f %-7.1f %s" % (1e6 * tt / steps, 1e6 * ct / steps, nc / steps, w))

# ---- the same step with a clock between the reference's calls (no profiler): which call costs what on the host
import collections
from cdv_slam_amd.update import _CorrLayer
acc = collections.OrderedDict()
def lap(name, t):
    now = time.perf_counter(); acc[name] = acc.get(name, 0.0) + now - t; return now
with torch.no_grad():
    for _ in range(steps):
        s = dp
        t = time.perf_counter()
        slot = (s.n - 1) % s.mem
        s.gmap_[(s.n - 1) % s.pmem] = s.new_tiles
        s.fmap1_[:, slot] = s.new_frame
        s.fmap2_[:, slot] = torch.nn.functional.avg_pool2d(s.new_frame[None], 4, 4)[0]
        t = lap("caller: ring writes + pool", t)
        s.append_again()
        t = lap("caller: torch.cat x3", t)
        coords = s.pops.transform(s.SE3(s.poses), s.patches, s.intrinsics, s.ii, s.jj, s.kk)
        t = lap("pops.transform", t)
        coords = coords.permute(0, 1, 4, 2, 3).contiguous()
        t = lap("caller: permute.contiguous", t)
        with torch.autocast("cuda", enabled=True):
            ii1 = s.kk % (s.M * s.pmem); jj1 = s.jj % s.mem; c1 = coords / 1; c4 = coords / 4
            t = lap("caller: % % / /", t)
            corr1 = _CorrLayer.apply(s.gmap, s.pyramid[0], c1, ii1, jj1, 3, 1, s.cuda_corr)
            t = lap("corr level 0 (apply)", t)
            corr2 = _CorrLayer.apply(s.gmap, s.pyramid[1], c4, ii1, jj1, 3, 1, s.cuda_corr)
            t = lap("corr level 1 (apply)", t)
            out = torch.stack([corr1, corr2], -1).view(1, len(ii1), -1)
            t = lap("stack + view", t)
            ix, jx = s.cuda_ba.neighbors(s.kk, s.jj)
            t = lap("cuda_ba.neighbors", t)
        t = lap("autocast exit", t)
        lmbda = torch.as_tensor([1e-4], device=s.dev)
        t = lap("caller: as_tensor(lmbda)", t)
        s.cuda_ba.forward(s.poses.data, s.patches, s.intrinsics, s.target, s.weight, lmbda, s.ii, s.jj, s.kk, s.M, s.t0, s.n, 2, False)
        t = lap("cuda_ba.forward", t)
torch.cuda.synchronize()
tot = sum(acc.values())
print("\nhost microseconds per step between the reference's calls (sum %.1f):" % (1e6 * tot / steps))
for k, v in acc.items():
    print("  %-32s %6.1f" % (k, 1e6 * v / steps))
