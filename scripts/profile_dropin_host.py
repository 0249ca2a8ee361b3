"""Host-side cost of the reference-shaped call sequence (update.DropinPath): enqueue time per step and a cProfile of it.
    python scripts/profile_dropin_host.py [config] [steps]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cdv_slam_amd import synth
from cdv_slam_amd.update import DropinPath

cfg = sys.argv[1] if len(sys.argv) > 1 else "default"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dev = torch.device("cuda:0")
st = synth.make_state(cfg, buffer_size=64, seed=1234)
dp = DropinPath(st, dev)
t_settle = time.perf_counter()
while time.perf_counter() - t_settle < 0.5:      # clocks settle (as bench.py does before its timed regions)
    dp.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    dp.step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("dropin %s: host enqueue %.1f us/step, total %.1f us/step" % (cfg, (t1 - t0) / steps * 1e6, (t2 - t0) / steps * 1e6))
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    dp.step()
pr.disable()
torch.cuda.synchronize()
ps = pstats.Stats(pr)
ps.sort_stats("tottime").print_stats(45)
