"""Time the fused correlation (alone, back to back) and the whole update step for the library build CDV_LIB selects
(experiment builds: make -C cdv_slam_amd/csrc VARIANT=name VFLAGS=...), and print a digest of the correlation output so
that variants can be compared with the product build:   CDV_LIB=.../libcdvslam_hip_name.so python scripts/exp_variants.py"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cdv_slam_amd import synth
from cdv_slam_amd.update import UpdatePath

dev = torch.device("cuda:0")
cfgs = sys.argv[1:] or ["default"]
for cfg in cfgs:
    kw = {"C": int(os.environ["CDV_EXP_C"])} if os.environ.get("CDV_EXP_C") else {}     # experiment: another feature width
    st = synth.make_state(cfg, **kw)
    up = UpdatePath(st, dev)
    coords = up.step()["coords"]
    out = up.corr_only(coords)
    torch.cuda.synchronize()
    digest = hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:12]
    ts = []
    for _ in range(7):   # 100 launches back to back per sample
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            up.corr_only(coords)
        e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 10.0)
    # whole step, 100 back to back
    for _ in range(20):
        up.step()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        up.step()
    e1.record(); e1.synchronize()
    print("%-28s %-8s corr us median %.1f min %.1f | step %.1f us | digest %s" % (
        os.path.basename(os.environ.get("CDV_LIB", "product")), cfg, np.median(ts), np.min(ts),
        e0.elapsed_time(e1) * 1e3 / 200, digest), flush=True)
