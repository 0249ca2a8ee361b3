"""Workload for the HBM-traffic PMC passes (run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE`):
N full update steps (each contains one corr_fused launch) and N planar->channels-last conversions of a whole
feature ring, a kernel of exactly known byte count used to check the FETCH_SIZE correction.
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dir> -- python3 scripts/profile_traffic.py [config]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cdv_slam_amd import synth, ops, _lib
from cdv_slam_amd.update import UpdatePath

cfg = sys.argv[1] if len(sys.argv) > 1 else "default"
dev = torch.device("cuda:0")
st = synth.make_state(cfg, buffer_size=64, seed=1234)
up = UpdatePath(st, dev)
planar = torch.as_tensor(st.fmap1, device=dev)
shadow = torch.zeros_like(up.fmap1)
lib = _lib.load()
N, C, H, W = planar.shape[-4:]
for _ in range(12):
    up.step()
    _lib.check(lib.cdv_fmap_to_nhwc(ops._p(planar), ops._p(shadow), N, C, H, W, 0, N, ops._stream()), "cdv_fmap_to_nhwc")
torch.cuda.synchronize()
print("known bytes of one ring conversion: read %d write %d" % (planar.numel() * 2, planar.numel() * 2))
