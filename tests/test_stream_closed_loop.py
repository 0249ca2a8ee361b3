"""Closed-loop parity over a frame stream (BASELINE.json: "ATE within 1e-4 of reference"; SURVEY.md 8(d): "closed-loop
synthetic trajectories", evaluate_tartan.py:63-70): the GPU stream runner and the oracle-driven runner (oracle/stream_py.py:
edges_py + orc_transform + orc_corr in the reference's half arithmetic + orc_fastba float32, the same operator stub) side by
side for >= 120 frames with keyframe drops.  Differences in the correlation (f32-accumulate MFMA against the reference's
half accumulation) feed back through the operator stub into targets, weights, poses and depths of every later frame.

Asserted: edge lists bit-identical at every frame (ii, jj, kk), the same keyframes, and the Sim(3)-aligned ATE-RMSE of the
camera centres between the two final trajectories <= 1e-4 scene units (the tolerance BASELINE.json states)."""
import numpy as np
import pytest
import torch

from cdv_slam_amd import metrics
from oracle.stream_py import StreamOracle, closed_loop
from tests.ba_checks import _log

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
CFG = dict(M=24, ht=192, wd=256, C=24, buffer_size=256)
ATE_TOL = 1e-4


def _pair(gain, **kw):
    from cdv_slam_amd.stream import StreamRunner
    run = StreamRunner(torch.device(DEV), gain=gain, **CFG, **kw)
    so = StreamOracle(gain=gain, **CFG, **kw)
    return run, so


@pytest.mark.parametrize("gain", [0.01, 0.25])
def test_closed_loop_stream_with_dropped_keyframes(gain):
    """every third frame the keyframe test drops frame n - 4 (the caller's decision, the same on both sides); gain = the
    operator stub's step in pixels (0.01: cdv_slam_amd/stream.py's default; 0.25: a stub the BA really follows)"""
    run, so = _pair(gain)
    res = closed_loop(run, so, frames=132, drop="pattern")
    assert res["edges_identical"], res.get("first_mismatch")
    assert res["frames"] == 132 and res["dropped"] >= 35 and res["keyframes"] >= 60
    ate = metrics.ate_rmse(res["poses_oracle"], res["poses_gpu"])
    moved = metrics.ate_rmse(res["poses_oracle"][:-1], res["poses_oracle"][1:])
    _log("closed_loop_pattern", "gain%g" % gain, {"ate": ate, "step_between_keyframes": moved, "frames": res["frames"],
                                                 "keyframes": res["keyframes"], "dropped": res["dropped"],
                                                 "t_maxdiff": np.abs(res["poses_oracle"][:, :3] - res["poses_gpu"][:, :3]).max()},
         {"ate": ATE_TOL})
    assert ate <= ATE_TOL, (ate, moved)
    assert np.abs(res["poses_oracle"][:, :3] - res["poses_gpu"][:, :3]).max() <= 10 * ATE_TOL
    rel = np.abs(res["patches_gpu"] - res["patches_oracle"]) / np.abs(res["patches_oracle"])
    assert np.median(rel) <= 1e-4


def test_closed_loop_stream_with_the_reference_keyframe_test():
    """the keyframe decision is the reference's own (slam.py:409-413): mean flow between the frames around n - 4 under
    KEYFRAME_THRESH, computed by each side from ITS state -- the decisions have to agree (the statistic is continuous in
    the poses: |difference| is reported and bounded), and with them the edge lists"""
    run, so = _pair(0.01, keyframe_thresh=2.5)
    res = closed_loop(run, so, frames=126, drop="flow")
    assert not res["decisions_differ"], res["decisions_differ"]
    assert res["motion_maxdiff"] <= 1e-3
    assert res["edges_identical"], res.get("first_mismatch")
    assert 10 <= res["dropped"] <= 110          # both outcomes of the test occur
    ate = metrics.ate_rmse(res["poses_oracle"], res["poses_gpu"])
    _log("closed_loop_flow", "gain0.01", {"ate": ate, "frames": res["frames"], "keyframes": res["keyframes"],
                                          "dropped": res["dropped"], "motion_maxdiff": res["motion_maxdiff"]}, {"ate": ATE_TOL})
    assert ate <= ATE_TOL
