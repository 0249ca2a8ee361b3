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


# ---- the same three runs with every size on the device (DeviceStreamRunner: no read-back inside a frame) ----------------------

def _pair_dev(gain, **kw):
    from cdv_slam_amd.stream import DeviceStreamRunner
    return DeviceStreamRunner(torch.device(DEV), gain=gain, **CFG, **kw), StreamOracle(gain=gain, **CFG, **kw)


@pytest.mark.parametrize("gain", [0.01, 0.25])
def test_device_stream_with_dropped_keyframes(gain):
    run, so = _pair_dev(gain)
    res = closed_loop(run, so, frames=132, drop="pattern")
    assert res["edges_identical"], res.get("first_mismatch")
    assert res["frames"] == 132 and res["dropped"] >= 35 and res["keyframes"] >= 60
    ate = metrics.ate_rmse(res["poses_oracle"], res["poses_gpu"])
    _log("device_stream_pattern", "gain%g" % gain, {"ate": ate, "frames": res["frames"], "keyframes": res["keyframes"],
                                                   "dropped": res["dropped"]}, {"ate": ATE_TOL})
    assert ate <= ATE_TOL
    rel = np.abs(res["patches_gpu"] - res["patches_oracle"]) / np.abs(res["patches_oracle"])
    assert np.median(rel) <= 1e-4
    # the inactive edges (removal-window pruning, slam.py:453-458) are the oracle's, in its order
    a = run.E_inac
    assert a == len(so.edges.ii_inac) and a > 0
    assert np.array_equal(run.ii_inac[:a].cpu().numpy(), so.edges.ii_inac) and np.array_equal(run.kk_inac[:a].cpu().numpy(), so.edges.kk_inac)
    assert np.array_equal(run.jj_inac[:a].cpu().numpy(), so.edges.jj_inac)
    # the point cloud of the patches inside the removal window (slam.py:524-526)
    n, M = res["keyframes"], so.M
    lo = max(n - so.rw, 0) * M
    got, want = run.points[lo:n * M].cpu().numpy(), so.points[lo:n * M]
    assert np.abs(got - want).max() <= 1e-3 * max(1.0, np.abs(want).max())


def test_device_stream_with_the_reference_keyframe_test():
    """the decision never leaves the device (slam.py:409-413 reads it back twice per frame): flow statistic -> compare ->
    conditional removal, index shift and buffer shift, all in the launches of cdv_stream_keyframe"""
    run, so = _pair_dev(0.01, keyframe_thresh=2.5)
    res = closed_loop(run, so, frames=126, drop="flow")
    assert not res["decisions_differ"], res["decisions_differ"]
    assert res["motion_maxdiff"] <= 1e-3
    assert res["edges_identical"], res.get("first_mismatch")
    assert 10 <= res["dropped"] <= 110
    ate = metrics.ate_rmse(res["poses_oracle"], res["poses_gpu"])
    _log("device_stream_flow", "gain0.01", {"ate": ate, "frames": res["frames"], "keyframes": res["keyframes"],
                                            "dropped": res["dropped"], "motion_maxdiff": res["motion_maxdiff"]}, {"ate": ATE_TOL})
    assert ate <= ATE_TOL


@pytest.mark.parametrize("drop", ["pattern", "flow"])
def test_device_stream_with_the_wide_optimisation_window(drop):
    """OPTIMIZATION_WINDOW 22 (default_cdvo++.yaml, BASELINE configs[4]) with every size on the device: the bundle adjustment
    runs on the 10 < N <= 32 path (ba_mid.hip) with its window read from the dynamic block -- 7 free poses at the stream's first
    update, growing to 22, inside launches laid out for 22.  Against the oracle-driven runner: edge lists bit-identical at
    every frame, the same keyframe decisions, ATE within BASELINE.json's tolerance, no failure event."""
    kw = dict(opt_window=22, keyframe_thresh=2.5) if drop == "flow" else dict(opt_window=22)
    run, so = _pair_dev(0.01, **kw)
    res = closed_loop(run, so, frames=110, drop=drop)
    if drop == "flow":
        assert not res["decisions_differ"], res["decisions_differ"]
        assert res["motion_maxdiff"] <= 1e-3
    assert res["edges_identical"], res.get("first_mismatch")
    assert res["frames"] == 110 and res["keyframes"] >= 40 and res["dropped"] >= 10
    ate = metrics.ate_rmse(res["poses_oracle"], res["poses_gpu"])
    _log("device_stream_window22", drop, {"ate": ate, "frames": res["frames"], "keyframes": res["keyframes"],
                                          "dropped": res["dropped"]}, {"ate": ATE_TOL})
    assert ate <= ATE_TOL
    rel = np.abs(res["patches_gpu"] - res["patches_oracle"]) / np.abs(res["patches_oracle"])
    assert np.median(rel) <= 1e-4
    assert run.events.counts() == [0, 0, 0, 0]


def test_device_stream_equals_the_host_sized_stream_and_does_not_synchronise():
    """DeviceStreamRunner against StreamRunner (one read-back per removal) on the benchmark frame size, same frames, same
    forced drops: edge lists and keyframe count identical, poses equal to rounding (the tiles' blend and the operator stub are
    other instantiations of the same arithmetic); and a frame of the device runner really enqueues without waiting: 40
    frames are issued in less host time than the device needs to run them"""
    import time
    from cdv_slam_amd.stream import DeviceStreamRunner, StreamRunner
    dev = torch.device(DEV)
    a, b = DeviceStreamRunner(dev, buffer_size=128), StreamRunner(dev, buffer_size=128)
    for f in range(70):
        inp = (a.pool[f % 4], a._draws[f, 0], a._draws[f, 1], a._draws[f, 2])
        a.frame(drop=(f % 3 == 2), inputs=inp)
        b.frame(drop=(f % 3 == 2), inputs=inp)
    n, E = a.counts()
    assert (n, E) == b.counts() and E > 30000
    ea, eb = a.edges, b.edges
    assert torch.equal(ea.ii, eb.ii) and torch.equal(ea.jj, eb.jj) and torch.equal(ea.kk, eb.kk)
    assert float((a.poses[:n] - b.poses[:n]).abs().max()) < 1e-5
    assert float((a.patches[:n * a.M] - b.patches[:n * a.M]).abs().max()) < 1e-4
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for f in range(70, 110):
        a.frame(drop=(f % 3 == 2))
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    assert t_enq < t_all and a.counts()[0] > n


def test_device_stream_two_frames_as_a_hipgraph():
    """With every size on the device a frame's launches do not depend on the host: two frames (24 launches: the edge lists are
    back in their first twin after two) captured as ONE hipGraph and replayed give the eager runner's bits -- edge lists,
    keyframe count, poses, depths -- with the reference's keyframe test deciding on the device, for AS LONG AS the graph is
    replayed: 90 pairs, over which the kept keyframes advance by more than the patch table's capacity (30 frames of ids), so
    that every slot of the table is met again by a new id under a replayed launch.  (Round 4 froze the table build's
    generation into the captured launches: from the first wrap on one of the two frames ran with its index in the error
    state and its BA skipped.)  No failure event may be counted on either runner."""
    from cdv_slam_amd.stream import DeviceStreamRunner
    dev = torch.device(DEV)
    kw = dict(buffer_size=256, pose_step=0.1)
    a, b = DeviceStreamRunner(dev, **kw), DeviceStreamRunner(dev, **kw)
    for f in range(20):
        a.frame(drop=None)
        b.frame(drop=None)
    while a.cur != 0:          # capture needs the lists in their first twin
        a.frame(drop=None)
        b.frame(drop=None)
    n_cap = a.counts()[0]
    replay = a.capture_pair()
    for pair in range(90):
        f = a.frames
        for k in range(2):
            dr = a._draws[(f + k) % a.N]
            a.stage_inputs(k, a.pool[(f + k) % 4], dr[0], dr[1], dr[2])
            b.frame(drop=None, inputs=(b.pool[(f + k) % 4], dr[0], dr[1], dr[2]))
        replay()
    n, E = a.counts()
    assert (n, E) == b.counts() and a.frames == b.frames and n < a.frames          # keyframes were dropped on the way
    assert n - n_cap > a.tcap // a.M, (n, n_cap)                                   # ... and the ids went round the table
    ea, eb = a.edges, b.edges
    assert torch.equal(ea.ii, eb.ii) and torch.equal(ea.jj, eb.jj) and torch.equal(ea.kk, eb.kk)
    assert torch.equal(a.poses[:n], b.poses[:n]) and torch.equal(a.patches[:n * a.M], b.patches[:n * a.M])
    assert a.E_inac == b.E_inac
    assert a.events.counts() == [0, 0, 0, 0] and b.events.counts() == [0, 0, 0, 0]


def test_device_stream_hipgraph_soak():
    """3,000 replays of ONE captured frame pair (6,000 frames; the reference's BUFFER_SIZE of 4096 keyframes) against the eager
    runner: the patch ids go round the table some sixty times, the patch and feature rings wrap a hundred times, the dynamic
    blocks alternate 3,000 times -- the same edge lists, keyframe count, poses and depths at the end, bit for bit, and not one
    failure event on either side."""
    from cdv_slam_amd.stream import DeviceStreamRunner
    dev = torch.device(DEV)
    kw = dict(buffer_size=4096, pose_step=0.1)
    a, b = DeviceStreamRunner(dev, **kw), DeviceStreamRunner(dev, **kw)
    for f in range(20):
        a.frame(drop=None)
        b.frame(drop=None)
    while a.cur != 0:
        a.frame(drop=None)
        b.frame(drop=None)
    replay = a.capture_pair()
    for pair in range(3000):
        f = a.frames
        for k in range(2):
            dr = a._draws[(f + k) % a.N]
            a.stage_inputs(k, a.pool[(f + k) % 4], dr[0], dr[1], dr[2])
            b.frame(drop=None, inputs=(b.pool[(f + k) % 4], dr[0], dr[1], dr[2]))
        replay()
    n, E = a.counts()
    assert (n, E) == b.counts() and a.frames == b.frames and a.frames > 6000 and n > 1500
    ea, eb = a.edges, b.edges
    assert torch.equal(ea.ii, eb.ii) and torch.equal(ea.jj, eb.jj) and torch.equal(ea.kk, eb.kk)
    assert torch.equal(a.poses[:n], b.poses[:n]) and torch.equal(a.patches[:n * a.M], b.patches[:n * a.M])
    assert a.E_inac == b.E_inac
    assert bool(torch.isfinite(a.poses[:n]).all())
    assert a.events.counts() == [0, 0, 0, 0] and b.events.counts() == [0, 0, 0, 0]


def test_device_stream_outlives_its_frame_buffer_when_keyframes_are_dropped():
    """what is bounded is the number of KEYFRAMES (slam.py bounds n, not the frames seen): a stream that drops two frames in
    three runs for several times buffer_size frames -- eagerly and as hipGraph replays alike -- and stops with the device's
    capacity word, not a host-side count of frames, once the keyframes really fill the buffers"""
    from cdv_slam_amd.stream import DeviceStreamRunner
    run = DeviceStreamRunner(torch.device(DEV), M=24, ht=192, wd=256, buffer_size=48)
    for f in range(12):
        run.frame(drop=False)
    for f in range(90):        # 102 frames through a 48-frame buffer: two in three are dropped, n ends at 12 + 30
        run.frame(drop=(f % 3 != 0))
    n, E = run.counts()
    assert run.frames == 102 and n == 42 and E > 0
    assert run.events.counts() == [0, 0, 0, 0]
    for f in range(60):        # now keep everything: the buffers fill up, the device says so
        run.frame(drop=False)
    with pytest.raises(RuntimeError, match="capacity exceeded"):
        run.counts()


def test_device_stream_reports_a_capacity_error_instead_of_writing_past_it():
    """the device cannot raise: when a frame's edges would not fit the edge buffers the begin launch sets the error word of
    the dynamic block and appends nothing; every later frame leaves the stream as it is; counts() raises on the host"""
    from cdv_slam_amd.stream import DeviceStreamRunner
    run = DeviceStreamRunner(torch.device(DEV), M=24, ht=192, wd=256, buffer_size=64)
    for f in range(6):
        run.frame(drop=False)
    n0, E0 = run.counts()
    run._desc.edge_capacity = E0 + 100          # the next frame's 2 r M edges do not fit any more
    guard = run._ii[run.cur, E0:E0 + 2000].clone()
    for f in range(3):
        run.frame(drop=False)
    with pytest.raises(RuntimeError, match="capacity exceeded"):
        run.counts()
    blk = run.dyn[run.slot].cpu()
    assert (int(blk[0]), int(blk[1])) == (n0, E0) and int(blk[7]) == 1
    assert torch.equal(run._ii[run.cur, E0:E0 + 2000], guard)      # nothing was appended
