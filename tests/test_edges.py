"""Edge bookkeeping: the numpy restatement of slam.py's append / remove / keyframe logic against
the replay SURVEY.md quotes the graph sizes from (CPU), and the device EdgeStore against that restatement, bit-exact, over
a stream of frames with random keyframe drops (GPU)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from cdv_slam_amd import synth            # noqa: E402
from oracle.edges_py import EdgesPy       # noqa: E402


def test_edges_oracle_reproduces_the_survey_replay():
    cfg = synth.CONFIGS["default"]
    M, r = cfg.M, cfg.patch_lifetime
    ix = np.repeat(np.arange(cfg.buffer_size), M)
    g = EdgesPy()
    for n in range(1, cfg.frames + 1):
        g.append_factors(*g.edges_forw(n, M, r), ix)
        g.append_factors(*g.edges_back(n, M, r), ix)
        if n < cfg.frames and n >= 8:
            g.keyframe(-1, n, M, ix, cfg.removal_window, drop=False)
    ii, jj, kk = synth.replay_edges(cfg)
    assert len(g.ii) == 47712
    assert np.array_equal(g.ii, ii) and np.array_equal(g.jj, jj) and np.array_equal(g.kk, kk)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,M,r,rw", [(0, 8, 5, 9), (1, 96, 13, 22)])
def test_edge_store_stream_bit_exact(seed, M, r, rw):
    import torch
    from cdv_slam_amd.edges import EdgeStore
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(seed)
    frames = 60
    ix_np = np.repeat(np.arange(frames + 8), M)
    ix = torch.as_tensor(ix_np, device=dev)
    st = EdgeStore(dev, capacity=M * (rw + 4) * 2 * r, net_dim=16)
    g = EdgesPy()
    n = 0
    for f in range(frames):
        n += 1
        # frame arrival (slam.py:697-709)
        E0 = len(g.ii)
        g.append_factors(*g.edges_forw(n, M, r), ix_np)
        g.append_factors(*g.edges_back(n, M, r), ix_np)
        added = st.append_frame(ix, n, M, r)
        assert added == len(g.ii) - E0 and st.E == len(g.ii)
        # update(): new target / weight / hidden state for every edge (slam.py:486-511)
        tgt = rng.standard_normal((st.E, 2)).astype(np.float32)
        wgt = rng.uniform(0, 1, (st.E, 2)).astype(np.float32)
        g.target, g.weight = tgt.copy(), wgt.copy()
        st.target[0].copy_(torch.as_tensor(tgt, device=dev)); st.weight[0].copy_(torch.as_tensor(wgt, device=dev))
        if n >= 8:
            drop = bool(rng.uniform() < 0.35) and n > 6
            k = n - 4                                          # KEYFRAME_INDEX = 4 (slam.py:409-417)
            n_new = g.keyframe(k, n, M, ix_np, rw, drop=drop)
            n_dev = st.keyframe(k, n, M, ix, rw, drop=drop)
            assert n_dev == n_new
            n = n_new
        for name in ("ii", "jj", "kk"):
            assert np.array_equal(getattr(st, name).cpu().numpy(), getattr(g, name)), (f, name)
        assert np.array_equal(st.target[0].cpu().numpy(), g.target) and np.array_equal(st.weight[0].cpu().numpy(), g.weight)
        a = st.E_inac
        assert a == len(g.ii_inac)
        for name in ("ii_inac", "jj_inac", "kk_inac", "target_inac", "weight_inac"):
            assert np.array_equal(getattr(st, name)[:a].cpu().numpy(), getattr(g, name)), (f, name)
    # hidden-state payload check: tag rows with their (kk, jj) and make sure a removal keeps rows with their edges
    tag = torch.stack([st.kk.to(torch.float16) % 997, st.jj.to(torch.float16)], 1)
    st.net[0][:, :2].copy_(tag)
    mask = torch.as_tensor(rng.uniform(size=st.E) < 0.5, device=dev)
    st.remove_factors(mask, store=False)
    assert torch.equal(st.net[0][:, 0], st.kk.to(torch.float16) % 997) and torch.equal(st.net[0][:, 1], st.jj.to(torch.float16))
    # generic append (loop-closure edges) and the concatenated view the global BA takes
    nk = torch.as_tensor(rng.integers(0, n * M, 50), device=dev)
    nj = torch.as_tensor(rng.integers(0, n, 50), device=dev)
    E0 = st.E
    st.append_factors(nk, nj, ix)
    assert st.E == E0 + 50 and torch.equal(st.kk[E0:], nk) and torch.equal(st.jj[E0:], nj) and torch.equal(st.ii[E0:], ix[nk])
    ft, fw, fi, fj, fk = st.full_edges()
    assert fi.numel() == st.E + st.E_inac and ft.shape == (1, st.E + st.E_inac, 2)


@pytest.mark.gpu
def test_stream_runner_end_to_end():
    """a whole synthetic stream (state write, edge append, prologue, correlation, BA, keyframe drops) stays finite and
    keeps exactly the edge lists the reference bookkeeping would"""
    import torch
    from cdv_slam_amd.stream import StreamRunner
    dev = torch.device("cuda:0")
    run = StreamRunner(dev, M=16, ht=192, wd=256, buffer_size=96, mem=36, pmem=36)
    g = EdgesPy()
    ix_np = np.repeat(np.arange(96), 16)
    n = 0
    for f in range(70):
        drop = f % 4 == 3
        n_dev, E = run.frame(drop=drop)
        n += 1
        g.append_factors(*g.edges_forw(n, 16, run.r), ix_np)
        g.append_factors(*g.edges_back(n, 16, run.r), ix_np)
        if n >= 8:
            n = g.keyframe(n - run.ki, n, 16, ix_np, run.rw, drop=drop and n > run.ki + 2)
        assert n_dev == n and E == len(g.ii)
        assert np.array_equal(run.edges.kk.cpu().numpy(), g.kk) and np.array_equal(run.edges.jj.cpu().numpy(), g.jj)
    assert run.n_updates == 70 - 7
    assert torch.isfinite(run.poses[:n]).all() and torch.isfinite(run.patches[:n * 16]).all()
    assert float(run.poses[:n, :3].abs().max()) < 50.0 and float((run.poses[:n, 3:].norm(dim=-1) - 1).abs().max()) < 1e-3


@pytest.mark.gpu
def test_frames_keyframe_shift_matches_the_reference_loop():
    """cdv_frames_keyframe_shift vs the Python loop of cdvslam/slam.py:431-441 (nine tensor copies per shifted frame):
    linear frame buffers (28-, 16- and 10,368-byte slots), rings with a modulus that the shift wraps around, an
    unaligned view, bit-exact"""
    import torch
    from cdv_slam_amd.edges import frames_keyframe_shift
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(3)
    N, M, mem, pmem = 64, 96, 12, 10
    poses = torch.randn((N, 7), generator=g).to(dev)
    intr = torch.randn((N, 4), generator=g).to(dev)
    patches = torch.randn((N * M, 3, 3, 3), generator=g).to(dev)
    tst = torch.arange(N, dtype=torch.int64).to(dev) * 3 + 1
    colors = torch.randint(0, 255, (N, M, 3), generator=g, dtype=torch.uint8).to(dev)   # 288-byte slots
    fmap = torch.randn((mem, 7, 9, 8), generator=g).half().to(dev)
    gmap = torch.randn((pmem * M, 8, 3, 3), generator=g).half().to(dev)
    odd = torch.randn((N * 5 + 1,), generator=g).to(dev)[1:].view(N, 5)                 # base only 4-byte aligned
    for k, n in [(40, 44), (3, 30), (20, 21), (7, 8), (0, 2), (50, 64)]:
        bufs = [poses, intr, patches, tst, colors, fmap, gmap, odd]
        want = [b.clone() for b in bufs]
        wp, wi, wpa, wt, wc, wf, wg, wo = want
        for i in range(k, n - 1):   # the reference loop, verbatim structure
            wt[i] = wt[i + 1]
            wc[i] = wc[i + 1]
            wp[i] = wp[i + 1]
            wpa[i * M:(i + 1) * M] = wpa[(i + 1) * M:(i + 2) * M]
            wi[i] = wi[i + 1]
            wg[(i % pmem) * M:(i % pmem + 1) * M] = wg[((i + 1) % pmem) * M:((i + 1) % pmem + 1) * M].clone()
            wf[i % mem] = wf[(i + 1) % mem].clone()
            wo[i] = wo[i + 1]
        frames_keyframe_shift([(poses, 0), (intr, 0), (patches.view(N, -1), 0), (tst, 0), (colors, 0), (fmap, mem),
                               (gmap.view(pmem, -1), pmem), (odd, 0)], k, n)
        torch.cuda.synchronize()
        for name, got, w in zip("poses intr patches tstamps colors fmap gmap odd".split(), bufs, want):
            assert torch.equal(got, w), (name, k, n)


def test_reduce_edges_host_selection():
    """loop.reduce_edges (optim_utils.py:23-60): increasing flow, pairs < 30 frames apart and flows >= 1000 skipped,
    a pick suppresses its neighbours i - nms .. i + nms for the same j, at most max_num_edges picks"""
    from cdv_slam_amd.loop import reduce_edges
    flow = np.array([5.0, 1.0, 2.0, 3.0, 1500.0, 0.5, 4.0])
    ii = np.array([10, 11, 12, 40, 5, 50, 14])
    jj = np.array([60, 60, 60, 60, 60, 60, 61])
    es = reduce_edges(flow, ii, jj, max_num_edges=1000, nms=1)
    # order of flow: (50,60) gap 10 -> skipped; (11,60) picked, suppresses 10..12 @ 60; (12,60) suppressed; (40,60) gap 20
    # skipped; (14,61) picked; (10,60) suppressed; (5,60) flow 1500 skipped
    assert es.tolist() == [[11, 60], [14, 61]]
    assert reduce_edges(flow, ii, jj, max_num_edges=1, nms=1).tolist() == [[11, 60]]
    assert reduce_edges(np.zeros(0), np.zeros(0, np.int64), np.zeros(0, np.int64), 10, 1).shape == (0, 2)
    es0 = reduce_edges(flow, ii, jj, max_num_edges=1000, nms=0)
    assert es0.tolist() == [[11, 60], [12, 60], [14, 61], [10, 60]]
