"""oracle/stream_py.py -- TEST INFRASTRUCTURE ONLY (tests/, bench.py's checker leg).

A whole frame stream on the CPU oracle: what `SLAM.__call__` does per frame once frames arrive (cdvslam/slam.py:612-740)
with the feature network and the update operator replaced by the SAME stubs cdv_slam_amd/stream.py uses, every step a
restatement of the reference, in the reference's arithmetic:

    new frame -> state buffers (slam.py:676-696), edges (append_factors(*edges_forw / *edges_back), :707-709)   edges_py.EdgesPy
    update()  (slam.py:480-526): reproject = orc_transform (projective_ops.py:53-113), corr = orc_corr in c10::Half
              arithmetic on the planar rings (correlation_kernel.cu:82-136,193-233; slam.py:316-323), stub operator,
              fastba = orc_fastba float32 (ba_cuda.cu:462-611), point cloud (slam.py:524-526)
    keyframe() (slam.py:408-458): motionmag from flow_mag (projective_ops.py:120-130, slam.py:399-406), frame k = n - 4
              dropped when the mean flow is under the threshold, buffers shifted (slam.py:429-441), removal-window pruning

parity unpinned as a whole (the reference has no stream test and cannot run here); its parts are pinned as DESIGN.md section
4 says.  Used to drive the GPU StreamRunner and this runner side by side over >= 120 frames (closed loop: differences in
the half-precision correlation feed back through the operator stub into the poses of later frames) and to compare edge
lists frame by frame and trajectories by Sim(3)-aligned ATE (evaluate_tartan.py:63-70)."""
import numpy as np

from . import oracle as O
from .edges_py import EdgesPy


def pool4(fmap16):
    """slam.py:682: F.avg_pool2d(fmap, 4, 4) on a half map -- float32 sum of the 16 values in window order, times 1/16,
    rounded to half"""
    C, h, w = fmap16.shape
    x = fmap16.astype(np.float32).reshape(C, h // 4, 4, w // 4, 4)
    s = np.zeros((C, h // 4, w // 4), np.float32)
    for a in range(4):
        for b in range(4):
            s = s + x[:, :, a, :, b]
    return (s * np.float32(1.0 / 16.0)).astype(np.float16)


def flow_mag(poses, patches, intrinsics, ii, jj, kk, beta, dtype=np.float32):
    """pops.flow_mag (projective_ops.py:120-130): beta * |full flow| + (1 - beta) * |translation-only flow| per patch
    pixel, and the validity (Z > 0.2 in both views)"""
    c0 = O.transform(poses, patches, intrinsics, ii, ii, kk, dtype=dtype)
    c1, val = O.transform(poses, patches, intrinsics, ii, jj, kk, valid=True, dtype=dtype)
    c2 = O.transform(poses, patches, intrinsics, ii, jj, kk, tonly=True, dtype=dtype)
    f1 = np.sqrt(((c1 - c0) ** 2).sum(-1))
    f2 = np.sqrt(((c2 - c0) ** 2).sum(-1))
    return beta * f1 + (1 - beta) * f2, val > 0.5


class StreamOracle:
    def __init__(self, M=96, ht=384, wd=512, C=24, mem=36, pmem=36, buffer_size=512, patch_lifetime=13, removal_window=22,
                 opt_window=10, keyframe_index=4, keyframe_thresh=12.5, corr_mode="ref", dtype=np.float32, gain=0.01, pose_step=0.05):
        self.pose_step = pose_step
        self.M, self.C, self.mem, self.pmem, self.N = M, C, mem, pmem, buffer_size
        self.h, self.w = ht // 4, wd // 4
        self.r, self.rw, self.ow, self.ki, self.kthresh = patch_lifetime, removal_window, opt_window, keyframe_index, keyframe_thresh
        self.corr_mode, self.dtype, self.gain = corr_mode, dtype, gain
        self.poses = np.zeros((self.N, 7), np.float32); self.poses[:, 6] = 1.0
        self.patches = np.zeros((self.N * M, 3, 3, 3), np.float32)
        self.intrinsics = np.tile(np.array([wd / 2.0, wd / 2.0, wd / 2.0, ht / 2.0], np.float32) / 4.0, (self.N, 1))
        self.ix = np.repeat(np.arange(self.N), M)
        self.fmap1 = np.zeros((mem, C, self.h, self.w), np.float16)
        self.fmap2 = np.zeros((mem, C, self.h // 4, self.w // 4), np.float16)
        self.gmap = np.zeros((pmem * M, C, 3, 3), np.float16)
        self.points = np.zeros((self.N * M, 3), np.float32)
        self.edges = EdgesPy()
        self.n = 0
        self.last_motion = None
        self.last = {}

    # -- stub of network.patchify (net_cdv.py:355-374): the caller supplies centres and depths, tiles are cut out of the features
    def _new_frame(self, fmap, cx, cy, d):
        M, n = self.M, self.n
        off = np.array([-1.0, 0.0, 1.0], np.float32)
        pt = np.empty((M, 3, 3, 3), np.float32)
        pt[:, 0] = cx[:, None, None] + off[None, None, :]
        pt[:, 1] = cy[:, None, None] + off[None, :, None]
        pt[:, 2] = d[:, None, None]
        self.patches[n * M:(n + 1) * M] = pt
        tiles = O.patchify(fmap, np.stack([cx, cy], -1).astype(np.float32), 1, "bilinear").astype(np.float16)   # [M,C,3,3]
        t0 = (n % self.pmem) * M
        self.gmap[t0:t0 + M] = tiles
        if n > 0:
            self.poses[n] = self.poses[n - 1]
            self.poses[n, 0] += np.float32(self.pose_step)

    def _update(self, fmap):
        M, n, e = self.M, self.n, self.edges
        slot = (n - 1) % self.mem
        self.fmap1[slot] = fmap                       # slam.py:679-682
        self.fmap2[slot] = pool4(fmap)
        coords = O.transform(self.poses, self.patches, self.intrinsics, e.ii, e.jj, e.kk)        # [E,3,3,2]
        coords = np.ascontiguousarray(coords.transpose(0, 3, 1, 2))                              # slam.py:329
        corr = O.slam_corr(self.gmap, self.fmap1, self.fmap2, coords, e.kk % (M * self.pmem), e.jj % self.mem, 3,
                           self.corr_mode).astype(np.float32)
        # the operator stub of cdv_slam_amd/stream.py: a small correction that depends on the correlation
        delta = np.float32(self.gain) * np.tanh(corr[:, :2])
        e.target = (coords[:, :, 1, 1] + delta).astype(np.float32)
        e.weight = (1.0 / (1.0 + np.exp(-corr[:, 2:4]))).astype(np.float32)
        t0 = max(1, n - self.ow)
        p, x, info = O.fastba(self.poses, self.patches, self.intrinsics[0], e.target, e.weight, 1e-4, e.ii, e.jj, e.kk, t0, n,
                              2, self.dtype)
        self.poses, self.patches = p.astype(np.float32), x.astype(np.float32)
        self.last = {"coords": coords, "corr": corr, "t0": t0, "info": info}
        # slam.py:524-526: the world points of every patch so far
        m = n * M
        Pinv = O.lie(O.SE3, "inv", self.poses[self.ix[:m]], dtype=np.float32)
        pt, K = self.patches[:m], self.intrinsics[self.ix[:m]]
        X0 = np.stack([(pt[:, 0, 1, 1] - K[:, 2]) / K[:, 0], (pt[:, 1, 1, 1] - K[:, 3]) / K[:, 1],
                       np.ones(m, np.float32), pt[:, 2, 1, 1]], -1).astype(np.float32)
        X = O.lie(O.SE3, "act4", Pinv, X0, dtype=np.float32)
        self.points[:m] = X[:, :3] / X[:, 3:]

    def motionmag(self, i, j):
        """slam.py:399-406"""
        e = self.edges
        k = (e.ii == i) & (e.jj == j)
        if not k.any():
            return float("nan")
        flow, _ = flow_mag(self.poses, self.patches, self.intrinsics, e.ii[k], e.jj[k], e.kk[k], 0.5)
        return float(flow.mean())

    def motion(self):
        """the keyframe test's statistic (slam.py:409-413): (motionmag(i, j) + motionmag(j, i)) / 2 for the two frames
        around k = n - KEYFRAME_INDEX"""
        i, j = self.n - self.ki - 1, self.n - self.ki + 1
        return 0.5 * (self.motionmag(i, j) + self.motionmag(j, i))

    def _keyframe(self, drop):
        M, n = self.M, self.n
        k = n - self.ki
        if drop:                                       # slam.py:429-441
            for i in range(k, n - 1):
                self.poses[i] = self.poses[i + 1]
                self.patches[i * M:(i + 1) * M] = self.patches[(i + 1) * M:(i + 2) * M]
                self.intrinsics[i] = self.intrinsics[i + 1]
                a, b = (i % self.pmem) * M, ((i + 1) % self.pmem) * M
                self.gmap[a:a + M] = self.gmap[b:b + M]
                self.fmap1[i % self.mem] = self.fmap1[(i + 1) % self.mem]
                self.fmap2[i % self.mem] = self.fmap2[(i + 1) % self.mem]
        self.n = self.edges.keyframe(k, n, M, self.ix, self.rw, drop=drop)

    def frame(self, fmap, cx, cy, d, drop=None):
        """one incoming frame.  drop: True / False = the caller decides whether frame n - 4 leaves (as StreamRunner's
        `drop` argument); None = the reference's own test, mean flow under KEYFRAME_THRESH (slam.py:413)"""
        self._new_frame(fmap, cx, cy, d)
        self.n += 1
        e = self.edges
        e.append_factors(*e.edges_forw(self.n, self.M, self.r), self.ix)
        e.append_factors(*e.edges_back(self.n, self.M, self.r), self.ix)
        self.last_motion = None
        if self.n >= 8:
            self._update(fmap)
            if drop is None:
                self.last_motion = self.motion()
                drop = self.last_motion < self.kthresh
            self._keyframe(bool(drop) and self.n > self.ki + 2)
        else:
            slot = (self.n - 1) % self.mem
            self.fmap1[slot] = fmap
            self.fmap2[slot] = pool4(fmap)
        return self.n, len(e.ii)


def closed_loop(run, so, frames, seed=1234, drop="pattern", check_edges=True, progress=None):
    """Drive the GPU stream runner `run` (cdv_slam_amd.stream.StreamRunner) and the oracle runner `so` side by side over
    `frames` frames from the same stubbed network outputs.  drop: "pattern" = frame n - 4 leaves on every third frame
    (the caller's decision, identical on both sides); "flow" = each side applies the reference's own test to ITS state
    (mean flow under KEYFRAME_THRESH, slam.py:409-413) and the decisions are compared.  Returns a dict: frames run,
    keyframes kept, edges, whether every frame's edge lists were bit-identical, the decisions that differed, the largest
    |motion| difference, and the final trajectories (poses of the keyframes) of both sides."""
    import torch
    rng = np.random.default_rng(seed)
    M, C, h, w = so.M, so.C, so.h, so.w
    pool = [(rng.standard_normal((C, h, w)) / 4).astype(np.float16) for _ in range(4)]
    pool_dev = [torch.as_tensor(p, device=run.dev) for p in pool]
    res = {"frames": 0, "edges_identical": True, "decisions_differ": [], "motion_maxdiff": 0.0, "dropped": 0}
    for f in range(frames):
        cx = rng.uniform(8, w - 8, M).astype(np.float32)
        cy = rng.uniform(8, h - 8, M).astype(np.float32)
        d = rng.uniform(0.25, 1.0, M).astype(np.float32)
        T = lambda a: torch.as_tensor(a, device=run.dev)
        want = (f % 3 == 2) if drop == "pattern" else None
        n0 = so.n
        n_o, E_o = so.frame(pool[f % 4], cx, cy, d, drop=want)
        run.frame(drop=want, inputs=(pool_dev[f % 4], T(cx), T(cy), T(d)))
        n_g, E_g = run.counts()
        res["dropped"] += int(n_o == n0)
        if drop == "flow" and so.last_motion is not None:
            res["motion_maxdiff"] = max(res["motion_maxdiff"], abs(so.last_motion - run.last_motion))
            if (so.last_motion < so.kthresh) != (run.last_motion < run.kthresh):
                res["decisions_differ"].append((f, so.last_motion, run.last_motion))
        same = (n_o == n_g and E_o == E_g)
        if same and check_edges:
            e = run.edges
            same = (np.array_equal(e.ii.cpu().numpy(), so.edges.ii) and np.array_equal(e.jj.cpu().numpy(), so.edges.jj)
                    and np.array_equal(e.kk.cpu().numpy(), so.edges.kk))
        if not same:
            res["edges_identical"] = False
            res["first_mismatch"] = f
            break
        res["frames"] = f + 1
        if progress is not None and f % 20 == 19:
            progress("closed loop: frame %d, %d keyframes, %d edges" % (f + 1, n_o, E_o))
    n = min(so.n, int(run.n))
    res.update(keyframes=n, edges=len(so.edges.ii), poses_oracle=so.poses[:n].copy(),
               poses_gpu=run.poses[:n].cpu().numpy(), patches_oracle=so.patches[:n * M, 2, 1, 1].copy(),
               patches_gpu=run.patches[:n * M, 2, 1, 1].cpu().numpy())
    return res
