"""A whole frame stream on the device path: what `SLAM.__call__` does per frame once the system is initialised
(cdvslam/slam.py:612-740), with the feature network and the update operator replaced by stubs.

    new frame -> state buffers, edges (EdgeStore.append_frame)                     slam.py:676-709
    update(): prologue [ingest | reproject | index] -> corr -> (stub operator) -> BA        slam.py:480-526
    keyframe(): optional drop of frame n - 4, removal-window pruning               slam.py:408-458

Used by scripts/bench_stream.py for the end-to-end frames/s of SURVEY.md 8(d)(iii) and as an integration test of the
pieces together (edge counts change every frame, buffers wrap around their rings)."""
import numpy as np
import torch

from . import ops
from .edges import EdgeStore, frames_keyframe_shift


class StreamRunner:
    def __init__(self, device, M=96, ht=384, wd=512, C=24, mem=36, pmem=36, buffer_size=512, patch_lifetime=13,
                 removal_window=22, opt_window=10, keyframe_index=4, seed=1234, loop_closure=False, max_edge_age=1000,
                 global_opt_freq=15, backend_thresh=64.0, pose_init=None, record_global=False, keyframe_thresh=12.5, gain=0.01):
        """loop_closure: the LOOP_CLOSURE configuration of the reference (default_cdvslam.yaml): the patch ring holds
        MAX_EDGE_AGE frames (slam.py:66-68), proximity loop edges are added every GLOBAL_OPT_FREQ frames
        (slam.py:699-705, patchgraph.py:71-97), edges that close loops survive the removal window (slam.py:453-457) and an
        update with long-range edges runs the GLOBAL bundle adjustment over inactive + active edges (slam.py:460-478,507).
        pose_init(n) -> 7 floats: the initial guess of frame n's pose (default: the previous pose moved forward)."""
        self.dev = device
        self.lc, self.max_edge_age, self.gof, self.backend_thresh = loop_closure, max_edge_age, global_opt_freq, backend_thresh
        self.last_global_ba = -1000
        self.ran_global_ba = {}
        self.n_global = 0
        self.pose_init, self.record_global, self.last_global = pose_init, record_global, None
        self.last_loop_edges = None    # the loop edges the latest frame appended (tests)
        if loop_closure:
            pmem = max_edge_age
        self.M, self.C, self.mem, self.pmem = M, C, mem, pmem
        self.h, self.w = ht // 4, wd // 4
        self.r, self.rw, self.ow, self.ki = patch_lifetime, removal_window, opt_window, keyframe_index
        self.N = buffer_size
        self.kthresh, self.gain = keyframe_thresh, gain      # KEYFRAME_THRESH (config.py:20); the operator stub's step
        self.last_motion = None
        g = torch.Generator(device="cpu").manual_seed(seed)
        self.g = g
        f32 = dict(dtype=torch.float32, device=device)
        self.poses = torch.zeros((self.N, 7), **f32); self.poses[:, 6] = 1.0
        self.patches = torch.zeros((self.N * M, 3, 3, 3), **f32)
        intr = torch.tensor([wd / 2.0, wd / 2.0, wd / 2.0, ht / 2.0]) / 4.0
        self.intrinsics = intr.to(device).repeat(self.N, 1).contiguous()
        self.ix = torch.arange(self.N, device=device).repeat_interleave(M)
        self.fmap1 = ops.alloc_fmap_ring(mem, C, self.h, self.w, device)
        self.fmap2 = ops.alloc_fmap_ring(mem, C, self.h // 4, self.w // 4, device)
        self.gmap = torch.zeros((pmem * M, C, 3, 3), dtype=torch.float16, device=device)
        self.gmap_pm = torch.zeros((pmem * M, 9, C), dtype=torch.float16, device=device)
        ecap = M * (removal_window + 6) * 2 * patch_lifetime + (M * 1000 if loop_closure else 0)
        self.edges = EdgeStore(device, capacity=ecap, net_dim=0, inactive_capacity=ecap + buffer_size * M * 2 * patch_lifetime)
        # without loop closure the live patch ids span the removal window (+ the frames of this update's bookkeeping): the
        # index is the two-launch patch table and the correlation reads the packed stream it writes; with loop closure
        # (patches of any age keep edges) the ranked index
        self.table = not loop_closure
        self.graph = ops.GraphIndex(device, E_cap=ecap, k_range=(removal_window + 8) * M + M * pmem,
                                    table_capacity=(removal_window + 8) * M if self.table else None)
        if self.table:
            self.coords_buf = torch.empty((1, ecap, 2, 3, 3), dtype=torch.float32, device=device)
            self.graph.bind_corr_stream(None, M * pmem, mem, pmem * M, mem)
        self.graph_full = None     # index over inactive + active edges (global BA), built on demand
        self.lmbda = torch.tensor([1e-4], **f32)
        self.n = 0
        self.n_updates = 0
        # a fixed pool of synthetic frames (stub feature network): feature maps and patch tiles
        self.pool = [(torch.randn((C, self.h, self.w), generator=g) / 4).half().to(device) for _ in range(4)]

    # -- stub of network.patchify (net_cdv.py:355-374): random patch centres, tiles cut out of the frame's features
    def _new_frame(self, inputs=None):
        """inputs: (fmap [C,h,w] f16, cx [M], cy [M], d [M]) on the device -- the stubbed network's outputs for this frame
        (tests hand the same ones to the oracle runner); None: drawn here"""
        M, n = self.M, self.n
        if inputs is None:
            fmap = self.pool[n % len(self.pool)]
            cx = (torch.rand(M, generator=self.g) * (self.w - 16) + 8).to(self.dev)
            cy = (torch.rand(M, generator=self.g) * (self.h - 16) + 8).to(self.dev)
            d = (torch.rand(M, generator=self.g) * 0.75 + 0.25).to(self.dev)
        else:
            fmap, cx, cy, d = inputs
        off = torch.tensor([-1.0, 0.0, 1.0], device=self.dev)
        pt = torch.empty((M, 3, 3, 3), device=self.dev)
        pt[:, 0] = cx[:, None, None] + off[None, None, :]
        pt[:, 1] = cy[:, None, None] + off[None, :, None]
        pt[:, 2] = d[:, None, None]
        self.patches[n * M:(n + 1) * M] = pt
        coords = torch.stack([cx, cy], -1)[None]
        tiles = ops.patchify_blend(fmap[None], coords, 1, "bilinear")[0].half()       # [M,C,3,3]
        t0 = (n % self.pmem) * M
        self.gmap[t0:t0 + M] = tiles
        if self.pose_init is not None:
            self.poses[n] = torch.as_tensor(self.pose_init(n), dtype=torch.float32, device=self.dev)
        elif n > 0:   # constant-position initialisation plus a small forward motion
            self.poses[n] = self.poses[n - 1]
            self.poses[n, 0] += 0.05
        return fmap, t0

    def _update(self, fmap, tile0):
        M, n, e = self.M, self.n, self.edges
        ii, jj, kk = e.ii, e.jj, e.kk      # ONE set of view objects for the whole update: the index is keyed on them
        if self.table:
            coords = ops.update_prologue_table(self.graph, fmap, self.fmap1, self.fmap2, (n - 1) % self.mem, self.gmap,
                                               self.gmap_pm, tile0, M, self.poses, self.patches, self.intrinsics, ii, jj, kk,
                                               coords_out=self.coords_buf)
            corr = ops.corr_fused_stream(self.gmap_pm, self.fmap1, self.fmap2, self.graph.corr_records_ptr(), e.E)
        else:
            coords = ops.update_prologue(self.graph, fmap, self.fmap1, self.fmap2, (n - 1) % self.mem, self.gmap, self.gmap_pm,
                                         tile0, M, self.poses, self.patches, self.intrinsics, ii, jj, kk)
            corr = ops.corr_fused(self.gmap_pm, self.fmap1, self.fmap2, coords, kk, jj, kmod=M * self.pmem, jmod=self.mem,
                                  pixel_major=True, order_ptr=self.graph.corr_order_ptr())   # edges grouped by target frame
        # stub of the update operator (net_cdv.py:66-107): a small correction that depends on the correlation
        delta = self.gain * torch.tanh(corr[0, :, :2].float())
        e.target[0].copy_(coords[0, :, :, 1, 1] + delta)
        e.weight[0].copy_(torch.sigmoid(corr[0, :, 2:4].float()))
        if self.lc and bool((e.ii < n - self.rw - 1).any()) and not self.ran_global_ba.get(n, False):
            self._global_ba()      # long-range edges exist: slam.py:507-510
        else:
            t0 = max(1, n - self.ow)
            ops.ba_forward(self.poses, self.patches, self.intrinsics, e.target, e.weight, self.lmbda, ii, jj, kk, M, t0,
                           n, 2, False, U_max=min(e.E, (self.rw + 8) * M + (1000 * M if self.lc else 0)), graph=self.graph)
        self.n_updates += 1

    def normalize(self):
        """PatchGraph.normalize (patchgraph.py:99-117): mean inverse depth 1, first pose the identity"""
        from .lietorch import SE3
        n, M = self.n, self.M
        s = self.patches[:n * M, 2].mean()
        self.patches[:n * M, 2] /= s
        self.poses[:n, :3] *= s
        self.poses[:n] = (SE3(self.poses[:n]) * SE3(self.poses[[0]]).inv()).data

    def _global_ba(self):
        """SLAM.__run_global_BA (slam.py:460-478): inactive + active edges, every pose from the oldest active source
        frame on is free"""
        e, n, M = self.edges, self.n, self.M
        target, weight, ii, jj, kk = e.full_edges()
        self.normalize()
        t0 = int(e.ii.min().item())
        if self.record_global:
            self.last_global = dict(poses=self.poses.clone(), patches=self.patches.clone(), target=target[0].clone(),
                                    weight=weight[0].clone(), ii=ii.clone(), jj=jj.clone(), kk=kk.clone(), t0=t0, n=n,
                                    E_active=e.E, E_inactive=e.E_inac)
        if self.graph_full is None or self.graph_full.E_cap < ii.numel():
            self.graph_full = ops.GraphIndex(self.dev, E_cap=int(ii.numel() * 1.5) + 1024, k_range=self.N * M)
        ops.ba_forward(self.poses, self.patches, self.intrinsics, target, weight, self.lmbda, ii, jj, kk, M, t0, n, 2, True,
                       U_max=min(int(ii.numel()), n * M), graph=self.graph_full)
        self.ran_global_ba[n] = True
        self.n_global += 1

    def _keyframe(self, drop):
        M, n = self.M, self.n
        k = n - self.ki
        if drop:   # frame k leaves: every frame buffer shifts down, one launch (slam.py:429-441)
            frames_keyframe_shift([(self.poses, 0), (self.intrinsics, 0), (self.patches.view(self.N, -1), 0),
                                   (self.gmap.view(self.pmem, -1), self.pmem), (self.gmap_pm.view(self.pmem, -1), self.pmem),
                                   (self.fmap1, self.mem), (self.fmap2, self.mem)], k, n)
        self.n = self.edges.keyframe(k, n, M, self.ix, self.rw, loop_closure=self.lc, opt_window=self.ow, drop=drop)

    def motion(self):
        """the keyframe test's statistic (slam.py:399-413): mean flow between the frames either side of k = n -
        KEYFRAME_INDEX, both directions; read back like the reference's two .item() calls"""
        from . import projective_ops as pops
        e, n = self.edges, self.n
        tot = 0.0
        for i, j in ((n - self.ki - 1, n - self.ki + 1), (n - self.ki + 1, n - self.ki - 1)):
            k = (e.ii == i) & (e.jj == j)
            flow, _ = pops.flow_mag(self.poses[None], self.patches[None], self.intrinsics[None], e.ii[k], e.jj[k], e.kk[k], beta=0.5)
            tot += flow.mean().item()
        return 0.5 * tot

    def frame(self, drop=False, inputs=None):
        """one incoming frame (slam.py:697-720 for an initialised system).  drop: True / False = the caller decides
        whether frame n - KEYFRAME_INDEX leaves; None = the reference's test (mean flow under KEYFRAME_THRESH, slam.py:413)"""
        if self.n + 1 >= self.N:
            raise RuntimeError("StreamRunner: frame buffer full")
        fmap, tile0 = self._new_frame(inputs)
        self.n += 1
        self.last_loop_edges = None
        if self.lc and self.n - self.last_global_ba >= self.gof:      # proximity loop edges (slam.py:699-705)
            from . import loop
            lk, lj = loop.edges_loop(self.poses, self.patches, self.intrinsics, self.ix, self.n, self.M,
                                     removal_window=self.rw, max_edge_age=self.max_edge_age, global_opt_freq=self.gof,
                                     keyframe_index=self.ki, backend_thresh=self.backend_thresh)
            if lk.numel() > 0:
                self.last_global_ba = self.n
                self.ran_global_ba[self.n] = False
                self.edges.append_factors(lk, lj, self.ix)
                self.last_loop_edges = (lk, lj)
        self.edges.append_frame(self.ix, self.n, self.M, self.r)
        if self.n >= 8:
            self._update(fmap, tile0)
            self.last_motion = None
            if drop is None:
                self.last_motion = self.motion()
                drop = self.last_motion < self.kthresh
            self._keyframe(bool(drop) and self.n > self.ki + 2)
        else:   # before initialisation only the rings are filled
            ops.fmap_ingest(fmap, self.fmap1, self.fmap2, (self.n - 1) % self.mem, gmap=self.gmap, gmap_pm=self.gmap_pm,
                            gmap_first=tile0, gmap_count=self.M)
        return self.n, self.edges.E
