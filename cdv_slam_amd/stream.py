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
                 removal_window=22, opt_window=10, keyframe_index=4, seed=1234):
        self.dev = device
        self.M, self.C, self.mem, self.pmem = M, C, mem, pmem
        self.h, self.w = ht // 4, wd // 4
        self.r, self.rw, self.ow, self.ki = patch_lifetime, removal_window, opt_window, keyframe_index
        self.N = buffer_size
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
        ecap = M * (removal_window + 6) * 2 * patch_lifetime
        self.edges = EdgeStore(device, capacity=ecap, net_dim=0, inactive_capacity=ecap + buffer_size * M * 2 * patch_lifetime)
        self.graph = ops.GraphIndex(device, E_cap=ecap, k_range=(removal_window + 8) * M + M * pmem)
        self.lmbda = torch.tensor([1e-4], **f32)
        self.n = 0
        self.n_updates = 0
        # a fixed pool of synthetic frames (stub feature network): feature maps and patch tiles
        self.pool = [(torch.randn((C, self.h, self.w), generator=g) / 4).half().to(device) for _ in range(4)]

    # -- stub of network.patchify (net_cdv.py:355-374): random patch centres, tiles cut out of the frame's features
    def _new_frame(self):
        M, n = self.M, self.n
        fmap = self.pool[n % len(self.pool)]
        cx = torch.rand(M, generator=self.g) * (self.w - 16) + 8
        cy = torch.rand(M, generator=self.g) * (self.h - 16) + 8
        d = torch.rand(M, generator=self.g) * 0.75 + 0.25
        off = torch.tensor([-1.0, 0.0, 1.0])
        pt = torch.empty((M, 3, 3, 3))
        pt[:, 0] = cx[:, None, None] + off[None, None, :]
        pt[:, 1] = cy[:, None, None] + off[None, :, None]
        pt[:, 2] = d[:, None, None]
        self.patches[n * M:(n + 1) * M] = pt.to(self.dev)
        coords = torch.stack([cx, cy], -1)[None].to(self.dev)
        tiles = ops.patchify_blend(fmap[None], coords, 1, "bilinear")[0].half()       # [M,C,3,3]
        t0 = (n % self.pmem) * M
        self.gmap[t0:t0 + M] = tiles
        if n > 0:   # constant-position initialisation plus a small forward motion
            self.poses[n] = self.poses[n - 1]
            self.poses[n, 0] += 0.05
        return fmap, t0

    def _update(self, fmap, tile0):
        M, n, e = self.M, self.n, self.edges
        coords = ops.update_prologue(self.graph, fmap, self.fmap1, self.fmap2, (n - 1) % self.mem, self.gmap, self.gmap_pm,
                                     tile0, M, self.poses, self.patches, self.intrinsics, e.ii, e.jj, e.kk)
        corr = ops.corr_fused(self.gmap_pm, self.fmap1, self.fmap2, coords, e.kk, e.jj, kmod=M * self.pmem, jmod=self.mem,
                              pixel_major=True)
        # stub of the update operator (net_cdv.py:66-107): a small correction that depends on the correlation
        delta = 0.01 * torch.tanh(corr[0, :, :2].float())
        e.target[0].copy_(coords[0, :, :, 1, 1] + delta)
        e.weight[0].copy_(torch.sigmoid(corr[0, :, 2:4].float()))
        t0 = max(1, n - self.ow)
        ops.ba_forward(self.poses, self.patches, self.intrinsics, e.target, e.weight, self.lmbda, e.ii, e.jj, e.kk, M, t0, n,
                       2, False, U_max=(self.rw + 8) * M, graph=self.graph)
        self.n_updates += 1

    def _keyframe(self, drop):
        M, n = self.M, self.n
        k = n - self.ki
        if drop:   # frame k leaves: every frame buffer shifts down, one launch (slam.py:429-441)
            frames_keyframe_shift([(self.poses, 0), (self.intrinsics, 0), (self.patches.view(self.N, -1), 0),
                                   (self.gmap.view(self.pmem, -1), self.pmem), (self.gmap_pm.view(self.pmem, -1), self.pmem),
                                   (self.fmap1, self.mem), (self.fmap2, self.mem)], k, n)
        self.n = self.edges.keyframe(k, n, M, self.ix, self.rw, drop=drop)

    def frame(self, drop=False):
        """one incoming frame (slam.py:697-720 for an initialised system)"""
        if self.n + 1 >= self.N:
            raise RuntimeError("StreamRunner: frame buffer full")
        fmap, tile0 = self._new_frame()
        self.n += 1
        self.edges.append_frame(self.ix, self.n, self.M, self.r)
        if self.n >= 8:
            self._update(fmap, tile0)
            self._keyframe(drop and self.n > self.ki + 2)
        else:   # before initialisation only the rings are filled
            ops.fmap_ingest(fmap, self.fmap1, self.fmap2, (self.n - 1) % self.mem, gmap=self.gmap, gmap_pm=self.gmap_pm,
                            gmap_first=tile0, gmap_count=self.M)
        return self.n, self.edges.E
