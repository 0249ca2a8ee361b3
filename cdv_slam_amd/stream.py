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
                 global_opt_freq=15, backend_thresh=64.0, pose_init=None, record_global=False, keyframe_thresh=12.5, gain=0.01,
                 pose_step=0.05):
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
        self.pose_step = pose_step
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
            self.poses[n, 0] += self.pose_step
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

    def counts(self):
        """(n, E): host integers here (DeviceStreamRunner has to ask the device)"""
        return self.n, self.edges.E


class _EdgeViews:
    """the live part of the device-resident edge lists, reference names (after a synchronisation: the owner asked for it)"""

    def __init__(self, owner, E):
        o = owner
        self.E = E
        c = o.cur
        self.ii, self.jj, self.kk = o._ii[c, :E], o._jj[c, :E], o._kk[c, :E]
        self.target, self.weight = o._target[c, :E][None], o._weight[c, :E][None]


class DeviceStreamRunner:
    """StreamRunner with every size on the device (include/cdvslam_hip.h "a frame stream whose sizes live on the DEVICE").

    The reference reads its keyframe decision back to the host (two .item() calls, cdvslam/slam.py:399-413) and from there on
    n and the number of edges are host integers; StreamRunner above does the same (one read-back per removal).  Here a frame
    is a fixed sequence of 12 launches and NO synchronisation:

        cdv_stream_frame_begin        n + 1, the frame's edges, patches_, pose guess, patch tiles          slam.py:676-709
        cdv_update_prologue_table_dyn ring ingest | table fill, then sort + neighbors + reprojection + packed stream
        cdv_corr_fused_stream_dyn     two-level correlation                                              slam.py:316-323
        cdv_stream_operator_stub      (stands where the Update operator runs)
        cdv_ba_forward_dyn            BA(iterations=2) over [max(1, n - OPTIMIZATION_WINDOW), n)            slam.py:512-515
        cdv_stream_keyframe           point cloud of the removal window | flow statistic -> decision ON THE DEVICE -> ONE
                                      compaction for both removals + index shift | frame-buffer shift  slam.py:399-458,524-526
    n, E, the decision live in a ring of "dynamic blocks"; the host only keeps an upper bound of E (from a pinned word the
    last launch of a frame writes) to dimension the launches.  Anything the host wants to KNOW (counts(), n, edges, poses of
    the keyframes) synchronises -- tests and the end of a run do, the frame loop does not.  Configurations: 3 x 3 patches,
    OPTIMIZATION_WINDOW <= 32 (<= 10: the window solver; beyond: the 10 < N <= 32 path, e.g. default_cdvo++.yaml's 22), no
    loop closure (StreamRunner serves that: its proximity search and greedy edge selection are host logic in the reference
    too, patchgraph.py:71-97)."""

    def __init__(self, device, M=96, ht=384, wd=512, C=24, mem=36, pmem=36, buffer_size=512, patch_lifetime=13,
                 removal_window=22, opt_window=10, keyframe_index=4, seed=1234, keyframe_thresh=12.5, gain=0.01, pose_step=0.05):
        import ctypes
        from . import _lib
        self.pose_step = pose_step
        if opt_window > 32:
            raise NotImplementedError("DeviceStreamRunner: OPTIMIZATION_WINDOW <= 32 (cdv_ba_forward_dyn)")
        self.lib = lib = _lib.load()
        self.dev = device
        self.M, self.C, self.mem, self.pmem, self.N = M, C, mem, pmem, buffer_size
        self.h, self.w = ht // 4, wd // 4
        self.r, self.rw, self.ow, self.ki = patch_lifetime, removal_window, opt_window, keyframe_index
        self.kthresh, self.gain = keyframe_thresh, gain
        f32 = dict(dtype=torch.float32, device=device)
        self.poses = torch.zeros((self.N, 7), **f32); self.poses[:, 6] = 1.0
        self.patches = torch.zeros((self.N * M, 3, 3, 3), **f32)
        intr = torch.tensor([wd / 2.0, wd / 2.0, wd / 2.0, ht / 2.0]) / 4.0
        self.intrinsics = intr.to(device).repeat(self.N, 1).contiguous()
        self.ix = torch.arange(self.N, device=device).repeat_interleave(M)
        self.points = torch.zeros((self.N * M, 3), **f32)
        self.fmap1 = ops.alloc_fmap_ring(mem, C, self.h, self.w, device)
        self.fmap2 = ops.alloc_fmap_ring(mem, C, self.h // 4, self.w // 4, device)
        self.gmap = torch.zeros((pmem * M, C, 3, 3), dtype=torch.float16, device=device)
        self.gmap_pm = torch.zeros((pmem * M, 9, C), dtype=torch.float16, device=device)
        self.ecap = ecap = M * (removal_window + 6) * 2 * patch_lifetime
        self.icap = icap = ecap + buffer_size * M * 2 * patch_lifetime
        z = lambda *s, dt=torch.int64: torch.zeros(s, dtype=dt, device=device)
        self._ii, self._jj, self._kk = z(2, ecap), z(2, ecap), z(2, ecap)                  # [0]: the lists, [1]: their twin
        self._target, self._weight = z(2, ecap, 2, dt=torch.float32), z(2, ecap, 2, dt=torch.float32)
        self.ii_inac, self.jj_inac, self.kk_inac = z(icap), z(icap), z(icap)
        self.target_inac, self.weight_inac = z(icap, 2, dt=torch.float32), z(icap, 2, dt=torch.float32)
        self.dyn = torch.zeros((8, 16), dtype=torch.int32, device=device)                  # ring of dynamic blocks
        self.slot, self.cur = 0, 0
        self.mirror = torch.zeros(1, dtype=torch.int64).pin_memory()                       # (frames << 32 | edges), written by the device
        self.ws = torch.zeros(lib.cdv_stream_workspace_bytes(ecap, M), dtype=torch.uint8, device=device)
        self.tcap = (removal_window + 8) * M
        self.graph = ops.GraphIndex(device, E_cap=ecap, k_range=(removal_window + 8) * M + M * pmem, table_capacity=self.tcap)
        self.coords_buf = torch.empty((1, ecap, 2, 3, 3), **f32)
        self.graph.bind_corr_stream(None, M * pmem, mem, pmem * M, mem)
        self.corr_out = torch.empty((1, ecap, 882), dtype=torch.float16, device=device)
        self.lmbda = torch.tensor([1e-4], **f32)
        self.ba_ws = ops.ba_private_workspace(device, ecap, self.tcap, opt_window, label="DeviceStreamRunner")
        self.events = self.ba_ws.events      # this stream's own failure events (ops.EventBlock; no synchronisation to read)
        if opt_window > 10 and M % 4 == 0 and M >= 16:
            # a frame's patches are M consecutive slots of the table (its capacity is a multiple of M): the 10 < N <= 32 path may
            # cut its workgroups per frame (cuda_ba.forward's PPF argument does the same for the drop-in caller)
            _lib.check(lib.cdv_ba_set_patches_per_frame(ctypes.c_void_p(self.ba_ws.data_ptr()), M), "cdv_ba_set_patches_per_frame")
        self.frames = 0
        # the stubbed feature network: a pool of feature maps and, drawn ON THE DEVICE once, every frame's patch centres / depths
        g = torch.Generator(device=device).manual_seed(seed)
        self.pool = [(torch.randn((C, self.h, self.w), generator=g, device=device) / 4).half() for _ in range(4)]
        u = torch.rand((self.N, 3, M), generator=g, device=device)
        self._draws = torch.stack([u[:, 0] * (self.w - 16) + 8, u[:, 1] * (self.h - 16) + 8, u[:, 2] * 0.75 + 0.25], 1).contiguous()
        # the frame buffers keyframe() shifts (slam.py:431-441), as one descriptor list
        bufs = [(self.poses, 0), (self.intrinsics, 0), (self.patches.view(self.N, -1), 0), (self.gmap.view(pmem, -1), pmem),
                (self.gmap_pm.view(pmem, -1), pmem), (self.fmap1, mem), (self.fmap2, mem)]
        self._bufs = (_lib.FrameBuf * len(bufs))()
        for a, (t, m) in zip(self._bufs, bufs):
            a.base, a.slot_bytes, a.modulus, a.reserved = t.data_ptr(), t[0].numel() * t.element_size(), int(m), 0
        self._nbufs = len(bufs)
        self._p = lambda t: ctypes.c_void_p(t.data_ptr())
        self._cast = ctypes.cast

    # ---- what the host may look at (synchronises) --------------------------------------------------------------
    def counts(self):
        """(n, E) after everything enqueued so far; raises if a capacity was exceeded on the device"""
        blk = self.dyn[self.slot].cpu()
        if int(blk[7]):
            raise RuntimeError("DeviceStreamRunner: capacity exceeded on the device (code %d): edges %d / %d, inactive %d / %d, "
                               "frames %d / %d" % (int(blk[7]), int(blk[1]), self.ecap, int(blk[2]), self.icap, int(blk[0]), self.N))
        return int(blk[0]), int(blk[1])

    n = property(lambda s: s.counts()[0])
    edges = property(lambda s: _EdgeViews(s, s.counts()[1]))
    E_inac = property(lambda s: int(s.dyn[s.slot, 2].item()))

    @property
    def last_motion(self):
        """the flow statistic of the last keyframe test (slam.py:409-413), None before the first one"""
        if self.frames < 8:
            return None
        p = self.lib.cdv_stream_motion(self._p(self.ws), self.ecap, self.M)
        off = (p - self.ws.data_ptr()) // 4
        return float(self.ws.view(torch.float32)[off].item())

    def _edge_bound(self):
        """an upper bound of the number of edges once the frame about to begin has arrived, without asking the device: what
        the last finished keyframe() left (pinned word) plus 2 r M per frame begun since"""
        v = int(self.mirror[0])
        seen_frame, seen_E = v >> 32, v & 0xFFFFFFFF
        return min(self.ecap, seen_E + (self.frames + 1 - seen_frame) * 2 * self.r * self.M)

    def _descriptor(self):
        """cdv_stream_desc: every buffer and size of the stream, once; a frame is then ONE ctypes call"""
        from . import _lib
        D = _lib.StreamDesc()
        D.M, D.C, D.H, D.W, D.mem, D.pmem, D.frames_capacity = self.M, self.C, self.h, self.w, self.mem, self.pmem, self.N
        D.patch_lifetime, D.removal_window, D.opt_window, D.keyframe_index = self.r, self.rw, self.ow, self.ki
        D.keyframe_thresh, D.gain, D.pose_step = self.kthresh, self.gain, self.pose_step
        D.slot, D.frames, D.cur = self.slot, self.frames, self.cur
        g = self.graph
        D.edge_capacity, D.inactive_capacity, D.table_capacity = self.ecap, self.icap, g.table_capacity
        D.graph_E_max, D.graph_k_range, D.graph_ws_bytes, D.ba_ws_bytes = g.E_cap, g.k_range, g.ws_bytes, self.ba_ws.numel()
        for name, t in (("poses", self.poses), ("patches", self.patches), ("intrinsics", self.intrinsics), ("points", self.points),
                        ("ix", self.ix), ("fmap1_nhwc", self.fmap1), ("fmap2_nhwc", self.fmap2), ("gmap_planar", self.gmap),
                        ("gmap_pm", self.gmap_pm), ("ii_inac", self.ii_inac), ("jj_inac", self.jj_inac), ("kk_inac", self.kk_inac),
                        ("target_inac", self.target_inac), ("weight_inac", self.weight_inac), ("coords", self.coords_buf),
                        ("corr_out", self.corr_out), ("lmbda", self.lmbda), ("dyn", self.dyn), ("ws", self.ws), ("graph_ws", g.ws),
                        ("ba_ws", self.ba_ws)):
            setattr(D, name, t.data_ptr())
        D.mirror_host = self.mirror.data_ptr()
        for name, t in (("ii", self._ii), ("jj", self._jj), ("kk", self._kk), ("target", self._target), ("weight", self._weight)):
            arr = getattr(D, name)
            arr[0], arr[1] = t[0].data_ptr(), t[1].data_ptr()
        D.n_bufs = self._nbufs
        for i in range(self._nbufs):
            D.bufs[i] = self._bufs[i]
        self._desc = D
        self._desc_ref = self._cast(ops.ctypes.pointer(D), ops.ctypes.c_void_p)
        V = ops.ctypes.c_void_p
        self._draw_ptrs = [tuple(V(self._draws.data_ptr() + 4 * self.M * (3 * f + k)) for k in range(3)) for f in range(self.N)]
        self._pool_ptrs = [V(t.data_ptr()) for t in self.pool]

    # ---- two frames as one hipGraph ------------------------------------------------------------------------------
    def capture_pair(self):
        """Capture TWO frames (24 launches) as one hipGraph and return replay(): with every size on the device nothing in a
        frame's launches depends on the host any more -- the dynamic blocks form a ring of two (a frame ends in the block it
        started from), the edge lists are back in their first twin after two frames, the launches are sized for the edge
        capacity.  The frames' inputs are read from fixed staging buffers (`stage_inputs(k, fmap, cx, cy, d)`, k = 0, 1) and
        the keyframe test is the reference's own (decided on the device).  Call after initialisation (>= 8 frames) with an
        even number of frames begun; the runner goes on in graph mode (frame() keeps working: same descriptor)."""
        if self.frames < 8 or self.cur != 0:
            raise RuntimeError("capture_pair: after initialisation, with the edge lists in their first twin (an even number of "
                               "frames since frame 8)")
        if not hasattr(self, "_desc"):
            self._descriptor()
        torch.cuda.synchronize()
        D = self._desc
        # continue from the current block in a ring of two: copy it to block 0
        self.dyn[0].copy_(self.dyn[self.slot].clone())
        D.slot, D.ring_blocks, D.fixed_bound = 0, 2, 1
        self.slot = 0
        M = self.M
        self._stage = [(torch.empty_like(self.pool[0]), torch.empty(3, M, dtype=torch.float32, device=self.dev)) for _ in range(2)]
        for k in range(2):
            self.stage_inputs(k, self.pool[k], *self._draws[(self.frames + k) % self.N])
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        frames0 = self.frames
        with torch.cuda.graph(g):
            for k in range(2):
                fm, dr = self._stage[k]
                self.frame(drop=None, inputs=(fm, dr[0], dr[1], dr[2]))
        torch.cuda.synchronize()
        # the capture enqueued nothing: the host counters it advanced are taken back, replay() advances them
        D.frames, D.slot, D.cur = frames0, 0, 0
        self.frames, self.slot, self.cur = frames0, 0, 0
        self._graph = g

        def replay():
            g.replay()
            self.frames += 2
            D.frames = self.frames
        return replay

    def stage_inputs(self, k, fmap, cx, cy, d):
        """the (stubbed) network outputs of frame k (0 or 1) of the next replayed pair"""
        fm, dr = self._stage[k]
        fm.copy_(fmap)
        dr[0].copy_(cx); dr[1].copy_(cy); dr[2].copy_(d)

    def frame(self, drop=False, inputs=None):
        """one incoming frame, enqueued (cdv_stream_frame: 12 launches behind ONE call); nothing is read back.  drop: None =
        the reference's keyframe test on the device, True / False = the caller decides; inputs: (fmap [C,h,w] f16, cx, cy,
        d [M]) on the device, default: the stub's own"""
        if not hasattr(self, "_desc"):
            self._descriptor()
        f = self.frames
        if inputs is None:
            fmap = self._pool_ptrs[f % len(self._pool_ptrs)]
            cx, cy, d = self._draw_ptrs[f % self.N]      # (the stub's draws repeat after buffer_size frames)
        else:
            self._hold = tuple(t.contiguous() for t in inputs)           # alive until the launches that read them are enqueued
            fmap, cx, cy, d = (self._p(t) for t in self._hold)
        force = -1 if drop is None else (1 if drop else 0)
        ops._lib.check(self.lib.cdv_stream_frame(self._desc_ref, fmap, cx, cy, d, force, ops._stream()), "cdv_stream_frame")
        self.frames, self.slot, self.cur = self._desc.frames, self._desc.slot, self._desc.cur
        g = self.graph
        g.is_table, g._key, g._nbr = True, None, None
