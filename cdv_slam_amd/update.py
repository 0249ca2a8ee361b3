"""The per-frame update hot path as one object: reproject -> 2-level correlation -> neighbors -> BA.

Mirrors the sequence of SLAM.update (cdvslam/slam.py:480-526) with the Update network replaced by
fixed `delta` / `weight` tensors: what bench.py times and what smoke() checks.

HBM-resident state (DESIGN.md "Data layout"):
  poses [N,7] f32, patches [N*M,3,3,3] f32, intrinsics [N,4] f32         (reference layouts)
  ii/jj/kk [E] int64, target/weight [E,2] f32
  gmap  [pmem*M, C, 3, 3] f16                                            (reference layout)
  fmap1 [mem, H+24, W+32, C] f16, fmap2 [mem, H/4+24, W/4+32, C] f16     padded CHANNELS-LAST rings
"""
import torch

from . import ops


class UpdatePath:
    def __init__(self, st, device, overlap=False, sorted_corr=True):
        """st: synth.SynthState (numpy).  Uploads the state once; step() then runs entirely on device.
        overlap: build the patch-graph index and the neighbor lists (they depend on the edge lists only) on a
        second HIP stream while the main stream reprojects and correlates; the streams join before the BA."""
        self.cfg = st.cfg
        self.dev = device
        self.overlap = overlap
        self.sorted_corr = sorted_corr   # correlation processes the edges grouped by target frame (ops.GraphIndex.corr_order_ptr)
        self.fused_prologue = True   # ingest + reproject + index histogram in one launch (cdv_update_prologue)
        self._aux = torch.cuda.Stream(device=device, priority=-1) if overlap else None  # high priority: tiny kernels
        t = lambda a, dt=None: torch.as_tensor(a, device=device) if dt is None else torch.as_tensor(a, dtype=dt, device=device)
        self.poses = t(st.poses).contiguous()
        self.patches = t(st.patches).contiguous()
        self.intrinsics = t(st.intrinsics).contiguous()
        self.ii, self.jj, self.kk = t(st.ii), t(st.jj), t(st.kk)
        self.target, self.weight = t(st.target).contiguous(), t(st.weight).contiguous()
        self.lmbda = torch.tensor([st.lmbda], dtype=torch.float32, device=device)
        self.t0, self.n = st.t0, st.n
        self.M = st.cfg.M
        self.E = st.E
        self._poses0, self._patches0 = self.poses.clone(), self.patches.clone()
        self.U_max = min(self.E, (st.cfg.removal_window + 2) * st.cfg.M) if not st.cfg.fully_connected \
            else st.cfg.frames * st.cfg.M
        # the table form of the index: one slot per patch id of the removal window (ids wrap around it), which is also the
        # number of per-patch rows the bundle adjustment's workspace is sized for
        self.graph = ops.GraphIndex(device, E_cap=self.E, k_range=st.cfg.buffer_size * st.cfg.M,
                                    table_capacity=min(self.U_max, st.cfg.buffer_size * st.cfg.M))
        self.has_features = st.fmap1 is not None
        if self.has_features:
            self.gmap = t(st.gmap).contiguous()
            mem, C, h, w = st.fmap1.shape
            self.fmap1 = ops.alloc_fmap_ring(mem, C, h, w, device)
            self.fmap2 = ops.alloc_fmap_ring(mem, C, h // 4, w // 4, device)
            planar = t(st.fmap1)
            for slot in range(mem):  # fill the rings exactly as the per-frame ingest does
                ops.fmap_ingest(planar[slot], self.fmap1, self.fmap2, slot)
            # the newest frame's planar features, re-ingested every step (slam.py:679-682)
            self.new_frame = planar[(st.n - 1) % mem].clone()
            self.new_slot = (st.n - 1) % mem
            self.kmod, self.jmod = st.cfg.M * st.cfg.pmem, st.cfg.mem
            self.corr_out = torch.empty((1, self.E, 882), dtype=torch.float16, device=device)
            # patch tiles in the pixel-major operand layout; the newest frame's M tiles are re-converted every step
            # together with its feature maps (what patchify + the ring writes do once per frame, net_cdv.py:355-374)
            self.gmap_pm = ops.gmap_to_pixel_major(self.gmap)
            self.new_tiles = ((st.n - 1) % st.cfg.pmem) * st.cfg.M
            # the correlation reads its inputs as one packed stream in processing order, written by the index build from
            # the coordinates the prologue leaves in this buffer (cdv_graph_bind_corr_stream); CDV_CORR_STREAM=0: the
            # separate order[] / coords / kk / jj reads
            import os
            self.corr_stream = os.environ.get("CDV_CORR_STREAM", "1") != "0" and st.cfg.C <= 32
            self.coords_buf = torch.empty((1, self.E, 2, 3, 3), dtype=torch.float32, device=device)
            if self.corr_stream:
                self.graph.bind_corr_stream(self.coords_buf, self.kmod, self.jmod, self.gmap_pm.shape[0], mem)

    def reset(self):
        self.poses.copy_(self._poses0)
        self.patches.copy_(self._patches0)

    # -- hipGraph replay of the whole update ----------------------------------------------------------
    def capture(self, warmup=3):
        """Capture one step() -- ~20 dependent launches, all on one stream, no host sync, fixed buffers --
        into a hipGraph (torch.cuda.CUDAGraph).  step_graph() then replays it with one host call."""
        for _ in range(warmup):
            self.step()
        torch.cuda.synchronize()
        self._graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph):
            self._graph_out = self.step()
        torch.cuda.synchronize()
        return self

    def step_graph(self):
        self._graph.replay()
        return self._graph_out

    def step(self, ingest=True, rebuild_graph=True, iterations=2):
        """One update.  Everything is enqueued on the current stream; no host synchronisation."""
        out = {}
        main = torch.cuda.current_stream()
        if (self.has_features and ingest and rebuild_graph and not self.overlap and self.fused_prologue and self.corr_stream
                and ops.prefer_table() and self.graph.table_capacity and self.n - self.t0 <= 32):
            # (graph.table_capacity == 0: this index has given its table up -- ops._table_still_fits -- and goes on ranked)
            # everything in front of the correlation in TWO launches: ring / tile ingest + the table's fill pass, then slot
            # sort + 3. neighbors (net_cdv.py:102 -> ba.cpp:59-97) + 1. reproject (slam.py:325-329) + the correlation's order
            # and packed input stream
            coords = ops.update_prologue_table(self.graph, self.new_frame, self.fmap1, self.fmap2, self.new_slot, self.gmap,
                                               self.gmap_pm, self.new_tiles, self.M, self.poses, self.patches,
                                               self.intrinsics, self.ii, self.jj, self.kk, coords_out=self.coords_buf)
            out["ix"], out["jx"] = self.graph.neighbors()
            self._stream_ready = True
        elif self.has_features and ingest and rebuild_graph and not self.overlap and self.fused_prologue:
            # ring / tile ingest, 1. reproject (slam.py:325-329) and the first launch of the patch-graph index side by
            # side in ONE launch, then the rest of the index build with 3. neighbors (net_cdv.py:102 -> ba.cpp:59-97)
            coords = ops.update_prologue(self.graph, self.new_frame, self.fmap1, self.fmap2, self.new_slot, self.gmap,
                                         self.gmap_pm, self.new_tiles, self.M, self.poses, self.patches, self.intrinsics,
                                         self.ii, self.jj, self.kk, coords_out=self.coords_buf)
            out["ix"], out["jx"] = self.graph.neighbors()
            self._stream_ready = self.corr_stream
        else:
            self._stream_ready = False
            # patch-graph index (shared by neighbors and BA) + 3. neighbors: launches that only need (jj, kk)
            if self.overlap:
                self._aux.wait_stream(main)      # the previous BA still reads the index this build overwrites
                with torch.cuda.stream(self._aux):
                    self.graph.build(self.jj, self.kk, force=rebuild_graph, with_neighbors=True, ii=self.ii)
                    out["ix"], out["jx"] = self.graph.neighbors()
            else:
                self.graph.build(self.jj, self.kk, force=rebuild_graph, with_neighbors=True, ii=self.ii)
                out["ix"], out["jx"] = self.graph.neighbors()
            if self.has_features and ingest:
                ops.fmap_ingest(self.new_frame, self.fmap1, self.fmap2, self.new_slot, gmap=self.gmap,
                                gmap_pm=self.gmap_pm, gmap_first=self.new_tiles, gmap_count=self.M)
            # 1. reproject (slam.py:325-329)
            coords = ops.transform(self.poses[None], self.patches[None], self.intrinsics[None], self.ii, self.jj,
                                   self.kk, layout_e2pp=True)
        out["coords"] = coords
        self.last_coords = coords
        # 2. correlation, both levels (slam.py:316-323)
        if self.has_features:
            out["corr"] = self.corr_only(coords)
        if self.overlap:
            main.wait_stream(self._aux)
            out["ix"].record_stream(main)
            out["jx"].record_stream(main)
        # 4. bundle adjustment (slam.py:512-515)
        ops.ba_forward(self.poses, self.patches, self.intrinsics, self.target, self.weight, self.lmbda, self.ii,
                       self.jj, self.kk, self.M, self.t0, self.n, iterations, False, U_max=self.U_max, graph=self.graph)
        return out

    # -- measurement helpers (bench.py) --------------------------------------------------------------
    def corr_only(self, coords):
        """just the fused correlation launch (dominant kernel) on the current stream"""
        # processing order by target frame, a by-product of this update's index build (prologue): each XCD's share of
        # the edges then works on ~3 frames' maps
        if getattr(self, "_stream_ready", False) and coords.data_ptr() == self.coords_buf.data_ptr():
            return ops.corr_fused_stream(self.gmap_pm, self.fmap1, self.fmap2, self.graph.corr_records_ptr(), self.E,
                                         out=self.corr_out)
        return ops.corr_fused(self.gmap_pm, self.fmap1, self.fmap2, coords, self.kk, self.jj, kmod=self.kmod,
                              jmod=self.jmod, out=self.corr_out, pixel_major=True,
                              order_ptr=self.graph.corr_order_ptr() if (self.sorted_corr and not self.overlap) else None)

    def stage_times(self, reps=20):
        """median microseconds per stage, each timed with HIP events on the current stream"""
        import numpy as np

        def timed(fn):
            ts = []
            for _ in range(reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fn()
                e1.record()
                e1.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            return float(np.median(ts))

        coords = self.last_coords if hasattr(self, "last_coords") else self.step()["coords"]
        res = {}
        if self.has_features:
            res["ingest"] = timed(lambda: ops.fmap_ingest(self.new_frame, self.fmap1, self.fmap2, self.new_slot,
                                                          gmap=self.gmap, gmap_pm=self.gmap_pm,
                                                          gmap_first=self.new_tiles, gmap_count=self.M))
        res["reproject"] = timed(lambda: ops.transform(self.poses[None], self.patches[None], self.intrinsics[None],
                                                       self.ii, self.jj, self.kk, layout_e2pp=True))
        res["graph_build"] = timed(lambda: self.graph.build(self.jj, self.kk, force=True, with_neighbors=True, ii=self.ii))
        if self.has_features:
            res["corr"] = timed(lambda: self.corr_only(coords))
        res["neighbors"] = timed(lambda: self.graph.neighbors())
        res["ba_2it"] = timed(lambda: ops.ba_forward(self.poses, self.patches, self.intrinsics, self.target,
                                                     self.weight, self.lmbda, self.ii, self.jj, self.kk, self.M,
                                                     self.t0, self.n, 2, False, U_max=self.U_max, graph=self.graph))
        if self.has_features:
            res["prologue_fused"] = timed(lambda: (ops.update_prologue(
                self.graph, self.new_frame, self.fmap1, self.fmap2, self.new_slot, self.gmap, self.gmap_pm, self.new_tiles,
                self.M, self.poses, self.patches, self.intrinsics, self.ii, self.jj, self.kk), self.graph.neighbors()))
            if getattr(self, "corr_stream", False):
                res["prologue_table"] = timed(lambda: (ops.update_prologue_table(
                    self.graph, self.new_frame, self.fmap1, self.fmap2, self.new_slot, self.gmap, self.gmap_pm, self.new_tiles,
                    self.M, self.poses, self.patches, self.intrinsics, self.ii, self.jj, self.kk, coords_out=self.coords_buf),
                    self.graph.neighbors()))
        res["step"] = timed(lambda: self.step())
        return res


class _CorrLayer(torch.autograd.Function):
    """the reference routes every correlation through a torch.autograd.Function that calls the extension
    (cdvslam/altcorr/correlation.py:4-13, `altcorr.corr` = CorrLayer.apply); this is that hop, own wording, forward only"""

    @staticmethod
    def forward(ctx, fmap1, fmap2, coords, ii, jj, radius, dropout, ext):
        out, = ext.forward(fmap1, fmap2, coords, ii, jj, radius)
        return out

    @staticmethod
    def backward(ctx, grad):
        raise NotImplementedError("training path (out of scope)")


class DropinPath:
    """The SAME update written the way the reference writes it -- SLAM.reproject / SLAM.corr (slam.py:316-329), the
    per-frame ring writes (slam.py:679-682), Update's fastba.neighbors (net_cdv.py:102) and fastba.BA (slam.py:512-515,
    fastba/ba.py:8) -- against the module names it imports (cuda_corr, cuda_ba, lietorch_backends, registered by
    install_dropin()) and on the reference's state layouts (planar rings [1,mem,C,h,w], gmap_ [pmem,M,C,3,3]).

    Round 4: what the unchanged caller really hands over, so that nothing here is easier than slam.py --
      * `gmap`, `poses`, `patches`, `intrinsics` are PROPERTIES returning a fresh .view() per access (slam.py:237-251;
        SLAM.corr reads `self.gmap` twice, :321-322);
      * the edge tensors are NEW objects every step: slam.py appends with torch.cat (:331-337), here the newest frame's
        edges are cut off and appended again -- same content, fresh storage, so the index is built once per step;
      * `ii % (M * pmem)`, `jj % mem`, `coords / 1`, `coords / 4` are computed per call as SLAM.corr does;
      * the two correlation calls go through an autograd.Function under autocast and no_grad (slam.py:486,
        altcorr/correlation.py:4-13,74-75); `lmbda` is made per call (slam.py:492);
      * the new frame's tiles are written into gmap_ every step (slam.py:676), so the tile shadow is converted once per step.
    What an unchanged slam.py gets; tests compare it with UpdatePath.step(), bench.py times it as `dropin_fps`."""

    def __init__(self, st, device):
        import importlib
        import cdv_slam_amd
        from . import projective_ops as pops
        from .lietorch import SE3
        # the configuration this SLAM object runs is known here: live patch ids span at most the removal window
        cdv_slam_amd.install_dropin(table_capacity=None if st.cfg.fully_connected
                                    else min(st.E, (st.cfg.removal_window + 2) * st.cfg.M))
        self.cuda_corr, self.cuda_ba = importlib.import_module("cuda_corr"), importlib.import_module("cuda_ba")
        self.pops, self.SE3 = pops, SE3
        t = lambda a: torch.as_tensor(a, device=device)
        cfg = st.cfg
        self.dev = device
        self.M, self.mem, self.pmem, self.n, self.t0, self.C = cfg.M, cfg.mem, cfg.pmem, st.n, st.t0, cfg.C
        self.N = cfg.buffer_size
        self.poses_ = t(st.poses).clone()                                  # pg.poses_ [N,7]
        self.patches_ = t(st.patches).clone().view(self.N, self.M, 3, 3, 3)    # pg.patches_ [N,M,3,3,3]
        self.intrinsics_ = t(st.intrinsics).clone()
        self.ii, self.jj, self.kk = t(st.ii), t(st.jj), t(st.kk)
        self.target, self.weight = t(st.target)[None].contiguous(), t(st.weight)[None].contiguous()
        self.fmap1_ = t(st.fmap1)[None].contiguous()                       # [1, mem, C, h, w]
        self.fmap2_ = torch.nn.functional.avg_pool2d(self.fmap1_[0], 4, 4)[None].contiguous()
        self.pyramid = (self.fmap1_, self.fmap2_)
        self.gmap_ = t(st.gmap).clone().view(self.pmem, self.M, cfg.C, 3, 3)
        self.new_frame = self.fmap1_[0, (st.n - 1) % self.mem].clone()
        self.new_tiles = self.gmap_[(st.n - 1) % self.pmem].clone()
        self.n_new = max(1, min(2 * cfg.patch_lifetime * cfg.M, st.E // 2))     # one frame's forward + backward edges
        self._poses0, self._patches0 = self.poses_.clone(), self.patches_.clone()

    # slam.py:237-251: every access is a fresh view object of the persistent buffer
    poses = property(lambda s: s.poses_.view(1, s.N, 7))
    patches = property(lambda s: s.patches_.view(1, s.N * s.M, 3, 3, 3))
    intrinsics = property(lambda s: s.intrinsics_.view(1, s.N, 4))
    gmap = property(lambda s: s.gmap_.view(1, s.pmem * s.M, s.C, 3, 3))

    def reset(self):
        self.poses_.copy_(self._poses0)
        self.patches_.copy_(self._patches0)

    def reproject(self):
        """SLAM.reproject (slam.py:325-329)"""
        coords = self.pops.transform(self.SE3(self.poses), self.patches, self.intrinsics, self.ii, self.jj, self.kk)
        return coords.permute(0, 1, 4, 2, 3).contiguous()

    def corr(self, coords):
        """SLAM.corr (slam.py:316-323), altcorr.corr = CorrLayer.apply (altcorr/correlation.py:74-75)"""
        ii1 = self.kk % (self.M * self.pmem)
        jj1 = self.jj % self.mem
        corr1 = _CorrLayer.apply(self.gmap, self.pyramid[0], coords / 1, ii1, jj1, 3, 1, self.cuda_corr)
        corr2 = _CorrLayer.apply(self.gmap, self.pyramid[1], coords / 4, ii1, jj1, 3, 1, self.cuda_corr)
        return torch.stack([corr1, corr2], -1).view(1, len(ii1), -1)

    def append_again(self):
        """the edge lists as a frame's append_factors leaves them (slam.py:331-337): new tensors from torch.cat"""
        c = self.ii.numel() - self.n_new
        self.jj = torch.cat([self.jj[:c], self.jj[c:]])
        self.kk = torch.cat([self.kk[:c], self.kk[c:]])
        self.ii = torch.cat([self.ii[:c], self.ii[c:]])

    @torch.no_grad()
    def step(self, ingest=True, iterations=2, pooled=None):
        """pooled: the new frame's level-1 map, if the caller wants a particular rounding of the 4x4 average (tests);
        default torch's avg_pool2d as slam.py:682"""
        out = {}
        if ingest:   # slam.py:676-682
            slot = (self.n - 1) % self.mem
            self.gmap_[(self.n - 1) % self.pmem] = self.new_tiles
            self.fmap1_[:, slot] = self.new_frame
            self.fmap2_[:, slot] = torch.nn.functional.avg_pool2d(self.new_frame[None], 4, 4)[0] if pooled is None else pooled
            self.append_again()
        coords = self.reproject()
        with torch.autocast("cuda", enabled=True):      # slam.py:486
            out["corr"] = self.corr(coords)
            # Update.forward (net_cdv.py:102)
            out["ix"], out["jx"] = self.cuda_ba.neighbors(self.kk, self.jj)
        out["coords"] = coords
        lmbda = torch.as_tensor([1e-4], device=self.dev)       # slam.py:492
        # fastba.BA (fastba/ba.py:8)
        out["ba"] = self.cuda_ba.forward(self.poses.data, self.patches, self.intrinsics, self.target, self.weight,
                                         lmbda, self.ii, self.jj, self.kk, self.M, self.t0, self.n, iterations, False)
        return out
