"""torch-level entry points over the C ABI (include/cdvslam_hip.h).

PyTorch is plumbing here: it owns device memory and the current HIP stream; every function below
only passes raw device pointers and sizes to libcdvslam_hip.so.  No CPU fallback exists -- a missing
library or a non-CUDA(ROCm) tensor raises.
"""
import ctypes
import os

import torch

from . import _lib

F16, F32, F64 = 0, 1, 2
_DT = {torch.float16: F16, torch.float32: F32, torch.float64: F64}
LIE_OPS = {"exp": 0, "log": 1, "inv": 2, "mul": 3, "adj": 4, "adjT": 5, "act": 6, "act4": 7, "matrix": 8}

# default index-range capacities of the graph workspace (reference buffers: BUFFER_SIZE = 4096 frames
# x 96 patches, cdvslam/config.py, patchgraph.py:25-29)
DEFAULT_K_RANGE = 4096 * 96


try:   # the raw handle of torch's current stream without building a Stream object (5 us per call through the public API:
    # an update through the drop-in names makes eight such calls)
    _raw_stream, _raw_device = torch._C._cuda_getCurrentRawStream, torch._C._cuda_getDevice
except AttributeError:   # a torch build without the raw accessors
    _raw_stream = _raw_device = None


def _stream():
    """the current stream's handle as an int (the signatures in _lib declare void*: ctypes converts, no object per call)"""
    if _raw_stream is not None:
        return _raw_stream(_raw_device())
    return torch.cuda.current_stream().cuda_stream


_ENV_RAW = getattr(os.environ, "_data", None)      # os._Environ keeps the encoded pairs in a plain dict (CPython, posix)
if not isinstance(_ENV_RAW, dict) or (len(_ENV_RAW) and not isinstance(next(iter(_ENV_RAW)), bytes)):
    _ENV_RAW = None


def _env(name, default):
    """os.environ.get(name, default), read at every call (the switches are documented as live) without the encode /
    decode round trip of os.environ: 0.1 us instead of 0.6 -- an update through the drop-in names asks ten times"""
    if _ENV_RAW is None:
        return os.environ.get(name, default)
    v = _ENV_RAW.get(name.encode())
    return default if v is None else v.decode()


def _p(t):
    """device address of a tensor for a void* parameter (an int: ctypes converts; None = NULL) -- an update through the drop-in
    names passes some fifty of them"""
    return None if t is None else t.data_ptr()


def _ident(t):
    """What a tensor IS for the caches below: the memory it views (address, shape, strides, dtype) and the version counter of
    that memory.  NOT the Python object: the reference hands over fresh views of its buffers at every access (SLAM.gmap,
    .poses, .patches are properties that call .view(), cdvslam/slam.py:237-251) and fresh edge tensors every frame
    (torch.cat, slam.py:331-337); a view shares its base's version counter, so `same ident` means `same bytes` as long as
    whoever compares also HOLDS a tensor on that storage (a freed address can be handed out again -- every cache below
    keeps the tensor it took the ident from).  A write through an alias with a version counter of its own (`t.data`,
    from_blob, dlpack) is invisible here, as it is to autograd."""
    return (t.data_ptr(), t._version, t.shape, t.stride(), t.dtype)


def _place(t):
    """_ident without the version: the place, whatever it holds at the moment"""
    return (t.data_ptr(), t.shape, t.stride(), t.dtype)


def touched(*ts):
    """tell the version counters that a kernel wrote these tensors through their raw pointers"""
    for t in ts:
        torch.autograd.graph.increment_version(t)


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("cdv_slam_amd: tensors must live on the GPU (HIP path only, no CPU fallback)")


def _sync_check():
    return _env("CDV_CHECK", "0") == "1"


def version():
    return _lib.load().cdv_version().decode()


# ---------------------------------------------------------------------------------------------------
# the compiled steady state of the drop-in modules (csrc/dropin_fast.cpp -> cdv_slam_amd/_dropin_fast.so)
# ---------------------------------------------------------------------------------------------------
# The reference binds its kernels with a torch extension (correlation.cpp:57-63, ba.cpp:183-188); the drop-in modules here
# are Python over the C ABI, and the caches below that make an unchanged slam.py fast cost 15-55 us of Python per call.  For
# the steady state -- the same call sequence on tensors that look like those of the update before -- the same bookkeeping
# runs compiled: the code below HANDS the state over ("arms") once it has served a complete pair / a complete neighbors +
# BA itself, asks the extension first at the next call, and TAKES the state back ("disarms") before any Python code touches
# the shared shadows or workspaces again.  Whatever the extension does not recognise it declines (None) and the Python code
# -- the authority on semantics -- serves the call.  CDV_DROPIN_FAST=0 (read at every call) switches the lane off.
_FAST_SYMS = ("cdv_shadows_sync", "cdv_corr_fused", "cdv_corr_level_checked_interleaved",
              "cdv_graph_build_table", "cdv_ba_workspace_bytes", "cdv_ba_forward", "cdv_transform")
_fast = None           # the bound extension module; False: not available (warned once)
_armed_pair = None     # (the _LevelPairing, ring entry A, ring entry B, the TileCache) whose state the extension holds
_armed_graph = None    # (token, the GraphIndex) whose workspace the extension builds in and runs the BA over


def _fast_mod():
    global _fast
    if _fast is None:
        try:
            from . import _dropin_fast as m
            lib = _lib.load()
            m.bind({n: ctypes.cast(getattr(lib, n), ctypes.c_void_p).value for n in _FAST_SYMS})
            _fast = m
            import atexit
            atexit.register(_fast_release)      # the extension holds tensors: they go before the interpreter tears torch down
        except ImportError as e:
            import warnings
            warnings.warn("cdv_slam_amd: the compiled drop-in bookkeeping (_dropin_fast.so, `make -C cdv_slam_amd/csrc`) is not "
                          "available (%s); the Python bookkeeping serves every call" % e, RuntimeWarning)
            _fast = False
    return _fast


def _fast_release():
    if _fast:
        _fast.drop_pending()
        _fast.disarm_pair()
        _fast.disarm_graph()


def fast_lane_enabled():
    return _env("CDV_DROPIN_FAST", "1") != "0" and bool(_fast_mod())


def _disarm_pair():
    """take the ring / tile shadows' state back from the extension (before Python code syncs or converts them itself)"""
    global _armed_pair
    if _armed_pair is None:
        return
    pairing, ea, eb, tiles = _armed_pair
    _armed_pair = None
    st = _fast.disarm_pair()
    pairing.fast_token = 0
    ea.pop("fast", None)
    eb.pop("fast", None)
    tiles.fast = False
    if st is not None:
        ea["version"], ea["parity"] = st["A_version"], st["A_parity"]
        eb["version"], eb["parity"] = st["B_version"], st["B_parity"]
        src = st["tiles_src"]
        tiles.src, tiles.ident = src, (src.data_ptr(), st["tiles_version"], src.shape, src.stride(), src.dtype)


def _disarm_graph(g=None):
    """take the per-device index workspace back (g given: only if it is the armed one), with the record of what it holds"""
    global _armed_graph
    if _armed_graph is None or (g is not None and _armed_graph[1] is not g):
        return
    g = _armed_graph[1]
    _armed_graph = None
    st = _fast.disarm_graph()
    g._key = g._nbr = None
    if st is not None and st["has_key"]:      # the index the extension left in the workspace, as build_table() would record it
        idn = lambda t, v: (t.data_ptr(), v, t.shape, t.stride(), t.dtype)
        jj, kk, ii = st["jj"], st["kk"], st["ii"]
        g._key = (idn(jj, st["jj_version"]), idn(kk, st["kk_version"]), None if ii is None else idn(ii, st["ii_version"]),
                  (jj, kk, ii), "table")
        g.E, g.is_table = kk.numel(), True
        if st["ix"] is not None:
            g._nbr = (st["ix"], st["jx"])


# ---------------------------------------------------------------------------------------------------
# patch-graph index
# ---------------------------------------------------------------------------------------------------

class GraphIndex:
    """Device workspace holding unique(kk) and the patch CSR (edges of every patch in (jj, id) order)."""

    def __init__(self, device, E_cap=1 << 16, k_range=DEFAULT_K_RANGE, table_capacity=None):
        """k_range: capacity of the ranked build's id range; table_capacity: slots of the table build (slot = id mod
        capacity) -- at least the ids between the oldest and newest patch with an edge, (REMOVAL_WINDOW + 2) x
        PATCHES_PER_FRAME for a SLAM object; None / 0: this workspace only ever holds the ranked index"""
        self.lib = _lib.load()
        self.device = device
        self.E_cap, self.k_range = 0, k_range
        self.table_capacity = min(k_range, 1 << 16, int(table_capacity)) if table_capacity else 0   # 0: ranked index only
        self.ws = None
        self._key = None
        self.E = 0
        self.is_table = False
        self.n_builds = 0          # index builds enqueued on this workspace (diagnostics, tests)
        self._events = None        # EventBlock of the bundle adjustments run over THIS index (created at the first one)
        self._graph_events_seen = 0
        self._reserve(E_cap)

    @property
    def events(self):
        """the failure events of the bundle adjustments that ran over this index (ops.EventBlock): counted per index, so that
        what happens to one path's graph -- a stream runner's table overflowing -- is nobody else's business"""
        if self._events is None:
            self._events = EventBlock(self.device, "GraphIndex@%x" % id(self))
        return self._events

    def _reserve(self, E):
        if self.ws is not None and E <= self.E_cap:
            return
        _disarm_graph(self)
        self.E_cap = max(E, int(self.E_cap * 1.5), 1024)
        nbytes = self.lib.cdv_graph_workspace_bytes(self.E_cap, self.k_range)
        self.ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        self.ws_bytes = nbytes
        # whoever allocates initialises: nothing depends on what the library remembers about this address
        _lib.check(self.lib.cdv_graph_workspace_init(_p(self.ws), nbytes, self.E_cap, self.k_range, _stream()),
                   "cdv_graph_workspace_init")
        if getattr(self, "_stream_cfg", None) is not None:     # the correlation stream binding lives with the workspace
            self.bind_corr_stream(*self._stream_cfg)
        self._key = None

    @staticmethod
    def _make_key(jj, kk, ii):
        """memory + version (_ident) of every tensor whose CONTENT is baked into the index: jj, kk, and ii when the build was
        given it (the per-patch edge records carry the source frames the N <= 32 bundle adjustment reads).  Strong
        references pin the storages, so the same ident implies the same content -- for the very objects and for any other
        view of the same memory (slam.py hands the same edge tensors to neighbors() and BA(), EdgeStore fresh views)."""
        return (_ident(jj), _ident(kk), None if ii is None else _ident(ii), (jj, kk, ii))

    def _same_key(self, key):
        """is the index in the workspace the one `key` describes?  A build WITH ii also serves a request without it
        (neighbors() after the prologue), and a build WITHOUT ii serves a request with any ii (cuda_ba.neighbors builds it,
        cuda_ba.forward then reads ii per edge: one more dependent load instead of a second build); a build with another or
        a modified ii does not serve a request with ii."""
        k = self._key
        if k is None or k[0] != key[0] or k[1] != key[1]:
            return False
        if key[2] is None or k[2] is None:
            return True      # nothing of ii is baked into an index built without it: the bundle adjustment reads ii itself then
        return k[2] == key[2]

    def corr_order_ptr(self):
        """device pointer (ctypes.c_void_p) of the correlation's processing order of the index currently in the
        workspace (edges grouped by target frame; cdv_graph_corr_order), for ops.corr_fused(order_ptr=...)"""
        p = self.lib.cdv_graph_corr_order(_p(self.ws))
        return ctypes.c_void_p(p) if p else None

    def bind_corr_stream(self, coords, kmod, jmod, Ng, slots, scale0=1.0):
        """cdv_graph_bind_corr_stream: the following builds on this workspace also write the correlation's packed input
        stream (coordinates + reduced ring indices in processing order) from `coords` [1,E,2,3,3] f32 -- a buffer that is
        written earlier on the stream (update_prologue(coords_out=...)).  coords None unbinds."""
        if coords is not None:
            _need_cuda(coords)
            if coords.dtype != torch.float32 or not coords.is_contiguous():
                raise TypeError("bind_corr_stream: coords must be a contiguous float32 buffer")
        _lib.check(self.lib.cdv_graph_bind_corr_stream(_p(self.ws), _p(coords), int(kmod), int(jmod), int(Ng), int(slots), float(scale0)),
                   "cdv_graph_bind_corr_stream")
        self._stream_coords = coords          # pinned: the builds read it
        self._stream_cfg = (coords, kmod, jmod, Ng, slots, scale0)

    def corr_records_ptr(self):
        p = self.lib.cdv_graph_corr_records(_p(self.ws))
        return ctypes.c_void_p(p) if p else None

    def build(self, jj, kk, force=False, with_neighbors=False, ii=None):
        """Enqueue the index build for (jj, kk).  Re-used when called again with the same, unmodified
        tensor objects (neighbors() and BA() of one update share one build).  with_neighbors: the build also
        produces fastba.neighbors(kk, jj) (picked up by neighbors() without another launch).  ii (optional): the
        source frames, copied into the per-patch edge records the bundle adjustment walks."""
        _disarm_graph(self)
        _need_cuda(jj, kk)
        if jj.dtype != torch.int64 or kk.dtype != torch.int64:
            raise TypeError("index tensors must be int64")
        jj, kk = jj.contiguous(), kk.contiguous()
        if ii is not None:
            ii = ii.contiguous()
        key = self._make_key(jj, kk, ii)
        if not force and not self.is_table and self._same_key(key):
            return self
        E = kk.numel()
        if jj.numel() != E:
            raise ValueError("jj and kk must have the same length")
        self.is_table = False
        self._reserve(E)
        self._nbr = None
        if ii is not None:
            _need_cuda(ii)
            if ii.dtype != torch.int64 or ii.numel() != E:
                raise TypeError("ii must be int64 with one entry per edge")
        ix = jx = None
        if with_neighbors and E > 0:
            ix = torch.empty(E, dtype=torch.int64, device=self.device)
            jx = torch.empty(E, dtype=torch.int64, device=self.device)
            self._nbr = (ix, jx)
        rc = self.lib.cdv_graph_build_edges(_p(ii), _p(jj), _p(kk), E, _p(self.ws), self.ws_bytes, self.E_cap,
                                            self.k_range, _p(ix), _p(jx), _stream())
        _lib.check(rc, "cdv_graph_build_edges")
        self.n_builds += 1
        self._key = key  # strong refs pin the tensors so that identity implies content
        self.E = E
        if _sync_check():
            self.meta()
        return self

    def index_for_ba(self, jj, kk, ii, N):
        """make sure the workspace holds an index of (ii, jj, kk) the bundle adjustment with N free poses can use: either
        form for N <= 32 (whatever is there and current is kept; otherwise the preferred one is built, CDV_INDEX), the ranked
        one for the global bundle adjustment"""
        _disarm_graph(self)
        if N > 32 or N < 1:      # global bundle adjustment, and the structure-only call (no free pose): ranked index
            return self.build(jj, kk, ii=ii)
        key = self._make_key(jj, kk, ii)
        if self._key is not None and self._same_key(key):
            return self
        return self.build_table(jj, kk, ii=ii) if (prefer_table() and self.table_capacity) else self.build(jj, kk, ii=ii)

    def build_table(self, jj, kk, ii=None, force=False, with_neighbors=False):
        """cdv_graph_build_table: the same index as a patch table (slot = patch id; two launches, no scan).  Serves
        neighbors(), the window / mid bundle adjustment (N <= 32) and the correlation's order + packed stream; not
        unique() (no ranks) and not the global bundle adjustment -- build() is there for those.  Same caching rule as
        build(): identity + version of the tensors."""
        _disarm_graph(self)
        _need_cuda(jj, kk)
        if not self.table_capacity:
            raise RuntimeError("GraphIndex.build_table: this workspace was created without a table capacity")
        if jj.dtype != torch.int64 or kk.dtype != torch.int64:
            raise TypeError("index tensors must be int64")
        jj, kk = jj.contiguous(), kk.contiguous()
        if ii is not None:
            ii = ii.contiguous()
        key = self._make_key(jj, kk, ii) + ("table",)
        if not force and self.is_table and self._same_key(key):
            return self
        E = kk.numel()
        if jj.numel() != E or (ii is not None and (ii.dtype != torch.int64 or ii.numel() != E)):
            raise ValueError("ii / jj / kk must be int64 tensors of one length")
        self._reserve(E)
        self._nbr = None
        ix = jx = None
        if with_neighbors and E > 0:
            ix = torch.empty(E, dtype=torch.int64, device=self.device)
            jx = torch.empty(E, dtype=torch.int64, device=self.device)
            self._nbr = (ix, jx)
        rc = self.lib.cdv_graph_build_table(_p(ii), _p(jj), _p(kk), E, _p(self.ws), self.ws_bytes, self.E_cap, self.k_range,
                                            self.table_capacity, _p(ix), _p(jx), _stream())
        _lib.check(rc, "cdv_graph_build_table")
        self.n_builds += 1
        self._key = key
        self.E = E
        self.is_table = True
        if _sync_check():
            self.meta()
        return self

    def table_arrays(self):
        """(deg [capacity] int32, plo [capacity] int32, records [chunks*32*16, 4] int32, overflow records [E_cap + 1, 4]
        int32, order [E_cap] int32, stream [E_cap, 24] int32, patch id per slot [capacity] int32): views into the
        workspace (tests, tools)"""
        off = (ctypes.c_int64 * 7)()
        _lib.check(self.lib.cdv_graph_table_offsets(self.E_cap, self.k_range, off), "cdv_graph_table_offsets")
        w32 = self.ws.view(torch.int32)
        o = [int(v) // 4 for v in off]
        R, chunks = self.table_capacity, (self.table_capacity + 15) // 16
        return (w32[o[0]:o[0] + R], w32[o[1]:o[1] + R], w32[o[2]:o[2] + chunks * 32 * 16 * 4].view(-1, 4),
                w32[o[3]:o[3] + (self.E_cap + 1) * 4].view(-1, 4), w32[o[4]:o[4] + self.E_cap],
                w32[o[5]:o[5] + self.E_cap * 24].view(-1, 24), w32[o[6]:o[6] + R])

    def meta(self):
        """(U, is_table, kmin, kmax, jmin, jmax, error, E) -- synchronises the stream."""
        m = (ctypes.c_int64 * 8)()
        _lib.check(self.lib.cdv_graph_read_meta_host(_p(self.ws), m, _stream()), "cdv_graph_read_meta_host")
        m = list(m)
        if m[6]:
            raise _lib.CdvError("patch-graph index: patch-id range exceeds the workspace capacity "
                                "(k_range=%d; got k in [%d,%d])" % (self.k_range, m[2], m[3]))
        return m

    def neighbors(self):
        _disarm_graph(self)
        if getattr(self, "_nbr", None) is not None:
            return self._nbr
        ix = torch.empty(self.E, dtype=torch.int64, device=self.device)
        jx = torch.empty(self.E, dtype=torch.int64, device=self.device)
        _lib.check(self.lib.cdv_neighbors(_p(self.ws), self.E, _p(ix), _p(jx), _stream()), "cdv_neighbors")
        self._nbr = (ix, jx)       # the index has not changed when somebody asks again
        return ix, jx

    def unique(self):
        """(kx, ku) == torch._unique(kk, sorted=True, return_inverse=True); one host sync for U."""
        _disarm_graph(self)
        if self.is_table:      # a table has no ranks: rebuild as the ranked index from the tensors it was built from
            jj, kk, ii = self._key[3]
            self.build(jj, kk, force=True, ii=ii)
        U = self.meta()[0] if self.E else 0
        kx = torch.empty(U, dtype=torch.int64, device=self.device)
        ku = torch.empty(self.E, dtype=torch.int64, device=self.device)
        _lib.check(self.lib.cdv_graph_get_unique(_p(self.ws), _p(kx), U, _p(ku), self.E, _stream()),
                   "cdv_graph_get_unique")
        return kx, ku


def update_prologue(graph, fmap_chw, fmap1_nhwc, fmap2_nhwc, slot, gmap, gmap_pm, gmap_first, gmap_count, poses, patches,
                    intrinsics, ii, jj, kk, layout_e2pp=True, coords_out=None):
    """cdv_update_prologue: ring / tile ingest of the new frame, reprojection of all edges and the start of the
    patch-graph index in ONE launch (then the rest of the index build, with neighbors).  Returns coords; the
    neighbors are picked up with graph.neighbors()."""
    lib = _lib.load()
    _need_cuda(fmap_chw, fmap1_nhwc, fmap2_nhwc, poses, patches, intrinsics, ii, jj, kk)
    C, H, W = fmap_chw.shape[-3:]
    E, P = kk.numel(), patches.shape[-1]
    if P != 3 or poses.dtype != torch.float32:
        raise TypeError("update_prologue: float32 state, 3x3 patches")
    dev = poses.device
    _disarm_graph(graph)
    graph._reserve(E)
    shape = (1, E, 2, P, P) if layout_e2pp else (1, E, P, P, 2)
    if coords_out is not None:      # a caller-owned buffer (the one bound with GraphIndex.bind_corr_stream)
        coords = coords_out.view(-1)[: E * 2 * P * P].view(shape)
    else:
        coords = torch.empty(shape, dtype=torch.float32, device=dev)
    ix = torch.empty(E, dtype=torch.int64, device=dev)
    jx = torch.empty(E, dtype=torch.int64, device=dev)
    Ng = gmap.numel() // (C * 9) if gmap is not None else 0
    rc = lib.cdv_update_prologue(_p(fmap_chw.contiguous()), _p(fmap1_nhwc), _p(fmap2_nhwc), int(slot), C, H, W, _p(gmap),
                                 _p(gmap_pm), Ng, int(gmap_first), int(gmap_count), _p(poses), _p(patches),
                                 _p(intrinsics), _p(ii), _p(jj), _p(kk), E, 1 if layout_e2pp else 0, _p(coords),
                                 _p(graph.ws), graph.ws_bytes, graph.E_cap, graph.k_range, _p(ix), _p(jx), _stream())
    _lib.check(rc, "cdv_update_prologue")
    graph.n_builds += 1
    graph._key = graph._make_key(jj, kk, ii)
    graph.E = E
    graph.is_table = False
    graph._nbr = (ix, jx)
    return coords


def update_prologue_table(graph, fmap_chw, fmap1_nhwc, fmap2_nhwc, slot, gmap, gmap_pm, gmap_first, gmap_count, poses, patches,
                          intrinsics, ii, jj, kk, coords_out=None):
    """cdv_update_prologue_table: everything in front of the correlation in TWO launches -- ring / tile ingest + the
    table's fill pass, then slot sort + neighbors + reprojection + the correlation's order and packed stream.  Returns
    coords [1,E,2,3,3]; neighbors with graph.neighbors(); the stream with graph.corr_records_ptr() (ring sizes must have
    been bound with graph.bind_corr_stream)."""
    lib = _lib.load()
    _need_cuda(fmap_chw, fmap1_nhwc, fmap2_nhwc, poses, patches, intrinsics, ii, jj, kk)
    C, H, W = fmap_chw.shape[-3:]
    E, P = kk.numel(), patches.shape[-1]
    if P != 3 or poses.dtype != torch.float32:
        raise TypeError("update_prologue_table: float32 state, 3x3 patches")
    dev = poses.device
    _disarm_graph(graph)
    graph._reserve(E)
    if coords_out is not None:
        coords = coords_out.view(-1)[: E * 18].view(1, E, 2, P, P)
    else:
        coords = torch.empty((1, E, 2, P, P), dtype=torch.float32, device=dev)
    ix = torch.empty(E, dtype=torch.int64, device=dev)
    jx = torch.empty(E, dtype=torch.int64, device=dev)
    Ng = gmap.numel() // (C * 9) if gmap is not None else 0
    rc = lib.cdv_update_prologue_table(_p(fmap_chw.contiguous()), _p(fmap1_nhwc), _p(fmap2_nhwc), int(slot), C, H, W, _p(gmap),
                                       _p(gmap_pm), Ng, int(gmap_first), int(gmap_count), _p(poses), _p(patches),
                                       _p(intrinsics), _p(ii), _p(jj), _p(kk), E, _p(coords), _p(graph.ws), graph.ws_bytes,
                                       graph.E_cap, graph.k_range, graph.table_capacity, _p(ix), _p(jx), _stream())
    _lib.check(rc, "cdv_update_prologue_table")
    graph.n_builds += 1
    graph._key = graph._make_key(jj, kk, ii) + ("table",)
    graph.E = E
    graph.is_table = True
    graph._nbr = (ix, jx)
    return coords


_graphs = {}


def prefer_table():
    """CDV_INDEX=ranked keeps the four-launch ranked index (unique ranks + CSR) as the form of the patch-graph index even
    where a table capacity is known; default: the two-launch patch table wherever one is.  Read at every call."""
    return _env("CDV_INDEX", "table") != "ranked"


_table_capacity = None


def configure_table(capacity):
    """Tell the per-device index workspaces that neighbors() / BA() create (the drop-in modules use them) how many patch
    ids can be live at once -- (REMOVAL_WINDOW + 2) x PATCHES_PER_FRAME of the SLAM object (cdvslam/slam.py:453-458; 2,304
    for default_cdvo.yaml) -- so that they use the two-launch table index (slot = id mod capacity).  None: ranked index
    (no assumption about the ids).  install_dropin(table_capacity=...) calls this; CDV_TABLE_CAPACITY sets the default."""
    global _table_capacity
    _table_capacity = int(capacity) if capacity else None
    _disarm_graph()
    _graphs.clear()


def table_capacity():
    if _table_capacity is not None:
        return _table_capacity
    env = _env("CDV_TABLE_CAPACITY", None)
    return int(env) if env else 0


def set_patch_capacity(n_patches):
    """capacity (number of patch ids: BUFFER_SIZE x PATCHES_PER_FRAME of the SLAM object, cdvslam/patchgraph.py:25-29) of
    the ranked index of the per-device workspaces created from now on by neighbors() / BA()"""
    global DEFAULT_K_RANGE
    DEFAULT_K_RANGE = int(n_patches)
    _disarm_graph()
    _graphs.clear()


def _device_graph(dev, **kw):
    g = _graphs.get(dev)
    if g is None:
        g = _graphs[dev] = GraphIndex(dev, **({"k_range": DEFAULT_K_RANGE, "table_capacity": table_capacity()} | kw))
    return g


def _table_still_fits(g, dev=None):
    """The table form of the index assumes that the live patch ids fit its capacity (install_dropin(table_capacity=...)).
    When they do not -- loop-closure or long-range edges, slam.py:507-510 -- the build flags a collision, neighbors() of
    that update are -1 and its BA is skipped and COUNTED (pinned event counters, no synchronisation).  Seen here before the
    next build: the workspace then gives the table up for good and goes on with the ranked index, which assumes nothing
    about the ids.  Only the events of bundle adjustments that ran over THIS index count (GraphIndex.events): another
    workspace's trouble on the same device does not cost this one its table."""
    if not g.table_capacity or g._events is None:
        return
    n = g._events.counts()[3]
    if n > g._graph_events_seen:
        g._graph_events_seen = n
        import warnings
        warnings.warn("cdv_slam_amd: the patch table (capacity %d) did not hold the live patch ids -- one update was skipped; "
                      "falling back to the ranked index on %s from now on" % (g.table_capacity, g.device), RuntimeWarning, stacklevel=3)
        g.table_capacity = 0
        g._key = None


def graph_for(jj, kk, ii=None, **kw):
    """Per-device shared GraphIndex, (re)built for (jj, kk) in the preferred form."""
    g = _device_graph(kk.device, **kw)
    _disarm_graph(g)
    _table_still_fits(g)
    key = g._make_key(jj.contiguous(), kk.contiguous(), None if ii is None else ii.contiguous())
    if g._key is not None and g._same_key(key):
        return g
    if prefer_table() and g.table_capacity:
        return g.build_table(jj, kk, ii=ii, with_neighbors=True)
    return g.build(jj, kk, ii=ii, with_neighbors=True)


def neighbors(kk, jj):
    """cuda_ba.neighbors(ii=kk, jj) (cdvslam/fastba/ba.cpp:59-97), fully on device."""
    if kk.numel() == 0:
        e = torch.empty(0, dtype=torch.int64, device=kk.device)
        return e, e.clone()
    ag = _armed_graph
    if ag is not None and _env("CDV_DROPIN_FAST", "1") != "0" and prefer_table() and not _sync_check():
        r = _fast.neighbors(ag[0], kk, jj, _stream())
        if r is not None:
            if type(r) is int:
                _lib.check(r, "cdv_graph_build_table")
            ag[1].n_builds += r[2]
            return r[0], r[1]
    return graph_for(jj, kk).neighbors()


# ---------------------------------------------------------------------------------------------------
# projective ops
# ---------------------------------------------------------------------------------------------------

def transform(poses, patches, intrinsics, ii, jj, kk, layout_e2pp=False, valid=False, jacobian=False, tonly=False):
    """Fused pops.transform (cdvslam/projective_ops.py:53-113).  poses [1,n,7] (tensor), patches
    [1,m,3,P,P], intrinsics [1,n,4] float32 -> coords [1,E,P,P,2] (or [1,E,2,P,P] if layout_e2pp)."""
    lib = _lib.load()
    _need_cuda(poses, patches, intrinsics, ii, jj, kk)
    if poses.dtype != torch.float32:
        raise TypeError("cdv_transform is float32 (the reference state buffers are float32, patchgraph.py:28-32)")
    poses, patches, intrinsics = poses.contiguous(), patches.contiguous(), intrinsics.contiguous()
    ii, jj, kk = ii.contiguous(), jj.contiguous(), kk.contiguous()
    E, P = ii.numel(), patches.shape[-1]
    dev = poses.device
    shape = (1, E, 2, P, P) if layout_e2pp else (1, E, P, P, 2)
    coords = torch.empty(shape, dtype=torch.float32, device=dev)
    vpx = torch.empty((1, E, P, P), dtype=torch.float32, device=dev) if valid else None
    v = Ji = Jj = Jz = None
    if jacobian:
        v = torch.empty((1, E), dtype=torch.float32, device=dev)
        Ji = torch.empty((1, E, 2, 6), dtype=torch.float32, device=dev)
        Jj = torch.empty((1, E, 2, 6), dtype=torch.float32, device=dev)
        Jz = torch.empty((1, E, 2, 1), dtype=torch.float32, device=dev)
    flags = (1 if layout_e2pp else 0) | (2 if tonly else 0)
    rc = lib.cdv_transform(_p(poses), _p(patches), _p(intrinsics), _p(ii), _p(jj), _p(kk), E, P, flags, _p(coords),
                           _p(vpx), _p(v), _p(Ji), _p(Jj), _p(Jz), _stream())
    _lib.check(rc, "cdv_transform")
    if jacobian:
        return coords, v, (Ji, Jj, Jz)
    if valid:
        return coords, vpx
    return coords


def fastba_reproject(poses, patches, intrinsics, ii, jj, kk):
    """cuda_ba.reproject (ba_cuda.cu:408-458, 614-646) -> [1,E,2,P,P]."""
    lib = _lib.load()
    _need_cuda(poses, patches, intrinsics, ii, jj, kk)
    poses, patches, intrinsics = poses.contiguous(), patches.contiguous(), intrinsics.contiguous()
    E, P = ii.numel(), patches.shape[-1]
    coords = torch.empty((1, E, 2, P, P), dtype=torch.float32, device=poses.device)
    rc = lib.cdv_fastba_reproject(_p(poses), _p(patches), _p(intrinsics), _p(ii.contiguous()), _p(jj.contiguous()),
                                  _p(kk.contiguous()), E, P, _p(coords), _stream())
    _lib.check(rc, "cdv_fastba_reproject")
    return coords


# ---------------------------------------------------------------------------------------------------
# altcorr
# ---------------------------------------------------------------------------------------------------

FMAP_PADX, FMAP_PADY = 16, 12   # CDV_FMAP_PADX / CDV_FMAP_PADY (include/cdvslam_hip.h)


def alloc_fmap_ring(slots, C, H, W, device):
    """Zero-initialised padded channels-last ring [slots, H + 2 PADY, W + 2 PADX, C] f16."""
    return torch.zeros((slots, H + 2 * FMAP_PADY, W + 2 * FMAP_PADX, C), dtype=torch.float16, device=device)


def fmap_interior(ring):
    """[slots, H, W, C] view of the image area of a padded ring."""
    return ring[:, FMAP_PADY:-FMAP_PADY, FMAP_PADX:-FMAP_PADX, :]


class NhwcCache:
    """padded channels-last shadows of planar feature rings somebody else writes (the reference's slam.py).  When the
    ring's version counter has moved, the shadow is re-synchronised by cdv_fmap_sync_nhwc: a fingerprint pass over the
    ring, then a conversion of only the slots that changed (slam.py:679-682 writes ONE slot per frame)."""

    def __init__(self, max_entries=4):
        self.entries = []
        self.max_entries = max_entries

    def get(self, fmap):
        place = _place(fmap)
        for ent in self.entries:
            if ent["place"] == place:      # the same memory, through whichever view object (ent["src"] pins the storage)
                if "fast" in ent:
                    _disarm_pair()         # the extension kept this shadow in step meanwhile: its version / parity come back
                if ent["version"] != fmap._version:
                    self._sync(ent)
                return ent["shadow"]
        C, H, W = fmap.shape[-3:]
        N = fmap.numel() // (C * H * W)
        lib = _lib.load()
        ent = {"src": fmap, "place": place, "version": None, "parity": 0,
               "shadow": torch.zeros(fmap.shape[:-3] + (H + 2 * FMAP_PADY, W + 2 * FMAP_PADX, C), dtype=fmap.dtype,
                                     device=fmap.device),
               "ws": torch.zeros(lib.cdv_fmap_sync_workspace_bytes(N), dtype=torch.uint8, device=fmap.device)}
        self._sync(ent)
        self.entries.append(ent)
        if len(self.entries) > self.max_entries:
            if "fast" in self.entries[0]:
                _disarm_pair()
            self.entries.pop(0)
        return ent["shadow"]

    def entry(self, fmap):
        """the shadow entry of this ring, or None"""
        place = _place(fmap)
        for ent in self.entries:
            if ent["place"] == place:
                return ent
        return None

    @staticmethod
    def _sync(ent):
        lib = _lib.load()
        fmap = ent["src"]
        C, H, W = fmap.shape[-3:]
        N = fmap.numel() // (C * H * W)
        _lib.check(lib.cdv_fmap_sync_nhwc(_p(fmap), _p(ent["shadow"]), N, C, H, W, _p(ent["ws"]), ent["parity"], _stream()),
                   "cdv_fmap_sync_nhwc")
        ent["parity"] ^= 1
        ent["version"] = fmap._version

    def converted_slots(self, fmap):
        """diagnostics: slots converted so far for this ring (synchronises)"""
        place = _place(fmap)
        for ent in self.entries:
            if ent["place"] == place:
                return int(ent["ws"][-64:-60].view(torch.int32).item())
        return 0


_nhwc = NhwcCache()


class TileCache:
    """pixel-major shadow [Ng, 9, C] of patch tiles somebody else keeps in the reference layout [1, Ng, C, 3, 3] (slam.py's
    gmap_): re-converted when the tensor's version counter has moved (1.5 MB in, 1.5 MB out at the default sizes)."""

    def __init__(self):
        self.src, self.ident, self.shadow = None, None, None
        self.n_converted = 0
        self.fast = False          # the extension holds (and keeps converting into) the shadow

    def get(self, gmap):
        """keyed on the memory + version (_ident), not on the Python object: slam.py's `gmap` property is a fresh view of
        gmap_ at every access (slam.py:249-251) and SLAM.corr reads it twice (:321-322)"""
        if self.fast:
            _disarm_pair()
        idn = _ident(gmap)
        if self.ident != idn or self.shadow is None:
            g = gmap[0] if gmap.dim() == 5 else gmap
            same_place = self.ident is not None and self.ident[0] == idn[0] and self.ident[2:] == idn[2:]
            self.shadow = gmap_to_pixel_major(g.contiguous(), out=self.shadow if (self.shadow is not None and same_place) else None)
            self.src, self.ident = gmap, idn      # src pins the storage the ident speaks of
            self.n_converted += 1
        return self.shadow


_tiles = TileCache()


def fmap_ingest(fmap_chw, fmap1_nhwc, fmap2_nhwc, slot, fmap1_nchw=None, fmap2_nchw=None, gmap=None, gmap_pm=None,
                gmap_first=0, gmap_count=0):
    """One new frame [C,H,W] f16 -> ring slot `slot` of the channels-last level-0 ring and its 4x4
    average pool into the level-1 ring (slam.py:681-682).  With gmap / gmap_pm also converts the frame's
    patch tiles gmap[gmap_first : gmap_first + gmap_count] ([.,C,3,3]) into the pixel-major array, same launch."""
    lib = _lib.load()
    _need_cuda(fmap_chw, fmap1_nhwc, fmap2_nhwc)
    C, H, W = fmap_chw.shape[-3:]
    fmap_chw = fmap_chw.contiguous()
    if gmap is None or gmap_pm is None:
        rc = lib.cdv_fmap_ingest(_p(fmap_chw), _p(fmap1_nhwc), _p(fmap2_nhwc), _p(fmap1_nchw), _p(fmap2_nchw),
                                 int(slot), C, H, W, _stream())
        _lib.check(rc, "cdv_fmap_ingest")
        return
    _need_cuda(gmap, gmap_pm)
    Ng = gmap.numel() // (C * 9)
    rc = lib.cdv_frame_ingest(_p(fmap_chw), _p(fmap1_nhwc), _p(fmap2_nhwc), _p(fmap1_nchw), _p(fmap2_nchw), int(slot),
                              C, H, W, _p(gmap), _p(gmap_pm), Ng, int(gmap_first), int(gmap_count), _stream())
    _lib.check(rc, "cdv_frame_ingest")


def gmap_to_pixel_major(gmap, out=None, first=0, count=None):
    """[Ng,C,3,3] f16 (reference layout of gmap_) -> [Ng,9,C]: the operand layout of the fused correlation."""
    lib = _lib.load()
    _need_cuda(gmap)
    if gmap.dtype != torch.float16:
        raise TypeError("gmap_to_pixel_major: float16 tiles")
    gmap = gmap.contiguous()
    C = gmap.shape[-3]
    Ng = gmap.numel() // (C * 9)
    if out is None:
        out = torch.empty((Ng, 9, C), dtype=torch.float16, device=gmap.device)
    count = Ng - first if count is None else count
    _lib.check(lib.cdv_gmap_to_pixel_major(_p(gmap), _p(out), Ng, C, int(first), int(count), _stream()),
               "cdv_gmap_to_pixel_major")
    return out


def corr_fused(gmap, fmap0_nhwc, fmap1_nhwc, coords, kk, jj, kmod=0, jmod=0, scales=(1.0, 4.0), order_ptr=None,
               out=None, pixel_major=False):
    """SLAM.corr (slam.py:316-323) in one launch.  gmap [Ng,C,3,3] f16 planar, fmapL_nhwc padded
    channels-last rings (alloc_fmap_ring), coords [1,E,2,3,3] f32 -> [1,E,882] f16 (fmap1_nhwc None -> one
    level, [1,E,441])."""
    lib = _lib.load()
    _need_cuda(gmap, fmap0_nhwc, coords, kk, jj)
    if gmap.dtype != torch.float16 or fmap0_nhwc.dtype != torch.float16 or coords.dtype != torch.float32:
        raise TypeError("corr_fused: gmap/fmap must be float16 and coords float32")
    nlev = 1 if fmap1_nhwc is None else 2
    E = kk.numel()
    C = gmap.shape[-1] if pixel_major else gmap.shape[-3]   # [Ng,9,C] (gmap_to_pixel_major) or [Ng,C,3,3]
    gmap, coords = gmap.contiguous(), coords.contiguous()
    Ng = gmap.numel() // (C * 9)
    slots = fmap0_nhwc.shape[-4]
    H0, W0 = fmap0_nhwc.shape[-3] - 2 * FMAP_PADY, fmap0_nhwc.shape[-2] - 2 * FMAP_PADX
    H1, W1 = (fmap1_nhwc.shape[-3] - 2 * FMAP_PADY, fmap1_nhwc.shape[-2] - 2 * FMAP_PADX) if nlev == 2 else (0, 0)
    if not fmap0_nhwc.is_contiguous() or (nlev == 2 and not fmap1_nhwc.is_contiguous()):
        raise RuntimeError("corr_fused: feature rings must be contiguous padded channels-last tensors")
    if out is None:
        out = torch.empty((1, E, 441 * nlev), dtype=torch.float16, device=gmap.device)
    rc = lib.cdv_corr_fused(_p(gmap), _p(fmap0_nhwc), _p(fmap1_nhwc), _p(coords), _p(kk.contiguous()),
                            _p(jj.contiguous()), order_ptr, _p(out), E, Ng, slots, C, H0, W0, H1, W1, float(scales[0]),
                            float(scales[1]), nlev, int(kmod), int(jmod), 1 if pixel_major else 0, _stream())
    _lib.check(rc, "cdv_corr_fused")
    return out


def pair_levels_enabled():
    """CDV_PAIR_LEVELS=0 switches the pairing of the two per-level cuda_corr.forward calls off (every call is then
    computed on its own, as the reference's extension does); default on.  Read at every call."""
    return _env("CDV_PAIR_LEVELS", "1") != "0"


def corr_fused_stream(gmap_pm, fmap0_nhwc, fmap1_nhwc, records_ptr, E, scales=(1.0, 4.0), out=None):
    """cdv_corr_fused_stream: SLAM.corr (slam.py:316-323) in one launch from the packed input stream the index build
    wrote (GraphIndex.bind_corr_stream / corr_records_ptr).  gmap_pm [Ng,9,C] pixel-major tiles -> [1,E,882] f16."""
    lib = _lib.load()
    _need_cuda(gmap_pm, fmap0_nhwc, fmap1_nhwc)
    C = gmap_pm.shape[-1]
    Ng = gmap_pm.numel() // (C * 9)
    slots = fmap0_nhwc.shape[-4]
    H0, W0 = fmap0_nhwc.shape[-3] - 2 * FMAP_PADY, fmap0_nhwc.shape[-2] - 2 * FMAP_PADX
    H1, W1 = fmap1_nhwc.shape[-3] - 2 * FMAP_PADY, fmap1_nhwc.shape[-2] - 2 * FMAP_PADX
    if out is None:
        out = torch.empty((1, E, 882), dtype=torch.float16, device=gmap_pm.device)
    rc = lib.cdv_corr_fused_stream(_p(gmap_pm), _p(fmap0_nhwc), _p(fmap1_nhwc), records_ptr, _p(out), E, Ng, slots, C, H0, W0,
                                   H1, W1, float(scales[0]), float(scales[1]), 1, _stream())
    _lib.check(rc, "cdv_corr_fused_stream")
    return out


class PairedLevel(torch.Tensor):
    """One pyramid level of a paired two-level correlation: a strided view (stride 2 along the level axis) of the
    interleaved buffer [1, E, 7, 7, 3, 3, 2] that cdv_corr_fused fills -- the very layout `torch.stack([corr1, corr2], -1)`
    produces (slam.py:323).  When exactly that stack is asked of the two views of ONE buffer, the buffer itself is the
    answer (same values, same shape, same strides, no 2 x 84 MB copy); every other operation sees an ordinary tensor."""

    @staticmethod
    def wrap(view, base, level):
        t = view.as_subclass(PairedLevel)
        t._pair_base, t._pair_level = base, level
        return t

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func is torch.stack and pair_levels_enabled():
            seq = args[0] if args else kwargs.get("tensors")
            dim = args[1] if len(args) > 1 else kwargs.get("dim", 0)
            if (isinstance(seq, (list, tuple)) and len(seq) == 2 and "out" not in kwargs
                    and all(isinstance(t, PairedLevel) and getattr(t, "_pair_base", None) is not None for t in seq)
                    and seq[0]._pair_base is seq[1]._pair_base and (seq[0]._pair_level, seq[1]._pair_level) == (0, 1)
                    and dim in (-1, seq[0]._pair_base.dim() - 1)
                    and seq[0].shape == seq[0]._pair_base.shape[:-1] and seq[1].shape == seq[1]._pair_base.shape[:-1]):
                _pairing.n_stacked += 1
                return seq[0]._pair_base
        with torch._C.DisableTorchFunctionSubclass():
            return func(*args, **kwargs)


class _LevelPairing:
    """The reference calls cuda_corr.forward twice per update: pyramid[0] with coords, then pyramid[1] with coords / 4
    (slam.py:321-322), and stacks the two results (slam.py:323).  Seen from here the second call repeats the per-edge
    work of the first, and the stack copies what one launch could have written in place.  Once the pattern has been
    observed (two consecutive calls on the same patch tiles and index tensors, on two rings whose sizes differ by a power
    of two), the call on the first ring computes BOTH levels in one launch (cdv_corr_fused) into one interleaved buffer
    [1, E, 7, 7, 3, 3, 2] and returns its level-0 view; the call that follows on the second ring only checks, per edge and
    on the device, that its coords are the first call's divided by the same power of two, recomputes the edges for which
    they are not (cdv_corr_level_checked_interleaved), and returns the level-1 view.  The views are PairedLevel tensors:
    torch.stack of the two along a new last axis returns the buffer they share.

    What is assumed, and what guards it (a documented mode, CDV_PAIR_LEVELS=0 turns all of it off):
      * the speculative level-1 result is only handed out to the call that IMMEDIATELY follows, on the very same
        memory at the same version (_ident: address, shape, strides, dtype, version counter of patch tiles, ii, jj -- NOT
        the Python objects: slam.py's `gmap` is a property that returns a fresh view per access, slam.py:249-251,321-322;
        the tensors of the first call are held alive by `pending`, so an address the allocator recycles cannot
        impersonate them) on the ring that was learned as the partner, unchanged since; any other call in between -- a
        third caller -- drops the pending result;
      * the learned partnership is keyed by the memory of ring A (at most four partnerships are remembered, each holding
        its two rings), re-validated (shapes, dtype, contiguity, ratio) before every paired launch;
      * the coords are checked per edge on the device, so a caller whose second call is not coords / ratio gets the
        recomputed values;
      * the stack shortcut needs the two views of one buffer, levels (0, 1), new last axis -- anything else is a real stack.
    A second call that never comes costs the speculative level and nothing else.  `n_fused` counts pairs served,
    `n_stacked` the stacks answered with the shared buffer."""

    def __init__(self):
        self.learned = {}      # _place(ring A) -> (ring A, ring B, ratio); insertion order = age
        self.last = None       # the previous call when it was an ordinary one
        self.pending = None
        self.n_fused = 0
        self.n_stacked = 0
        self.fast_token = 0    # != 0: the extension holds this pairing's rings and tiles (_arm / _disarm_pair)

    @staticmethod
    def _same(held, now):
        """every tensor recorded at the first call views the same memory, unmodified since (the held tensor pins it)"""
        return all(idn == _ident(u) for (t, idn), u in zip(held, now))

    @staticmethod
    def _hold(*ts):
        return tuple((t, _ident(t)) for t in ts)

    def _partner(self, ringA):
        ent = self.learned.get(_place(ringA))
        if ent is None:
            return None, 0
        rb, ratio = ent[1], ent[2]
        ok = (rb.is_contiguous() and rb.dtype == torch.float16 and rb.shape[:3] == ringA.shape[:3]
              and rb.shape[3] * ratio == ringA.shape[3] and rb.shape[4] * ratio == ringA.shape[4])
        return (rb, ratio) if ok else (None, 0)

    def _arm(self, tiles, ringA, ringB, ratio):
        """a complete pair has just been served from here: from the next call on the extension serves this pattern"""
        global _armed_pair
        if not fast_lane_enabled() or tiles.dim() != 5 or not tiles.is_contiguous() or _tiles.shadow is None:
            return
        ea, eb = _nhwc.entry(ringA), _nhwc.entry(ringB)
        if ea is None or eb is None or _tiles.ident != _ident(tiles):
            return
        _disarm_pair()
        ring = lambda e: {"src": e["src"], "shadow": e["shadow"], "ws": e["ws"], "version": e["version"], "parity": e["parity"]}
        try:
            self.fast_token = _fast.arm_pair({"A": ring(ea), "B": ring(eb), "ratio": int(ratio), "tiles_src": tiles,
                                              "tiles_pm": _tiles.shadow, "tiles_version": _tiles.ident[1]})
        except RuntimeError:      # shapes the extension does not serve (a ring with a batch, ...): Python goes on serving them
            self.fast_token = 0
            return
        ea["fast"] = eb["fast"] = True
        _tiles.fast = True
        _armed_pair = (self, ea, eb, _tiles)

    def call(self, fmap1, fmap2, coords, ii, jj):
        lib = _lib.load()
        E = coords.shape[1]
        if self.fast_token:
            _fast.drop_pending()
        pend, self.pending = self.pending, None
        last, self.last = self.last, None
        if not pair_levels_enabled():
            return None
        # ---- the second call of a pair whose first call computed both levels
        if (pend is not None and E == pend["E"] and self._same(pend["held"], (fmap1, ii, jj, fmap2))):
            shadow = _nhwc.get(fmap2)          # in step already (synchronised by the first call; the version has not moved)
            N2, H2, W2 = fmap2.shape[1], fmap2.shape[3], fmap2.shape[4]
            C = fmap1.shape[2]
            g = fmap1 if fmap1.is_contiguous() else fmap1[0].contiguous()    # batch 1: the same bytes either way
            rc = lib.cdv_corr_level_checked_interleaved(_p(g), _p(shadow), _p(coords), _p(pend["coords"]), 1.0 / pend["ratio"],
                                                        _p(ii), _p(jj), _p(pend["buf"]), 1, E, g.numel() // (C * 9), N2, C, H2, W2,
                                                        1.0, 0, 0, 0, _stream())
            _lib.check(rc, "cdv_corr_level_checked_interleaved")
            self.n_fused += 1
            if not self.fast_token and g is fmap1:
                self._arm(fmap1, pend["ringA"], fmap2, pend["ratio"])
            return PairedLevel.wrap(pend["buf"][..., 1], pend["buf"], 1)
        # ---- a first call whose second is expected: both levels now, interleaved as the stack will want them
        ringB, ratio = self._partner(fmap2)
        if ringB is not None and E == ii.numel() == jj.numel() and E > 0:
            sa, sb = _nhwc.get(fmap2), _nhwc.get(ringB)
            buf = torch.empty((1, E, 7, 7, 3, 3, 2), dtype=torch.float16, device=fmap1.device)
            corr_fused(_tiles.get(fmap1), sa[0], sb[0], coords, ii, jj, scales=(1.0, float(ratio)), out=buf, pixel_major=True)
            self.pending = {"held": self._hold(fmap1, ii, jj, ringB), "E": E, "coords": coords, "ratio": ratio, "buf": buf,
                            "ringA": fmap2}
            return PairedLevel.wrap(buf[..., 0], buf, 0)
        # ---- an ordinary call: remember it, and learn the pairing from two in a row on the same tiles and indices
        if last is not None and E == last["E"] and self._same(last["held"], (fmap1, ii, jj)):
            ra = last["ring"]
            if _place(ra) != _place(fmap2) and ra.shape[:3] == fmap2.shape[:3] and fmap2.dtype == torch.float16:
                h, w, H, W = fmap2.shape[3], fmap2.shape[4], ra.shape[3], ra.shape[4]
                ratio = H // h if h > 0 else 0
                if ratio >= 2 and (ratio & (ratio - 1)) == 0 and h * ratio == H and w * ratio == W:
                    self.learned.pop(_place(ra), None)
                    while len(self.learned) >= 4:      # the oldest partnership goes (and with it the hold on its rings)
                        self.learned.pop(next(iter(self.learned)))
                    self.learned[_place(ra)] = (ra, fmap2, ratio)
        self.last = {"held": self._hold(fmap1, ii, jj), "E": E, "ring": fmap2}
        return None


_pairing = _LevelPairing()


def corr_forward(fmap1, fmap2, coords, ii, jj, radius):
    """cuda_corr.forward (cdvslam/altcorr/correlation.cpp:35-42): fmap1 [B,N1,C,P,P], fmap2
    [B,N2,C,H2,W2], coords [B,M,2,P,P] f32 -> [B,M,2r+1 (x),2r+1 (y),P,P]."""
    pr = _pairing
    if pr.fast_token and _env("CDV_DROPIN_FAST", "1") != "0" and pair_levels_enabled():
        r = _fast.corr(pr.fast_token, fmap1, fmap2, coords, ii, jj, radius, _stream())
        if r is not None:
            if type(r) is int:
                _lib.check(r, "cuda_corr.forward")
            if r[3] & 4:
                _tiles.n_converted += 1
            pr.n_fused += r[2]
            return PairedLevel.wrap(r[0], r[1], r[2])
    lib = _lib.load()
    _need_cuda(fmap1, fmap2, coords, ii, jj)
    if fmap1.dtype != fmap2.dtype or fmap1.dtype not in (torch.float16, torch.float32):
        raise TypeError("corr: feature maps must both be float16 or float32")
    if coords.dtype != torch.float32:
        raise TypeError("corr: coords must be float32 (correlation_kernel.cu:86)")
    B, N1, C, P = fmap1.shape[0], fmap1.shape[1], fmap1.shape[2], fmap1.shape[3]
    N2, H2, W2 = fmap2.shape[1], fmap2.shape[3], fmap2.shape[4]
    M = coords.shape[1]
    D1 = 2 * radius + 1
    coords = coords.contiguous()
    ii, jj = ii.contiguous(), jj.contiguous()
    fast = (fmap1.dtype == torch.float16 and radius == 3 and P == 3 and C % 8 == 0 and C <= 128
            and fmap2.is_contiguous())
    if fast:
        if B == 1 and C <= 32:
            o = _pairing.call(fmap1, fmap2, coords, ii, jj)
            if o is not None:
                return o
        shadow = _nhwc.get(fmap2)
        outs = []
        for b in range(B):
            o = corr_fused(fmap1[b], shadow[b], None, coords[b:b + 1], ii, jj)
            outs.append(o.view(1, M, D1, D1, P, P))
        return outs[0] if B == 1 else torch.cat(outs, 0)
    fmap1, fmap2 = fmap1.contiguous(), fmap2.contiguous()
    out = torch.empty((B, M, D1, D1, P, P), dtype=fmap1.dtype, device=fmap1.device)
    for b in range(B):
        rc = lib.cdv_corr_fwd(_p(fmap1[b]), _p(fmap2[b]), _p(coords[b]), _p(ii), _p(jj), _p(out[b]), M, N1, N2, C, P,
                              H2, W2, radius, _DT[fmap1.dtype], _stream())
        _lib.check(rc, "cdv_corr_fwd")
    return out


def patchify_forward(net, coords, radius):
    """cuda_corr.patchify_forward (correlation.cpp:49-52): net [B,C,H,W], coords [B,M,2] -> [B,M,C,D,D]."""
    lib = _lib.load()
    _need_cuda(net, coords)
    if net.dtype not in (torch.float16, torch.float32):
        raise TypeError("patchify: net must be float16 or float32")
    net, coords = net.contiguous(), coords.contiguous().float()
    B, C, H, W = net.shape
    M, D = coords.shape[1], 2 * radius + 2
    out = torch.empty((B, M, C, D, D), dtype=net.dtype, device=net.device)
    rc = lib.cdv_patchify_fwd(_p(net), _p(coords), _p(out), B, M, C, H, W, radius, _DT[net.dtype], _stream())
    _lib.check(rc, "cdv_patchify_fwd")
    return out


def patchify_blend(net, coords, radius, mode):
    """altcorr.patchify with mode 'bilinear' / 'upperleft' (correlation.py:51-71) in one launch."""
    lib = _lib.load()
    _need_cuda(net, coords)
    if net.dtype not in (torch.float16, torch.float32):
        raise TypeError("patchify: net must be float16 or float32")
    net, coords = net.contiguous(), coords.contiguous().float()
    B, C, H, W = net.shape
    M = coords.shape[1]
    d = 1 if mode == "upperleft" else 2 * radius + 1
    out = torch.empty((B, M, C, d, d), dtype=net.dtype if mode == "upperleft" else torch.float32, device=net.device)
    rc = lib.cdv_patchify_blend(_p(net), _p(coords), _p(out), B, M, C, H, W, radius, 2 if mode == "upperleft" else 1,
                                _DT[net.dtype], _stream())
    _lib.check(rc, "cdv_patchify_blend")
    return out


def patchify_multi(jobs, coords):
    """The altcorr.patchify calls of one new frame (net_cdv.py:355-374) in ONE launch (cdv_patchify_multi).
    jobs: list of dicts {net [C,H,W] or [1,C,H,W] f16/f32, radius, mode 'bilinear'|'upperleft', scale (sx, sy) or
    float (default 1), offset (ox, oy) or float (default 0)}: job j = altcorr.patchify(net, (coords + offset) * scale,
    radius, mode).  coords [1,M,2] or [M,2] f32 (x, y).  Returns the list of outputs, [1,M,C,d,d] each."""
    import ctypes
    lib = _lib.load()
    if len(jobs) > _lib.MAX_PATCHIFY_JOBS:
        raise ValueError("patchify_multi: more than %d jobs" % _lib.MAX_PATCHIFY_JOBS)
    coords = coords.reshape(-1, 2).contiguous().float()
    _need_cuda(coords)
    M = coords.shape[0]
    arr = (_lib.PatchifyJob * max(len(jobs), 1))()
    outs, keep = [], []
    pair = lambda v, d: (float(v), float(v)) if not isinstance(v, (tuple, list)) else (float(v[0]), float(v[1]))
    for a, job in zip(arr, jobs):
        net = job["net"]
        net = net[0] if net.dim() == 4 else net
        if net.dtype not in (torch.float16, torch.float32):
            raise TypeError("patchify: net must be float16 or float32")
        net = net.contiguous()
        _need_cuda(net)
        keep.append(net)
        C, H, W = net.shape
        mode = job.get("mode", "bilinear")
        r = int(job["radius"])
        d = 1 if mode == "upperleft" else 2 * r + 1
        out = torch.empty((1, M, C, d, d), dtype=net.dtype if mode == "upperleft" else torch.float32, device=net.device)
        outs.append(out)
        (sx, sy), (ox, oy) = pair(job.get("scale", 1.0), 1.0), pair(job.get("offset", 0.0), 0.0)
        a.net, a.out, a.C, a.H, a.W, a.radius = net.data_ptr(), out.data_ptr(), C, H, W, r
        a.mode, a.dtype, a.sx, a.sy, a.ox, a.oy = (2 if mode == "upperleft" else 1), _DT[net.dtype], sx, sy, ox, oy
    rc = lib.cdv_patchify_multi(ctypes.cast(arr, ctypes.c_void_p), len(jobs), _p(coords), M, _stream())
    _lib.check(rc, "cdv_patchify_multi")
    return outs


def flow_mag(poses, patches, intrinsics, ii, jj, kk, beta):
    """pops.flow_mag (projective_ops.py:120-130) fused: -> (flow [1,E,P,P] f32, valid [1,E,P,P] bool)"""
    lib = _lib.load()
    _need_cuda(poses, patches, intrinsics, ii, jj, kk)
    poses, patches, intrinsics = poses.contiguous(), patches.contiguous(), intrinsics.contiguous()
    E, P = ii.numel(), patches.shape[-1]
    flow = torch.empty((1, E, P, P), dtype=torch.float32, device=poses.device)
    val = torch.empty((1, E, P, P), dtype=torch.uint8, device=poses.device)
    rc = lib.cdv_flow_mag(_p(poses), _p(patches), _p(intrinsics), _p(ii.contiguous()), _p(jj.contiguous()),
                          _p(kk.contiguous()), E, P, float(beta), _p(flow), _p(val), _stream())
    _lib.check(rc, "cdv_flow_mag")
    return flow, val.bool()


def point_cloud(poses, patches, intrinsics, ix):
    """pops.point_cloud (projective_ops.py:115-117) fused: patches [1,M,3,P,P], ix [M] -> [1,M,P,P,4]"""
    lib = _lib.load()
    _need_cuda(poses, patches, intrinsics, ix)
    poses, patches, intrinsics = poses.contiguous(), patches.contiguous(), intrinsics.contiguous()
    M, P = ix.numel(), patches.shape[-1]
    pts = torch.empty((1, M, P, P, 4), dtype=torch.float32, device=poses.device)
    rc = lib.cdv_point_cloud(_p(poses), _p(patches), _p(intrinsics), _p(ix.contiguous()), M, P, _p(pts), _stream())
    _lib.check(rc, "cdv_point_cloud")
    return pts


# ---------------------------------------------------------------------------------------------------
# fastba
# ---------------------------------------------------------------------------------------------------

_ba_ws = {}
_ba_ppf = {}          # (device, workspace address) -> the PPF hint the workspace currently carries
_ba_bound = {}        # workspace address -> the EventBlock its kernels currently count into
BA_EVENTS = ("reduced system not positive definite", "more unique patches than U_max (update skipped)",
             "in-launch hand-off timed out (update not applied)", "patch-graph index in its range-error state (update skipped)")


class EventBlock:
    """Four int32 event counters (order of BA_EVENTS) in pinned host memory that the BA kernels bump
    (cdv_ba_bind_status_counters): readable without synchronising.  One block per patch-graph index (GraphIndex.events) and
    one per private BA workspace (ba_private_workspace): an event belongs to the path it happened on.  The words come out of
    pinned pools that are never handed back -- a kernel still in flight when its owner dies writes to memory that stays
    what it was."""
    _pools = []      # [(pinned tensor [256, 4], rows used)]
    _all = []        # every block ever made: (device, row view) -- ba_event_counts(device) sums them

    def __init__(self, device, label=""):
        if not EventBlock._pools or EventBlock._pools[-1][1] >= 256:
            EventBlock._pools.append([torch.zeros((256, 4), dtype=torch.int32).pin_memory(), 0])
        pool = EventBlock._pools[-1]
        self.cnt = pool[0][pool[1]]
        self._np = self.cnt.numpy()        # the same four words (a read is 0.3 us this way, 1.5 through the tensor)
        pool[1] += 1
        self.device, self.label = torch.device(device), label
        self.seen = [0, 0, 0, 0]           # what has been warned about
        EventBlock._all.append((self.device, self.cnt))

    def ptr(self):
        return ctypes.c_void_p(self.cnt.data_ptr())

    def counts(self):
        return self._np.tolist()

    def bind(self, ws):
        """the BA launches on workspace `ws` count into this block from now on"""
        if _ba_bound.get(ws.data_ptr()) is not self:
            _lib.check(_lib.load().cdv_ba_bind_status_counters(_p(ws), self.ptr()), "cdv_ba_bind_status_counters")
            _ba_bound[ws.data_ptr()] = self

    def report(self):
        """non-blocking: warn about failure events counted since the last look (the update they belong to is an earlier one)"""
        now = self._np.tolist()
        if now == self.seen:
            return
        for i in range(4):
            if now[i] > self.seen[i]:
                import warnings
                warnings.warn("cdv_slam_amd BA on %s (%s): %s -- %d new event(s); ops.ba_status() / CDV_CHECK=1 raise at the call"
                              % (self.device, self.label, BA_EVENTS[i], now[i] - self.seen[i]), RuntimeWarning, stacklevel=4)
                self.seen[i] = now[i]


_ba_need = {}         # (E, U_max, N) -> cdv_ba_workspace_bytes: asked once per shape, not once per call


def _arm_graph(g, ba_ws, ppf):
    """the per-device table index has just served a complete BA from here: from the next call on the extension serves
    cuda_ba.neighbors / cuda_ba.forward on it (same workspaces, same event block, same PPF hint) until something it does
    not recognise comes along"""
    global _armed_graph
    k = g._key      # (ident jj, ident kk, ident ii | None, (jj, kk, ii), "table")
    jj, kk, ii = k[3]
    nbr = getattr(g, "_nbr", None)
    ev = g.events
    _disarm_graph()
    tok = _fast.arm_graph({"ws": g.ws, "ba_ws": ba_ws, "ws_bytes": g.ws_bytes, "E_cap": g.E_cap, "k_range": g.k_range,
                           "table_capacity": g.table_capacity, "ppf": ppf, "events_ptr": ev.cnt.data_ptr(), "events_seen": list(ev.seen),
                           "jj": jj, "kk": kk, "ii": ii, "jj_version": k[0][1], "kk_version": k[1][1],
                           "ii_version": None if k[2] is None else k[2][1],
                           "ix": None if nbr is None else nbr[0], "jx": None if nbr is None else nbr[1]})
    _armed_graph = (tok, g)


def _ba_workspace(dev, E, U_max, N):
    lib = _lib.load()
    need = _ba_need.get((E, U_max, N))
    if need is None:
        if len(_ba_need) > 4096:
            _ba_need.clear()
        need = _ba_need[(E, U_max, N)] = lib.cdv_ba_workspace_bytes(E, U_max, max(N, 1))
    ws = _ba_ws.get(dev)
    if ws is None or ws.numel() < need:
        if ws is not None:
            _ba_bound.pop(ws.data_ptr(), None)
        ws = _ba_ws[dev] = torch.empty(int(need * 1.25) + 4096, dtype=torch.uint8, device=dev)
        _lib.check(lib.cdv_ba_workspace_init(_p(ws), _stream()), "cdv_ba_workspace_init")   # whoever allocates initialises
        _ba_bound.pop(ws.data_ptr(), None)      # (an address the allocator hands out again carries no binding)
    return ws


def ba_private_workspace(dev, E, U_max, N, label="private BA workspace"):
    """a bundle-adjustment workspace of the caller's own (the per-device one above is re-allocated when somebody asks for
    more): initialised, with an event block of its own -- `ws.events` (ops.EventBlock)"""
    lib = _lib.load()
    ws = torch.empty(int(lib.cdv_ba_workspace_bytes(E, U_max, max(N, 1))) + 4096, dtype=torch.uint8, device=dev)
    _lib.check(lib.cdv_ba_workspace_init(_p(ws), _stream()), "cdv_ba_workspace_init")
    _ba_bound.pop(ws.data_ptr(), None)
    ws.events = EventBlock(dev, label)
    ws.events.bind(ws)
    return ws


def ba_event_counts(device=None):
    """Failure events the BA kernels have counted on `device` since start-up, over ALL workspaces, WITHOUT synchronising: a
    list of four ints in the order of BA_EVENTS.  (Per path: GraphIndex.events.counts(), ws.events.counts().)"""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    tot = [0, 0, 0, 0]
    for d, row in EventBlock._all:
        if d == dev:
            for i, v in enumerate(row.tolist()):
                tot[i] += int(v)
    return tot


def ba_status(device=None, raise_on_error=True):
    """Status words of the last BA on `device` (cdv_ba_status; synchronises the current stream):
    (cholesky, overflow, hand-off, graph).  Raises CdvError naming the failure unless raise_on_error is False."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    ws = _ba_ws.get(dev)
    if ws is None:
        return (0, 0, 0, 0)
    info = (ctypes.c_int32 * 4)()
    rc = _lib.load().cdv_ba_status(_p(ws), info, _stream())
    if rc != 0 and raise_on_error:
        _lib.check(rc, "bundle adjustment")
    return tuple(int(v) for v in info)


def ba_forward(poses, patches, intrinsics, target, weight, lmbda, ii, jj, kk, PPF, t0, t1, iterations,
               eff_impl=False, debug=False, U_max=None, graph=None):
    """cuda_ba.forward (cdvslam/fastba/ba.cpp:31-45): in place on poses / patches, returns []."""
    ag = _armed_graph
    if (ag is not None and graph is None and not debug and torch.is_tensor(lmbda) and _env("CDV_DROPIN_FAST", "1") != "0"
            and prefer_table() and not _sync_check()):
        r = _fast.ba(ag[0], poses, patches, intrinsics, target, weight, lmbda, ii, jj, kk, int(PPF) if PPF else 0, int(t0), int(t1),
                     int(iterations), _stream())
        if r is not None:
            if type(r) is int:
                _lib.check(r, "cdv_ba_forward")
            ag[1].n_builds += r[0]
            return []
    _disarm_graph()        # whoever runs a BA from here may re-bind, grow or re-index what the extension was working on
    lib = _lib.load()
    _need_cuda(poses, patches, intrinsics, target, weight, ii, jj, kk)
    for t in (poses, patches, intrinsics):
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise TypeError("BA: poses / patches / intrinsics must be contiguous float32 (updated in place)")
    N = int(t1) - int(t0)
    # eff_impl only selects the reference's block-sparse E storage (ba_cuda.cu:506-509): the numbers are those of the
    # dense path.  Here the library picks its own path by N (<= 10, <= 32, global BA up to 1024 free poses).
    if N > 1024:
        raise NotImplementedError("BA with more than 1024 free poses")
    E, P = kk.numel(), patches.shape[-1]
    dev = poses.device
    if E == 0:
        return []
    # only pointers travel: a contiguous float32 tensor goes as it is (the reshapes / casts below cost 2 us each on the host)
    if target.dtype != torch.float32 or not target.is_contiguous():
        target = target.reshape(-1, 2).float().contiguous()
    if weight.dtype != torch.float32 or not weight.is_contiguous():
        weight = weight.reshape(-1, 2).float().contiguous()
    if not torch.is_tensor(lmbda):
        lmbda = torch.tensor([float(lmbda)], dtype=torch.float32, device=dev)
    if lmbda.dtype != torch.float32 or not lmbda.is_contiguous() or lmbda.device != dev:
        lmbda = lmbda.reshape(-1).float().contiguous().to(dev)
    ii, jj, kk = ii.contiguous(), jj.contiguous(), kk.contiguous()
    g = graph if graph is not None else _device_graph(dev)
    _table_still_fits(g)
    g.index_for_ba(jj, kk, ii, N)
    g.events.report()
    if U_max is None:
        U_max = min(E, patches.numel() // (3 * P * P))
    if g.is_table:
        U_max = max(U_max, g.table_capacity)      # the slab kernels work through every slot of the table
    ws = _ba_workspace(dev, E, U_max, N)
    g.events.bind(ws)        # this call's failure events are this index's
    ppf = int(PPF) if (PPF is not None and int(PPF) > 0) else 0      # cuda_ba.forward's PPF: a hint for the 10 < N <= 32 path
    if _ba_ppf.get((dev, ws.data_ptr())) != ppf:
        _lib.check(lib.cdv_ba_set_patches_per_frame(_p(ws), ppf), "cdv_ba_set_patches_per_frame")
        _ba_ppf[(dev, ws.data_ptr())] = ppf
    dbg = None
    if debug:
        n6, Us = 6 * N, (U_max + 63) // 64 * 64
        dbg = torch.zeros(n6 * n6 + 2 * n6 + 3 * Us + n6 * Us + 64, dtype=torch.float32, device=dev)
    rc = lib.cdv_ba_forward(_p(poses), _p(patches), _p(intrinsics), _p(target), _p(weight), _p(lmbda), _p(ii), _p(jj),
                            _p(kk), E, P, int(t0), int(t1), int(iterations), _p(g.ws), _p(ws), ws.numel(), U_max,
                            _p(dbg), _stream())
    _lib.check(rc, "cdv_ba_forward")
    if _sync_check() and iterations > 0:
        ba_status(dev)        # raises CdvError: not positive definite / U_max exceeded / hand-off lost / graph range
    if graph is None and not debug and g.is_table and 1 <= N <= 32 and g._key is not None and fast_lane_enabled():
        _arm_graph(g, ws, ppf)
    if debug:
        n6, Us = 6 * N, (U_max + 63) // 64 * 64
        o = 0
        out = {}
        for name, size, shape in (("S", n6 * n6, (n6, n6)), ("y", n6, (n6,)), ("dX", n6, (N, 6)), ("dZ", Us, (Us,)),
                                  ("C", Us, (Us,)), ("u", Us, (Us,)), ("E", n6 * Us, (n6, Us))):
            out[name] = dbg[o:o + size].view(shape)
            o += size
        if g.is_table:
            # per-patch rows are indexed by slot (id mod capacity) here; hand them out by unique rank as the ranked index does
            pos = torch.unique(kk) % g.table_capacity
            for name in ("dZ", "C", "u"):
                v = torch.zeros_like(out[name]); v[:pos.numel()] = out[name][pos]; out[name] = v
            v = torch.zeros_like(out["E"]); v[:, :pos.numel()] = out["E"][:, pos]; out["E"] = v
        return out
    return []


# ---------------------------------------------------------------------------------------------------
# lietorch
# ---------------------------------------------------------------------------------------------------

_LIE_OUT = {"exp": "N", "log": "K", "inv": "N", "mul": "N", "adj": "K", "adjT": "K", "act": 3, "act4": 4, "matrix": 16}


def lie_op(group_id, op, x, y=None):
    """lietorch_backends.<op>(group_id, x[, y]) forward on flat contiguous [n, dim] rows."""
    lib = _lib.load()
    _need_cuda(x, y)
    if x.dtype not in (torch.float32, torch.float64):
        raise TypeError("lietorch ops: float32 or float64")
    if not x.is_contiguous() or (y is not None and not y.is_contiguous()):
        raise RuntimeError("lietorch ops: inputs must be contiguous (lietorch.cpp:7)")
    if group_id not in (1, 3):
        raise NotImplementedError("only SO3 (1) and SE3 (3) are on the update path")
    N, K = (7, 6) if group_id == 3 else (4, 3)
    n = x.shape[0]
    od = _LIE_OUT[op]
    od = N if od == "N" else K if od == "K" else od
    z = torch.empty((n, od), dtype=x.dtype, device=x.device)
    rc = lib.cdv_lie_op(group_id, LIE_OPS[op], _DT[x.dtype], n, _p(x), _p(y), _p(z), _stream())
    _lib.check(rc, "cdv_lie_op")
    return z.view(n, 4, 4) if op == "matrix" else z
