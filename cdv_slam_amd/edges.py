"""Device-resident edge lists of the patch graph: the bookkeeping either side of the update path.

Mirrors what cdvslam/slam.py does with `torch.cat` and boolean-mask indexing on `pg.ii / jj / kk / target / weight /
net` -- append_factors (:331-337) with __edges_forw / __edges_back (:528-541), remove_factors (:339-354) and
keyframe()'s removals and index shift (:408-458) -- on fixed-capacity buffers: appending writes in place, removing
is a stable compaction into a twin buffer (bit-exact with the reference's order), nothing is reallocated.
The number of edges is data-dependent after a removal: one 8-byte read-back per removal, as the reference's mask
indexing implies too."""
import ctypes

import torch

from . import _lib
from .ops import _p, _stream, touched


class EdgeStore:
    def __init__(self, device, capacity, net_dim=0, net_dtype=torch.float16, inactive_capacity=None):
        self.lib = _lib.load()
        self.dev, self.cap = device, int(capacity)
        self.icap = int(inactive_capacity if inactive_capacity is not None else 4 * capacity)
        z = lambda *s, dt=torch.int64: torch.zeros(s, dtype=dt, device=device)
        # twin buffers: [2][cap]
        self._ii, self._jj, self._kk = z(2, self.cap), z(2, self.cap), z(2, self.cap)
        self._target, self._weight = z(2, self.cap, 2, dt=torch.float32), z(2, self.cap, 2, dt=torch.float32)
        self._net = z(2, self.cap, net_dim, dt=net_dtype) if net_dim else None
        self.ii_inac, self.jj_inac, self.kk_inac = z(self.icap), z(self.icap), z(self.icap)
        self.target_inac, self.weight_inac = z(self.icap, 2, dt=torch.float32), z(self.icap, 2, dt=torch.float32)
        self._ws = torch.zeros(self.lib.cdv_edges_workspace_bytes(self.cap), dtype=torch.uint8, device=device)
        self._counts = torch.zeros(2, dtype=torch.int32).pin_memory()
        self.cur, self.E, self.E_inac = 0, 0, 0

    # -- views of the live part, reference names --------------------------------------------------------------
    ii = property(lambda s: s._ii[s.cur, :s.E])
    jj = property(lambda s: s._jj[s.cur, :s.E])
    kk = property(lambda s: s._kk[s.cur, :s.E])
    target = property(lambda s: s._target[s.cur, :s.E][None])      # [1,E,2]
    weight = property(lambda s: s._weight[s.cur, :s.E][None])
    net = property(lambda s: None if s._net is None else s._net[s.cur, :s.E][None])

    def _tail(self, n):
        if self.E + n > self.cap:
            raise RuntimeError("EdgeStore: capacity %d exceeded" % self.cap)

    def append_frame(self, ix, n, M, r):
        """append_factors(*edges_forw()); append_factors(*edges_back()) for the frame that just arrived
        (slam.py:707-709): one launch; the new edges' hidden state is zero (slam.py:336-337)."""
        added = ctypes.c_int64(0)
        c = self.cur
        rc = self.lib.cdv_edges_frame(_p(self._ii[c]), _p(self._jj[c]), _p(self._kk[c]), _p(ix), self.E, self.cap, int(n),
                                      int(M), int(r), ctypes.byref(added), _stream())
        _lib.check(rc, "cdv_edges_frame")
        touched(self._ii, self._jj, self._kk)      # written through raw pointers: whoever keys a cache on these views must see it
        self._zero_tail(added.value)
        self.E += added.value
        return added.value

    def append_factors(self, new_k, new_j, ix):
        """append_factors(ii=new_k, jj=new_j) (slam.py:331-337) for arbitrary edges, e.g. loop closure"""
        n = new_k.numel()
        self._tail(n)
        c = self.cur
        rc = self.lib.cdv_edges_append(_p(self._ii[c]), _p(self._jj[c]), _p(self._kk[c]), _p(ix), _p(new_k.contiguous()),
                                       _p(new_j.contiguous()), self.E, n, self.cap, _stream())
        _lib.check(rc, "cdv_edges_append")
        touched(self._ii, self._jj, self._kk)
        self._zero_tail(n)
        self.E += n

    def _zero_tail(self, n):
        if n:
            c = self.cur
            self._target[c, self.E:self.E + n].zero_()
            self._weight[c, self.E:self.E + n].zero_()
            if self._net is not None:
                self._net[c, self.E:self.E + n].zero_()

    def remove_factors(self, mask, store):
        """remove_factors(m, store) (slam.py:339-354): m [E] bool, True = drop; store keeps the dropped edges with
        their target / weight as inactive edges (used by the global BA, slam.py:462-468)."""
        if self.E == 0:
            return 0
        m8 = mask.to(torch.uint8).contiguous()
        c, o = self.cur, 1 - self.cur
        if store and self.E_inac + self.E > self.icap:
            raise RuntimeError("EdgeStore: inactive capacity %d exceeded" % self.icap)
        nb = 0 if self._net is None else self._net.shape[-1] * self._net.element_size()
        P = lambda t: None if t is None else _p(t)
        rc = self.lib.cdv_edges_remove(
            _p(m8), self.E, _p(self._ws), _p(self._ii[c]), _p(self._jj[c]), _p(self._kk[c]), _p(self._target[c]),
            _p(self._weight[c]), P(None if self._net is None else self._net[c]), nb, _p(self._ii[o]), _p(self._jj[o]),
            _p(self._kk[o]), _p(self._target[o]), _p(self._weight[o]), P(None if self._net is None else self._net[o]),
            _p(self.ii_inac) if store else None, _p(self.jj_inac) if store else None, _p(self.kk_inac) if store else None,
            _p(self.target_inac) if store else None, _p(self.weight_inac) if store else None, self.E_inac,
            ctypes.c_void_p(self._counts.data_ptr()), _stream())
        _lib.check(rc, "cdv_edges_remove")
        touched(self._ii, self._jj, self._kk, self._target, self._weight)
        torch.cuda.current_stream().synchronize()
        kept, removed = int(self._counts[0]), int(self._counts[1])
        self.cur, self.E = o, kept
        if store:
            self.E_inac += removed
        return removed

    def keyframe_shift(self, k, M):
        """index shift after frame k was dropped (slam.py:425-427)"""
        c = self.cur
        _lib.check(self.lib.cdv_edges_keyframe_shift(_p(self._ii[c]), _p(self._jj[c]), _p(self._kk[c]), self.E, int(k),
                                                     int(M), _stream()), "cdv_edges_keyframe_shift")
        touched(self._ii, self._jj, self._kk)

    def keyframe(self, k, n, M, ix, removal_window, loop_closure=False, opt_window=10, drop=True):
        """the edge part of SLAM.keyframe() (slam.py:408-458): if `drop`, frame k leaves the graph (its edges removed
        without storing, indices above it shifted; the caller shifts the frame buffers and decrements n); then edges
        whose source frame left the removal window become inactive.  Returns the new n."""
        if drop:
            self.remove_factors((self.ii == k) | (self.jj == k), store=False)
            self.keyframe_shift(k, M)
            n -= 1
        to_remove = ix[self.kk] < n - removal_window
        if loop_closure:
            lc = ((self.jj - self.ii) > 30) & (self.jj > (n - opt_window))
            to_remove = to_remove & ~lc
        self.remove_factors(to_remove, store=True)
        return n

    def full_edges(self):
        """(target, weight, ii, jj, kk) over inactive + active edges, as __run_global_BA concatenates them (slam.py:462-468)"""
        a = self.E_inac
        return (torch.cat((self.target_inac[:a][None], self.target), 1), torch.cat((self.weight_inac[:a][None], self.weight), 1),
                torch.cat((self.ii_inac[:a], self.ii)), torch.cat((self.jj_inac[:a], self.jj)),
                torch.cat((self.kk_inac[:a], self.kk)))


def frames_keyframe_shift(bufs, k, n):
    """keyframe(): the frame buffers after frame k is dropped (cdvslam/slam.py:431-441) in ONE launch:
    for i = k .. n - 2: buf[slot(i)] = buf[slot(i + 1)].  bufs: list of (tensor, modulus); the tensor's dim 0 is the
    slot axis (a frame buffer: modulus 0, slot(i) = i; a ring of m slots: modulus m, slot(i) = i % m).  A tensor whose
    slots are groups of rows (patches_ viewed as [N * M, ...]) is passed reshaped to [slots, ...]."""
    import ctypes
    from . import _lib
    from .ops import _stream
    lib = _lib.load()
    if len(bufs) > _lib.MAX_FRAME_BUFS:
        raise ValueError("frames_keyframe_shift: more than %d buffers" % _lib.MAX_FRAME_BUFS)
    arr = (_lib.FrameBuf * max(len(bufs), 1))()
    for a, (t, m) in zip(arr, bufs):
        if not t.is_cuda or not t.is_contiguous():
            raise RuntimeError("frames_keyframe_shift: buffers must be contiguous device tensors")
        a.base, a.slot_bytes, a.modulus, a.reserved = t.data_ptr(), t[0].numel() * t.element_size(), int(m), 0
    _lib.check(lib.cdv_frames_keyframe_shift(ctypes.cast(arr, ctypes.c_void_p), len(bufs), int(k), int(n), _stream()),
               "cdv_frames_keyframe_shift")
