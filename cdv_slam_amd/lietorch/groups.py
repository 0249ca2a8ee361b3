"""Lie-group value types over the HIP backend (cdv_slam_amd/csrc/lie.hip), forward only.

The operator surface the update path's callers use -- `SE3(data)`, `X * Y`, `X * points`, `X[:, idx]`, `.inv()`,
`.log()`, `.retr(a)`, `.adjT(a)`, `.matrix()`, `SE3.exp(a)`, `SE3.Identity(...)`, `cat` / `stack` -- with the meaning
the reference gives those names (cdvslam/lietorch/groups.py:52-301 for the names, broadcasting.py:11-33 for the
batch rule): `data` is [..., embedded_dim] (SE3: tx ty tz qx qy qz qw, SO3: qx qy qz qw), binary ops broadcast size-1
batch dimensions, `retr(a) = Exp(a) * X`.

Own construction: a group is a row of `_SPECS` (id, tangent / embedded width); every Lie operation is a row of
`_LIE_OPS` (backend op, arity, whether the result is a group element) turned into a method by `_install_lie_ops`;
tensor-like helpers are generated from `_DATA_PASSTHROUGH`.  Every operation is one launch of `cdv_lie_op` on flat
contiguous rows; broadcasting is an `expand` view made contiguous once (no `repeat`).
"""

import torch

from .. import ops

# name -> (backend group id, tangent width, embedded width)          ids: lietorch/src/lietorch.cpp:286-316
_SPECS = {"SO3": (1, 3, 4), "SE3": (3, 6, 7)}

# method name -> (backend op, number of tensor operands besides self, result is a group element)
_LIE_OPS = {
    "log": ("log", 0, False),
    "inv": ("inv", 0, True),
    "mul": ("mul", 1, True),
    "adj": ("adj", 1, False),
    "adjT": ("adjT", 1, False),
    "matrix": ("matrix", 0, False),      # [..., 4, 4]
}

# tensor methods forwarded to `data` and re-wrapped
_DATA_PASSTHROUGH = ("detach", "cpu", "cuda", "float", "double", "to", "clone", "contiguous")


def _rows(t):
    """[..., w] -> contiguous [n, w]"""
    return t.reshape(-1, t.shape[-1]).contiguous()


def _launch(gid, op, x, y=None):
    """one backend launch on broadcast batch dims; returns [batch..., out...]"""
    if y is None:
        out = ops.lie_op(gid, op, _rows(x))
        batch = tuple(x.shape[:-1])
    else:
        if x.dim() != y.dim():
            raise ValueError("lietorch: operands need the same number of dimensions (got %s and %s)"
                             % (tuple(x.shape), tuple(y.shape)))
        batch = tuple(torch.broadcast_shapes(x.shape[:-1], y.shape[:-1]))
        out = ops.lie_op(gid, op, _rows(x.expand(batch + x.shape[-1:])), _rows(y.expand(batch + y.shape[-1:])))
    return out.view(batch + tuple(out.shape[1:]))


def _batch(args):
    if len(args) == 1 and isinstance(args[0], (tuple, list, torch.Size)):
        return tuple(args[0])
    return tuple(int(a) for a in args)


class LieGroup:
    """Base of the group value types; concrete groups come from `_SPECS`."""
    group_name = group_id = manifold_dim = embedded_dim = None

    def __init__(self, data):
        self.data = data

    def __repr__(self):
        return "%s(batch=%s, dtype=%s, device=%s)" % (self.group_name, tuple(self.shape), self.dtype, self.device)

    shape = property(lambda self: self.data.shape[:-1])
    device = property(lambda self: self.data.device)
    dtype = property(lambda self: self.data.dtype)
    tangent_shape = property(lambda self: self.data.shape[:-1] + (self.manifold_dim,))

    def vec(self):
        return self.data

    # ---- construction -----------------------------------------------------------------------------------
    @classmethod
    def Identity(cls, *batch_shape, device=None, dtype=None, **_ignored):
        batch = _batch(batch_shape)
        row = torch.zeros(cls.embedded_dim, device=device, dtype=dtype or torch.float32)
        row[-1] = 1.0                                            # unit quaternion, zero translation
        return cls(row.expand(batch + (cls.embedded_dim,)).clone())

    @classmethod
    def IdentityLike(cls, G):
        return cls.Identity(G.shape, device=G.device, dtype=G.dtype)

    @classmethod
    def InitFromVec(cls, data):
        return cls(data)

    @classmethod
    def Random(cls, *batch_shape, sigma=1.0, **tensor_kwargs):
        return cls.exp(sigma * torch.randn(_batch(batch_shape) + (cls.manifold_dim,), **tensor_kwargs))

    @classmethod
    def exp(cls, a):
        return cls(_launch(cls.group_id, "exp", a))

    # ---- operations that are not a single table row --------------------------------------------------------
    def retr(self, a):
        """Exp(a) * X"""
        return type(self)(_launch(self.group_id, "mul", _launch(self.group_id, "exp", a), self.data))

    def act(self, p):
        width = p.shape[-1]
        if width not in (3, 4):
            raise ValueError("act: points must have 3 or 4 components")
        return _launch(self.group_id, "act" if width == 3 else "act4", self.data, p)

    def translation(self):
        """the element applied to the homogeneous origin, [..., 4]"""
        origin = self.data.new_zeros((1,) * (self.data.dim() - 1) + (4,))
        origin[..., 3] = 1.0
        return _launch(self.group_id, "act4", self.data, origin)

    def quaternion(self):
        return self.data[..., -4:]

    def __mul__(self, other):
        if isinstance(other, LieGroup):
            return self.mul(other)
        if torch.is_tensor(other):
            return self.act(other)
        return NotImplemented

    # ---- batch-dimension helpers -----------------------------------------------------------------------
    def view(self, dims):
        return type(self)(self.data.view(tuple(dims) + (self.embedded_dim,)))

    def unbind(self, dim=0):
        return [type(self)(x) for x in self.data.unbind(dim=dim)]

    def __getitem__(self, index):
        return type(self)(self.data[index])

    def __setitem__(self, index, item):
        self.data[index] = item.data if isinstance(item, LieGroup) else item


def _install_lie_ops(cls):
    def make(op, extra, group_result):
        if extra == 0:
            def method(self):
                out = _launch(self.group_id, op, self.data)
                return type(self)(out) if group_result else out
        else:
            def method(self, other):
                out = _launch(self.group_id, op, self.data, other.data if isinstance(other, LieGroup) else other)
                return type(self)(out) if group_result else out
        return method

    for name, (op, extra, group_result) in _LIE_OPS.items():
        m = make(op, extra, group_result)
        m.__name__ = name
        setattr(cls, name, m)

    def forward(name):
        def method(self, *args, **kwargs):
            return type(self)(getattr(self.data, name)(*args, **kwargs))
        method.__name__ = name
        return method

    for name in _DATA_PASSTHROUGH:
        setattr(cls, name, forward(name))


_install_lie_ops(LieGroup)


def _make_group(name):
    gid, k, n = _SPECS[name]
    return type(name, (LieGroup,), dict(group_name=name, group_id=gid, manifold_dim=k, embedded_dim=n))


_SO3Base, _SE3Base = _make_group("SO3"), _make_group("SE3")


class SO3(_SO3Base):
    def __init__(self, data):
        # SO3(an SE3) takes its rotation
        super().__init__(data.data[..., 3:7] if isinstance(data, LieGroup) and data.group_id == 3 else data)


class SE3(_SE3Base):
    def __init__(self, data):
        # SE3(an SO3) is the rotation with zero translation
        if isinstance(data, LieGroup) and data.group_id == 1:
            q = data.data
            data = torch.cat([q.new_zeros(q.shape[:-1] + (3,)), q], -1)
        super().__init__(data)

    def scale(self, s):
        """translation scaled by s (one factor per element), rotation kept"""
        out = self.data.clone()
        out[..., :3] *= s.unsqueeze(-1)
        return SE3(out)


def _gather(groups):
    return type(groups[0]), [g.data for g in groups]


def cat(group_list, dim):
    cls, parts = _gather(group_list)
    return cls(torch.cat(parts, dim=dim))


def stack(group_list, dim):
    cls, parts = _gather(group_list)
    return cls(torch.stack(parts, dim=dim))
