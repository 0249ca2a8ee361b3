"""Host-side mirror of the reference's Lie-group wrappers (cdvslam/lietorch/groups.py:52-301,
broadcasting.py:11-33), forward only, on the HIP backend (cdv_slam_amd/csrc/lie.hip).

Semantics kept: `data` is [..., embedded_dim] (SE3: tx ty tz qx qy qz qw); binary ops broadcast
size-1 batch dimensions; `X * Y` composes, `X * p` acts on 3- or 4-vectors; `retr(a) = Exp(a) * X`;
`matrix()` returns 4x4 matrices; indexing / view / cat / stack operate on the batch dimensions.
Broadcasting uses expand + one contiguous copy instead of the reference's `repeat`.
"""
import numpy as np
import torch

from .. import ops


def _flat2(x, y):
    """Broadcast the batch dims of x [..., dx] and y [..., dy]; returns flat contiguous rows + batch shape."""
    if x.dim() != y.dim():
        raise ValueError("lietorch: operands must have the same number of dimensions "
                         "(got %s and %s)" % (tuple(x.shape), tuple(y.shape)))
    bs = torch.broadcast_shapes(x.shape[:-1], y.shape[:-1])
    xf = x.expand(bs + x.shape[-1:]).reshape(-1, x.shape[-1]).contiguous()
    yf = y.expand(bs + y.shape[-1:]).reshape(-1, y.shape[-1]).contiguous()
    return xf, yf, tuple(bs)


class LieGroup:
    group_name = None
    group_id = None
    manifold_dim = None
    embedded_dim = None
    id_elem = None

    def __init__(self, data):
        self.data = data

    def __repr__(self):
        return "{}: size={}, device={}, dtype={}".format(self.group_name, self.shape, self.device, self.dtype)

    # -- basic attributes ---------------------------------------------------------------------------
    @property
    def shape(self):
        return self.data.shape[:-1]

    @property
    def device(self):
        return self.data.device

    @property
    def dtype(self):
        return self.data.dtype

    @property
    def tangent_shape(self):
        return self.data.shape[:-1] + (self.manifold_dim,)

    def vec(self):
        return self.data

    # -- constructors -------------------------------------------------------------------------------
    @classmethod
    def Identity(cls, *batch_shape, **kwargs):
        if isinstance(batch_shape[0], (tuple, list, torch.Size)):
            batch_shape = tuple(batch_shape[0])
        numel = int(np.prod(batch_shape))
        data = cls.id_elem.reshape(1, -1)
        if 'device' in kwargs:
            data = data.to(kwargs['device'])
        if 'dtype' in kwargs:
            data = data.type(kwargs['dtype'])
        return cls(data.repeat(numel, 1)).view(tuple(batch_shape))

    @classmethod
    def IdentityLike(cls, G):
        return cls.Identity(G.shape, device=G.data.device, dtype=G.data.dtype)

    @classmethod
    def InitFromVec(cls, data):
        return cls(data)

    @classmethod
    def Random(cls, *batch_shape, sigma=1.0, **kwargs):
        if isinstance(batch_shape[0], (tuple, list, torch.Size)):
            batch_shape = tuple(batch_shape[0])
        xi = torch.randn(tuple(batch_shape) + (cls.manifold_dim,), **kwargs)
        return cls.exp(sigma * xi)

    # -- backend dispatch ---------------------------------------------------------------------------
    @classmethod
    def _unary(cls, op, x):
        out = ops.lie_op(cls.group_id, op, x.reshape(-1, x.shape[-1]).contiguous())
        return out.view(tuple(x.shape[:-1]) + tuple(out.shape[1:]))

    @classmethod
    def _binary(cls, op, x, y):
        xf, yf, bs = _flat2(x, y)
        out = ops.lie_op(cls.group_id, op, xf, yf)
        return out.view(bs + tuple(out.shape[1:]))

    @classmethod
    def exp(cls, x):
        return cls(cls._unary("exp", x))

    def log(self):
        return self._unary("log", self.data)

    def inv(self):
        return self.__class__(self._unary("inv", self.data))

    def mul(self, other):
        return self.__class__(self._binary("mul", self.data, other.data))

    def retr(self, a):
        dX = self._unary("exp", a)
        return self.__class__(self._binary("mul", dX, self.data))

    def adj(self, a):
        return self._binary("adj", self.data, a)

    def adjT(self, a):
        return self._binary("adjT", self.data, a)

    def act(self, p):
        if p.shape[-1] == 3:
            return self._binary("act", self.data, p)
        if p.shape[-1] == 4:
            return self._binary("act4", self.data, p)
        raise ValueError("act: points must have 3 or 4 components")

    def matrix(self):
        """4x4 matrices [..., 4, 4] (groups.py:180-184 builds them by acting on the identity columns)."""
        return self._unary("matrix", self.data)

    def translation(self):
        p = torch.as_tensor([0.0, 0.0, 0.0, 1.0], dtype=self.dtype, device=self.device)
        p = p.view([1] * (len(self.data.shape) - 1) + [4, ])
        return self._binary("act4", self.data, p)

    def quaternion(self):
        return self.data[..., -4:] if self.group_id == 1 else self.data[..., 3:7]

    # -- tensor-like helpers ------------------------------------------------------------------------
    def detach(self):
        return self.__class__(self.data.detach())

    def view(self, dims):
        return self.__class__(self.data.view(tuple(dims) + (self.embedded_dim,)))

    def __mul__(self, other):
        if isinstance(other, LieGroup):
            return self.mul(other)
        if isinstance(other, torch.Tensor):
            return self.act(other)
        return NotImplemented

    def __getitem__(self, index):
        return self.__class__(self.data[index])

    def __setitem__(self, index, item):
        self.data[index] = item.data

    def to(self, *args, **kwargs):
        return self.__class__(self.data.to(*args, **kwargs))

    def cpu(self):
        return self.__class__(self.data.cpu())

    def cuda(self):
        return self.__class__(self.data.cuda())

    def float(self, device=None):
        return self.__class__(self.data.float())

    def double(self, device=None):
        return self.__class__(self.data.double())

    def unbind(self, dim=0):
        return [self.__class__(x) for x in self.data.unbind(dim=dim)]


class SO3(LieGroup):
    group_name = 'SO3'
    group_id = 1
    manifold_dim = 3
    embedded_dim = 4
    id_elem = torch.as_tensor([0.0, 0.0, 0.0, 1.0])

    def __init__(self, data):
        if isinstance(data, SE3):
            data = data.data[..., 3:7]
        super().__init__(data)


class SE3(LieGroup):
    group_name = 'SE3'
    group_id = 3
    manifold_dim = 6
    embedded_dim = 7
    id_elem = torch.as_tensor([0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0])

    def __init__(self, data):
        if isinstance(data, SO3):
            data = torch.cat([torch.zeros_like(data.data[..., :3]), data.data], -1)
        super().__init__(data)

    def scale(self, s):
        t, q = self.data.split([3, 4], -1)
        return SE3(torch.cat([t * s.unsqueeze(-1), q], dim=-1))


def cat(group_list, dim):
    return group_list[0].__class__(torch.cat([X.data for X in group_list], dim=dim))


def stack(group_list, dim):
    return group_list[0].__class__(torch.stack([X.data for X in group_list], dim=dim))
