"""Host-side logic that needs no GPU: the Python operator surface above the C ABI with the backend launch replaced by
the CPU oracle (monkeypatched here, in the test only), checked against fixtures the reference's own Python produced."""
import os

import numpy as np
import pytest
import torch

from oracle import oracle as O


@pytest.fixture
def lie_backend_on_oracle(monkeypatch):
    from cdv_slam_amd import ops

    def lie_op(group_id, op, x, y=None):
        dt = np.float64 if x.dtype == torch.float64 else np.float32
        out = O.lie(group_id, op, x.numpy(), None if y is None else y.numpy(), dtype=dt)
        return torch.from_numpy(out)

    monkeypatch.setattr(ops, "lie_op", lie_op)


def test_lietorch_layer_vs_reference_python(golden_dir, lie_backend_on_oracle):
    """cdv_slam_amd.lietorch (own construction) gives what the reference's groups.py / broadcasting.py gave on the same
    inputs (tests/golden/lietorch_py.npz): broadcasting of size-1 batch dims, op wiring, result types"""
    from cdv_slam_amd.lietorch import SE3, SO3, cat, stack
    g = np.load(os.path.join(golden_dir, "lietorch_py.npz"))
    a, b, p4 = (torch.from_numpy(g[k]) for k in ("a", "b", "p4"))
    X, Y = SE3.exp(a), SE3.exp(b)
    tol = 1e-12
    assert isinstance(X, SE3) and X.shape == (3, 4) and X.tangent_shape == (3, 4, 6)
    assert np.allclose(X.data.numpy(), g["X"], atol=tol)
    assert isinstance(X * Y, SE3) and np.allclose((X * Y).data.numpy(), g["XY"], atol=tol)
    assert np.allclose(X.inv().data.numpy(), g["Xinv"], atol=tol)
    assert torch.is_tensor(X.log()) and np.allclose(X.log().numpy(), g["logX"], atol=tol)
    assert np.allclose((X[:, :, None] * p4).numpy(), g["act4"], atol=tol)
    assert np.allclose(X.matrix().numpy(), g["matrix"], atol=tol)
    assert np.allclose(X.adjT(a).numpy(), g["adjT"], atol=tol)
    assert np.allclose(X.adj(a).numpy(), g["adj"], atol=tol)
    assert np.allclose(X.retr(a).data.numpy(), g["retr"], atol=tol)
    # helpers on the batch dimensions
    I = SE3.Identity(2, 3, dtype=torch.float64)
    assert I.shape == (2, 3) and torch.equal(I.data[..., :6], torch.zeros(2, 3, 6, dtype=torch.float64)) and bool((I.data[..., 6] == 1).all())
    I.data[0, 0, 0] = 5.0                                  # rows do not alias each other
    assert float(I.data[1, 2, 0]) == 0.0
    assert SE3.IdentityLike(X).shape == X.shape and SE3.IdentityLike(X).dtype == X.dtype
    assert cat([X, X], 0).shape == (6, 4) and stack([X, X], 0).shape == (2, 3, 4)
    assert X.view((12,)).shape == (12,) and len(X.unbind(0)) == 3 and X[1].shape == (4,)
    Z = SE3(X.data.clone()); Z[0] = Y[0].data.expand(4, 7); assert torch.equal(Z.data[0], Y.data[0].expand(4, 7))
    assert X.float().dtype == torch.float32 and X.detach().data.data_ptr() == X.data.data_ptr()
    R = SO3(X)
    assert torch.equal(R.data, X.data[..., 3:]) and torch.equal(SE3(R).data[..., :3], torch.zeros(3, 4, 3, dtype=torch.float64))
    assert np.allclose(X.translation()[..., :3].numpy(), g["X"][..., :3], atol=tol) and X.translation().shape == (3, 4, 4)
    s = torch.full((3, 4), 2.0, dtype=torch.float64)
    assert torch.equal(X.scale(s).data[..., :3], 2 * X.data[..., :3]) and torch.equal(X.scale(s).data[..., 3:], X.data[..., 3:])
    with pytest.raises(ValueError):
        X * Y[0]                                            # differing numbers of dimensions, as the reference asserts
    with pytest.raises(ValueError):
        X.act(torch.zeros(3, 4, 5, dtype=torch.float64))
    assert "SE3" in repr(X)


def test_projective_ops_refuses_what_the_kernels_do_not_serve():
    """no composed fallback: float64 / batch > 1 / depth=True / non-SE3 raise instead of running torch code"""
    from cdv_slam_amd import projective_ops as pops
    from cdv_slam_amd.lietorch import SE3, SO3
    idx = torch.zeros(1, dtype=torch.long)
    patches, intr = torch.zeros(1, 1, 3, 3, 3), torch.ones(1, 1, 4)
    with pytest.raises(NotImplementedError):
        pops.transform(SE3(torch.zeros(1, 1, 7, dtype=torch.float64)), patches, intr, idx, idx, idx)
    with pytest.raises(NotImplementedError):
        pops.transform(SE3(torch.zeros(2, 1, 7)), patches, intr, idx, idx, idx)
    with pytest.raises(NotImplementedError):
        pops.transform(SE3(torch.zeros(1, 1, 7)), patches, intr, idx, idx, idx, depth=True)
    with pytest.raises(NotImplementedError):
        pops.flow_mag(SO3(torch.zeros(1, 1, 4)), patches, intr, idx, idx, idx)
    with pytest.raises(RuntimeError):                      # float32 / batch 1 on the CPU reaches the HIP-only check
        pops.transform(SE3(torch.zeros(1, 1, 7)), patches, intr, idx, idx, idx)


def test_iproj_proj_pinhole_maps():
    """pops.iproj / pops.proj against the closed forms (projective_ops.py:19-50): per-frame intrinsics, the 0.1 clamp"""
    from cdv_slam_amd import projective_ops as pops
    g = torch.Generator().manual_seed(3)
    patches = torch.rand((1, 5, 3, 3, 3), generator=g) * 20
    intr = torch.rand((1, 5, 4), generator=g) * 10 + 20
    X = pops.iproj(patches, intr)
    assert X.shape == (1, 5, 3, 3, 4)
    for n in range(5):
        fx, fy, cx, cy = intr[0, n]
        assert torch.allclose(X[0, n, ..., 0], (patches[0, n, 0] - cx) / fx) and torch.allclose(X[0, n, ..., 1], (patches[0, n, 1] - cy) / fy)
        assert torch.equal(X[0, n, ..., 2], torch.ones(3, 3)) and torch.equal(X[0, n, ..., 3], patches[0, n, 2])
    Y = X.clone()
    Y[..., 2] = torch.linspace(-0.5, 2.0, 45).view(1, 5, 3, 3)
    uvd = pops.proj(Y, intr, depth=True)
    uv = pops.proj(Y, intr)
    assert uv.shape == (1, 5, 3, 3, 2) and torch.equal(uvd[..., :2], uv)
    d = 1.0 / Y[..., 2].clamp(min=0.1)
    assert torch.allclose(uvd[..., 2], d)
    for n in range(5):
        fx, fy, cx, cy = intr[0, n]
        assert torch.allclose(uv[0, n, ..., 0], fx * (d[0, n] * Y[0, n, ..., 0]) + cx)
        assert torch.allclose(uv[0, n, ..., 1], fy * (d[0, n] * Y[0, n, ..., 1]) + cy)


def test_switches_are_read_live():
    """the environment switches (CDV_CHECK, CDV_INDEX, CDV_PAIR_LEVELS, CDV_TABLE_CAPACITY) are documented as live: a change
    of os.environ is seen by the next call, although they are read without os.environ's encode / decode round trip"""
    import os
    from cdv_slam_amd import ops
    keep = {k: os.environ.get(k) for k in ("CDV_CHECK", "CDV_PAIR_LEVELS", "CDV_INDEX")}
    try:
        os.environ["CDV_CHECK"] = "1"
        assert ops._sync_check()
        os.environ["CDV_CHECK"] = "0"
        assert not ops._sync_check()
        os.environ.pop("CDV_PAIR_LEVELS", None)
        assert ops.pair_levels_enabled()
        os.environ["CDV_PAIR_LEVELS"] = "0"
        assert not ops.pair_levels_enabled()
        os.environ["CDV_INDEX"] = "ranked"
        assert not ops.prefer_table()
        del os.environ["CDV_INDEX"]
        assert ops.prefer_table()
        assert ops._env("CDV_NO_SUCH_SWITCH", "dflt") == "dflt"
    finally:
        for k, v in keep.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
