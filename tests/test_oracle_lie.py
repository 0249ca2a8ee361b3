"""Property tests that pin the oracle's SO3/SE3 arithmetic.

These are the reference's own forward tests (cdvslam/lietorch/run_tests.py:16-52: exp-log,
inverse, adjoint identity, act vs 4x4 matrix; float64, atol 1e-8) restated against the oracle,
plus the golden vectors of the reference's Python layer (groups.py broadcasting / op wiring).
"""
import os

import numpy as np
import pytest

from oracle import oracle as O

GROUPS = [(O.SO3, 3, 4), (O.SE3, 6, 7)]


@pytest.mark.parametrize("group,K,N", GROUPS)
def test_exp_log(group, K, N):
    rng = np.random.default_rng(0)
    a = 0.2 * rng.standard_normal((5040, K))
    b = O.lie(group, "log", O.lie(group, "exp", a, dtype=np.float64), dtype=np.float64)
    assert np.allclose(a, b, atol=1e-8)


@pytest.mark.parametrize("group,K,N", GROUPS)
def test_inv(group, K, N):
    rng = np.random.default_rng(1)
    X = O.lie(group, "exp", 0.1 * rng.standard_normal((120, K)), dtype=np.float64)
    a = O.lie(group, "log", O.lie(group, "mul", X, O.lie(group, "inv", X, dtype=np.float64), dtype=np.float64),
              dtype=np.float64)
    assert np.allclose(a, 0, atol=1e-8)


@pytest.mark.parametrize("group,K,N", GROUPS)
def test_adj(group, K, N):
    """X * Exp(a) == Exp(Adj_X a) * X   (run_tests.py:30-41)"""
    rng = np.random.default_rng(2)
    X = O.lie(group, "exp", rng.standard_normal((120, K)), dtype=np.float64)
    a = rng.standard_normal((120, K))
    b = O.lie(group, "adj", X, a, dtype=np.float64)
    Y1 = O.lie(group, "mul", X, O.lie(group, "exp", a, dtype=np.float64), dtype=np.float64)
    Y2 = O.lie(group, "mul", O.lie(group, "exp", b, dtype=np.float64), X, dtype=np.float64)
    c = O.lie(group, "log", O.lie(group, "mul", Y1, O.lie(group, "inv", Y2, dtype=np.float64), dtype=np.float64),
              dtype=np.float64)
    assert np.allclose(c, 0, atol=1e-8)


@pytest.mark.parametrize("group,K,N", GROUPS)
def test_act_vs_matrix(group, K, N):
    rng = np.random.default_rng(3)
    X = O.lie(group, "exp", rng.standard_normal((60, K)), dtype=np.float64)
    p = rng.standard_normal((60, 3))
    p1 = O.lie(group, "act", X, p, dtype=np.float64)
    T = O.lie(group, "matrix", X, dtype=np.float64)
    ph = np.concatenate([p, np.ones((60, 1))], -1)
    p2 = np.einsum("nij,nj->ni", T, ph)[:, :3]
    assert np.allclose(p1, p2, atol=1e-8)
    p4 = rng.standard_normal((60, 4))
    assert np.allclose(O.lie(group, "act4", X, p4, dtype=np.float64), np.einsum("nij,nj->ni", T, p4), atol=1e-8)


def test_adjT_is_transpose_of_adj():
    rng = np.random.default_rng(4)
    X = O.lie(O.SE3, "exp", rng.standard_normal((40, 6)), dtype=np.float64)
    Ad = np.stack([O.lie(O.SE3, "adj", X, np.tile(np.eye(6)[c], (40, 1)), dtype=np.float64) for c in range(6)], -1)
    a = rng.standard_normal((40, 6))
    assert np.allclose(O.lie(O.SE3, "adjT", X, a, dtype=np.float64), np.einsum("nji,nj->ni", Ad, a), atol=1e-12)


def test_small_angle_branches_f32():
    a = np.zeros((3, 6), np.float32)
    a[1, 3:] = 1e-7
    a[2, 3:] = 1e-4
    X = O.lie(O.SE3, "exp", a)
    assert np.allclose(np.linalg.norm(X[:, 3:], axis=1), 1, atol=1e-6)
    assert np.allclose(O.lie(O.SE3, "log", X), a, atol=1e-6)


def test_lietorch_python_layer_golden(golden_dir):
    """groups.py op wiring + broadcasting, captured from the reference's Python files."""
    g = np.load(os.path.join(golden_dir, "lietorch_py.npz"))
    a, b = g["a"], g["b"]
    X = O.lie(O.SE3, "exp", a.reshape(-1, 6), dtype=np.float64).reshape(3, 4, 7)
    Y = O.lie(O.SE3, "exp", b.reshape(-1, 6), dtype=np.float64).reshape(3, 1, 7)
    assert np.array_equal(X, g["X"]) and np.array_equal(Y, g["Y"])
    Yb = np.broadcast_to(Y, (3, 4, 7)).reshape(-1, 7)
    assert np.array_equal(O.lie(O.SE3, "mul", X.reshape(-1, 7), Yb, dtype=np.float64).reshape(3, 4, 7), g["XY"])
    assert np.array_equal(O.lie(O.SE3, "inv", X.reshape(-1, 7), dtype=np.float64).reshape(3, 4, 7), g["Xinv"])
    assert np.array_equal(O.lie(O.SE3, "log", X.reshape(-1, 7), dtype=np.float64).reshape(3, 4, 6), g["logX"])
    Xb = np.broadcast_to(X[:, :, None], (3, 4, 5, 7)).reshape(-1, 7)
    assert np.array_equal(O.lie(O.SE3, "act4", Xb, g["p4"].reshape(-1, 4), dtype=np.float64).reshape(3, 4, 5, 4),
                          g["act4"])
    assert np.allclose(O.lie(O.SE3, "matrix", X.reshape(-1, 7), dtype=np.float64).reshape(3, 4, 4, 4), g["matrix"],
                       atol=1e-15)
    assert np.array_equal(O.lie(O.SE3, "adjT", X.reshape(-1, 7), a.reshape(-1, 6), dtype=np.float64).reshape(3, 4, 6),
                          g["adjT"])
    dX = O.lie(O.SE3, "exp", a.reshape(-1, 6), dtype=np.float64)
    assert np.array_equal(O.lie(O.SE3, "mul", dX, X.reshape(-1, 7), dtype=np.float64).reshape(3, 4, 7), g["retr"])
