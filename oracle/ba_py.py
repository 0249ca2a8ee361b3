"""oracle/ba_py.py -- TEST INFRASTRUCTURE ONLY.

numpy restatement of the reference's differentiable Python bundle adjustment
(cdvslam/ba.py:86-185 `BA`, helpers :40-76) -- the "CPU Gauss-Newton" of BASELINE.json
configs[0].  Forward values only (no autograd).  Jacobians come from the C oracle's
pops.transform restatement (projective_ops.py:53-108).

Pinned against the reference's own ba.py executed under the shimmed import
(tests/golden/make_golden.py -> tests/golden/ba_py_*.npz).
"""
import numpy as np

from . import oracle as O


def _scatter_mat(A, ii, jj, n, m):
    """safe_scatter_add_mat (ba.py:40-42): A [E,p,q] summed into [n*m,p,q] for in-range (ii,jj)."""
    v = (ii >= 0) & (jj >= 0) & (ii < n) & (jj < m)
    out = np.zeros((n * m,) + A.shape[1:], A.dtype)
    np.add.at(out, ii[v] * m + jj[v], A[v])
    return out


def _scatter_vec(b, ii, n):
    """safe_scatter_add_vec (ba.py:44-46)."""
    v = (ii >= 0) & (ii < n)
    out = np.zeros((n,) + b.shape[1:], b.dtype)
    np.add.at(out, ii[v], b[v])
    return out


def BA(poses, patches, intrinsics, targets, weights, lmbda, ii, jj, kk, bounds, ep=100.0, fixedp=1,
       structure_only=False, dtype=np.float32):
    """poses [n,7], patches [m,3,P,P], intrinsics [n,4], targets/weights [E,2] -> (poses, patches, info)."""
    poses = np.asarray(poses, dtype).reshape(-1, 7)
    P = patches.shape[-1]
    patches = np.asarray(patches, dtype).reshape(-1, 3, P, P)
    intrinsics = np.asarray(intrinsics, dtype).reshape(-1, 4)
    targets = np.asarray(targets, dtype).reshape(-1, 2)
    weights = np.asarray(weights, dtype).reshape(-1, 2)
    ii, jj, kk = (np.asarray(x, np.int64) for x in (ii, jj, kk))

    n = int(max(ii.max(), jj.max())) + 1
    coords, v, (Ji, Jj, Jz) = O.transform(poses, patches, intrinsics, ii, jj, kk, jacobian=True, dtype=dtype)
    c = coords[:, P // 2, P // 2, :]
    r = targets - c
    v = v * (np.linalg.norm(r, axis=-1) < 250).astype(dtype)          # ba.py:98
    in_bounds = (c[:, 0] > bounds[0]) & (c[:, 1] > bounds[1]) & (c[:, 0] < bounds[2]) & (c[:, 1] < bounds[3])
    v = v * in_bounds.astype(dtype)                                     # ba.py:100-106

    r = (v[:, None] * r)[..., None]                                     # [E,2,1]
    w = (v[:, None] * weights)[..., None]                               # [E,2,1]

    wJiT = np.swapaxes(w * Ji, 1, 2)
    wJjT = np.swapaxes(w * Jj, 1, 2)
    wJzT = np.swapaxes(w * Jz, 1, 2)
    Bii, Bij = wJiT @ Ji, wJiT @ Jj
    Bji, Bjj = wJjT @ Ji, wJjT @ Jj
    Eik, Ejk = wJiT @ Jz, wJjT @ Jz
    vi, vj = wJiT @ r, wJjT @ r

    n = n - fixedp
    ii = ii - fixedp
    jj = jj - fixedp
    kx, ku = np.unique(kk, return_inverse=True)
    m = len(kx)

    B = (_scatter_mat(Bii, ii, ii, n, n) + _scatter_mat(Bij, ii, jj, n, n)
         + _scatter_mat(Bji, jj, ii, n, n) + _scatter_mat(Bjj, jj, jj, n, n)).reshape(n, n, 6, 6)
    E = (_scatter_mat(Eik, ii, ku, n, m) + _scatter_mat(Ejk, jj, ku, n, m)).reshape(n, m, 6, 1)
    C = _scatter_vec(wJzT @ Jz, ku, m)                                  # [m,1,1]
    vv = (_scatter_vec(vi, ii, n) + _scatter_vec(vj, jj, n)).reshape(n, 6)
    ww = _scatter_vec(wJzT @ r, ku, m)                                  # [m,1,1]

    Q = (1.0 / (C + dtype(lmbda))).astype(dtype)                        # ba.py:151
    Ed = np.transpose(E[..., 0], (0, 2, 1)).reshape(6 * n, m)           # dense [6n, m]
    info = 0
    if structure_only or n == 0:
        dZ = (Q * ww).reshape(m)
        dX = None
    else:
        q = Q.reshape(m)
        Bd = np.transpose(B, (0, 2, 1, 3)).reshape(6 * n, 6 * n)
        S = Bd - (Ed * q) @ Ed.T
        y = vv.reshape(6 * n) - (Ed * q) @ ww.reshape(m)
        A = S + (dtype(ep) + dtype(1e-4) * S) * np.eye(6 * n, dtype=dtype)  # block_solve, ba.py:66-73
        try:
            L = np.linalg.cholesky(A.astype(np.float64)).astype(dtype) if dtype == np.float64 else \
                np.linalg.cholesky(A)
            z = np.linalg.solve(L, y)
            dX = np.linalg.solve(L.T, z).astype(dtype)
        except np.linalg.LinAlgError:                                   # ba.py:16-20: zeros on failure
            info = 1
            dX = np.zeros(6 * n, dtype)
        dZ = q * (ww.reshape(m) - Ed.T @ dX)
        dX = dX.reshape(n, 6)

    disps = patches[:, 2].copy()
    disps[kx] = disps[kx] + dZ.reshape(m, 1, 1)                         # disp_retr, ba.py:49-51
    disps = np.clip(disps, 1e-3, 10.0).astype(dtype)                    # ba.py:179
    patches = patches.copy()
    patches[:, 2] = disps
    if dX is not None and n > 0:
        upd = np.zeros((poses.shape[0], 6), dtype)
        upd[fixedp + np.arange(n)] = dX
        dG = O.lie(O.SE3, "exp", upd, dtype=dtype)                      # poses.retr(...), groups.py:157-160
        poses = O.lie(O.SE3, "mul", dG, poses, dtype=dtype)
    return poses, patches, info
