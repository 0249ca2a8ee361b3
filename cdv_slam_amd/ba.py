"""Host-side mirror of the reference's Python bundle adjustment, cdvslam/ba.py:86-185 (`BA`) -- forward values, on
the GPU.  Same signature and the same semantics where they differ from fastba (ba_cuda.cu): `ep` damping argument
(ba.py:66-73), 250 px residual gate (:98), explicit `bounds` (:100-106), inverse-depth clamp to [1e-3, 10] (:179),
`fixedp` leading poses held fixed, out-of-place result, zeros when the Cholesky factorisation fails (:16-20).

The per-edge work -- reprojection, validity and the three Jacobian blocks -- is ONE launch of the fused HIP kernel
(cdv_transform with Jacobians; the reference composes ~20 torch ops); the scatter sums and the dense 6n x 6n solve are
plain torch ops on the same device (this entry point serves the 10-frame graph of BASELINE.json configs[0] and
the training-time caller net_cdv.py:559, not the per-frame hot path -- that one is fastba.BA).
No autograd: inference only."""
import torch

from . import ops
from .lietorch import SE3


def _scatter_blocks(A, ri, ci, n, m):
    """safe_scatter_add_mat (ba.py:40-42): blocks A [E,p,q] summed into [n,m,p,q] for in-range (ri, ci)."""
    ok = (ri >= 0) & (ci >= 0) & (ri < n) & (ci < m)
    out = torch.zeros((n * m,) + tuple(A.shape[1:]), dtype=A.dtype, device=A.device)
    out.index_add_(0, (ri[ok] * m + ci[ok]), A[ok])
    return out.view(n, m, *A.shape[1:])


def _scatter_rows(b, ri, n):
    """safe_scatter_add_vec (ba.py:44-46)."""
    ok = (ri >= 0) & (ri < n)
    out = torch.zeros((n,) + tuple(b.shape[1:]), dtype=b.dtype, device=b.device)
    out.index_add_(0, ri[ok], b[ok])
    return out


def BA(poses, patches, intrinsics, targets, weights, lmbda, ii, jj, kk, bounds, ep=100.0, PRINT=False, fixedp=1,
       structure_only=False):
    """poses: SE3 [1,n,7]; patches [1,m,3,P,P]; intrinsics [1,n,4]; targets / weights [1,E,2] -> (poses: SE3, patches)"""
    pdata = poses.data if isinstance(poses, SE3) else poses
    n = int(max(int(ii.max()), int(jj.max()))) + 1
    coords, v, (Ji, Jj, Jz) = ops.transform(pdata, patches, intrinsics, ii, jj, kk, jacobian=True)
    P = coords.shape[3]
    r = targets - coords[..., P // 2, P // 2, :]                       # [1,E,2]
    v = v * (r.norm(dim=-1) < 250).float()                             # ba.py:98
    c = coords[..., P // 2, P // 2, :]
    in_bounds = (c[..., 0] > bounds[0]) & (c[..., 1] > bounds[1]) & (c[..., 0] < bounds[2]) & (c[..., 1] < bounds[3])
    v = v * in_bounds.float()                                          # ba.py:100-106
    # From here on float64: the normal equations are summed with atomics (index_add_) whose order changes from run to
    # run, and at ep = 1 the system is weakly damped -- in float32 that shows as 1e-4 run-to-run differences in the
    # update.  Jacobians and residuals are the float32 values of the fused kernel, as in the reference.
    f64 = torch.float64
    r = (v[..., None] * r)[0].unsqueeze(-1).to(f64)                    # [E,2,1]
    w = (v[..., None] * weights)[0].unsqueeze(-1).to(f64)              # [E,2,1]
    Ji, Jj, Jz = Ji[0].to(f64), Jj[0].to(f64), Jz[0].to(f64)
    wJiT, wJjT, wJzT = (w * Ji).transpose(1, 2), (w * Jj).transpose(1, 2), (w * Jz).transpose(1, 2)
    Bii, Bij, Bji, Bjj = wJiT @ Ji, wJiT @ Jj, wJjT @ Ji, wJjT @ Jj
    Eik, Ejk = wJiT @ Jz, wJjT @ Jz
    vi, vj = wJiT @ r, wJjT @ r

    n = n - fixedp
    i2, j2 = ii - fixedp, jj - fixedp
    kx, ku = torch.unique(kk, return_inverse=True)
    m = kx.numel()
    B = (_scatter_blocks(Bii, i2, i2, n, n) + _scatter_blocks(Bij, i2, j2, n, n)
         + _scatter_blocks(Bji, j2, i2, n, n) + _scatter_blocks(Bjj, j2, j2, n, n))          # [n,n,6,6]
    E = _scatter_blocks(Eik, i2, ku, n, m) + _scatter_blocks(Ejk, j2, ku, n, m)              # [n,m,6,1]
    C = _scatter_rows(wJzT @ Jz, ku, m).view(m)
    vv = (_scatter_rows(vi, i2, n) + _scatter_rows(vj, j2, n)).view(6 * n)
    ww = _scatter_rows(wJzT @ r, ku, m).view(m)
    lm = lmbda.reshape(-1)[0].to(f64) if torch.is_tensor(lmbda) else float(lmbda)
    Q = 1.0 / (C + lm)                                                 # ba.py:151

    if structure_only or n == 0:
        dZ = Q * ww
        dX = None
    else:
        Ed = E[..., 0].permute(0, 2, 1).reshape(6 * n, m)              # dense [6n, m]
        Bd = B.permute(0, 2, 1, 3).reshape(6 * n, 6 * n)
        S = Bd - (Ed * Q) @ Ed.t()
        y = vv - (Ed * Q) @ ww
        A = S + (ep + 1e-4 * S) * torch.eye(6 * n, dtype=S.dtype, device=S.device)   # block_solve, ba.py:66-73
        L, info = torch.linalg.cholesky_ex(A)
        if int(info) != 0:                                             # CholeskySolver, ba.py:16-20: zeros on failure
            dX = torch.zeros(6 * n, dtype=S.dtype, device=S.device)
        else:
            dX = torch.cholesky_solve(y[:, None], L)[:, 0]
        dZ = Q * (ww - Ed.t() @ dX)
        dX = dX.view(n, 6)

    disps = patches[0, :, 2].clone()
    disps[kx] = disps[kx] + dZ.view(m, 1, 1).to(disps.dtype)           # disp_retr, ba.py:49-51
    new_patches = patches.clone()
    new_patches[0, :, 2] = disps.clamp(min=1e-3, max=10.0)             # ba.py:179
    if dX is not None and n > 0:
        upd = torch.zeros((pdata.shape[-2], 6), dtype=pdata.dtype, device=pdata.device)
        upd[fixedp + torch.arange(n, device=pdata.device)] = dX.to(pdata.dtype)
        new = SE3.exp(upd[None]) * SE3(pdata.reshape(1, -1, 7))       # poses.retr(...), groups.py:157-160
        return (new if isinstance(poses, SE3) else new.data), new_patches
    return poses, new_patches
