"""Trajectory error in the form the reference evaluates it: ATE-RMSE of the translation part after a Sim(3) alignment
(evaluate_tartan.py:63-70: evo main_ape.ape(..., pose_relation=translation_part, align=True, correct_scale=True)).

evo is not part of this repository; the alignment is the closed form it implements (S. Umeyama, "Least-squares
estimation of transformation parameters between two point patterns", PAMI 13(4), 1991): with mu, sigma^2 the means and
variances and Sigma = 1/n sum (y_i - mu_y)(x_i - mu_x)^T = U D V^T,  R = U S V^T  (S = diag(1, 1, det(U) det(V))),
c = trace(D S) / sigma_x^2,  t = mu_y - c R mu_x.
"""
import numpy as np


def umeyama_alignment(x, y, with_scale=True):
    """x, y: [3, n] point sets; returns (R [3,3], t [3], c) minimising sum |y_i - (c R x_i + t)|^2"""
    x, y = np.asarray(x, np.float64), np.asarray(y, np.float64)
    if x.shape != y.shape or x.shape[0] != 3:
        raise ValueError("umeyama_alignment: two [3, n] arrays")
    n = x.shape[1]
    mx, my = x.mean(1), y.mean(1)
    xc, yc = x - mx[:, None], y - my[:, None]
    var_x = (xc ** 2).sum() / n
    cov = yc @ xc.T / n
    U, D, Vt = np.linalg.svd(cov)
    S = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        S[2, 2] = -1.0
    R = U @ S @ Vt
    c = float(np.trace(np.diag(D) @ S) / var_x) if with_scale and var_x > 0 else 1.0
    t = my - c * R @ mx
    return R, t, c


def camera_centres(poses):
    """poses [n, 7] world-to-camera (t, q_xyzw) as the patch graph stores them -> camera centres [n, 3] = the
    translation part of poses.inv() (what SLAM.terminate returns, slam.py:305-307)"""
    p = np.asarray(poses, np.float64)
    t, q = p[:, :3], p[:, 3:7] / np.linalg.norm(p[:, 3:7], axis=1, keepdims=True)
    x, y, z, w = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                  2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                  2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1).reshape(-1, 3, 3)
    return -np.einsum("nji,nj->ni", R, t)       # -R^T t


def ate_rmse(poses_ref, poses_est, with_scale=True):
    """ATE-RMSE of the estimated trajectory against the reference one after Sim(3) (with_scale) / SE(3) alignment"""
    a, b = camera_centres(poses_ref), camera_centres(poses_est)
    R, t, c = umeyama_alignment(b.T, a.T, with_scale)
    err = a - (c * (R @ b.T).T + t)
    return float(np.sqrt((err ** 2).sum(1).mean()))
