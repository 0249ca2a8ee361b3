"""TEST INFRASTRUCTURE: the stated BA tolerances and the gauge-free comparisons used where a
graph leaves a weak direction in the reduced system.

Two classes of synthetic graph (conditioning measured with the float64 oracle, `cond` = ratio of the extreme
eigenvalues of the damped S of iteration 0; `f32 oracle` = what the reference's arithmetic in float32, sequential sums,
deviates from float64 by after two iterations -- the floor any float32 implementation sits on):

  well-conditioned   small (cond 4e2), default (3e2), stress (2e3): a fixed-pose window anchors scale;
                     f32 oracle: t 1-2e-7, q 1e-7, inverse depth 3e-7
  weak scale gauge   init (1e5), pr1 (8e4), global (6e4), global_l (8e4): one fixed pose only, so monocular scale is held
                     by the +1.0 damping of ba_cuda.cu:589 alone; f32 oracle: t 2-4e-5, q 0.7-14e-7, inverse depth 0.5-17e-5

For the second class the raw translation / depth bound cannot be tighter than float32 allows along that one direction,
so the bounds that carry the parity claim there are the gauge-free ones (all at the 1e-5 level or below):
Sim(3)-aligned ATE, the reprojection cost after the update, the error of dX inside the well-determined eigen-subspace
of S, and the backward error of the solve.
"""
import json
import os

import numpy as np

from cdv_slam_amd import metrics
from oracle import oracle as O

# against the float64 oracle after BA(iterations=2): absolute translation / quaternion error, relative inverse-depth
# error (of max(|d|, 1e-2)), Sim(3)-aligned ATE-RMSE over all frames, relative reprojection cost
BA_TOL = {
    "small":    dict(t=1e-5, q=1e-6, d=1e-4, ate=1e-6, cost=1e-6),
    "default":  dict(t=1e-5, q=1e-6, d=1e-4, ate=1e-6, cost=1e-6),
    "stress":   dict(t=1e-5, q=1e-6, d=1e-4, ate=1e-6, cost=1e-6),
    "init":     dict(t=5e-5, q=1e-6, d=2e-4, ate=1e-6, cost=1e-6),
    "pr1":      dict(t=3e-5, q=1e-6, d=1e-4, ate=1e-6, cost=1e-6),
    "global":   dict(t=1e-4, q=5e-6, d=2e-4, ate=5e-6, cost=1e-6),
    "global_l": dict(t=1e-4, q=5e-6, d=2e-4, ate=1e-5, cost=1e-6),
    "global_xl": dict(t=2e-4, q=1e-5, d=5e-4, ate=2e-5, cost=1e-6),
}
# iteration 0: dX error inside the eigen-subspace of S with eigenvalue >= 1e-2 of the largest (absolute, rad / scene
# units; |dX| is 3e-3 .. 3e-2), and the relative residual of the solve |S dX - y| / |y| with the kernel's own S, y, dX
DX_STRONG_TOL = 5e-6
DX_WEAK_TOL = {"small": 1e-5, "default": 1e-5, "stress": 2e-5, "init": 1e-4, "pr1": 5e-5, "global": 5e-4, "global_l": 5e-4,
               "global_xl": 1e-3}
SOLVE_RESIDUAL_TOL = 2e-5
DZ_TOL = {"small": 1e-5, "default": 1e-5, "stress": 1e-5, "init": 2.5e-4, "pr1": 1e-4, "global": 1e-4, "global_l": 1e-4,
          "global_xl": 5e-4}


def _log(kind, name, got, tol):
    """measured values next to their bounds, one JSON line per check, when CDV_TEST_LOG names a file (profiles/ keeps
    the GPU box's)"""
    path = os.environ.get("CDV_TEST_LOG")
    if path:
        with open(path, "a") as f:
            f.write(json.dumps({"check": kind, "graph": name, "measured": {k: float(v) for k, v in got.items()},
                                "bound": {k: float(v) for k, v in tol.items()}}) + "\n")


def reprojection_cost(poses, patches, st):
    """sum w |target - x1|^2 over all edges, evaluated in float64 (fastba's projection, ba_cuda.cu:299-300)"""
    c = O.fastba_reproject(np.asarray(poses, np.float64), np.asarray(patches, np.float64), st.intrinsics[0], st.ii, st.jj,
                           st.kk, dtype=np.float64)
    P = c.shape[-1]
    r = st.target.astype(np.float64) - c[:, :, P // 2, P // 2]
    return float((st.weight.astype(np.float64) * r * r).sum())


def check_end_state(name, st, poses, patches, p64, x64):
    """poses / patches after BA(2 it) on the GPU against the float64 oracle's, with the table's numbers"""
    tol = BA_TOL[name]
    et = np.abs(poses[:, :3] - p64[:, :3]).max()
    eq = np.abs(poses[:, 3:] - p64[:, 3:]).max()
    d, d64 = patches[:, 2, 0, 0].astype(np.float64), x64[:, 2, 0, 0]
    ed = (np.abs(d - d64) / np.maximum(np.abs(d64), 1e-2)).max()      # relative error per patch (of max(|d|, 1e-2))
    ate = metrics.ate_rmse(p64[:st.n], poses[:st.n])
    c64 = reprojection_cost(p64, x64, st)
    cg = reprojection_cost(poses, patches, st)
    ec = abs(cg - c64) / c64
    got = dict(t=et, q=eq, d=ed, ate=ate, cost=ec)
    print("BA end state [%s]: " % name + "  ".join("%s %.2e (<= %.0e)" % (k, got[k], tol[k]) for k in got))
    _log("end_state", name, got, tol)
    for k in got:
        assert got[k] <= tol[k], (name, k, got[k], tol[k])
    return got


def check_iteration0(name, dbg, o64):
    """dX / dZ of iteration 0: error split by the eigen-subspaces of the float64 S; backward error of the solve"""
    S64 = o64["S"]
    w, V = np.linalg.eigh((S64 + S64.T) / 2)
    strong = w >= 1e-2 * w[-1]
    dX = dbg["dX"].astype(np.float64).reshape(-1)
    coef = V.T @ (dX - o64["dX"].reshape(-1))
    e_strong = np.abs(coef[strong]).max()
    e_weak = np.abs(coef[~strong]).max() if (~strong).any() else 0.0
    S = np.tril(dbg["S"].astype(np.float64))       # the lower triangle is what every path accumulates and factors
    S = S + np.tril(S, -1).T
    y = dbg["y"].astype(np.float64)
    res = np.linalg.norm(S @ dX - y) / np.linalg.norm(y)
    U = len(o64["dZ"])
    e_dz = np.abs(dbg["dZ"][:U].astype(np.float64) - o64["dZ"]).max()
    print("BA iteration 0 [%s]: dX strong %.2e (<= %.0e) weak %.2e (<= %.0e), solve residual %.2e (<= %.0e), dZ %.2e (<= %.0e)"
          % (name, e_strong, DX_STRONG_TOL, e_weak, DX_WEAK_TOL[name], res, SOLVE_RESIDUAL_TOL, e_dz, DZ_TOL[name]))
    _log("iteration0", name, dict(dX_strong=e_strong, dX_weak=e_weak, solve_residual=res, dZ=e_dz),
         dict(dX_strong=DX_STRONG_TOL, dX_weak=DX_WEAK_TOL[name], solve_residual=SOLVE_RESIDUAL_TOL, dZ=DZ_TOL[name]))
    assert e_strong <= DX_STRONG_TOL, (name, e_strong)
    assert e_weak <= DX_WEAK_TOL[name], (name, e_weak)
    assert res <= SOLVE_RESIDUAL_TOL, (name, res)
    assert e_dz <= DZ_TOL[name], (name, e_dz)
