"""Drop-in for the reference's `cuda_ba` extension (cdvslam/fastba/ba.cpp:183-188)."""
from cdv_slam_amd import ops


def forward(poses, patches, intrinsics, target, weight, lmbda, ii, jj, kk, PPF, t0, t1, iterations, eff_impl):
    """ba (ba.cpp:31-45): in-place BA, returns an empty list like the reference."""
    return ops.ba_forward(poses, patches, intrinsics, target, weight, lmbda, ii, jj, kk, PPF, t0, t1, iterations,
                          eff_impl)


def neighbors(ii, jj):
    """neighbors (ba.cpp:59-97) -> [ix, jx]"""
    ix, jx = ops.neighbors(ii, jj)
    return [ix, jx]


def reproject(poses, patches, intrinsics, ii, jj, kk):
    """reproject (ba.cpp:50-57) -> Tensor [1,E,2,P,P]"""
    P = patches.shape[-1]
    return ops.fastba_reproject(poses.view(-1, 7), patches.view(-1, 3, P, P), intrinsics.view(-1, 4), ii, jj, kk)


def solve_system(J_Ginv_i, J_Ginv_j, ii, jj, res, ep, lm, freen):
    raise NotImplementedError("cuda_ba.solve_system (Sim3 pose-graph solve of the classic loop closure, "
                              "ba.cpp:120-181) is outside the update hot path")
