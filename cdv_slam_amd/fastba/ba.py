"""Host-side mirror of cdvslam/fastba/ba.py."""
from .. import ops


def neighbors(ii, jj):
    """Temporal neighbour edges per patch (cdvslam/fastba/ba.cpp:59-97): ii = patch ids (kk), jj = frames."""
    return ops.neighbors(ii, jj)


def reproject(poses, patches, intrinsics, ii, jj, kk):
    """cuda_ba.reproject (ba_cuda.cu:614-646)."""
    P = patches.shape[-1]
    return ops.fastba_reproject(poses.view(-1, 7), patches.view(-1, 3, P, P), intrinsics.view(-1, 4), ii, jj, kk)


def BA(poses, patches, intrinsics, target, weight, lmbda, ii, jj, kk, t0, t1, M, iterations, eff_impl=False):
    """In-place bundle adjustment (ba.py:7-8 -> ba_cuda.cu:462-611).  `poses` may be the raw [1,N,7] tensor."""
    data = poses.data if hasattr(poses, "data") else poses
    return ops.ba_forward(data, patches, intrinsics, target, weight, lmbda, ii, jj, kk, M, t0, t1, iterations,
                          eff_impl)
