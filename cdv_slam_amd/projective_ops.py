"""`projective_ops` operator surface (names and argument meaning of cdvslam/projective_ops.py) on the fused HIP kernels.

`transform` (+ validity, Jacobians, translation-only), `flow_mag` and `point_cloud` are ONE launch each
(cdv_transform / cdv_flow_mag / cdv_point_cloud) where the reference composes gathers, lietorch Inv / Mul / Act4 and
elementwise ops (projective_ops.py:53-130).  The kernels serve what the update path hands them -- float32 state, batch
1, SE3 poses -- and anything else raises: there is no composed fallback.  `iproj` / `proj` are the two pinhole maps as
plain tensor expressions for callers that want them on their own.
"""
import torch

from . import ops

MIN_DEPTH = 0.2


def _fc(intrinsics):
    """[b,n,4] -> focal lengths and principal point as [b,n,2,1,1] each"""
    k = intrinsics[..., None, None]
    return k[:, :, 0:2], k[:, :, 2:4]


def iproj(patches, intrinsics):
    """pixel grid + inverse depth -> homogeneous rays: patches [b,n,3,P,P], intrinsics [b,n,4] -> [b,n,P,P,4]
    (x - cx) / fx, (y - cy) / fy, 1, d    (projective_ops.py:19-29)"""
    f, c = _fc(intrinsics)
    rays = (patches[:, :, 0:2] - c) / f
    d = patches[:, :, 2:3]
    return torch.cat([rays, torch.ones_like(d), d], dim=2).movedim(2, -1)


def proj(X, intrinsics, depth=False):
    """homogeneous points [b,n,P,P,4] -> pixels [b,n,P,P,2] (or 3 with the inverse depth), 1 / max(Z, 0.1) as the
    reference clamps it (projective_ops.py:32-50)"""
    f, c = _fc(intrinsics)
    inv_z = X[..., 2].clamp(min=0.1).reciprocal()
    uv = f.movedim(2, -1) * (X[..., 0:2] * inv_z[..., None]) + c.movedim(2, -1)
    return torch.cat([uv, inv_z[..., None]], dim=-1) if depth else uv


def _kernel_pose_rows(poses, what):
    """the [1,n,7] float32 tensor the kernels read, or an error naming what is not served"""
    # a tensor, this package's LieGroup, or the caller's own group object (the reference's lietorch.SE3 when only
    # projective_ops is substituted, install_dropin(package=)): anything that carries its rows in `.data` and names its group
    group = not torch.is_tensor(poses)
    data = getattr(poses, "data", None) if group else poses
    if group and getattr(poses, "group_id", None) != 3:
        raise NotImplementedError("%s: SE3 poses only (Sim3 belongs to loop closure, out of scope)" % what)
    if not torch.is_tensor(data):
        raise TypeError("%s: poses must be an SE3 or a tensor" % what)
    if data.dtype != torch.float32 or data.dim() != 3 or data.shape[0] != 1:
        raise NotImplementedError("%s: the HIP path serves float32 poses of batch 1 (got %s %s); no composed fallback"
                                  % (what, data.dtype, tuple(data.shape)))
    return data


def transform(poses, patches, intrinsics, ii, jj, kk, depth=False, valid=False, jacobian=False, tonly=False):
    """patch k of frame i seen from frame j (projective_ops.py:53-113), one launch.

    -> coords [1,E,P,P,2]; valid=True: also (Z > 0.2) per pixel [1,E,P,P]; jacobian=True:
    (coords, (Z > 0.2) at the centre [1,E], (Ji [1,E,2,6], Jj [1,E,2,6], Jz [1,E,2,1]))."""
    if depth:
        raise NotImplementedError("transform(depth=True) is not on the update path")
    if not (valid or jacobian or tonly) and ops._fast and ops._env("CDV_DROPIN_FAST", "1") != "0":
        # the compiled lane (csrc/dropin_fast.cpp): the same checks, allocation and launch as below without the Python around them
        group = not torch.is_tensor(poses)
        data = getattr(poses, "data", None) if group else poses
        if torch.is_tensor(data) and data.is_cuda and not (group and getattr(poses, "group_id", None) != 3):
            r = ops._fast.transform(data, patches, intrinsics, ii, jj, kk, ops._stream())
            if r is not None:
                if type(r) is int:
                    ops._lib.check(r, "cdv_transform")
                return r.permute(0, 1, 3, 4, 2)
    if not (valid or jacobian):
        # every inference caller turns the result into [1,E,2,P,P] at once (`.permute(0, 1, 4, 2, 3).contiguous()`,
        # slam.py:328-329, loop_closure/long_term.py:118-129): the kernel writes THAT layout and the [1,E,P,P,2] tensor handed
        # back is a permuted view of it -- same values, same shape, and the caller's permute + contiguous() is then the
        # buffer itself instead of a 3.4 MB copy kernel
        return ops.transform(_kernel_pose_rows(poses, "transform"), patches, intrinsics, ii, jj, kk, layout_e2pp=True,
                             tonly=tonly).permute(0, 1, 3, 4, 2)
    return ops.transform(_kernel_pose_rows(poses, "transform"), patches, intrinsics, ii, jj, kk, layout_e2pp=False,
                         valid=valid, jacobian=jacobian, tonly=tonly)


def reproject(poses, patches, intrinsics, ii, jj, kk):
    """SLAM.reproject (cdvslam/slam.py:325-329): the coords in the [1,E,2,P,P] layout the correlation reads, written
    directly in that layout"""
    return ops.transform(_kernel_pose_rows(poses, "reproject"), patches, intrinsics, ii, jj, kk, layout_e2pp=True)


def point_cloud(poses, patches, intrinsics, ix):
    """world points of patches [1,m,3,P,P] whose frames are ix [m] -> [1,m,P,P,4] (projective_ops.py:115-117)"""
    if patches.dtype != torch.float32 or patches.shape[1] != ix.numel():
        raise NotImplementedError("point_cloud: float32 patches, one frame index per patch")
    return ops.point_cloud(_kernel_pose_rows(poses, "point_cloud"), patches, intrinsics, ix)


def flow_mag(poses, patches, intrinsics, ii, jj, kk, beta=0.3):
    """keyframe-test flow magnitude, beta * |full flow| + (1 - beta) * |translation-only flow|, and the validity mask
    (projective_ops.py:120-130) -> ([1,E,P,P] f32, [1,E,P,P] bool)"""
    if patches.dtype != torch.float32:
        raise NotImplementedError("flow_mag: float32 patches")
    return ops.flow_mag(_kernel_pose_rows(poses, "flow_mag"), patches, intrinsics, ii, jj, kk, beta)
