"""Host-side mirror of cdvslam/projective_ops.py on the fused HIP reprojection kernel.

`transform` is ONE kernel launch (cdv_transform) instead of the reference's gather + lietorch Inv /
Mul / Act4 + elementwise chain (projective_ops.py:53-113); the remaining helpers are thin torch code.
"""
import torch

from . import ops
from .lietorch import SE3

MIN_DEPTH = 0.2


def extract_intrinsics(intrinsics):
    return intrinsics[..., None, None, :].unbind(dim=-1)


def coords_grid(ht, wd, **kwargs):
    y, x = torch.meshgrid(torch.arange(ht).to(**kwargs).float(), torch.arange(wd).to(**kwargs).float(),
                          indexing="ij")
    return torch.stack([x, y], dim=-1)


def iproj(patches, intrinsics):
    """inverse projection (projective_ops.py:19-29): patches [b,n,3,P,P], intrinsics [b,n,4] -> [b,n,P,P,4]"""
    x, y, d = patches.unbind(dim=2)
    fx, fy, cx, cy = intrinsics[..., None, None].unbind(dim=2)
    return torch.stack([(x - cx) / fx, (y - cy) / fy, torch.ones_like(d), d], dim=-1)


def proj(X, intrinsics, depth=False):
    """projection with d = 1 / Z.clamp(min=0.1) (projective_ops.py:32-50)"""
    X, Y, Z, W = X.unbind(dim=-1)
    fx, fy, cx, cy = intrinsics[..., None, None].unbind(dim=2)
    d = 1.0 / Z.clamp(min=0.1)
    x = fx * (d * X) + cx
    y = fy * (d * Y) + cy
    if depth:
        return torch.stack([x, y, d], dim=-1)
    return torch.stack([x, y], dim=-1)


def _pose_data(poses):
    return poses.data if isinstance(poses, SE3) else poses


def transform(poses, patches, intrinsics, ii, jj, kk, depth=False, valid=False, jacobian=False, tonly=False):
    """projective transform of patch k from frame i to frame j (projective_ops.py:53-113).

    Returns coords [b,E,P,P,2]; with valid=True also (X1.z > 0.2) [b,E,P,P]; with jacobian=True
    (coords, (Z > 0.2) [b,E], (Ji [b,E,2,6], Jj [b,E,2,6], Jz [b,E,2,1]))."""
    data = _pose_data(poses)
    if not isinstance(poses, SE3) and not torch.is_tensor(poses):
        raise NotImplementedError("transform: only SE3 poses are on the update path (Sim3 is loop closure)")
    if depth or data.dtype != torch.float32 or data.shape[0] != 1:
        return _transform_composed(SE3(data), patches, intrinsics, ii, jj, kk, depth, valid, jacobian, tonly)
    return ops.transform(data, patches, intrinsics, ii, jj, kk, layout_e2pp=False, valid=valid, jacobian=jacobian,
                         tonly=tonly)


def reproject(poses, patches, intrinsics, ii, jj, kk):
    """SLAM.reproject (cdvslam/slam.py:325-329): transform(...).permute(0,1,4,2,3).contiguous() -> [1,E,2,P,P],
    written directly in that layout."""
    return ops.transform(_pose_data(poses), patches, intrinsics, ii, jj, kk, layout_e2pp=True)


def _transform_composed(poses, patches, intrinsics, ii, jj, kk, depth, valid, jacobian, tonly):
    """General path out of individual Lie ops (float64 poses, batch > 1, depth=True)."""
    if jacobian:
        raise NotImplementedError("transform(jacobian=True) is fused for float32 / batch 1 only")
    X0 = iproj(patches[:, kk], intrinsics[:, ii])
    Gij = poses[:, jj] * poses[:, ii].inv()
    if tonly:
        Gij.data[..., 3:] = torch.as_tensor([0, 0, 0, 1], device=Gij.device, dtype=Gij.dtype)
    X1 = Gij[:, :, None, None] * X0
    x1 = proj(X1, intrinsics[:, jj], depth)
    if valid:
        return x1, (X1[..., 2] > 0.2).float()
    return x1


def point_cloud(poses, patches, intrinsics, ix):
    """world points of patches (projective_ops.py:115-117)"""
    data = _pose_data(poses)
    if (data.dtype == torch.float32 and data.shape[0] == 1 and patches.dtype == torch.float32
            and patches.shape[1] == ix.numel()):
        return ops.point_cloud(data, patches, intrinsics, ix)     # one launch
    poses = poses if isinstance(poses, SE3) else SE3(poses)
    return poses[:, ix, None, None].inv() * iproj(patches, intrinsics[:, ix])


def flow_mag(poses, patches, intrinsics, ii, jj, kk, beta=0.3):
    """flow magnitude used by the keyframe test (projective_ops.py:120-130)"""
    data = _pose_data(poses)
    if data.dtype == torch.float32 and data.shape[0] == 1 and patches.dtype == torch.float32:
        return ops.flow_mag(data, patches, intrinsics, ii, jj, kk, beta)   # one launch instead of three + torch ops
    coords0 = transform(poses, patches, intrinsics, ii, ii, kk)
    coords1, val = transform(poses, patches, intrinsics, ii, jj, kk, tonly=False, valid=True)
    coords2 = transform(poses, patches, intrinsics, ii, jj, kk, tonly=True)
    flow1 = (coords1 - coords0).norm(dim=-1)
    flow2 = (coords2 - coords0).norm(dim=-1)
    return beta * flow1 + (1 - beta) * flow2, (val > 0.5)
