"""Host-side mirror of cdvslam/altcorr/correlation.py (inference forward only)."""
import torch

from .. import ops


def _no_grad_inputs(*ts):
    if torch.is_grad_enabled() and any(t.requires_grad for t in ts if torch.is_tensor(t)):
        raise NotImplementedError("altcorr backward (training) is out of scope of the HIP update path")


def corr(fmap1, fmap2, coords, ii, jj, radius=1, dropout=1):
    """Local correlation volume + bilinear blend (correlation.py:74-75 -> CorrLayer.forward :6-13).
    Returns [B, M, 2r+1 (x), 2r+1 (y), P, P]."""
    _no_grad_inputs(fmap1, fmap2)
    return ops.corr_forward(fmap1, fmap2, coords, ii, jj, radius)


def patchify(net, coords, radius, mode='bilinear'):
    """Extract (2r+1)^2 patches around coords (correlation.py:51-71)."""
    _no_grad_inputs(net)
    if mode in ('bilinear', 'upperleft') and net.is_cuda and net.dim() == 4 and coords.dim() == 3:
        return ops.patchify_blend(net, coords, radius, mode)      # gather + blend in one launch
    patches = ops.patchify_forward(net, coords, radius)
    if mode == 'bilinear':
        offset = (coords - coords.floor()).to(net.device)
        dx, dy = offset[:, :, None, None, None].unbind(dim=-1)
        d = 2 * radius + 1
        x00 = (1 - dy) * (1 - dx) * patches[..., :d, :d]
        x01 = (1 - dy) * (dx) * patches[..., :d, 1:]
        x10 = (dy) * (1 - dx) * patches[..., 1:, :d]
        x11 = (dy) * (dx) * patches[..., 1:, 1:]
        return x00 + x01 + x10 + x11
    if mode == 'upperleft':
        return patches[..., :1, :1]
    return patches
