"""`altcorr` operator surface (names of cdvslam/altcorr/correlation.py:51-75), inference forward only, on the HIP kernels."""
import torch

from .. import ops

_MODES = ("bilinear", "upperleft")


def _forward_only(*ts):
    if torch.is_grad_enabled() and any(t.requires_grad for t in ts if torch.is_tensor(t)):
        raise NotImplementedError("altcorr backward (training) is out of scope of the HIP update path")


def corr(fmap1, fmap2, coords, ii, jj, radius=1, dropout=1):
    """Local correlation volume + bilinear blend (correlation.py:74-75 -> CorrLayer.forward :6-13).
    Returns [B, M, 2r+1 (x), 2r+1 (y), P, P]."""
    _forward_only(fmap1, fmap2)
    return ops.corr_forward(fmap1, fmap2, coords, ii, jj, radius)


def patchify(net, coords, radius, mode='bilinear'):
    """(2r+1)^2 samples around coords [B,M,2] of net [B,C,H,W] (correlation.py:51-71): 'bilinear' blends the four
    neighbouring gathers (float32 result), 'upperleft' keeps the one sample at floor(coords), any other mode returns
    the raw (2r+2)^2 gather.  Gather and blend are one launch (cdv_patchify_blend / cdv_patchify_fwd)."""
    _forward_only(net)
    if mode in _MODES:
        return ops.patchify_blend(net, coords, radius, mode)
    return ops.patchify_forward(net, coords, radius)
