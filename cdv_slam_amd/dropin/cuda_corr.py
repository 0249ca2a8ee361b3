"""Drop-in for the reference's `cuda_corr` extension (cdvslam/altcorr/correlation.cpp:57-63)."""
from cdv_slam_amd import ops


def forward(fmap1, fmap2, coords, ii, jj, radius):
    """corr_forward (correlation.cpp:35-42) -> [corr]"""
    return [ops.corr_forward(fmap1, fmap2, coords, ii, jj, int(radius))]


def backward(fmap1, fmap2, coords, ii, jj, corr_grad, radius):
    raise NotImplementedError("cuda_corr.backward is the training path (out of scope: inference runs under no_grad)")


def patchify_forward(net, coords, radius):
    """patchify_forward (correlation.cpp:49-52) -> [patches]"""
    return [ops.patchify_forward(net, coords, int(radius))]


def patchify_backward(net, coords, gradient, radius):
    raise NotImplementedError("cuda_corr.patchify_backward is the training path (out of scope)")
