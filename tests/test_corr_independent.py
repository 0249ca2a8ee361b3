"""The corr oracle (oracle/cdv_oracle.c::orc_corr -- the checker of rows a1 / a2) cross-checked against a second,
independently written torch-CPU statement (tests/corr_torch_ref.py): the reference holds no vectors for altcorr
(SURVEY.md 8c), so two separate readings of correlation_kernel.cu:82-136,213-232 have to agree instead --
bit for bit where the arithmetic is fully specified (the float16 path with its per-step rounding, the float32 path
with its sequential sums), to rounding for the float64 truth."""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests.corr_torch_ref import corr_torch, slam_corr_torch


def _case(seed, M=60, C=24, H2=20, W2=28, N2=3, Ng=10, far=True):
    rng = np.random.default_rng(seed)
    fmap2 = (rng.standard_normal((N2, C, H2, W2)) / 4).astype(np.float16)
    gmap = (rng.standard_normal((Ng, C, 3, 3)) / 4).astype(np.float16)
    cx, cy = rng.uniform(-5, W2 + 5, M), rng.uniform(-5, H2 + 5, M)
    if far:
        cx[:4] = [-40.0, 3 * W2, 0.0, W2 - 1.0]       # far outside, exactly on the first / last column
        cy[:4] = [5.0, -3.0 * H2, 0.0, H2 - 1.0]
    sc = rng.uniform(0.2, 3.0, M)
    off = np.arange(3.0) - 1
    coords = np.empty((M, 2, 3, 3), np.float32)
    coords[:, 0] = cx[:, None, None] + sc[:, None, None] * off[None, None, :]
    coords[:, 1] = cy[:, None, None] + sc[:, None, None] * off[None, :, None]
    coords[5] = np.round(coords[5])                    # integer coordinates: dx = dy = 0
    us = rng.integers(0, Ng, M).astype(np.int64)
    vs = rng.integers(0, N2, M).astype(np.int64)
    return gmap, fmap2, coords, us, vs


@pytest.mark.parametrize("radius", [1, 3])
def test_half_path_bit_exact(radius):
    """mode "ref": float16 products and partial sums in channel order, float16 four-slice blend"""
    gmap, fmap2, coords, us, vs = _case(1)
    a = O.corr(gmap, fmap2, coords, us, vs, radius, "ref")
    b = corr_torch(gmap, fmap2, coords, us, vs, radius, "ref").numpy()
    assert a.dtype == b.dtype == np.float16 and a.shape == b.shape == (len(us), 2 * radius + 1, 2 * radius + 1, 3, 3)
    assert np.array_equal(a.view(np.uint16), b.view(np.uint16))
    assert np.abs(a.astype(np.float32)).max() > 0.5      # not a comparison of zeros


@pytest.mark.parametrize("radius", [1, 3])
def test_f32_path_bit_exact(radius):
    gmap, fmap2, coords, us, vs = _case(2)
    g32, f32 = gmap.astype(np.float32), fmap2.astype(np.float32)
    a = O.corr(g32, f32, coords, us, vs, radius, "f32")
    b = corr_torch(g32, f32, coords, us, vs, radius, "f32").numpy()
    assert a.dtype == b.dtype == np.float32
    assert np.array_equal(a, b)


def test_truth_mode_to_rounding():
    gmap, fmap2, coords, us, vs = _case(3)
    for g, f in ((gmap, fmap2), (gmap.astype(np.float32), fmap2.astype(np.float32))):
        a = O.corr(g, f, coords, us, vs, 3, "truth")
        b = corr_torch(g, f, coords, us, vs, 3, "truth").numpy()
        assert a.dtype == b.dtype == np.float64
        assert np.abs(a - b).max() <= 1e-13 * max(1.0, np.abs(a).max())


def test_out_of_bounds_rule_and_layout():
    """hand-made map: the value at (row, col) of channel 0 encodes its position, the tile is a one-hot on channel 0, so
    the raw correlation IS the sampled pixel and the output layout / out-of-bounds rule can be read off directly"""
    H2, W2, C = 6, 7, 8
    fmap2 = np.zeros((1, C, H2, W2), np.float32)
    fmap2[0, 0] = 10.0 * np.arange(H2)[:, None] + np.arange(W2)[None, :] + 1.0      # > 0 everywhere inside
    gmap = np.zeros((1, C, 3, 3), np.float32)
    gmap[0, 0] = 1.0
    coords = np.zeros((1, 2, 3, 3), np.float32)
    coords[0, 0], coords[0, 1] = 2.0, 3.0          # x = 2, y = 3 for every patch pixel, integer: no blend
    for fn in (lambda: O.corr(gmap, fmap2, coords, np.zeros(1, np.int64), np.zeros(1, np.int64), 3, "f32"),
               lambda: corr_torch(gmap, fmap2, coords, [0], [0], 3, "f32").numpy()):
        out = fn()[0]                               # [x offset][y offset][3][3]
        for ox in range(7):
            for oy in range(7):
                col, row = 2 + ox - 3, 3 + oy - 3
                want = fmap2[0, 0, row, col] if (0 <= row < H2 and 0 <= col < W2) else 0.0
                assert out[ox, oy, 1, 1] == want, (ox, oy)


def test_two_level_stack_on_a_synthetic_state():
    """SLAM.corr on the tiny seeded state: both statements, feature order (x off, y off, i0, j0, level)"""
    from cdv_slam_amd import synth
    st = synth.make_state("tiny")
    coords = np.ascontiguousarray(O.transform(st.poses, st.patches, st.intrinsics, st.ii, st.jj, st.kk)
                                  .transpose(0, 3, 1, 2))
    a = O.slam_corr(st.gmap, st.fmap1, st.fmap2, coords, st.ii1, st.jj1, 3, "ref")
    b = slam_corr_torch(st.gmap, st.fmap1, st.fmap2, coords, st.ii1, st.jj1, 3, "ref").numpy()
    assert a.shape == b.shape == (st.E, 882)
    assert np.array_equal(a.view(np.uint16), b.view(np.uint16))
    t = O.slam_corr(st.gmap, st.fmap1, st.fmap2, coords, st.ii1, st.jj1, 3, "truth")
    tb = slam_corr_torch(st.gmap, st.fmap1, st.fmap2, coords, st.ii1, st.jj1, 3, "truth").numpy()
    assert np.abs(t - tb).max() <= 1e-13 * max(1.0, np.abs(t).max())
    # the half path stays inside the envelope the GPU tests allow against the truth
    tol = 2.0 ** -8 * np.abs(t).max() + 2.0 ** -10
    assert np.abs(a.astype(np.float64) - t).max() <= 4 * tol
