"""TEST INFRASTRUCTURE: a second, independently written statement of altcorr.corr on torch-CPU tensors.

It shares no code with oracle/cdv_oracle.c::orc_corr: the window samples are fetched with one torch.gather over a
flattened map, the channel contraction is a vectorised loop over channels (so that the summation ORDER and the
per-step rounding of the reference kernel, correlation_kernel.cu:121-131, are reproduced by torch's own float16 /
float32 arithmetic), and the bilinear blend is the four-slice expression of correlation_kernel.cu:213-232 evaluated
by torch in the maps' dtype.  tests/test_corr_independent.py cross-checks the C oracle's three modes against it.
"""
import torch


def corr_torch(fmap1, fmap2, coords, us, vs, radius, mode):
    """fmap1 [N1,C,P,P], fmap2 [N2,C,H2,W2], coords [M,2,P,P] float32, us/vs [M] int64.
    mode "ref":   float16 maps, every product and every partial sum rounded to float16, float16 blend
         "f32":   float32 maps, sequential float32 sums, float32 blend
         "truth": float64 sums; blend weights are those the maps' dtype would hold (float16 weights for float16 maps)
    -> [M, 2r+1 (x), 2r+1 (y), P, P], the permuted layout cuda_corr.forward returns."""
    fmap1, fmap2 = torch.as_tensor(fmap1), torch.as_tensor(fmap2)
    coords = torch.as_tensor(coords, dtype=torch.float32)
    us, vs = torch.as_tensor(us, dtype=torch.int64), torch.as_tensor(vs, dtype=torch.int64)
    M, _, P, _ = coords.shape
    N2, C, H2, W2 = fmap2.shape
    D = 2 * radius + 2
    x, y = coords[:, 0], coords[:, 1]                                   # [M,P,P]
    off = torch.arange(D) - radius
    rows = y.floor().long()[:, None, None] + off[None, :, None, None, None]      # [M,D,1,P,P]
    cols = x.floor().long()[:, None, None] + off[None, None, :, None, None]      # [M,1,D,P,P]
    inside = (rows >= 0) & (rows < H2) & (cols >= 0) & (cols < W2)               # [M,D,D,P,P]
    flat = (rows.clamp(0, H2 - 1) * W2 + cols.clamp(0, W2 - 1)).expand(M, D, D, P, P).reshape(M, 1, -1)
    work = {"ref": torch.float16, "f32": torch.float32, "truth": torch.float64}[mode]
    acc = torch.zeros((M, D, D, P, P), dtype=work)
    tiles = fmap1[us]                                                    # [M,C,P,P]
    maps = fmap2.reshape(N2, C, H2 * W2)
    for c in range(C):                                                   # channel order of the reference's loop
        samples = torch.gather(maps[vs, c][:, None], 2, flat).reshape(M, D, D, P, P)
        a, b = tiles[:, c][:, None, None].to(work), samples.to(work)
        acc = acc + a * b            # float16: the product and the sum are each rounded (c10::Half operators)
    acc = torch.where(inside, acc, torch.zeros((), dtype=work))
    wdt = fmap1.dtype if mode != "f32" else torch.float32
    dx = (x - x.floor()).to(wdt)[:, None, None].to(work if mode == "truth" else wdt)
    dy = (y - y.floor()).to(wdt)[:, None, None].to(work if mode == "truth" else wdt)
    d = D - 1
    out = (1 - dx) * (1 - dy) * acc[:, :d, :d]
    out = out + dx * (1 - dy) * acc[:, :d, 1:]
    out = out + (1 - dx) * dy * acc[:, 1:, :d]
    out = out + dx * dy * acc[:, 1:, 1:]
    return out.permute(0, 2, 1, 3, 4).contiguous()


def slam_corr_torch(gmap, fmap1, fmap2, coords, ii1, jj1, radius=3, mode="ref"):
    """SLAM.corr (slam.py:316-323): levels at coords / 1 and coords / 4, stacked last -> [E, 2 (2r+1)^2 P^2]"""
    coords = torch.as_tensor(coords, dtype=torch.float32)
    c1 = corr_torch(gmap, fmap1, coords / 1, ii1, jj1, radius, mode)
    c2 = corr_torch(gmap, fmap2, coords / 4, ii1, jj1, radius, mode)
    return torch.stack([c1, c2], -1).reshape(coords.shape[0], -1)
