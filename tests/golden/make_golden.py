#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by EXECUTING the reference's own Python files.

Run in the build container only (needs /root/reference, which never travels to the GPU box):

    python tests/golden/make_golden.py

What runs verbatim from the reference (imported from /root/reference, not copied):
    cdvslam/projective_ops.py, cdvslam/ba.py, cdvslam/lietorch/{groups,group_ops,broadcasting}.py
What is injected so those files import on this CPU-only image (ordinary ModuleNotFoundError
otherwise -- no permission was denied):
    torch_scatter      -> scatter_sum via index_add_ (used at ba.py:42,46,51,56)
    cuda_ba, cuda_corr -> empty placeholders (imported by cdvslam/fastba, cdvslam/altcorr at import time only)
    lietorch_backends  -> forward group ops backed by THIS repo's CPU oracle (oracle/lie_impl.h)
So the fixtures pin the reference's Python-level algorithm (pops.transform, jacobians, ba.BA
control flow, gates, scatter layout, broadcasting) -- the Lie arithmetic inside is our restatement,
which is pinned separately by the reference's property tests (tests/test_oracle_lie.py).

Fixtures are DATA ONLY: seeded inputs + the outputs the reference code produced.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"

from oracle import oracle as O  # noqa: E402
from cdv_slam_amd import synth  # noqa: E402


def _install_shims():
    ts = types.ModuleType("torch_scatter")

    def scatter_sum(src, index, dim=-1, out=None, dim_size=None):
        dim = dim % src.dim()
        if dim_size is None:
            dim_size = int(index.max()) + 1 if index.numel() else 0
        shape = list(src.shape)
        shape[dim] = dim_size
        res = torch.zeros(shape, dtype=src.dtype, device=src.device)
        return res.index_add_(dim, index.to(src.device).long(), src)

    ts.scatter_sum = scatter_sum
    ts.scatter_mean = None
    sys.modules["torch_scatter"] = ts
    sys.modules["cuda_ba"] = types.ModuleType("cuda_ba")
    sys.modules["cuda_ba"].neighbors = None
    sys.modules["cuda_ba"].reproject = None
    sys.modules["cuda_corr"] = types.ModuleType("cuda_corr")

    lb = types.ModuleType("lietorch_backends")

    def _np(t):
        return t.detach().cpu().numpy()

    def _mk(op, nin):
        def f(group_id, *inputs):
            dt = np.float64 if inputs[0].dtype == torch.float64 else np.float32
            arrs = [_np(x).astype(dt) for x in inputs[:nin]]
            out = O.lie(group_id, op, *arrs, dtype=dt)
            return torch.from_numpy(out).to(inputs[0].dtype)
        return f

    for name, op, nin in (("expm", "exp", 1), ("logm", "log", 1), ("inv", "inv", 1), ("mul", "mul", 2),
                          ("adj", "adj", 2), ("adjT", "adjT", 2), ("act", "act", 2), ("act4", "act4", 2),
                          ("as_matrix", "matrix", 1)):
        setattr(lb, name, _mk(op, nin))
    for name in ("expm_backward", "logm_backward", "inv_backward", "mul_backward", "adj_backward",
                 "adjT_backward", "act_backward", "act4_backward", "projector", "Jinv"):
        setattr(lb, name, None)
    sys.modules["lietorch_backends"] = lb


def main():
    _install_shims()
    sys.path.insert(0, REF)
    from cdvslam import projective_ops as pops
    from cdvslam.lietorch import SE3
    from cdvslam import ba as refba

    out = {}

    # ---- pops.transform / jacobian / valid / tonly / flow_mag / point_cloud (tiny graph) ----
    for dt, tdt in (("f32", torch.float32), ("f64", torch.float64)):
        st = synth.make_state("tiny", features=False)
        poses = torch.from_numpy(st.poses).to(tdt)[None]
        patches = torch.from_numpy(st.patches).to(tdt)[None]
        intr = torch.from_numpy(st.intrinsics).to(tdt)[None]
        # per-frame intrinsics differ slightly so that the ii / jj intrinsics roles are pinned
        intr = intr * (1 + 0.01 * torch.arange(intr.shape[1], dtype=tdt)[None, :, None])
        ii, jj, kk = (torch.from_numpy(x) for x in (st.ii, st.jj, st.kk))
        with torch.no_grad():
            x1 = pops.transform(SE3(poses), patches, intr, ii, jj, kk)
            x1j, v, (Ji, Jj, Jz) = pops.transform(SE3(poses), patches, intr, ii, jj, kk, jacobian=True)
            x1v, val = pops.transform(SE3(poses), patches, intr, ii, jj, kk, valid=True)
            x1t = pops.transform(SE3(poses), patches, intr, ii, jj, kk, tonly=True)
            fm, fv = pops.flow_mag(SE3(poses), patches, intr, ii, jj, kk, beta=0.5)
            m = st.n * st.cfg.M
            ix = torch.arange(m) // st.cfg.M
            pc = pops.point_cloud(SE3(poses), patches[:, :m], intr, ix)
        np.savez_compressed(
            os.path.join(HERE, "pops_transform_%s.npz" % dt),
            poses=poses[0].numpy(), patches=patches[0].numpy(), intrinsics=intr[0].numpy(),
            ii=st.ii, jj=st.jj, kk=st.kk, coords=x1[0].numpy(), coords_jac=x1j[0].numpy(), valid=v[0].numpy(),
            Ji=Ji[0].numpy(), Jj=Jj[0].numpy(), Jz=Jz[0].numpy(), coords_valid=x1v[0].numpy(),
            validpx=val[0].numpy(), coords_tonly=x1t[0].numpy(), flow_mag=fm[0].numpy(),
            flow_valid=fv[0].numpy(), point_cloud=pc[0].numpy(), point_cloud_ix=ix.numpy())
        out["pops_" + dt] = x1.shape

    # ---- reference ba.BA on the tiny graph (fully connected variant + sliding-window variant) ----
    for tag, kw in (("fc", dict(frames=5, M=6, fully_connected=True)), ("win", dict())):
        st = synth.make_state("tiny", features=False, **kw)
        tdt = torch.float32
        poses = torch.from_numpy(st.poses).to(tdt)[None]
        patches = torch.from_numpy(st.patches).to(tdt)[None]
        intr = torch.from_numpy(st.intrinsics).to(tdt)[None]
        ii, jj, kk = (torch.from_numpy(x) for x in (st.ii, st.jj, st.kk))
        target = torch.from_numpy(st.target)[None]
        weight = torch.from_numpy(st.weight)[None]
        h, w = st.cfg.ht // st.cfg.res, st.cfg.wd // st.cfg.res
        bounds = [-64, -64, w + 64, h + 64]
        res = {}
        for ep in (1.0, 100.0):
            with torch.no_grad():
                P2, X2 = refba.BA(SE3(poses.clone()), patches.clone(), intr, target, weight,
                                  torch.as_tensor([1e-4]), ii, jj, kk, bounds, ep=ep, fixedp=1)
                # second iteration feeds the first one's output, as the training loop does
                P3, X3 = refba.BA(P2, X2, intr, target, weight, torch.as_tensor([1e-4]), ii, jj, kk, bounds,
                                  ep=ep, fixedp=1)
            res["poses_ep%g" % ep] = P2.data[0].numpy()
            res["patches_ep%g" % ep] = X2[0].numpy()
            res["poses2_ep%g" % ep] = P3.data[0].numpy()
            res["patches2_ep%g" % ep] = X3[0].numpy()
        with torch.no_grad():
            Ps, Xs = refba.BA(SE3(poses.clone()), patches.clone(), intr, target, weight, torch.as_tensor([1e-4]),
                              ii, jj, kk, bounds, ep=1.0, fixedp=1, structure_only=True)
        res["patches_structure_only"] = Xs[0].numpy()
        np.savez_compressed(os.path.join(HERE, "ba_py_%s.npz" % tag), poses=st.poses, patches=st.patches,
                            intrinsics=st.intrinsics, target=st.target, weight=st.weight, ii=st.ii, jj=st.jj,
                            kk=st.kk, bounds=np.array(bounds, np.float32), **res)
        out["ba_" + tag] = len(st.ii)

    # ---- lietorch python layer: broadcasting + op wiring (groups.py) ----
    g = torch.Generator().manual_seed(1234)
    a = 0.3 * torch.randn(3, 4, 6, generator=g, dtype=torch.float64)
    b = 0.3 * torch.randn(3, 1, 6, generator=g, dtype=torch.float64)
    p4 = torch.randn(3, 4, 5, 4, generator=g, dtype=torch.float64)
    X, Y = SE3.exp(a), SE3.exp(b)
    np.savez_compressed(
        os.path.join(HERE, "lietorch_py.npz"), a=a.numpy(), b=b.numpy(), p4=p4.numpy(),
        X=X.data.numpy(), Y=Y.data.numpy(), XY=(X * Y).data.numpy(), Xinv=X.inv().data.numpy(),
        logX=X.log().numpy(), act4=(X[:, :, None] * p4).numpy(), matrix=X.matrix().numpy(),
        adjT=X.adjT(a).numpy(), adj=X.adj(a).numpy(), retr=X.retr(a).data.numpy())
    out["lietorch_py"] = tuple(X.data.shape)
    print("golden fixtures written:", out)


def _patch_inputs(st):
    """patches of the synthetic states are a 3x3 coordinate grid with one inverse depth per patch (synth.make_state):
    the fixtures keep the three x values, the three y values and d per patch; tests/golden_util.py rebuilds the planes"""
    xy = np.stack([st.patches[:, 0, 0, :], st.patches[:, 1, :, 0]], 1).copy()      # [P, 2, 3]
    d = st.patches[:, 2, 1, 1].copy()
    rebuilt = np.empty_like(st.patches)
    rebuilt[:, 0] = xy[:, 0, None, :]
    rebuilt[:, 1] = xy[:, 1, :, None]
    rebuilt[:, 2] = d[:, None, None]
    assert np.array_equal(rebuilt, st.patches)
    return xy, d


def bench_size():
    """Round 3: fixtures at benchmark size, produced by the reference's own Python files.
    (i)   cdvslam/ba.py:86-185 on BASELINE configs[0] (synth `pr1`: 10 frames x 96 patches, fully connected,
          E = 9,600), ep = 1.0, two successive calls                                   -> ba_py_pr1.npz
    (ii)  cdvslam/altcorr/correlation.py:51-71 `patchify` (bilinear / upperleft / raw; r = 0, 1, 3; f16 and f32
          maps) with cuda_corr.patchify_forward backed by the oracle's gather           -> patchify_py.npz
    (iii) cdvslam/projective_ops.py:53-130 on every 6th edge of the `small` graph with per-frame intrinsics
                                                                                        -> pops_small_f32.npz"""
    _install_shims()
    sys.path.insert(0, REF)

    def patchify_forward(net, coords, radius):
        out = np.stack([O.patchify_raw(net[b].numpy(), coords[b].numpy(), radius) for b in range(net.shape[0])])
        return [torch.from_numpy(out)]

    sys.modules["cuda_corr"].patchify_forward = patchify_forward
    from cdvslam import projective_ops as pops
    from cdvslam.lietorch import SE3
    from cdvslam import ba as refba
    from cdvslam.altcorr import correlation as refcorr

    out = {}
    # ---- (i) ba.py on configs[0] --------------------------------------------------------------------
    st = synth.make_state("pr1", features=False)
    tdt = torch.float32
    poses = torch.from_numpy(st.poses).to(tdt)[None]
    patches = torch.from_numpy(st.patches).to(tdt)[None]
    intr = torch.from_numpy(st.intrinsics).to(tdt)[None]
    ii, jj, kk = (torch.from_numpy(x) for x in (st.ii, st.jj, st.kk))
    target = torch.from_numpy(st.target)[None]
    weight = torch.from_numpy(st.weight)[None]
    h, w = st.cfg.ht // st.cfg.res, st.cfg.wd // st.cfg.res
    bounds = [-64, -64, w + 64, h + 64]
    with torch.no_grad():
        P2, X2 = refba.BA(SE3(poses.clone()), patches.clone(), intr, target, weight, torch.as_tensor([1e-4]), ii, jj,
                          kk, bounds, ep=1.0, fixedp=1)
        P3, X3 = refba.BA(P2, X2, intr, target, weight, torch.as_tensor([1e-4]), ii, jj, kk, bounds, ep=1.0, fixedp=1)
        _, Xs = refba.BA(SE3(poses.clone()), patches.clone(), intr, target, weight, torch.as_tensor([1e-4]), ii, jj,
                         kk, bounds, ep=1.0, fixedp=1, structure_only=True)
    c, d = _patch_inputs(st)
    P = st.n * st.cfg.M
    assert int(st.kk.max()) < P
    for X in (X2, X3, Xs):                       # ba.py writes one inverse depth to all nine pixels (ba.py:176-180)
        assert torch.equal(X[0, :P, 2], X[0, :P, 2, :1, :1].expand(-1, 3, 3))
        assert torch.equal(X[0, :, :2], patches[0, :, :2])
    np.savez_compressed(
        os.path.join(HERE, "ba_py_pr1.npz"), poses=st.poses, patch_xy=c, patch_d=d, intrinsics=st.intrinsics,
        target=st.target, weight=st.weight, frames=np.int64(st.n), M=np.int64(st.cfg.M),
        bounds=np.array(bounds, np.float32), poses1=P2.data[0].numpy(), d1=X2[0, :, 2, 0, 0].numpy(),
        poses2=P3.data[0].numpy(), d2=X3[0, :, 2, 0, 0].numpy(), d_structure_only=Xs[0, :, 2, 0, 0].numpy())
    out["ba_py_pr1"] = len(st.ii)

    # ---- (ii) altcorr.patchify, Python layer ---------------------------------------------------------
    g = torch.Generator().manual_seed(1234)
    B, C, H, W, M = 2, 8, 24, 32, 24
    net16 = (torch.randn((B, C, H, W), generator=g) / 4).half()
    net32 = torch.randn((B, 3, H, W), generator=g)
    coords = torch.stack([torch.rand((B, M), generator=g) * (W + 6) - 3, torch.rand((B, M), generator=g) * (H + 6) - 3], -1)
    coords[0, 0] = torch.tensor([0.0, 0.0])                 # integer coordinates, the four map corners, far outside
    coords[0, 1] = torch.tensor([W - 1.0, H - 1.0])
    coords[0, 2] = torch.tensor([-0.25, H - 0.5])
    coords[0, 3] = torch.tensor([W + 20.0, -20.0])
    res = dict(net16=net16.numpy(), net32=net32.numpy(), coords=coords.numpy())
    with torch.no_grad():
        for tag, net in (("f16", net16), ("f32", net32)):
            for r in (0, 1, 3):
                for mode in ("bilinear", "upperleft", "raw"):
                    y = refcorr.patchify(net, coords, r, mode=mode)
                    res["%s_r%d_%s" % (tag, r, mode)] = y.numpy()
    np.savez_compressed(os.path.join(HERE, "patchify_py.npz"), **res)
    out["patchify_py"] = len(res)

    # ---- (iii) projective_ops on the `small` graph, per-frame intrinsics ------------------------------
    st = synth.make_state("small", features=False)
    poses = torch.from_numpy(st.poses)[None]
    patches = torch.from_numpy(st.patches)[None]
    intr = torch.from_numpy(st.intrinsics)[None]
    intr = intr * (1 + 0.004 * torch.arange(intr.shape[1], dtype=torch.float32)[None, :, None])
    sel = np.arange(0, st.E, 6)
    ii, jj, kk = (torch.from_numpy(x[sel]) for x in (st.ii, st.jj, st.kk))
    with torch.no_grad():
        x1j, v, (Ji, Jj, Jz) = pops.transform(SE3(poses), patches, intr, ii, jj, kk, jacobian=True)
        x1 = pops.transform(SE3(poses), patches, intr, ii, jj, kk)
        x1v, val = pops.transform(SE3(poses), patches, intr, ii, jj, kk, valid=True)
        fm, fv = pops.flow_mag(SE3(poses), patches, intr, ii, jj, kk, beta=0.5)
        m = st.n * st.cfg.M
        ix = torch.arange(m) // st.cfg.M
        pc = pops.point_cloud(SE3(poses), patches[:, :m], intr, ix)
    assert torch.equal(x1, x1v)
    c, d = _patch_inputs(st)
    np.savez_compressed(
        os.path.join(HERE, "pops_small_f32.npz"), poses=st.poses, patch_xy=c, patch_d=d, intrinsics=intr[0].numpy(),
        ii=ii.numpy(), jj=jj.numpy(), kk=kk.numpy(), coords=x1[0].numpy(), coords_jac_centre=x1j[0].numpy()[:, 1, 1],
        valid=v[0].numpy(), Ji=Ji[0].numpy(), Jj=Jj[0].numpy(), Jz=Jz[0].numpy(), validpx=val[0].numpy(),
        flow_mag=fm[0].numpy(), flow_valid=fv[0].numpy(), point_cloud_centre=pc[0].numpy()[:, 1, 1])
    out["pops_small"] = len(sel)
    for f in ("ba_py_pr1.npz", "patchify_py.npz", "pops_small_f32.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")
    print("benchmark-size fixtures written:", out)


def corr_pin():
    """Round 4: the correlation pinned as far as anything of the reference that RUNS here allows.
    corr_forward_kernel + corr_cuda_forward (altcorr/correlation_kernel.cu:82-136,193-233) are CUDA; the one piece of the
    reference that executes in this container and shares their sampling convention is `altcorr.patchify(net, coords, r,
    'bilinear')` (altcorr/correlation.py:51-71): floor(coords) - r .. + r + 1 integer samples, out-of-range samples zero,
    then the four-slice blend with the SAME weights and the same roles of dx / dy as correlation_kernel.cu:213-230.  The
    correlation is linear in the feature map, so

        corr[e, y_off, x_off, i0, j0] = sum_c gmap[ii[e], c, i0, j0] * patchify(fmap[jj[e]], coords[e, :, i0, j0] / s, 3)[c, y_off, x_off]

    with patchify's blend computed BY THE REFERENCE'S PYTHON in float64, contracted here, and permuted to [x_off][y_off] as
    correlation_kernel.cu:232 returns it.  cuda_corr.patchify_forward (a CUDA kernel, correlation_kernel.cu:16-47) is
    stood in for by the torch gather below (integer samples, zero outside) -- that gather is this repository's reading,
    everything after it is the reference executing.  -> corr_pin.npz (inputs + float64 result, both pyramid levels)"""
    _install_shims()
    sys.path.insert(0, REF)

    def patchify_forward(net, coords, radius):
        B, C, H, W = net.shape
        M, D = coords.shape[1], 2 * radius + 2
        fx, fy = torch.floor(coords[..., 0]).long(), torch.floor(coords[..., 1]).long()
        out = torch.zeros((B, M, C, D, D), dtype=net.dtype)
        bidx = torch.arange(B)[:, None].expand(B, M)
        for a in range(D):
            for b in range(D):
                i, j = fy + (a - radius), fx + (b - radius)
                ok = (i >= 0) & (i < H) & (j >= 0) & (j < W)
                v = net[bidx, :, i.clamp(0, H - 1), j.clamp(0, W - 1)]          # [B, M, C]
                out[:, :, :, a, b] = torch.where(ok[..., None], v, torch.zeros_like(v))
        return [out]

    sys.modules["cuda_corr"].patchify_forward = patchify_forward
    from cdvslam.altcorr import correlation as refcorr

    rng = np.random.default_rng(20261004)
    mem, C, H, W, Ng, E = 4, 24, 24, 32, 40, 144
    fmap1 = (rng.standard_normal((mem, C, H, W)) / 4).astype(np.float16)
    # the level-1 map as slam.py:682 makes it: 4x4 average of the level-0 map, stored in half
    fmap2 = torch.nn.functional.avg_pool2d(torch.from_numpy(fmap1).float(), 4, 4).half().numpy()
    gmap = (rng.standard_normal((Ng, C, 3, 3)) / 4).astype(np.float16)
    cx, cy = rng.uniform(4, W - 4, E), rng.uniform(4, H - 4, E)
    sc = rng.uniform(0.5, 1.6, E)
    # borders, corners, half-outside, far outside, integer coordinates, a degenerate patch
    cx[:12] = [0.0, W - 1.0, -0.25, W - 0.5, 1.5, W - 2.5, -3.0, W + 2.75, -40.0, 3.0 * W, 10.0, 11.0]
    cy[:12] = [0.0, H - 1.0, H - 0.5, -0.25, -2.0, H + 1.25, 5.5, 7.25, 6.0, -3.0 * H, 12.0, 13.0]
    off = np.arange(3.0) - 1
    coords = np.empty((E, 2, 3, 3), np.float32)
    coords[:, 0] = cx[:, None, None] + sc[:, None, None] * off[None, None, :]
    coords[:, 1] = cy[:, None, None] + sc[:, None, None] * off[None, :, None]
    coords[10] = np.round(coords[10])                   # dx = dy = 0
    coords[11] = coords[11, :, 1:2, 1:2]                # all nine pixels on one point
    # a rotated / sheared patch: x depends on the row, y on the column (the roles of i0 / j0 must not be swapped)
    coords[12, 0] = 14.3 + 0.9 * off[None, None, :] + 0.4 * off[None, :, None]
    coords[12, 1] = 9.6 - 0.3 * off[None, None, :] + 1.1 * off[None, :, None]
    ii = rng.integers(0, Ng, E).astype(np.int64)
    jj = rng.integers(0, mem, E).astype(np.int64)

    res = {}
    with torch.no_grad():
        for lvl, (fm, s) in enumerate(((fmap1, 1.0), (fmap2, 4.0))):
            net = torch.from_numpy(fm).double()
            g = torch.from_numpy(gmap).double()
            out = torch.zeros((E, 7, 7, 3, 3), dtype=torch.float64)
            c = torch.from_numpy(coords) / s              # float32 division, as slam.py:321-322 does it
            for e in range(E):
                pts = c[e].reshape(2, 9).T[None].contiguous()          # [1, 9, (x, y)]: pixel p = i0 * 3 + j0
                samp = refcorr.patchify(net[jj[e]][None], pts, 3, mode="bilinear")[0]       # [9, C, 7 (y_off), 7 (x_off)]
                f1 = g[ii[e]].reshape(C, 9)                              # [C, p]
                blended = torch.einsum("cp,pcyx->yxp", f1, samp)        # [y_off, x_off, p]
                out[e] = blended.permute(1, 0, 2).reshape(7, 7, 3, 3)   # correlation_kernel.cu:232: (x_off, y_off, i0, j0)
            res["corr%d" % lvl] = out.numpy()
    np.savez_compressed(os.path.join(HERE, "corr_pin.npz"), fmap1=fmap1, fmap2=fmap2, gmap=gmap, coords=coords, ii=ii, jj=jj,
                        **res)
    print("corr_pin.npz", os.path.getsize(os.path.join(HERE, "corr_pin.npz")), "bytes;",
          "max |corr0| %.3f, max |corr1| %.3f" % (np.abs(res["corr0"]).max(), np.abs(res["corr1"]).max()))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "bench-size":
        bench_size()
    elif len(sys.argv) > 1 and sys.argv[1] == "corr-pin":
        corr_pin()
    else:
        main()
