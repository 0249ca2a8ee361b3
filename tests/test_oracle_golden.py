"""Pin the oracle's projective ops and ba.py restatement against golden vectors produced by the
reference's OWN Python files (tests/golden/make_golden.py; cdvslam/projective_ops.py:53-130,
cdvslam/ba.py:86-185 executed verbatim under a shimmed import)."""
import os

import numpy as np
import pytest

from oracle import ba_py
from oracle import oracle as O


@pytest.mark.parametrize("tag,dtype,tol", [("f32", np.float32, 2e-5), ("f64", np.float64, 1e-11)])
def test_pops_transform_golden(golden_dir, tag, dtype, tol):
    g = np.load(os.path.join(golden_dir, "pops_transform_%s.npz" % tag))
    args = (g["poses"], g["patches"], g["intrinsics"], g["ii"], g["jj"], g["kk"])
    coords = O.transform(*args, dtype=dtype)
    assert coords.dtype == g["coords"].dtype
    assert np.allclose(coords, g["coords"], rtol=0, atol=tol * 100)
    c2, v, (Ji, Jj, Jz) = O.transform(*args, jacobian=True, dtype=dtype)
    assert np.array_equal(v, g["valid"])
    for a, b in ((Ji, g["Ji"]), (Jj, g["Jj"]), (Jz, g["Jz"])):
        assert np.allclose(a, b, rtol=tol * 10, atol=tol * np.abs(b).max())
    c3, vpx = O.transform(*args, valid=True, dtype=dtype)
    assert np.array_equal(vpx, g["validpx"])
    ct = O.transform(*args, tonly=True, dtype=dtype)
    assert np.allclose(ct, g["coords_tonly"], rtol=0, atol=tol * 100)


def test_flow_mag_and_point_cloud_golden(golden_dir):
    """pops.flow_mag (projective_ops.py:120-130) and point_cloud (:115-117) composed from oracle ops."""
    g = np.load(os.path.join(golden_dir, "pops_transform_f64.npz"))
    dt = np.float64
    args = (g["poses"], g["patches"], g["intrinsics"])
    ii, jj, kk = g["ii"], g["jj"], g["kk"]
    c0 = O.transform(*args, ii, ii, kk, dtype=dt)
    c1, val = O.transform(*args, ii, jj, kk, valid=True, dtype=dt)
    c2 = O.transform(*args, ii, jj, kk, tonly=True, dtype=dt)
    flow = 0.5 * np.linalg.norm(c1 - c0, axis=-1) + 0.5 * np.linalg.norm(c2 - c0, axis=-1)
    assert np.allclose(flow, g["flow_mag"], atol=1e-9)
    assert np.array_equal(val > 0.5, g["flow_valid"])
    ix = g["point_cloud_ix"]
    Pinv = O.lie(O.SE3, "inv", g["poses"][ix], dtype=dt)
    patches = g["patches"][:len(ix)]
    K = g["intrinsics"][ix]
    X0 = np.stack([(patches[:, 0] - K[:, 2, None, None]) / K[:, 0, None, None],
                   (patches[:, 1] - K[:, 3, None, None]) / K[:, 1, None, None],
                   np.ones_like(patches[:, 2]), patches[:, 2]], -1)
    pc = O.lie(O.SE3, "act4", np.repeat(Pinv, 9, axis=0), X0.reshape(-1, 4), dtype=dt).reshape(X0.shape)
    assert np.allclose(pc, g["point_cloud"], atol=1e-10)


@pytest.mark.parametrize("tag", ["fc", "win"])
def test_ba_py_golden(golden_dir, tag):
    g = np.load(os.path.join(golden_dir, "ba_py_%s.npz" % tag))
    for ep in (1.0, 100.0):
        P2, X2, info = ba_py.BA(g["poses"], g["patches"], g["intrinsics"], g["target"], g["weight"], 1e-4,
                                g["ii"], g["jj"], g["kk"], g["bounds"], ep=ep, fixedp=1)
        assert info == 0
        # float32 Gauss-Newton: the ep=1.0 system is weakly damped, so f32 rounding of the solve
        # (reference: ATen cholesky, here: numpy) shows at the 1e-5 level on ~0.07 updates
        tol = 1e-4 if ep == 1.0 else 2e-5
        assert np.allclose(P2, g["poses_ep%g" % ep], atol=tol)
        assert np.allclose(X2, g["patches_ep%g" % ep], rtol=10 * tol, atol=tol)
        P3, X3, info = ba_py.BA(P2, X2, g["intrinsics"], g["target"], g["weight"], 1e-4,
                                g["ii"], g["jj"], g["kk"], g["bounds"], ep=ep, fixedp=1)
        assert np.allclose(P3, g["poses2_ep%g" % ep], atol=3 * tol)
        assert np.allclose(X3, g["patches2_ep%g" % ep], rtol=30 * tol, atol=3 * tol)
    _, Xs, _ = ba_py.BA(g["poses"], g["patches"], g["intrinsics"], g["target"], g["weight"], 1e-4,
                        g["ii"], g["jj"], g["kk"], g["bounds"], ep=1.0, fixedp=1, structure_only=True)
    assert np.allclose(Xs, g["patches_structure_only"], rtol=2e-4, atol=2e-5)


def test_fastba_matches_ba_py_where_gates_coincide():
    """fastba (ba_cuda.cu) == ba.py(ep=1.0) on states where their differing gates do not fire:
    residual < 128 px, all points in bounds, Z > 0.2, depth stays in (1e-3, 10), single intrinsics."""
    from cdv_slam_amd import synth
    st = synth.make_state("tiny", features=False, frames=5, M=6, fully_connected=True)
    h, w = st.cfg.ht // st.cfg.res, st.cfg.wd // st.cfg.res
    bounds = [-64, -64, w + 64, h + 64]
    dt = np.float64
    P1, X1, _ = ba_py.BA(st.poses, st.patches, st.intrinsics, st.target, st.weight, 1e-4, st.ii, st.jj, st.kk,
                         bounds, ep=1.0, fixedp=1, dtype=dt)
    P2, X2, info = O.fastba(st.poses, st.patches, st.intrinsics[0], st.target, st.weight, 1e-4, st.ii, st.jj,
                            st.kk, 1, st.n, iterations=1, dtype=dt)
    assert info == 0
    # ba.py renormalises every pose through lietorch; fastba leaves fixed poses untouched
    assert np.allclose(P1, P2, atol=1e-9)
    # depth clamps differ by design: ba.py clamps to [1e-3, 10] (ba.py:179), fastba to
    # d > 20 -> 1, max(d, 1e-4) (ba_cuda.cu:219-221); compare patches where neither fires
    ok = (X1[:, 2, 0, 0] > 1.001e-3) & (X1[:, 2, 0, 0] < 9.99)
    assert ok.sum() >= 0.7 * len(ok)
    assert np.allclose(X1[ok], X2[ok], atol=1e-9)


def test_ate_metric_umeyama():
    """metrics.ate_rmse (evaluate_tartan.py:63-70): a trajectory moved by a known Sim(3) aligns back exactly; noise on the
    camera centres comes out as its RMS; without scale correction a scaled copy does not align"""
    from cdv_slam_amd import metrics
    rng = np.random.default_rng(2)
    n = 50
    q = rng.normal(size=(n, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    poses = np.concatenate([rng.normal(size=(n, 3)), q], 1)
    c0 = metrics.camera_centres(poses)
    # centres of identity-rotation poses are -t
    ident = np.concatenate([rng.normal(size=(n, 3)), np.tile([0, 0, 0, 1.0], (n, 1))], 1)
    assert np.allclose(metrics.camera_centres(ident), -ident[:, :3])
    # a similarity applied to the camera centres: rebuild poses with identity rotations at the moved centres
    A = np.linalg.qr(rng.normal(size=(3, 3)))[0]
    A *= np.sign(np.linalg.det(A))
    moved = 2.5 * (A @ c0.T).T + np.array([1.0, -2.0, 0.5])
    ref = np.concatenate([-c0, np.tile([0, 0, 0, 1.0], (n, 1))], 1)
    est = np.concatenate([-moved, np.tile([0, 0, 0, 1.0], (n, 1))], 1)
    assert metrics.ate_rmse(ref, est) < 1e-12
    assert metrics.ate_rmse(ref, est, with_scale=False) > 0.1
    R, t, c = metrics.umeyama_alignment(c0.T, moved.T)
    assert np.allclose(R, A) and np.allclose(t, [1.0, -2.0, 0.5]) and abs(c - 2.5) < 1e-12
    noisy = np.concatenate([-(c0 + rng.normal(0, 1e-3, c0.shape)), np.tile([0, 0, 0, 1.0], (n, 1))], 1)
    e = metrics.ate_rmse(ref, noisy)
    assert 1.2e-3 < e < 2.2e-3     # sqrt(3) * 1e-3, minus what the alignment absorbs


# ---------------------------------------------------------------------------------------------------
# round 3: fixtures at benchmark size (tests/golden/make_golden.py bench-size)
# ---------------------------------------------------------------------------------------------------

def test_ba_py_reference_run_on_configs0_pins_both_oracles():
    """BASELINE configs[0] (10 frames x 96 patches fully connected, E = 9,600) as the reference's OWN cdvslam/ba.py:86-185
    computed it, two successive calls at ep = 1.0 (== fastba's damping, ba_cuda.cu:589):
      * oracle/ba_py.py (the restatement of ba.py) reproduces both calls and the structure-only branch;
      * the fastba restatement (oracle/fastba_impl.h, ba_cuda.cu:232-611) reproduces them too -- on this state none of the
        gates the two differ in fires (residual < 128 px, every centre in bounds, inverse depth inside (1e-3, 10)) --
        at the float32 level of the reference's arithmetic: the float32 instantiation to 1e-5, the float64 one within
        the weak-gauge bounds of tests/ba_checks.py BA_TOL['pr1'] (the reference itself is float32)."""
    from tests import golden_util, ba_checks
    g = golden_util.load_ba_pr1()
    args = (g["intrinsics"], g["target"], g["weight"], 1e-4, g["ii"], g["jj"], g["kk"], g["bounds"])
    P1, X1, info = ba_py.BA(g["poses"], g["patches"], *args, ep=1.0, fixedp=1)
    assert info == 0
    assert np.abs(P1 - g["poses1"]).max() < 1e-5 and np.abs(X1[:, 2, 0, 0] - g["d1"]).max() < 2e-5
    P2, X2, info = ba_py.BA(P1, X1, *args, ep=1.0, fixedp=1)
    assert np.abs(P2 - g["poses2"]).max() < 2e-5 and np.abs(X2[:, 2, 0, 0] - g["d2"]).max() < 5e-5
    _, Xs, _ = ba_py.BA(g["poses"], g["patches"], *args, ep=1.0, fixedp=1, structure_only=True)
    assert np.abs(Xs[:, 2, 0, 0] - g["d_structure_only"]).max() < 1e-6
    # the gates in which fastba and ba.py differ do not fire here
    for d in (g["d1"], g["d2"]):
        used = d[np.unique(g["kk"])]
        assert used.min() > 1.001e-3 and used.max() < 9.99
    c = O.fastba_reproject(g["poses"], g["patches"], g["intrinsics"][0], g["ii"], g["jj"], g["kk"], dtype=np.float64)[:, :, 1, 1]
    b = g["bounds"]
    assert (c[:, 0] > b[0]).all() and (c[:, 1] > b[1]).all() and (c[:, 0] < b[2]).all() and (c[:, 1] < b[3]).all()
    assert np.linalg.norm(g["target"] - c, axis=-1).max() < 128
    n = int(g["frames"])
    tol = ba_checks.BA_TOL["pr1"]
    for dt, (tt, tq, td) in ((np.float32, (1e-5, 1e-6, 5e-5)), (np.float64, (tol["t"], tol["q"], tol["d"]))):
        for it, pk, dk in ((1, "poses1", "d1"), (2, "poses2", "d2")):
            p, x, info = O.fastba(g["poses"], g["patches"], g["intrinsics"][0], g["target"], g["weight"], 1e-4, g["ii"],
                                  g["jj"], g["kk"], 1, n, it, dt)
            assert info == 0
            assert np.abs(p[:, :3] - g[pk][:, :3]).max() <= tt * it, (dt, it)
            assert np.abs(p[:, 3:] - g[pk][:, 3:]).max() <= tq, (dt, it)
            assert (np.abs(x[:, 2, 0, 0] - g[dk]) / np.maximum(np.abs(g[dk]), 1e-2)).max() <= td * it, (dt, it)


def test_patchify_python_layer_golden(golden_dir):
    """altcorr.patchify (cdvslam/altcorr/correlation.py:51-71) as the reference's own Python computed it on top of the
    gather: bilinear / upperleft / raw, r = 0, 1, 3, float16 and float32 maps, corner / outside coordinates."""
    g = np.load(os.path.join(golden_dir, "patchify_py.npz"))
    for tag in ("f16", "f32"):
        net = g["net16" if tag == "f16" else "net32"]
        for r in (0, 1, 3):
            for mode in ("bilinear", "upperleft", "raw"):
                want = g["%s_r%d_%s" % (tag, r, mode)]
                got = np.stack([O.patchify(net[b], g["coords"][b], r, mode) for b in range(net.shape[0])])
                assert got.dtype == want.dtype and got.shape == want.shape, (tag, r, mode)
                if mode == "bilinear":
                    assert np.allclose(got, want, rtol=0, atol=1e-6), (tag, r, mode)
                else:
                    assert np.array_equal(got, want), (tag, r, mode)


def test_pops_small_golden():
    """projective_ops.py:53-130 on every 6th edge of the `small` graph, per-frame intrinsics (ii's for iproj, jj's for
    proj): coords, validity, Jacobians, flow_mag, point_cloud as the reference's own file computed them."""
    from tests import golden_util
    g = golden_util.load_pops_small()
    args = (g["poses"], g["patches"], g["intrinsics"], g["ii"], g["jj"], g["kk"])
    coords = O.transform(*args, dtype=np.float32)
    assert np.abs(coords - g["coords"]).max() < 1e-3
    c2, v, (Ji, Jj, Jz) = O.transform(*args, jacobian=True, dtype=np.float32)
    assert np.array_equal(v, g["valid"])
    assert np.abs(c2[:, 1, 1] - g["coords_jac_centre"]).max() < 1e-3
    for a, b in ((Ji, g["Ji"]), (Jj, g["Jj"]), (Jz, g["Jz"])):
        assert np.allclose(a, b, rtol=2e-4, atol=2e-5 * np.abs(b).max())
    c3, vpx = O.transform(*args, valid=True, dtype=np.float32)
    assert np.array_equal(vpx, g["validpx"])


def test_corr_oracle_vs_reference_run_pin(golden_dir):
    """Rows a1 / a2: the correlation oracle against `corr_pin.npz` -- the reference's own `altcorr.patchify(..., 'bilinear')`
    (cdvslam/altcorr/correlation.py:51-71, executed by tests/golden/make_golden.py corr-pin) contracted with the patch
    features: its blend weights, its roles of dx / dy, its zero for out-of-range integer samples, the [x][y] order of
    correlation_kernel.cu:232 -- on borders, corners, far-outside, integer and sheared patches, both pyramid levels.
    Still unpinned by anything that runs: the half-precision accumulate order of correlation_kernel.cu:121-131 and the
    cast of dx / dy to half (:218-219); mode "ref" restates those and is only held inside the half envelope here."""
    z = np.load(os.path.join(golden_dir, "corr_pin.npz"))
    g32 = z["gmap"].astype(np.float32)
    for lvl, (fm, s) in enumerate(((z["fmap1"], 1.0), (z["fmap2"], 4.0))):
        c = (z["coords"] / np.float32(s)).astype(np.float32)
        want = z["corr%d" % lvl]
        top = np.abs(want).max()
        assert top > 0.2
        # float64 accumulation of the same products (the half maps are exact in float32); the reference's weights are
        # float32 products (1 - dy) * (1 - dx), hence 1e-7 and not 1e-15
        truth = O.corr(g32, fm.astype(np.float32), c, z["ii"], z["jj"], 3, "truth")
        assert truth.shape == want.shape == (len(z["ii"]), 7, 7, 3, 3)
        assert np.abs(truth - want).max() <= 2e-7 * top
        # the [x][y] order and the i0 / j0 roles are really pinned: the transposed readings are far off
        assert np.abs(truth.transpose(0, 2, 1, 3, 4) - want).max() > 0.1 * top
        assert np.abs(truth.transpose(0, 1, 2, 4, 3) - want).max() > 0.1 * top
        f32 = O.corr(g32, fm.astype(np.float32), c, z["ii"], z["jj"], 3, "f32")
        assert np.abs(f32 - want).max() <= 1e-5 * top
        half = O.corr(z["gmap"], fm, c, z["ii"], z["jj"], 3, "ref").astype(np.float64)
        assert np.abs(half - want).max() <= 2.0 ** -8 * top + 2.0 ** -10
        # far outside the map: exact zeros in every reading
        for a in (truth, f32, half):
            assert not a[8].any() and not a[9].any() and not want[8].any()
    both = O.slam_corr(z["gmap"].astype(np.float32), z["fmap1"].astype(np.float32), z["fmap2"].astype(np.float32), z["coords"],
                       z["ii"], z["jj"], 3, "truth")
    want = np.stack([z["corr0"], z["corr1"]], -1).reshape(len(z["ii"]), -1)      # slam.py:323
    assert np.abs(both - want).max() <= 2e-7 * np.abs(want).max()


@pytest.mark.parametrize("window", [None, 10])
def test_block_sparse_E_of_the_reference_equals_its_dense_branch(window):
    """Row a12: cuda_ba.forward has two branches for the Schur solve -- dense E (ba_cuda.cu:583-592) and, with eff_impl, the
    block-sparse `EfficentE` (block_e.cu:38-300; ba_cuda.cu:380-383, 567-580).  The product computes the dense quantities and
    does not reproduce EfficentE's index structures; here the two branches are restated side by side (oracle/fastba_impl.h,
    oracle/block_e_py.py) and give the same damped S, y, dX, dZ on a graph with 25 free poses (the global BA's shape) and on a
    windowed one (poses below t0 fixed: their rows are dropped at product time, block_e.cu:171-184) -- so the float64 oracle
    the HIP global path is held against IS the eff_impl branch's result too.  A wrong reading of the two block roles
    (ba_cuda.cu:382-383: the self block takes -w Jz Ji, the (i, j) block +w Jz Jj) is far off."""
    from cdv_slam_amd import synth
    from oracle.block_e_py import EfficentE, solve_eff_impl
    kw = dict(frames=26, opt_window=10 ** 6, removal_window=10 ** 6) if window is None else dict(frames=26, opt_window=window)
    st = synth.make_state("small", features=False, **kw)
    N = st.n - st.t0
    assert N == (25 if window is None else window)
    _, _, info, o = O.fastba(st.poses, st.patches, st.intrinsics[0], st.target, st.weight, st.lmbda, st.ii, st.jj, st.kk,
                             st.t0, st.n, 1, np.float64, debug=True)
    assert info == 0
    r, w, Jz, Ji, Jj = O.fastba_edges(st.poses, st.patches, st.intrinsics[0], st.target, st.weight, st.ii, st.jj, st.kk)
    assert (st.kk // st.cfg.M == st.ii).all()                       # EfficentE's assumption: a patch's frame is its source frame
    be = EfficentE(st.ii, st.jj, o["kx"], st.cfg.M, st.t0)
    # ij_xself as torch::_unique(cat(ii n + jj, ii n + ii)) gives it (block_e.cu:45-50): bit-exact integer bookkeeping
    nf = int(max(st.ii.max(), st.jj.max())) + 1
    uq = np.unique(np.concatenate([st.ii * nf + st.jj, st.ii * nf + st.ii]))
    assert np.array_equal(uq[be.ij_xself[0]], st.ii * nf + st.jj) and np.array_equal(uq[be.ij_xself[1]], st.ii * nf + st.ii)
    deg = [len(np.unique(np.concatenate([st.jj[st.ii == i], [i]]))) for i in np.unique(st.ii)]
    assert len(be.index_tensor) == sum(d * d for d in deg)          # block_e.cu:100-103
    be.fill(st.kk, w, Jz, Ji, Jj)
    S, y, dX, dZ = solve_eff_impl(o["B"], o["v"], o["C"], o["u"], st.lmbda, be, N)
    for key, got, want in (("S", S, o["S"]), ("y", y, o["y"]), ("dX", dX, o["dX"].reshape(-1)), ("dZ", dZ, o["dZ"])):
        assert np.abs(got - want).max() <= 1e-12 * np.abs(want).max(), key
    # the products one by one against the dense E
    Q = 1.0 / (o["C"] + st.lmbda)
    assert np.abs(be.computeEQEt(N, Q) - (o["E"] * Q) @ o["E"].T).max() <= 1e-12 * np.abs(o["B"]).max()
    assert np.abs(be.computeEv(N, Q * o["u"]) - o["E"] @ (Q * o["u"])).max() <= 1e-12 * np.abs(o["v"]).max()
    assert np.abs(be.computeEtv(len(Q), dX) - o["E"].T @ dX).max() <= 1e-12 * np.abs(o["u"]).max()
    # negative control: the two block roles swapped
    be.ij_xself = be.ij_xself[::-1].copy()
    be.fill(st.kk, w, Jz, Ji, Jj)
    assert np.abs(be.computeEQEt(N, Q) - (o["E"] * Q) @ o["E"].T).max() > 1e-2 * np.abs((o["E"] * Q) @ o["E"].T).max()
