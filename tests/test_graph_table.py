"""The patch-graph index as a TABLE (cdv_graph_build_table / cdv_update_prologue_table: two launches, no scan) against its
checkers: numpy (per-patch lists in (jj, edge id) order == std::stable_sort by jj, ba.cpp:84-86), the CPU oracle
(fastba.neighbors, ba.cpp:59-97) and the ranked index build (cdv_graph_build*) -- all integer work, bit-exact."""
import numpy as np
import pytest
import torch

from cdv_slam_amd import ops, synth
from oracle import oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def T(a):
    return torch.as_tensor(np.ascontiguousarray(a), device=DEV)


def table_lists(g, E):
    """{patch id: [(edge, ii, jj), ...] in table order} from the workspace (first 32 records from the table, longer lists
    from the overflow segment)"""
    deg, plo, recs, ovf, order, stream, kid = (a.cpu().numpy() for a in g.table_arrays())
    out = {}
    assert ((deg > 0) == (kid >= 0)).all()
    for s in np.nonzero(deg)[0]:
        d = int(deg[s])
        if d <= 32:
            rows = [recs[((s >> 4) * 32 + t) * 16 + (s & 15)] for t in range(d)]
        else:
            rows = [ovf[plo[s] + t] for t in range(d)]
            first = [recs[((s >> 4) * 32 + t) * 16 + (s & 15)] for t in range(32)]
            assert np.array_equal(np.array(first), np.array(rows[:32]))           # the table holds the first 32 of them
        rows = np.array(rows)
        k = int(kid[s])
        assert (rows[:, 3] == k).all() and k % g.table_capacity == s             # one patch per slot, slot = id mod capacity
        out[k] = rows[:, :3]
    return out, order[:E], stream[:E]


def check_table(ii, jj, kk, k_range, bind=None, cap=None):
    if cap is None:
        cap = min(k_range, (int(kk.max()) - int(kk.min()) + 16) // 16 * 16)      # just the live id range: ids wrap around it
    g = ops.GraphIndex(torch.device(DEV), E_cap=len(kk), k_range=k_range, table_capacity=cap)
    if bind is not None:
        g.bind_corr_stream(*bind)
    g.build_table(T(jj), T(kk), ii=None if ii is None else T(ii), with_neighbors=True)
    ix, jx = g.neighbors()
    ix_o, jx_o = O.neighbors(kk, jj)
    assert np.array_equal(ix.cpu().numpy(), ix_o) and np.array_equal(jx.cpu().numpy(), jx_o)
    lists, order, stream = table_lists(g, len(kk))
    # numpy: the edges of every patch ordered by (jj, edge id)
    perm = np.lexsort((np.arange(len(kk)), jj, kk))
    ks, first = np.unique(kk[perm], return_index=True)
    assert sorted(lists) == [int(k) for k in ks]
    bounds = list(first) + [len(kk)]
    for n, k in enumerate(ks):
        es = perm[bounds[n]:bounds[n + 1]]
        got = lists[int(k)]
        assert np.array_equal(got[:, 0], es) and np.array_equal(got[:, 2], jj[es])
        assert np.array_equal(got[:, 1], ii[es] if ii is not None else np.full(len(es), -1))
    m = g.meta()
    assert m[0] == len(ks) and m[1] == 1 and m[2] == ks.min() and m[3] == ks.max() and m[6] == 0 and m[7] == len(kk)
    # the correlation's processing order: a permutation of the edges, grouped by target frame mod 32
    assert np.array_equal(np.sort(order), np.arange(len(kk)))
    bins = jj[order] & 31
    assert (np.diff(bins) >= 0).all()
    # a second build on the same workspace (cursors zero again, generation moved on) gives the same index
    g.build_table(T(jj), T(kk), ii=None if ii is None else T(ii), with_neighbors=True, force=True)
    ix2, jx2 = g.neighbors()
    assert torch.equal(ix, ix2) and torch.equal(jx, jx2)
    lists2, _, _ = table_lists(g, len(kk))
    assert all(np.array_equal(lists[k], lists2[k]) for k in lists)
    # ... and agrees with the ranked build's neighbors on the same workspace, which then serves unique() again
    kx, ku = g.unique()
    kx_o, ku_o = O.unique(kk)
    assert np.array_equal(kx.cpu().numpy(), kx_o) and np.array_equal(ku.cpu().numpy(), ku_o) and not g.is_table
    return g, order, stream


@pytest.mark.parametrize("name", ["tiny", "small", "pr1", "default", "stress"])
def test_table_index_bit_exact(name):
    cfg = synth.CONFIGS[name]
    ii, jj, kk = synth.replay_edges(cfg)
    check_table(ii, jj, kk, cfg.buffer_size * cfg.M)
    check_table(None, jj, kk, cfg.buffer_size * cfg.M, cap=min(cfg.buffer_size * cfg.M, 1 << 16))   # a roomy table: no wrap


def test_table_index_irregular_and_overflowing_patches():
    """random multigraphs: duplicate (k, j) edges, gaps in the id range (ids without an edge between live ones), patches
    with more than 32 edges (overflow segment) next to ordinary ones, a single edge, one patch owning every edge"""
    rng = np.random.default_rng(11)
    E = 20000
    kk = rng.choice(np.arange(100, 9000, 3), size=E).astype(np.int64)
    jj = rng.integers(5, 300, size=E).astype(np.int64)
    ii = rng.integers(0, 300, size=E).astype(np.int64)
    check_table(ii, jj, kk, 9000)
    kk = np.concatenate([rng.integers(40, 60, 1500), rng.integers(1000, 3000, 6000)]).astype(np.int64)   # ~75 edges per patch in [40, 60)
    jj = rng.integers(0, 64, len(kk)).astype(np.int64)
    ii = rng.integers(0, 64, len(kk)).astype(np.int64)
    sh = rng.permutation(len(kk))
    check_table(ii[sh], jj[sh], kk[sh], 4096)
    check_table(None, np.array([3], np.int64), np.array([7], np.int64), 64)
    check_table(None, rng.integers(0, 20, 120).astype(np.int64), np.full(120, 42, np.int64), 64)
    # one workspace re-used for graphs with different id ranges (shrinking and growing, far beyond the capacity: ids wrap)
    g = ops.GraphIndex(torch.device(DEV), E_cap=4096, k_range=5000, table_capacity=4000)
    for lo, hi, E in ((100, 4000, 3000), (2000, 2100, 500), (70000, 73999, 4096), (4500, 4600, 50), (1_000_000_000, 1_000_003_000, 3000)):
        kk = rng.integers(lo, hi, E).astype(np.int64)
        jj = rng.integers(0, 40, E).astype(np.int64)
        g.build_table(T(jj), T(kk), with_neighbors=True)
        ix, jx = g.neighbors()
        ix_o, jx_o = O.neighbors(kk, jj)
        assert np.array_equal(ix.cpu().numpy(), ix_o) and np.array_equal(jx.cpu().numpy(), jx_o)
        m = g.meta()
        assert m[0] == len(np.unique(kk)) and m[2] == kk.min() and m[3] == kk.max()


def test_table_index_error_states():
    """two live ids in one slot (the capacity is smaller than the live id range), a negative id, or a patch with more edges
    than the sort launch serves (128): the index reports its error state, neighbors say "none", and the next well-formed
    build on the workspace is fine again"""
    import os
    rng = np.random.default_rng(5)
    g = ops.GraphIndex(torch.device(DEV), E_cap=4096, k_range=4096, table_capacity=1000)
    good_k, good_j = rng.integers(5000, 6000, 2000).astype(np.int64), rng.integers(0, 30, 2000).astype(np.int64)
    e = np.arange(2000)
    for bad_k in (np.where(e == 77, good_k[78] + 1000, good_k), np.where(e == 5, -1, good_k), np.where(e < 200, 5321, good_k),
                  np.where(e == 3, good_k[3] + (1 << 40), good_k)):
        os.environ["CDV_CHECK"] = "0"
        g.build_table(T(good_j), T(bad_k.astype(np.int64)), with_neighbors=True, force=True)
        ix, jx = g.neighbors()
        assert bool((ix == -1).all()), int((ix != -1).sum())
        assert bool((jx == -1).all()), int((jx != -1).sum())
        with pytest.raises(ops._lib.CdvError):
            g.meta()
        g.build_table(T(good_j), T(good_k), with_neighbors=True, force=True)
        ix, jx = g.neighbors()
        ix_o, jx_o = O.neighbors(good_k, good_j)
        assert np.array_equal(ix.cpu().numpy(), ix_o) and np.array_equal(jx.cpu().numpy(), jx_o) and g.meta()[6] == 0


def test_table_stream_from_bound_coordinates():
    """cdv_graph_build_table with a bound coordinate buffer writes the correlation's packed input stream: record p belongs
    to edge order[p] and carries its 18 coordinates, the reduced ring indices (slam.py:319-320) and the floor extremes"""
    st = synth.make_state("small", features=False)
    rng = np.random.default_rng(2)
    coords = (rng.uniform(-5, 70, (1, st.E, 2, 3, 3))).astype(np.float32)
    coords[0, 7] = 1e9
    coords[0, 8] = -1e9
    ct = T(coords)
    kmod, jmod, Ng, slots = st.cfg.M * st.cfg.pmem, st.cfg.mem, (st.cfg.M * st.cfg.pmem * 3) // 4, st.cfg.mem
    g, order, stream = check_table(st.ii, st.jj, st.kk, st.cfg.buffer_size * st.cfg.M, bind=(ct, kmod, jmod, Ng, slots))
    # check_table ended with a ranked build (unique()): build the table once more for the stream it leaves
    g.build_table(T(st.jj), T(st.kk), ii=T(st.ii), force=True)
    _, order, stream = table_lists(g, st.E)
    e = stream[:, 18]
    assert np.array_equal(e, order)
    assert np.array_equal(stream[:, :18].view(np.float32), coords[0].reshape(st.E, 18)[e])
    kq, jq = st.kk[e] % kmod, st.jj[e] % jmod
    ok = kq < Ng
    assert np.array_equal(stream[:, 19].astype(np.int64)[ok], kq[ok]) and np.array_equal(stream[:, 20].astype(np.int64)[ok], jq[ok])
    assert (stream[:, 19][~ok] == -1).all() and (stream[:, 20][~ok] == -1).all() and (~ok).any()
    fl = lambda v: np.clip(np.floor(v), -30000, 30000).astype(np.int64)
    x, y = fl(coords[0, e, 0].reshape(st.E, 9)), fl(coords[0, e, 1].reshape(st.E, 9))
    bx, by = stream[:, 21].astype(np.int64), stream[:, 22].astype(np.int64)
    s16 = lambda v: ((v & 0xffff) ^ 0x8000) - 0x8000
    assert np.array_equal(s16(bx), x.min(1)) and np.array_equal(bx >> 16, x.max(1))
    assert np.array_equal(s16(by), y.min(1)) and np.array_equal(by >> 16, y.max(1))


@pytest.mark.parametrize("name", ["small", "default"])
def test_prologue_table_equals_separate_launches(name):
    """cdv_update_prologue_table (2 launches) == ring ingest + cdv_transform + table build issued one by one: rings and
    pixel-major tiles bit for bit, coordinates bit for bit, neighbors, and the stream carries those coordinates"""
    st = synth.make_state(name)
    dev = torch.device(DEV)
    mem, C, h, w = st.fmap1.shape
    M = st.cfg.M

    def rings():
        f1, f2 = ops.alloc_fmap_ring(mem, C, h, w, dev), ops.alloc_fmap_ring(mem, C, h // 4, w // 4, dev)
        for s in range(mem - 1):
            ops.fmap_ingest(T(st.fmap1[s]), f1, f2, s)
        return f1, f2

    gmap = T(st.gmap)
    new_frame, slot, tiles = T(st.fmap1[mem - 1]), mem - 1, ((st.n - 1) % st.cfg.pmem) * M
    poses, patches, intr = T(st.poses), T(st.patches), T(st.intrinsics)
    ii, jj, kk = T(st.ii), T(st.jj), T(st.kk)
    # one by one
    a1, a2 = rings()
    pm_a = ops.gmap_to_pixel_major(gmap)
    pm_a[tiles:tiles + M] = 0
    ops.fmap_ingest(new_frame, a1, a2, slot, gmap=gmap, gmap_pm=pm_a, gmap_first=tiles, gmap_count=M)
    coords_a = ops.transform(poses[None], patches[None], intr[None], ii, jj, kk, layout_e2pp=True)
    # fused
    b1, b2 = rings()
    pm_b = ops.gmap_to_pixel_major(gmap)
    pm_b[tiles:tiles + M] = 0
    g = ops.GraphIndex(dev, E_cap=st.E, k_range=st.cfg.buffer_size * M, table_capacity=(st.cfg.removal_window + 2) * M)
    g.bind_corr_stream(None, M * st.cfg.pmem, st.cfg.mem, gmap.shape[0], mem)
    coords_b = ops.update_prologue_table(g, new_frame, b1, b2, slot, gmap, pm_b, tiles, M, poses, patches, intr, ii, jj, kk)
    assert torch.equal(a1, b1) and torch.equal(a2, b2) and torch.equal(pm_a, pm_b)
    # (the two kernels run the same formulas; the compiler contracts their multiply-adds differently: last-bit differences)
    assert float((coords_a - coords_b).abs().max()) < 1e-4
    want = O.transform(st.poses, st.patches, st.intrinsics, st.ii, st.jj, st.kk, dtype=np.float64).transpose(0, 3, 1, 2)
    assert np.abs(coords_b[0].cpu().numpy() - want).max() < 1e-3
    ix, jx = g.neighbors()
    ix_o, jx_o = O.neighbors(st.kk, st.jj)
    assert np.array_equal(ix.cpu().numpy(), ix_o) and np.array_equal(jx.cpu().numpy(), jx_o)
    _, order, stream = table_lists(g, st.E)
    assert np.array_equal(np.sort(order), np.arange(st.E)) and np.array_equal(stream[:, 18], order)
    assert np.array_equal(stream[:, :18].view(np.float32), coords_b[0].cpu().numpy().reshape(st.E, 18)[order])


# ---------------------------------------------------------------------------------------------------
# bundle adjustment on the table index
# ---------------------------------------------------------------------------------------------------

def _ba(st, form, iterations=2, debug=False, cap=None):
    dev = torch.device(DEV)
    if cap is None:
        cap = (int(st.kk.max()) - int(st.kk.min()) + 16) // 16 * 16            # just the live id range
    g = ops.GraphIndex(dev, E_cap=st.E, k_range=st.cfg.buffer_size * st.cfg.M, table_capacity=cap)
    ii, jj, kk = T(st.ii), T(st.jj), T(st.kk)
    (g.build_table if form == "table" else g.build)(jj, kk, ii=ii)
    poses, patches = T(st.poses).clone(), T(st.patches).clone()
    res = ops.ba_forward(poses, patches, T(st.intrinsics), T(st.target), T(st.weight), torch.tensor([st.lmbda], device=dev), ii, jj,
                         kk, st.cfg.M, st.t0, st.n, iterations, False, graph=g, debug=debug)
    torch.cuda.synchronize()
    assert g.is_table == (form == "table")          # the index that was there has been used, not replaced
    assert ops.ba_status(raise_on_error=False) == (0, 0, 0, 0)
    return poses.cpu().numpy(), patches.cpu().numpy(), res


@pytest.mark.parametrize("name", ["small", "init", "pr1", "default", "stress", "mid15", "mid19_m5", "mid11", "mid32"])
def test_ba_on_the_table_against_the_ranked_index_and_the_oracle(name):
    """the same bundle adjustment over the table's slots (N <= 10 and 10 < N <= 32 paths): the stated bounds against the
    float64 oracle (tests/ba_checks.py), the ranked index's result to rounding (other chunk boundaries and slab order,
    same sums), identical bits from run to run and for a roomier table whose slots do not wrap"""
    from tests import ba_checks
    from tests.test_gpu_parity import _make
    st, tol = _make(name)
    pt, xt, _ = _ba(st, "table")
    pr, xr, _ = _ba(st, "ranked")
    # two float32 summation orders of the same sums (chunk boundaries, slab order); the 192-unknown system of mid32 carries
    # the difference a little further than the smaller ones
    rnd = (5e-6, 5e-5) if name == "mid32" else (2e-6, 2e-5)
    assert np.abs(pt - pr).max() < rnd[0] and np.abs(xt - xr).max() < rnd[1]
    assert not np.array_equal(pt, st.poses)
    p64, x64, info = O.fastba(st.poses, st.patches, st.intrinsics[0], st.target, st.weight, st.lmbda, st.ii, st.jj, st.kk, st.t0,
                              st.n, 2, np.float64)
    ba_checks.check_end_state(tol, st, pt, xt, p64, x64)
    pt2, xt2, _ = _ba(st, "table")
    assert np.array_equal(pt, pt2) and np.array_equal(xt, xt2)
    pw, xw, _ = _ba(st, "table", cap=min(st.cfg.buffer_size * st.cfg.M, 1 << 16))
    assert np.abs(pw - pr).max() < rnd[0] and np.abs(xw - xr).max() < rnd[1]
    # iteration-0 intermediates through the table (per-patch rows handed out by unique rank)
    _, _, dbg = _ba(st, "table", iterations=1, debug=True)
    _, _, info, o = O.fastba(st.poses, st.patches, st.intrinsics[0], st.target, st.weight, st.lmbda, st.ii, st.jj, st.kk, st.t0,
                             st.n, 1, np.float64, debug=True)
    U = len(o["kx"])
    for key, got, want in (("S", dbg["S"], o["S"]), ("y", dbg["y"], o["y"]), ("C", dbg["C"][:U], o["C"]),
                           ("u", dbg["u"][:U], o["u"]), ("E", dbg["E"][:, :U], o["E"])):
        assert np.abs(got.cpu().numpy() - want).max() <= 1e-4 * np.abs(want).max(), key


@pytest.mark.parametrize("variant", ["small", "mid15"])
def test_ba_on_a_table_with_ids_that_have_no_edge(variant):
    """patches inside the live id range WITHOUT any edge (never produced by slam.py: ids are renumbered when a frame is
    dropped): they take a row of the table's span, contribute nothing, and are not retracted -- also when their inverse
    depth lies outside the clamps of patch_retr (ba_cuda.cu:219-221); against the float64 oracle like every other graph"""
    from tests import ba_checks
    from tests.test_gpu_parity import _make
    st, tol = _make(variant)
    rng = np.random.default_rng(3)
    ids = np.unique(st.kk)
    dead = rng.choice(ids[5:-5], size=len(ids) // 7, replace=False)
    keep = ~np.isin(st.kk, dead)
    st.ii, st.jj, st.kk = st.ii[keep].copy(), st.jj[keep].copy(), st.kk[keep].copy()
    st.target, st.weight = st.target[keep].copy(), st.weight[keep].copy()
    st.patches = st.patches.copy()
    st.patches[dead[::2], 2] = 50.0                  # > 20: patch_retr would set it to 1
    st.patches[dead[1::2], 2] = 1e-6                 # < 1e-4: patch_retr would raise it
    poses, patches, dbg = _ba(st, "table", iterations=1, debug=True)
    _, _, info, o = O.fastba(st.poses, st.patches, st.intrinsics[0], st.target, st.weight, st.lmbda, st.ii, st.jj, st.kk,
                             st.t0, st.n, 1, np.float64, debug=True)
    U = len(o["kx"])
    for key, got, want in (("S", dbg["S"], o["S"]), ("y", dbg["y"], o["y"]), ("C", dbg["C"][:U], o["C"]),
                           ("u", dbg["u"][:U], o["u"]), ("E", dbg["E"][:, :U], o["E"])):
        assert np.abs(got.cpu().numpy() - want).max() <= 1e-4 * np.abs(want).max(), key
    poses, patches, _ = _ba(st, "table")
    assert np.array_equal(patches[dead], st.patches[dead])
    p64, x64, _ = O.fastba(st.poses, st.patches, st.intrinsics[0], st.target, st.weight, st.lmbda, st.ii, st.jj, st.kk, st.t0,
                           st.n, 2, np.float64)
    ba_checks.check_end_state(tol, st, poses, patches, p64, x64)
    pr, xr, _ = _ba(st, "ranked")
    assert np.abs(pr - poses).max() < 2e-6          # other chunking than the ranked index: same numbers to rounding


def test_ba_with_overflowing_patches_on_the_table():
    """patches with more than 32 edges (first 32 records in the table, all of them in the overflow segment): duplicated
    edges push some patches of the `small` graph to ~60 edges"""
    from tests import ba_checks
    st = synth.make_state("small", features=False)
    rng = np.random.default_rng(9)
    heavy = np.unique(st.kk)[10:40]
    extra = np.flatnonzero(np.isin(st.kk, heavy))
    sel = np.concatenate([np.arange(st.E), extra, extra[rng.random(len(extra)) < 0.5]])
    rng.shuffle(sel)
    st.ii, st.jj, st.kk = st.ii[sel].copy(), st.jj[sel].copy(), st.kk[sel].copy()
    st.target, st.weight = st.target[sel].copy(), st.weight[sel].copy()
    assert np.bincount(st.kk).max() > 40
    poses, patches, _ = _ba(st, "table")
    p64, x64, _ = O.fastba(st.poses, st.patches, st.intrinsics[0], st.target, st.weight, st.lmbda, st.ii, st.jj, st.kk, st.t0,
                           st.n, 2, np.float64)
    ba_checks.check_end_state("small", st, poses, patches, p64, x64)
    pr, xr, _ = _ba(st, "ranked")
    assert np.abs(pr - poses).max() < 2e-6 and np.abs(xr - patches).max() < 2e-5


def test_global_ba_needs_the_ranked_index_and_gets_it():
    """N > 32 (the global bundle adjustment) works on unique ranks: the C ABI refuses a table there, ops.ba_forward builds
    the ranked index"""
    st = synth.make_state("global", features=False)
    dev = torch.device(DEV)
    g = ops.GraphIndex(dev, E_cap=st.E, k_range=st.cfg.buffer_size * st.cfg.M, table_capacity=st.cfg.buffer_size * st.cfg.M)
    ii, jj, kk = T(st.ii), T(st.jj), T(st.kk)
    g.build_table(jj, kk, ii=ii)
    poses, patches = T(st.poses).clone(), T(st.patches).clone()
    lib = ops._lib.load()
    ws = ops._ba_workspace(dev, st.E, len(st.patches), st.n - st.t0)
    rc = lib.cdv_ba_forward(ops._p(poses), ops._p(patches), ops._p(T(st.intrinsics)), ops._p(T(st.target)), ops._p(T(st.weight)),
                            ops._p(torch.tensor([st.lmbda], device=dev)), ops._p(ii), ops._p(jj), ops._p(kk), st.E, 3, st.t0, st.n, 2,
                            ops._p(g.ws), ops._p(ws), ws.numel(), len(st.patches), None, ops._stream())
    assert rc != 0 and b"patch table" in lib.cdv_last_error()
    ops.ba_forward(poses, patches, T(st.intrinsics), T(st.target), T(st.weight), torch.tensor([st.lmbda], device=dev), ii, jj, kk,
                   st.cfg.M, st.t0, st.n, 2, True, graph=g)
    torch.cuda.synchronize()
    assert not g.is_table and not torch.equal(poses, T(st.poses))


def test_dropin_table_gives_way_to_the_ranked_index_when_the_ids_do_not_fit():
    """install_dropin(table_capacity=...) with a capacity the live patch ids outgrow (what loop-closure / long-range edges do,
    slam.py:507-510): the update that meets the collision is skipped and counted -- and the next one finds the count (no
    synchronisation: pinned counters), gives the table up on that device and goes through the ranked index: neighbors and
    the bundle adjustment equal the ranked index's, the caller is told once"""
    import os
    import warnings
    import cdv_slam_amd
    st = synth.make_state("small", features=False)
    dev = torch.device(DEV)
    ii, jj, kk = T(st.ii), T(st.jj), T(st.kk)
    span = int(st.kk.max() - st.kk.min()) + 1
    args = lambda po, pa: (po, pa, T(st.intrinsics), T(st.target), T(st.weight), torch.tensor([st.lmbda], device=dev), ii, jj, kk,
                           st.cfg.M, st.t0, st.n, 2, False)
    want_p, want_x = T(st.poses).clone(), T(st.patches).clone()
    ops.configure_table(None)
    ops.ba_forward(*args(want_p, want_x))                       # the ranked index
    ix_o, jx_o = O.neighbors(st.kk, st.jj)
    cuda_corr, cuda_ba, _ = cdv_slam_amd.install_dropin(table_capacity=span // 2)     # two live ids per slot
    os.environ["CDV_CHECK"] = "0"
    try:
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            poses, patches = T(st.poses).clone(), T(st.patches).clone()
            ix, jx = cuda_ba.neighbors(kk, jj)
            assert bool((ix == -1).all())                                           # the table's error state
            cuda_ba.forward(*args(poses, patches))
            torch.cuda.synchronize()
            assert torch.equal(poses, T(st.poses))                                  # skipped, and counted
            # the next update: fresh edge tensors as slam.py hands them over
            jj2, kk2 = jj.clone(), kk.clone()
            ix, jx = cuda_ba.neighbors(kk2, jj2)
            assert np.array_equal(ix.cpu().numpy(), ix_o) and np.array_equal(jx.cpu().numpy(), jx_o)
            cuda_ba.forward(poses, patches, *args(None, None)[2:7], jj2, kk2, *args(None, None)[9:])
            torch.cuda.synchronize()
        assert any("falling back to the ranked index" in str(x.message) for x in w)
        assert not ops._device_graph(dev).is_table
        # (the index neighbors() built carries no source frames, the first run's did: two instantiations of the same kernels,
        # equal up to the compiler's multiply-add contractions)
        assert float((poses - want_p).abs().max()) < 1e-6 and float((patches - want_x).abs().max()) < 1e-5
        assert not torch.equal(poses, T(st.poses))
    finally:
        os.environ["CDV_CHECK"] = "1"
        ops.configure_table(None)


def test_table_build_captured_as_a_hipgraph_serves_ids_that_move_round_the_table():
    """The two launches of a table build carry nothing of the host but pointers and sizes (the build's generation is a device
    word): captured ONCE into a hipGraph and replayed over edge lists whose patch ids move on by a quarter of the table's
    capacity per replay -- every slot meets new ids under the very same launches, several times round -- every replay gives
    the oracle's neighbors, bit for bit.  In between a replay whose ids do NOT fit the table (two live ids per slot): the
    error state, neighbors -1; the replay after it is clean again -- the error word belongs to one build and is cleared by
    the build before.  (Round 4 froze a host-side generation into the launches: the first wrap put every later replay into
    the error state.)"""
    import os
    os.environ["CDV_CHECK"] = "0"      # (CDV_CHECK=1 reads the index's status back after every build: not capturable)
    dev = torch.device(DEV)
    rng = np.random.default_rng(3)
    cap, E = 64, 900
    g = ops.GraphIndex(dev, E_cap=E, k_range=4096, table_capacity=cap)
    jj_d = torch.zeros(E, dtype=torch.int64, device=dev)
    kk_d = torch.zeros(E, dtype=torch.int64, device=dev)

    def edges(lo, span):
        kk = lo + rng.integers(0, span, E)
        jj = rng.integers(0, 30, E)
        return kk.astype(np.int64), jj.astype(np.int64)

    kk, jj = edges(100, cap)
    kk_d.copy_(T(kk)); jj_d.copy_(T(jj))
    g.build_table(jj_d, kk_d, with_neighbors=True, force=True)          # warm-up (one-time initialisation outside the capture)
    torch.cuda.synchronize()
    cg = torch.cuda.CUDAGraph()
    with torch.cuda.graph(cg):
        g.build_table(jj_d, kk_d, with_neighbors=True, force=True)
        ix, jx = g.neighbors()
    torch.cuda.synchronize()
    for rep in range(14):
        bad = rep in (5, 9)
        kk, jj = edges(100 + 16 * rep, 2 * cap if bad else cap)
        if bad:
            assert len(np.unique(kk % cap)) < len(np.unique(kk))         # two live ids in one slot
        kk_d.copy_(T(kk)); jj_d.copy_(T(jj))
        cg.replay()
        torch.cuda.synchronize()
        if bad:
            assert bool((ix == -1).all()) and bool((jx == -1).all()), rep
            with pytest.raises(ops._lib.CdvError, match="range"):
                g.meta()
        else:
            ix_o, jx_o = O.neighbors(kk, jj)
            assert np.array_equal(ix.cpu().numpy(), ix_o) and np.array_equal(jx.cpu().numpy(), jx_o), rep
            assert g.meta()[6] == 0


@pytest.mark.parametrize("n_max", [10, 22, 32])
def test_ba_forward_dyn_equals_the_static_call_for_every_window_inside_its_bound(n_max):
    """cdv_ba_forward_dyn lays its launches out for N_max free poses and takes the window [t0, t0 + N) from a dynamic block on
    the device: for N_max = 10 (window kernels) and 22 / 32 (the 10 < N <= 32 kernels) and several actual windows -- the full
    one, a smaller one, the 7 free poses of a stream's first update -- poses and patches come out BIT FOR BIT as from
    cdv_ba_forward(t0, t0 + N) on the same table, whenever the static call takes the same kernels (N > 10 on the wide
    bounds), and within the BA's rounding bounds where it takes the window kernels instead (N <= 10 inside a bound of 22)."""
    import ctypes
    os_environ = __import__("os").environ
    os_environ["CDV_CHECK"] = "0"
    from cdv_slam_amd import _lib
    lib = _lib.load()
    dev = torch.device(DEV)
    frames = n_max + 4
    st = synth.make_state("small", features=False, frames=frames, opt_window=n_max, removal_window=frames + 2, buffer_size=frames + 8)
    M = st.cfg.M
    cap = (st.cfg.removal_window + 2) * M
    g = ops.GraphIndex(dev, E_cap=st.E, k_range=st.cfg.buffer_size * M, table_capacity=cap)
    ii, jj, kk = T(st.ii), T(st.jj), T(st.kk)
    g.build_table(jj, kk, ii=ii, force=True)
    tgt, wgt, intr, lm = T(st.target), T(st.weight), T(st.intrinsics), torch.tensor([st.lmbda], device=DEV)
    ws = ops.ba_private_workspace(dev, st.E, cap, n_max)
    assert lib.cdv_ba_set_patches_per_frame(ctypes.c_void_p(ws.data_ptr()), M) == 0     # as ops.ba_forward tells its workspace (PPF)
    dyn = torch.zeros(16, dtype=torch.int32, device=dev)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    for N in sorted({n_max, max(n_max - 5, 1), 7}):
        t0 = st.n - N
        dyn[4], dyn[5], dyn[0], dyn[1] = t0, N, st.n, st.E
        pd, xd = T(st.poses).clone(), T(st.patches).clone()
        rc = lib.cdv_ba_forward_dyn(P(pd), P(xd), P(intr), P(tgt), P(wgt), P(lm), P(ii), P(jj), P(kk), st.E, 3, n_max, P(dyn), 2,
                                    P(g.ws), P(ws), ws.numel(), cap, ops._stream())
        assert rc == 0, lib.cdv_last_error()
        ps, xs = T(st.poses).clone(), T(st.patches).clone()
        ops.ba_forward(ps, xs, intr, tgt, wgt, lm, ii, jj, kk, M, t0, st.n, 2, False, U_max=cap, graph=g)
        torch.cuda.synchronize()
        assert not torch.equal(pd, T(st.poses))
        same_kernels = (N <= 10) == (n_max <= 10)
        if same_kernels:
            assert torch.equal(pd, ps) and torch.equal(xd, xs), (n_max, N)
        else:
            assert float((pd - ps).abs().max()) < 2e-6 and float((xd - xs).abs().max()) < 2e-5, (n_max, N)
        assert ws.events.counts() == [0, 0, 0, 0]
