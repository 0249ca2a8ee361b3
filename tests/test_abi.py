"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/cdvslam_hip.h declares, argument validation returns error codes (no exit()), and the Python
operator surface has the reference's names.  No GPU compute is launched here."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "cdvslam_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cdv_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from cdv_slam_amd import _lib
    lib = _lib.load()
    names = _declared_functions()
    assert len(names) >= 17
    for n in names:
        assert hasattr(lib, n), "missing export: " + n
    assert set(names) == set(_lib.SIGNATURES), "ctypes table and header disagree"
    assert b"gfx950" in lib.cdv_version()


def test_argument_errors_are_codes_not_exits():
    from cdv_slam_amd import _lib
    lib = _lib.load()
    # unsupported group id -> CDV_ERR_UNSUPPORTED before anything touches the GPU
    rc = lib.cdv_lie_op(2, 0, 1, 4, None, None, None, None)
    assert rc == -4 and b"SO3" in lib.cdv_last_error()
    rc = lib.cdv_corr_fused(None, None, None, None, None, None, None, None, 10, 1, 1, 24, 8, 8, 2, 2, 1.0, 4.0, 3, 0,
                            0, 0, None)
    assert rc == -2
    rc = lib.cdv_graph_build(None, None, 10, None, 0, 16, 16, None)
    assert rc == -2
    assert lib.cdv_graph_workspace_bytes(1000, 100) > 1000 * 4 * 3
    assert lib.cdv_ba_workspace_bytes(1000, 100, 10) > 4 * 3660 * 4  # BA_REPL copies of [S | y]
    # more than 1024 free poses is a clean error; 40 free poses (global BA) get past that check to the next one
    rc = lib.cdv_ba_forward(None, None, None, None, None, None, None, None, None, 10, 3, 0, 2000, 2, None, None, 0, 10,
                            None, None)
    assert rc == -4 and b"1024" in lib.cdv_last_error()
    rc = lib.cdv_ba_forward(None, None, None, None, None, None, None, None, None, 10, 3, 0, 40, 2, None, None, 0, 10,
                            None, None)
    assert rc == -2 and b"graph_ws" in lib.cdv_last_error()
    # the global-BA workspace holds the dense E and the Cholesky working matrix
    assert lib.cdv_ba_workspace_bytes(100000, 10000, 300) > 6 * 300 * 10000 * 4 + (6 * 300) ** 2 * 4


def test_operator_surface_names():
    import cdv_slam_amd
    from cdv_slam_amd import altcorr, fastba, lietorch, projective_ops
    assert callable(altcorr.corr) and callable(altcorr.patchify)
    assert callable(fastba.BA) and callable(fastba.neighbors) and callable(fastba.reproject)
    for n in ("SE3", "SO3", "cat", "stack"):
        assert hasattr(lietorch, n)
    for n in ("transform", "iproj", "proj", "point_cloud", "flow_mag", "reproject"):
        assert hasattr(projective_ops, n)
    cc, cb, lb = cdv_slam_amd.install_dropin()
    for n in ("forward", "backward", "patchify_forward", "patchify_backward"):
        assert hasattr(cc, n)
    for n in ("forward", "neighbors", "reproject", "solve_system"):
        assert hasattr(cb, n)
    for n in ("expm", "logm", "inv", "mul", "adj", "adjT", "act", "act4", "as_matrix", "projector", "Jinv",
              "expm_backward", "act4_backward"):
        assert hasattr(lb, n)
    assert lietorch.SE3.group_id == 3 and lietorch.SO3.group_id == 1


def test_hip_path_refuses_cpu_tensors():
    import torch
    from cdv_slam_amd import ops
    with pytest.raises(RuntimeError):
        ops.lie_op(3, "exp", torch.zeros(2, 6))
    with pytest.raises(RuntimeError):
        ops.neighbors(torch.zeros(4, dtype=torch.long), torch.zeros(4, dtype=torch.long))


def test_product_never_imports_oracle():
    """the oracle is test infrastructure: nothing under cdv_slam_amd/ may reference it"""
    pkg = os.path.join(ROOT, "cdv_slam_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
                assert "libcdv_oracle" not in txt, f


def test_synth_graph_sizes():
    from cdv_slam_amd import synth
    import numpy as np
    for name, E, U, N in (("default", 47712, 2208, 10), ("stress", 97412, 4508, 22), ("init", 6144, 768, 7),
                          ("pr1", 9600, 960, 9)):
        ii, jj, kk = synth.replay_edges(synth.CONFIGS[name])
        assert len(ii) == E and len(np.unique(kk)) == U
    st = synth.make_state("default", features=False)
    assert st.n - st.t0 == 10


def test_ba_workspace_accounts_for_the_pair_index():
    """the global bundle adjustment (more than 32 free poses) builds a frame-pair index of the call's edges inside its
    workspace: cdv_ba_workspace_bytes grows with E_max there, and only there (host arithmetic, no GPU call)"""
    from cdv_slam_amd import _lib
    lib = _lib.load()
    small = [lib.cdv_ba_workspace_bytes(E, 4096, 10) for E in (1000, 1000000)]
    assert small[0] == small[1]
    big = [lib.cdv_ba_workspace_bytes(E, 4096, 64) for E in (1000, 100000, 1000000)]
    assert big[0] < big[1] < big[2]
    # ~250 bytes per edge of index workspace + 8 of keys, and at most (N + 1)(N + 2) / 2 pair slots of 384 bytes
    per_edge = (big[2] - big[1]) / 900000.0
    assert 100.0 < per_edge < 600.0


def test_stream_descriptor_layout_matches_the_header(tmp_path):
    """cdv_stream_desc is handed over by address: the ctypes mirror must have the C compiler's layout"""
    import ctypes
    import subprocess
    from cdv_slam_amd import _lib
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "%s"\nint main(void){printf("%%zu %%zu %%zu %%zu %%zu %%zu\\n", '
                   'sizeof(cdv_stream_desc), offsetof(cdv_stream_desc, slot), offsetof(cdv_stream_desc, edge_capacity), '
                   'offsetof(cdv_stream_desc, ii), offsetof(cdv_stream_desc, mirror_host), offsetof(cdv_stream_desc, bufs));return 0;}\n'
                   % os.path.join(ROOT, "include", "cdvslam_hip.h"))
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-o", str(exe), str(src)])
    got = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    D = _lib.StreamDesc
    assert got == [ctypes.sizeof(D), D.slot.offset, D.edge_capacity.offset, D.ii.offset, D.mirror_host.offset, D.bufs.offset]


def test_compiled_dropin_bookkeeping_loads_binds_and_declines_unarmed():
    """cdv_slam_amd/_dropin_fast.so (csrc/dropin_fast.cpp: the steady state of the drop-in modules' per-call bookkeeping,
    compiled against torch's C++ API as the reference's own bindings are, correlation.cpp:57-63 / ba.cpp:183-188): it
    imports without a GPU, binds every C-ABI entry point it calls from the library _lib.load() opened, and -- nothing armed --
    declines every call, so that the Python code serves it."""
    import torch
    from cdv_slam_amd import _lib, ops
    m = ops._fast_mod()
    assert m, "the extension was not built (make -C cdv_slam_amd/csrc)"
    for n in ("bind", "arm_pair", "disarm_pair", "drop_pending", "corr", "arm_graph", "disarm_graph", "neighbors", "ba", "transform"):
        assert callable(getattr(m, n)), n
    assert set(ops._FAST_SYMS) <= set(_lib.SIGNATURES)
    assert ops.fast_lane_enabled()
    os.environ["CDV_DROPIN_FAST"] = "0"
    try:
        assert not ops.fast_lane_enabled()          # read at every call
    finally:
        del os.environ["CDV_DROPIN_FAST"]
    x, i = torch.zeros(4), torch.zeros(4, dtype=torch.int64)
    assert m.disarm_pair() is None and m.disarm_graph() is None
    assert m.corr(1, x, x, x, i, i, 3, 0) is None
    assert m.neighbors(1, i, i, 0) is None
    assert m.ba(1, x, x, x, x, x, x, i, i, i, 96, 0, 10, 2, 0) is None
    assert m.transform(x.view(1, 1, 4), x, x, i, i, i, 0) is None      # host tensors: not served (and nothing is launched)
    # handing over state that is not what the lane serves is refused at the hand-over, not at the next call
    with pytest.raises(RuntimeError):
        m.arm_pair({"A": {"src": x, "shadow": x, "ws": x, "version": 0, "parity": 0}, "B": {"src": x, "shadow": x, "ws": x, "version": 0, "parity": 0},
                    "ratio": 4, "tiles_src": x, "tiles_pm": x, "tiles_version": 0})
    assert m.disarm_pair() is None


def test_install_dropin_can_answer_the_packages_projective_ops(tmp_path, monkeypatch):
    """install_dropin(package=...): `from . import projective_ops as pops` inside the package's slam.py (cdvslam/slam.py:7)
    resolves to cdv_slam_amd.projective_ops, and the fused transform accepts the package's OWN pose objects (anything with a
    `.data` tensor and group_id 3).  Shown on a stand-in package with the same import line and no projective_ops.py of its own
    that could be picked up instead."""
    import sys
    import torch
    import cdv_slam_amd
    from cdv_slam_amd import projective_ops as ours
    pkg = tmp_path / "standin_slam"
    pkg.mkdir()
    (pkg / "__init__.py").write_text("")
    (pkg / "slam.py").write_text("from . import projective_ops as pops\n")
    monkeypatch.syspath_prepend(str(tmp_path))
    cdv_slam_amd.install_dropin(package="standin_slam")
    try:
        import importlib
        slam = importlib.import_module("standin_slam.slam")
        assert slam.pops is ours
        assert sys.modules["cuda_corr"].forward and sys.modules["cuda_ba"].neighbors and sys.modules["lietorch_backends"]

        class TheirSE3:                 # the reference's lietorch.SE3 carries its rows in .data and names its group (groups.py:268)
            group_id = 3
            def __init__(self, data):
                self.data = data
        rows = ours._kernel_pose_rows(TheirSE3(torch.zeros(1, 4, 7)), "transform")
        assert rows.shape == (1, 4, 7)
        TheirSE3.group_id = 4           # Sim3: not served
        with pytest.raises(NotImplementedError):
            ours._kernel_pose_rows(TheirSE3(torch.zeros(1, 4, 7)), "transform")
        with pytest.raises(TypeError):
            ours._kernel_pose_rows(type("NoRows", (), {"group_id": 3})(), "transform")
    finally:
        for n in ("standin_slam", "standin_slam.slam", "standin_slam.projective_ops"):
            sys.modules.pop(n, None)
