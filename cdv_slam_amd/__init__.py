"""cdv_slam_amd -- MI355X (gfx950) implementation of CDV-SLAM's per-frame update hot path.

Layout (only what the path needs):
  csrc/                hand-written HIP kernels + the C ABI (include/cdvslam_hip.h) -> libcdvslam_hip.so
  ops.py               torch plumbing over the C ABI (device pointers + current HIP stream)
  altcorr/ fastba/ lietorch/ projective_ops.py
                       host-side mirror of the reference operator surface
                       (cdvslam/altcorr, cdvslam/fastba, cdvslam/lietorch, cdvslam/projective_ops.py)
  dropin/              modules named cuda_corr / cuda_ba / lietorch_backends with the reference's pybind
                       signatures, so cdvslam/slam.py and net_cdv.py run unchanged
  update.py            the fused per-frame update path used by bench.py (reproject -> corr -> neighbors -> BA)
  synth.py             seeded synthetic patch-graph states
"""

__all__ = ["ops", "altcorr", "fastba", "lietorch", "projective_ops", "ba", "synth", "install_dropin"]


def install_dropin(table_capacity=None, package=None):
    """Register cuda_corr, cuda_ba and lietorch_backends in sys.modules (reference import names:
    cdvslam/altcorr/correlation.py:2, cdvslam/fastba/ba.py:2, cdvslam/lietorch/group_ops.py:1).
    table_capacity (optional): the number of patch ids that can be live at once, (REMOVAL_WINDOW + 2) * PATCHES_PER_FRAME of
    the configuration the SLAM object runs (2,304 for default_cdvo.yaml): cuda_ba.neighbors / forward then use the
    two-launch table form of the patch-graph index (ops.configure_table); without it the ranked index, which assumes
    nothing about the ids.
    package (optional, e.g. "cdvslam"; call BEFORE importing it): also answer `<package>.projective_ops` with
    cdv_slam_amd.projective_ops, so that `from . import projective_ops as pops` in the package's slam.py (slam.py:7) gets the
    one-launch transform / flow_mag / point_cloud instead of composing ~15 lietorch and torch launches per call
    (projective_ops.py:53-130) over lietorch_backends.  The pose argument may be the package's own SE3 object: anything with
    a `.data` tensor and group_id 3 is read (projective_ops._kernel_pose_rows)."""
    import sys
    from . import ops, projective_ops
    from .dropin import cuda_ba, cuda_corr, lietorch_backends
    if table_capacity is not None:
        ops.configure_table(table_capacity)
    sys.modules.setdefault("cuda_corr", cuda_corr)
    sys.modules.setdefault("cuda_ba", cuda_ba)
    sys.modules.setdefault("lietorch_backends", lietorch_backends)
    if package:
        sys.modules.setdefault(package + ".projective_ops", projective_ops)
    return cuda_corr, cuda_ba, lietorch_backends
