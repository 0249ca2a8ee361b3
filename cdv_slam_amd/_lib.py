"""ctypes loader for libcdvslam_hip.so (the C-ABI HIP library, include/cdvslam_hip.h).

There is NO CPU fallback: if the shared library is missing or a call fails, this raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# CDV_LIB selects another build of the same library (the diagnostic in-kernel-stamp build); default = product
LIB_PATH = os.environ.get("CDV_LIB") or os.path.join(_HERE, "libcdvslam_hip.so")

_vp, _i32, _i64, _sz, _f32 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_size_t, ctypes.c_float

# name -> (restype, argtypes); mirrors include/cdvslam_hip.h one to one
SIGNATURES = {
    "cdv_last_error": (ctypes.c_char_p, []),
    "cdv_version": (ctypes.c_char_p, []),
    "cdv_workspace_forget": (None, [_vp]),
    "cdv_corr_fwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "cdv_fmap_padded_elems": (_sz, [_i64, _i32, _i32, _i32]),
    "cdv_fmap_to_nhwc": (_i32, [_vp, _vp, _i64, _i32, _i32, _i32, _i64, _i64, _vp]),
    "cdv_fmap_sync_workspace_bytes": (_sz, [_i64]),
    "cdv_fmap_sync_nhwc": (_i32, [_vp, _vp, _i64, _i32, _i32, _i32, _vp, _i32, _vp]),
    "cdv_shadows_sync": (_i32, [_vp, _i32, _vp, _vp, _i64, _i32, _vp]),
    "cdv_fmap_ingest": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "cdv_corr_fused": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _i32,
                              _f32, _f32, _i32, _i64, _i64, _i32, _vp]),
    "cdv_corr_fused_split": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _i32,
                                    _f32, _f32, _i64, _i64, _i32, _vp]),
    "cdv_corr_level_checked": (_i32, [_vp, _vp, _vp, _vp, _f32, _vp, _vp, _vp, _i32, _i64, _i64, _i64, _i32, _i32, _i32, _f32,
                                      _i64, _i64, _i32, _vp]),
    "cdv_corr_level_checked_interleaved": (_i32, [_vp, _vp, _vp, _vp, _f32, _vp, _vp, _vp, _i32, _i64, _i64, _i64, _i32, _i32, _i32,
                                                  _f32, _i64, _i64, _i32, _vp]),
    "cdv_graph_workspace_init": (_i32, [_vp, _sz, _i64, _i64, _vp]),
    "cdv_ba_workspace_init": (_i32, [_vp, _vp]),
    "cdv_gmap_to_pixel_major": (_i32, [_vp, _vp, _i64, _i32, _i64, _i64, _vp]),
    "cdv_frame_ingest": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _i64, _i64, _i64, _vp]),
    "cdv_patchify_fwd": (_i32, [_vp, _vp, _vp, _i32, _i64, _i32, _i32, _i32, _i32, _i32, _vp]),
    "cdv_loop_flow": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _vp, _vp]),
    "cdv_patchify_multi": (_i32, [_vp, _i32, _vp, _i64, _vp]),
    "cdv_patchify_blend": (_i32, [_vp, _vp, _vp, _i32, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "cdv_edges_frame": (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _i32, _vp, _vp]),
    "cdv_edges_append": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp]),
    "cdv_edges_workspace_bytes": (_sz, [_i64]),
    "cdv_edges_remove": (_i32, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                _vp, _vp, _vp, _i64, _vp, _vp]),
    "cdv_edges_keyframe_shift": (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, _vp]),
    "cdv_frames_keyframe_shift": (_i32, [_vp, _i32, _i32, _i32, _vp]),
    "cdv_flow_mag": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _f32, _vp, _vp, _vp]),
    "cdv_point_cloud": (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp]),
    "cdv_transform": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "cdv_fastba_reproject": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp]),
    "cdv_graph_workspace_bytes": (_sz, [_i64, _i64]),
    "cdv_graph_build": (_i32, [_vp, _vp, _i64, _vp, _sz, _i64, _i64, _vp]),
    "cdv_graph_build_neighbors": (_i32, [_vp, _vp, _i64, _vp, _sz, _i64, _i64, _vp, _vp, _vp]),
    "cdv_graph_build_edges": (_i32, [_vp, _vp, _vp, _i64, _vp, _sz, _i64, _i64, _vp, _vp, _vp]),
    "cdv_graph_build_table": (_i32, [_vp, _vp, _vp, _i64, _vp, _sz, _i64, _i64, _i64, _vp, _vp, _vp]),
    "cdv_graph_table_offsets": (_i32, [_i64, _i64, _vp]),
    "cdv_update_prologue_table": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _i64, _i64, _i64,
                                         _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _sz, _i64, _i64, _i64, _vp, _vp, _vp]),
    "cdv_update_prologue": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _i64, _i64, _i64,
                                   _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _sz, _i64, _i64, _vp, _vp, _vp]),
    "cdv_graph_read_meta_host": (_i32, [_vp, _vp, _vp]),
    "cdv_graph_corr_order": (_vp, [_vp]),
    "cdv_graph_bind_corr_stream": (_i32, [_vp, _vp, _i64, _i64, _i64, _i64, _f32]),
    "cdv_graph_corr_records": (_vp, [_vp]),
    "cdv_corr_fused_stream": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _f32, _f32,
                                     _i32, _vp]),
    "cdv_graph_get_unique": (_i32, [_vp, _vp, _i64, _vp, _i64, _vp]),
    "cdv_neighbors": (_i32, [_vp, _i64, _vp, _vp, _vp]),
    "cdv_ba_workspace_bytes": (_sz, [_i64, _i64, _i32]),
    "cdv_ba_forward": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp, _vp, _sz,
                              _i64, _vp, _vp]),
    "cdv_ba_status": (_i32, [_vp, _vp, _vp]),
    "cdv_ba_bind_status_counters": (_i32, [_vp, _vp]),
    "cdv_ba_test_handoff": (_i32, [_i32]),
    "cdv_ba_factor_ticket": (_i32, [_i32, _i32, _vp]),
    "cdv_ba_set_patches_per_frame": (_i32, [_vp, _i32]),
    "cdv_lie_op": (_i32, [_i32, _i32, _i32, _i64, _vp, _vp, _vp, _vp]),
    # ---- a frame stream whose sizes live on the device
    "cdv_update_prologue_table_dyn": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _i64, _i64, _vp, _vp, _vp,
                                             _vp, _vp, _vp, _i64, _vp, _vp, _vp, _sz, _i64, _i64, _i64, _vp]),
    "cdv_corr_fused_stream_dyn": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _f32,
                                         _f32, _i32, _vp]),
    "cdv_ba_forward_dyn": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp, _i32, _vp, _vp, _sz,
                                  _i64, _vp]),
    "cdv_stream_workspace_bytes": (_sz, [_i64, _i32]),
    "cdv_stream_frame_begin": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp,
                                      _vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _vp, _vp]),
    "cdv_stream_operator_stub": (_i32, [_vp, _vp, _vp, _i32, _vp, _vp, _f32, _i64, _vp]),
    "cdv_stream_points": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp]),
    "cdv_stream_keyframe": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                   _vp, _vp, _vp, _i64, _i64, _i64, _i32, _i32, _i32, _f32, _i32, _vp, _i32, _vp, _vp, _vp, _vp]),
    "cdv_stream_motion": (_vp, [_vp, _i64, _i32]),
    "cdv_stream_frame": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _vp]),
}

class FrameBuf(ctypes.Structure):
    """cdv_frame_buf (include/cdvslam_hip.h): one per-frame buffer of cdv_frames_keyframe_shift"""
    _fields_ = [("base", ctypes.c_void_p), ("slot_bytes", ctypes.c_int64), ("modulus", ctypes.c_int32),
                ("reserved", ctypes.c_int32)]


MAX_FRAME_BUFS = 16


def _stream_desc_fields():
    P, I, L, F = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_float
    return ([(n, I) for n in ("M", "C", "H", "W", "mem", "pmem", "frames_capacity", "patch_lifetime", "removal_window",
                              "opt_window", "keyframe_index", "n_bufs")]
            + [(n, F) for n in ("keyframe_thresh", "gain", "pose_step")]
            + [("slot", I), ("frames", I), ("cur", I), ("ring_blocks", I), ("fixed_bound", I)]
            + [(n, L) for n in ("edge_capacity", "inactive_capacity", "table_capacity", "graph_E_max", "graph_k_range")]
            + [("graph_ws_bytes", ctypes.c_size_t), ("ba_ws_bytes", ctypes.c_size_t)]
            + [(n, P) for n in ("poses", "patches", "intrinsics", "points", "ix", "fmap1_nhwc", "fmap2_nhwc", "gmap_planar", "gmap_pm")]
            + [(n, P * 2) for n in ("ii", "jj", "kk", "target", "weight")]
            + [(n, P) for n in ("ii_inac", "jj_inac", "kk_inac", "target_inac", "weight_inac", "coords", "corr_out", "lmbda", "dyn",
                                "ws", "graph_ws", "ba_ws", "mirror_host")]
            + [("bufs", FrameBuf * 16)])


class StreamDesc(ctypes.Structure):
    """cdv_stream_desc (include/cdvslam_hip.h): the buffers and sizes of a device-resident frame stream, field for field"""
    _fields_ = _stream_desc_fields()


class ShadowRing(ctypes.Structure):
    """cdv_shadow_ring (include/cdvslam_hip.h): one planar ring and its channels-last shadow for cdv_shadows_sync"""
    _fields_ = [("src_nchw", ctypes.c_void_p), ("dst_nhwc", ctypes.c_void_p), ("ws", ctypes.c_void_p), ("N", ctypes.c_int64),
                ("C", ctypes.c_int32), ("H", ctypes.c_int32), ("W", ctypes.c_int32), ("parity", ctypes.c_int32)]


class PatchifyJob(ctypes.Structure):
    """cdv_patchify_job (include/cdvslam_hip.h): one altcorr.patchify call of cdv_patchify_multi"""
    _fields_ = [("net", ctypes.c_void_p), ("out", ctypes.c_void_p), ("C", ctypes.c_int), ("H", ctypes.c_int),
                ("W", ctypes.c_int), ("radius", ctypes.c_int), ("mode", ctypes.c_int), ("dtype", ctypes.c_int),
                ("sx", ctypes.c_float), ("sy", ctypes.c_float), ("ox", ctypes.c_float), ("oy", ctypes.c_float)]


MAX_PATCHIFY_JOBS = 8

_lib = None


class CdvError(RuntimeError):
    pass


def load():
    """Load the HIP library (once).  Raises ImportError with build instructions if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "cdv_slam_amd: %s not found. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C cdv_slam_amd/csrc`). There is no CPU fallback for the HIP path." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().cdv_last_error()
        raise CdvError("%s failed (rc=%d): %s" % (what, rc, msg.decode() if msg else "?"))
