"""Proximity loop-closure edges for the global bundle adjustment: PatchGraph.edges_loop (cdvslam/patchgraph.py:71-97)
and its greedy selection reduce_edges (cdvslam/loop_closure/optim_utils.py:23-60, a numba loop in the reference).

The candidate search -- every recent frame against every old patch, ~10^6 reprojections -- is one HIP launch
(cdv_loop_flow: one wave per (target frame, source frame) pair); the selection over the few thousand surviving pairs is
sequential by construction (greedy non-maximum suppression in order of flow) and runs on the host, as in the reference.
"""
import numpy as np
import torch

from . import _lib
from .ops import _need_cuda, _p, _stream


def loop_flow(poses, patches, intrinsics, ix, M, j0, nj, f0, nf, beta=0.5):
    """mean flow of the M patch centres of frames f0 .. f0+nf-1 in frames j0 .. j0+nj-1 -> [nj, nf] f32 (inf: too few valid)"""
    lib = _lib.load()
    _need_cuda(poses, patches, intrinsics, ix)
    poses, patches, intrinsics, ix = poses.contiguous(), patches.contiguous(), intrinsics.contiguous(), ix.contiguous()
    P = patches.shape[-1]
    out = torch.empty((nj, nf), dtype=torch.float32, device=poses.device)
    rc = lib.cdv_loop_flow(_p(poses), _p(patches), _p(intrinsics), _p(ix), int(M), int(P), int(j0), int(nj), int(f0),
                           int(nf), float(beta), _p(out), _stream())
    _lib.check(rc, "cdv_loop_flow")
    return out


def reduce_edges(flow_mag, ii, jj, max_num_edges, nms):
    """optim_utils.py:23-60: candidates in order of increasing flow; skip pairs less than 30 frames apart, infinite
    flow, or suppressed ones; a chosen (i, j) suppresses (i - nms .. i + nms, j); stop after max_num_edges + 1 picks
    (the reference compares len(es), which holds a leading sentinel, with max_num_edges)."""
    flow_mag, ii, jj = np.asarray(flow_mag), np.asarray(ii), np.asarray(jj)
    es = []
    if ii.size == 0:
        return np.zeros((0, 2), dtype=np.int64)
    Ni, Nj = int(ii.max()) + 1, int(jj.max()) + 1
    ignore = np.zeros((Ni, Nj), dtype=bool)
    for idx in np.argsort(flow_mag):
        if len(es) + 1 > max_num_edges:
            break
        i, j = int(ii[idx]), int(jj[idx])
        if (j - i) < 30 or flow_mag[idx] >= 1000 or ignore[i, j]:
            continue
        es.append((i, j))
        ignore[max(i - nms, 0):min(i + nms, Ni - 1) + 1, j] = True
    return np.asarray(es, dtype=np.int64).reshape((-1, 2))


def edges_loop(poses, patches, intrinsics, ix, n, M, removal_window=22, max_edge_age=1000, global_opt_freq=15,
               keyframe_index=4, backend_thresh=64.0, max_num_edges=1000, nms=1):
    """PatchGraph.edges_loop (patchgraph.py:71-97): edges from old patches to new frames -> (kk, jj) int64 on the device.
    poses [N,7], patches [N*M,3,P,P], intrinsics [N,4], ix [N*M] (frame of a patch), n frames so far."""
    dev = poses.device
    l = n - removal_window                      # upper bound for "old" patches
    empty = torch.empty(0, dtype=torch.int64, device=dev)
    if l <= 0:
        return empty, empty
    j0, j1 = max(n - global_opt_freq, 0), n - keyframe_index
    f0 = max(l - max_edge_age, 0)
    nj, nf = j1 - j0, l - f0
    if nj <= 0 or nf <= 0:
        return empty, empty
    flow = loop_flow(poses, patches, intrinsics, ix, M, j0, nj, f0, nf, beta=0.5)
    mask = flow < backend_thresh                # [nj, nf], candidates in (j, f) order like flatmeshgrid(jj, kk, 'ij')
    fm = flow[mask].cpu().numpy()
    jl, fl = torch.nonzero(mask, as_tuple=True)
    i_frames = ix[(f0 + fl) * M].cpu().numpy()  # ii[::M][mask]
    j_frames = (j0 + jl).cpu().numpy()
    es = reduce_edges(fm, i_frames, j_frames, max_num_edges=max_num_edges, nms=nms)
    if len(es) == 0:
        return empty, empty
    edges = torch.as_tensor(es, device=dev)
    ii = edges[:, 0:1].repeat(1, M)
    jj = edges[:, 1:2].repeat(1, M)
    kk = ii * M + torch.arange(M, device=dev)
    return kk.flatten(), jj.flatten()
