"""TEST INFRASTRUCTURE: loaders for the benchmark-size fixtures of tests/golden/make_golden.py (`bench-size`)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rebuild_patches(xy, d):
    """[P,2,3] grid values + [P] inverse depth -> patches [P,3,3,3] (x plane, y plane, inverse depth plane)"""
    P = len(d)
    patches = np.empty((P, 3, 3, 3), np.float32)
    patches[:, 0] = xy[:, 0, None, :]
    patches[:, 1] = xy[:, 1, :, None]
    patches[:, 2] = d[:, None, None]
    return patches


def load_ba_pr1():
    """BASELINE configs[0]: inputs of the reference's ba.py run + its outputs after one and two calls"""
    g = dict(np.load(os.path.join(GOLDEN, "ba_py_pr1.npz")))
    n, M = int(g["frames"]), int(g["M"])
    g["patches"] = rebuild_patches(g["patch_xy"], g["patch_d"])
    kk, jj = np.meshgrid(np.arange(n * M), np.arange(n), indexing="ij")     # fully connected: every patch to every frame
    g["kk"], g["jj"] = kk.reshape(-1).astype(np.int64), jj.reshape(-1).astype(np.int64)
    g["ii"] = g["kk"] // M
    return g


def load_pops_small():
    g = dict(np.load(os.path.join(GOLDEN, "pops_small_f32.npz")))
    g["patches"] = rebuild_patches(g["patch_xy"], g["patch_d"])
    return g
