"""oracle/oracle.py -- TEST INFRASTRUCTURE ONLY.

numpy/ctypes front-end of the CPU oracle (oracle/cdv_oracle.c).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
package (cdv_slam_amd/) never does.

Every function mirrors one operator of the reference's update hot path; the C file's header
lists the reference file:line each one restates and the parity-pinning status.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libcdv_oracle.so")

OPS = {"exp": 0, "log": 1, "inv": 2, "mul": 3, "adj": 4, "adjT": 5, "act": 6, "act4": 7, "matrix": 8}
SO3, SE3 = 1, 3


def build(force=False):
    """Compile the oracle with gcc (seconds)."""
    src = [os.path.join(_HERE, f) for f in ("cdv_oracle.c", "lie_impl.h", "fastba_impl.h")]
    if (not force and os.path.exists(_LIB_PATH)
            and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in src)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libcdv_oracle.so"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.orc_unique.restype = ctypes.c_long
    return _lib


def num_threads():
    """threads used by the edge-parallel oracle loops (corr, transform); BA assembly/solve are serial"""
    return int(lib().orc_num_threads())


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def _suffix(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return "f32"
    if dtype == np.float64:
        return "f64"
    raise TypeError(dtype)


# ------------------------------------------------------------------------------------------
# Lie groups (lietorch semantics: cdvslam/lietorch/include/so3.h, se3.h)
# ------------------------------------------------------------------------------------------

def lie(group, op, x, y=None, dtype=np.float32):
    """Batched forward op on flat [n, dim] rows (lietorch_backends.<op> semantics)."""
    N, K = (7, 6) if group == SE3 else (4, 3)
    x = _c(x, dtype)
    n = x.shape[0]
    out_dim = {"exp": N, "log": K, "inv": N, "mul": N, "adj": K, "adjT": K, "act": 3, "act4": 4, "matrix": 16}[op]
    z = np.empty((n, out_dim), dtype=dtype)
    y = None if y is None else _c(y, dtype)
    fn = getattr(lib(), "orc_lie_" + _suffix(dtype))
    rc = fn(ctypes.c_int(group), ctypes.c_int(OPS[op]), ctypes.c_long(n), _p(x), _p(y), _p(z))
    if rc != 0:
        raise RuntimeError("orc_lie failed rc=%d" % rc)
    if op == "matrix":
        z = z.reshape(n, 4, 4)
    return z


# ------------------------------------------------------------------------------------------
# projective ops (cdvslam/projective_ops.py:53-113)
# ------------------------------------------------------------------------------------------

def transform(poses, patches, intrinsics, ii, jj, kk, jacobian=False, valid=False, tonly=False,
              dtype=np.float32):
    """poses [n,7], patches [m,3,P,P], intrinsics [n,4] -> coords [E,P,P,2] (+ extras)."""
    poses, patches, intrinsics = _c(poses, dtype), _c(patches, dtype), _c(intrinsics, dtype)
    ii, jj, kk = _c(ii, np.int64), _c(jj, np.int64), _c(kk, np.int64)
    E, P = len(ii), patches.shape[-1]
    coords = np.empty((E, P, P, 2), dtype=dtype)
    validpx = np.empty((E, P, P), dtype=dtype) if valid else None
    v = Ji = Jj = Jz = None
    if jacobian:
        v = np.empty((E,), dtype=dtype)
        Ji = np.empty((E, 2, 6), dtype=dtype)
        Jj = np.empty((E, 2, 6), dtype=dtype)
        Jz = np.empty((E, 2, 1), dtype=dtype)
    fn = getattr(lib(), "orc_transform_" + _suffix(dtype))
    fn(_p(poses), _p(patches), _p(intrinsics), _p(ii), _p(jj), _p(kk), ctypes.c_long(E), ctypes.c_int(P),
       ctypes.c_int(int(tonly)), _p(coords), _p(validpx), _p(v), _p(Ji), _p(Jj), _p(Jz))
    if jacobian:
        return coords, v, (Ji, Jj, Jz)
    if valid:
        return coords, validpx
    return coords


def fastba_reproject(poses, patches, intrinsics, ii, jj, kk, dtype=np.float32):
    """fastba `reproject` kernel (ba_cuda.cu:408-458) -> [E,2,P,P]."""
    poses, patches, intrinsics = _c(poses, dtype), _c(patches, dtype), _c(intrinsics, dtype)
    ii, jj, kk = _c(ii, np.int64), _c(jj, np.int64), _c(kk, np.int64)
    E, P = len(ii), patches.shape[-1]
    coords = np.empty((E, 2, P, P), dtype=dtype)
    fn = getattr(lib(), "orc_fastba_reproject_" + _suffix(dtype))
    fn(_p(poses), _p(patches), _p(intrinsics), _p(ii), _p(jj), _p(kk), ctypes.c_long(E), ctypes.c_int(P), _p(coords))
    return coords


# ------------------------------------------------------------------------------------------
# index bookkeeping
# ------------------------------------------------------------------------------------------

def unique(x):
    """torch._unique(x, sorted=True, return_inverse=True) -> (values, inverse)."""
    x = _c(x, np.int64)
    u = np.empty_like(x)
    inv = np.empty_like(x)
    U = lib().orc_unique(ctypes.c_long(len(x)), _p(x), _p(u), _p(inv))
    return u[:U].copy(), inv


def neighbors(ii, jj):
    """fastba.neighbors (ba.cpp:59-97): ii = patch ids, jj = frame ids -> (ix, jx)."""
    ii, jj = _c(ii, np.int64), _c(jj, np.int64)
    ix = np.empty_like(ii)
    jx = np.empty_like(ii)
    lib().orc_neighbors(ctypes.c_long(len(ii)), _p(ii), _p(jj), _p(ix), _p(jx))
    return ix, jx


# ------------------------------------------------------------------------------------------
# fastba BA (ba_cuda.cu:462-611, dense-E path)
# ------------------------------------------------------------------------------------------

def fastba(poses, patches, intrinsics, target, weight, lmbda, ii, jj, kk, t0, t1, iterations=2,
           dtype=np.float32, debug=False):
    """Returns (poses_new [n,7], patches_new [m,3,P,P], info[, dbg dict of iteration-0 B,E,C,v,u,S,y,dX,dZ])."""
    poses = _c(poses, dtype).reshape(-1, 7).copy()
    P = patches.shape[-1]
    patches = _c(patches, dtype).reshape(-1, 3, P, P).copy()
    intrinsics = _c(intrinsics, dtype).reshape(-1, 4)
    target = _c(target, dtype).reshape(-1, 2)
    weight = _c(weight, dtype).reshape(-1, 2)
    ii, jj, kk = _c(ii, np.int64), _c(jj, np.int64), _c(kk, np.int64)
    kx, ku = unique(kk)
    E, U, N = len(ii), len(kx), t1 - t0
    n6 = 6 * N
    dbg = None
    if debug:
        dbg = np.zeros(2 * n6 * n6 + n6 * U + 2 * U + 3 * n6 + U + 8, dtype=dtype)
    fn = getattr(lib(), "orc_fastba_" + _suffix(dtype))
    fn.restype = ctypes.c_int
    lm = ctypes.c_float(float(lmbda)) if np.dtype(dtype) == np.float32 else ctypes.c_double(float(lmbda))
    info = fn(_p(poses), _p(patches), _p(intrinsics), _p(target), _p(weight), lm, _p(ii), _p(jj), _p(kk),
              _p(kx), _p(ku), ctypes.c_long(E), ctypes.c_long(U), ctypes.c_int(P), ctypes.c_int(t0),
              ctypes.c_int(t1), ctypes.c_int(iterations), _p(dbg))
    if not debug:
        return poses, patches, info
    o = 0
    out = {"kx": kx, "ku": ku}
    for name, size, shape in (("B", n6 * n6, (n6, n6)), ("E", n6 * U, (n6, U)), ("C", U, (U,)), ("v", n6, (n6,)),
                              ("u", U, (U,)), ("S", n6 * n6, (n6, n6)), ("y", n6, (n6,)), ("dX", n6, (N, 6)),
                              ("dZ", U, (U,))):
        out[name] = dbg[o:o + size].reshape(shape).copy()
        o += size
    return poses, patches, info, out


def fastba_edges(poses, patches, intrinsics, target, weight, ii, jj, kk, dtype=np.float64):
    """the per-edge factor of ba_cuda.cu:262-342 on its own: (r [E,2], w [E,2] masked, Jz [E,2], Ji [E,2,6], Jj [E,2,6])"""
    poses = _c(poses, dtype).reshape(-1, 7)
    P = patches.shape[-1]
    patches = _c(patches, dtype).reshape(-1, 3, P, P)
    intrinsics = _c(intrinsics, dtype).reshape(-1, 4)
    target, weight = _c(target, dtype).reshape(-1, 2), _c(weight, dtype).reshape(-1, 2)
    ii, jj, kk = _c(ii, np.int64), _c(jj, np.int64), _c(kk, np.int64)
    E = len(ii)
    r, w, Jz = (np.zeros((E, 2), dtype) for _ in range(3))
    Ji, Jj = np.zeros((E, 2, 6), dtype), np.zeros((E, 2, 6), dtype)
    getattr(lib(), "orc_fastba_edges_" + _suffix(dtype))(_p(poses), _p(patches), _p(intrinsics), _p(target), _p(weight), _p(ii),
                                                         _p(jj), _p(kk), ctypes.c_long(E), ctypes.c_int(P), _p(r), _p(w), _p(Jz),
                                                         _p(Ji), _p(Jj))
    return r, w, Jz, Ji, Jj


# ------------------------------------------------------------------------------------------
# altcorr (correlation_kernel.cu, correlation.py)
# ------------------------------------------------------------------------------------------

def corr(fmap1, fmap2, coords, us, vs, radius, mode="ref"):
    """altcorr.corr on one level.

    fmap1 [N1,C,P,P], fmap2 [N2,C,H2,W2], coords [M,2,P,P] float32, us/vs [M] int64.
    mode "ref"  : float16 in, reference-faithful half arithmetic, float16 out
         "f32"  : float32 in/out (MIXED_PRECISION False path)
         "truth": float16 or float32 in, float64 accumulate/out
    Returns [M, D-1 (x), D-1 (y), P, P]   (the permuted layout the reference returns).
    """
    M = coords.shape[0]
    H, W = coords.shape[2], coords.shape[3]
    C, H2, W2 = fmap2.shape[1], fmap2.shape[2], fmap2.shape[3]
    D1 = 2 * radius + 1
    coords = _c(coords, np.float32)
    us, vs = _c(us, np.int64), _c(vs, np.int64)
    if mode == "ref":
        f1, f2 = _c(fmap1, np.float16), _c(fmap2, np.float16)
        out = np.empty((M, D1, D1, H, W), dtype=np.float16)
        m = 0
    elif mode == "f32":
        f1, f2 = _c(fmap1, np.float32), _c(fmap2, np.float32)
        out = np.empty((M, D1, D1, H, W), dtype=np.float32)
        m = 1
    elif mode == "truth":
        if fmap2.dtype == np.float16:
            f1, f2 = _c(fmap1, np.float16), _c(fmap2, np.float16)
            m = 2
        else:
            f1, f2 = _c(fmap1, np.float32), _c(fmap2, np.float32)
            m = 3
        out = np.empty((M, D1, D1, H, W), dtype=np.float64)
    else:
        raise ValueError(mode)
    assert f1.shape[1] == C and f1.shape[2] == H and f1.shape[3] == W
    lib().orc_corr(ctypes.c_int(m), _p(f1), _p(f2), _p(coords), _p(us), _p(vs), ctypes.c_long(M), ctypes.c_int(C),
                   ctypes.c_int(H), ctypes.c_int(W), ctypes.c_int(H2), ctypes.c_int(W2), ctypes.c_int(radius), _p(out))
    return out


def slam_corr(gmap, fmap1, fmap2, coords, ii1, jj1, radius=3, mode="ref"):
    """SLAM.corr (cdvslam/slam.py:316-323): two pyramid levels stacked -> [E, 2*(2r+1)^2*P*P]."""
    c1 = corr(gmap, fmap1, coords / np.float32(1), ii1, jj1, radius, mode)
    c2 = corr(gmap, fmap2, coords / np.float32(4), ii1, jj1, radius, mode)
    return np.stack([c1, c2], -1).reshape(coords.shape[0], -1)


def patchify_raw(net, coords, radius):
    """cuda_corr.patchify_forward: net [C,H,W], coords [M,2] -> [M,C,D,D]."""
    net = np.ascontiguousarray(net)
    coords = _c(coords, np.float32)
    C, H, W = net.shape
    M, D = coords.shape[0], 2 * radius + 2
    out = np.empty((M, C, D, D), dtype=net.dtype)
    lib().orc_patchify(ctypes.c_int(net.dtype.itemsize), _p(net), _p(coords), ctypes.c_long(M), ctypes.c_int(C),
                       ctypes.c_int(H), ctypes.c_int(W), ctypes.c_int(radius), _p(out))
    return out


def patchify(net, coords, radius, mode="bilinear"):
    """altcorr.patchify (cdvslam/altcorr/correlation.py:51-71).  The blend runs in numpy in the
    promoted dtype of (offset float32, patches), as torch type promotion does."""
    patches = patchify_raw(net, coords, radius)
    if mode == "bilinear":
        coords = _c(coords, np.float32)
        off = coords - np.floor(coords)
        dx = off[:, 0][:, None, None, None]
        dy = off[:, 1][:, None, None, None]
        d = 2 * radius + 1
        p = patches.astype(np.float32)
        x00 = (1 - dy) * (1 - dx) * p[..., :d, :d]
        x01 = (1 - dy) * (dx) * p[..., :d, 1:]
        x10 = (dy) * (1 - dx) * p[..., 1:, :d]
        x11 = (dy) * (dx) * p[..., 1:, 1:]
        return x00 + x01 + x10 + x11
    if mode == "upperleft":
        return patches[..., :1, :1]
    return patches
