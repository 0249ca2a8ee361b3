"""Synthetic TartanAir-shaped patch-graph states for tests and bench.py.

Reproduces the *harness slice* of the reference front-end -- the edge bookkeeping of
``SLAM.__edges_forw/__edges_back`` (cdvslam/slam.py:528-541), ``append_factors``
(slam.py:331-337) and the REMOVAL_WINDOW culling in ``keyframe`` (slam.py:453-458, no
keyframe drops) -- and fills the state buffers with the seeded synthetic data specified in
BASELINE.md section 2 (512x384 stream, seed 1234).  Pure numpy, no GPU, no oracle.
"""
from dataclasses import dataclass, field

import numpy as np


@dataclass
class GraphConfig:
    name: str = "default"
    frames: int = 40            # frames replayed (n at the time of the update)
    M: int = 96                 # PATCHES_PER_FRAME
    patch_lifetime: int = 13    # PATCH_LIFETIME
    removal_window: int = 22    # REMOVAL_WINDOW
    opt_window: int = 10        # OPTIMIZATION_WINDOW
    ht: int = 384
    wd: int = 512
    res: int = 4                # RES
    C: int = 24                 # DIMF
    mem: int = 36               # fmap ring slots (slam.py:65)
    pmem: int = 36              # patch ring slots
    buffer_size: int = 64       # BUFFER_SIZE (4096 in the reference; only rows < frames are touched)
    fully_connected: bool = False  # PR1 graph: every patch to every frame (incl. self)
    seed: int = 1234


CONFIGS = {
    # BASELINE.json configs[0]: 10 frames x 96 patches fully connected, first pose fixed
    "pr1": GraphConfig(name="pr1", frames=10, fully_connected=True, buffer_size=16),
    # initialisation: n == 8, t0 = 1 (slam.py:712-716)
    "init": GraphConfig(name="init", frames=8, buffer_size=16),
    # default steady state: E = 47,712, U = 2,208, N = 10
    "default": GraphConfig(name="default", frames=40),
    # stress: --opts PATCHES_PER_FRAME 196 OPTIMIZATION_WINDOW 22 -> E = 97,412, U = 4,508, N = 22
    "stress": GraphConfig(name="stress", frames=40, M=196, opt_window=22),
    # small graphs for fast CPU tests
    "tiny": GraphConfig(name="tiny", frames=6, M=8, ht=128, wd=160, mem=8, pmem=8, buffer_size=8),
    "small": GraphConfig(name="small", frames=30, M=16, ht=192, wd=256, buffer_size=40),
    # global bundle adjustment (slam.py:460-478): inactive edges kept, every pose but the first free -> N = 79, E = 30k
    "global": GraphConfig(name="global", frames=80, M=16, ht=192, wd=256, buffer_size=96, removal_window=10 ** 6,
                          opt_window=10 ** 6),
    # the same with three pose panels' worth of free poses in a wider graph: N = 139
    "global_l": GraphConfig(name="global_l", frames=140, M=12, ht=192, wd=256, buffer_size=160, removal_window=10 ** 6,
                            opt_window=10 ** 6),
    # the global BA at the reference's scale (slam.py:460-478 with MAX_EDGE_AGE = 1000): N = 299 free poses, full-size
    # frames, 96 patches per frame -> U = 28,800 patches, E = 0.7 M edges
    "global_xl": GraphConfig(name="global_xl", frames=300, buffer_size=304, removal_window=10 ** 6, opt_window=10 ** 6),
}


def replay_edges(cfg: GraphConfig):
    """Edge lists (ii, jj, kk) as they stand when update() runs at n == cfg.frames."""
    M, r = cfg.M, cfg.patch_lifetime
    if cfg.fully_connected:
        kk, jj = np.meshgrid(np.arange(cfg.frames * M), np.arange(cfg.frames), indexing="ij")
        kk, jj = kk.reshape(-1), jj.reshape(-1)
        return (kk // M).astype(np.int64), jj.astype(np.int64), kk.astype(np.int64)
    jj = np.zeros(0, np.int64)
    kk = np.zeros(0, np.int64)
    for n in range(1, cfg.frames + 1):
        # forward edges: patches of frames [n-r, n-1) -> newest frame n-1   (slam.py:528-534)
        fk = np.arange(M * max(n - r, 0), M * max(n - 1, 0), dtype=np.int64)
        fj = np.full_like(fk, n - 1)
        # backward edges: patches of newest frame -> frames [n-r, n), patch-outer / frame-inner
        bk, bj = np.meshgrid(np.arange(M * (n - 1), M * n, dtype=np.int64),
                             np.arange(max(n - r, 0), n, dtype=np.int64), indexing="ij")
        kk = np.concatenate([kk, fk, bk.reshape(-1)])
        jj = np.concatenate([jj, fj, bj.reshape(-1)])
        if n < cfg.frames and n >= 8:
            # keyframe(): drop edges whose source frame left the removal window (slam.py:453-458)
            keep = (kk // M) >= n - cfg.removal_window
            kk, jj = kk[keep], jj[keep]
    return (kk // M).astype(np.int64), jj, kk


# -- minimal float64 SE3 helpers (host-side data generation only) ---------------------------

def _quat_mul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by,
                     aw * by + ay * bw + az * bx - ax * bz,
                     aw * bz + az * bw + ax * by - ay * bx,
                     aw * bw - ax * bx - ay * by - az * bz])


def _quat_rot(q, p):
    uv = 2.0 * np.cross(q[:3], p)
    return p + q[3] * uv + np.cross(q[:3], uv)


def _se3_exp(xi):
    tau, phi = xi[:3], xi[3:]
    th = np.linalg.norm(phi)
    if th < 1e-8:
        q = np.array([*(0.5 * phi), 1.0])
        V = np.eye(3)
    else:
        q = np.array([*(np.sin(th / 2) / th * phi), np.cos(th / 2)])
        K = np.array([[0, -phi[2], phi[1]], [phi[2], 0, -phi[0]], [-phi[1], phi[0], 0]])
        V = np.eye(3) + (1 - np.cos(th)) / th**2 * K + (th - np.sin(th)) / th**3 * K @ K
    return V @ tau, q / np.linalg.norm(q)


def _vq_mul(a, b):
    ax, ay, az, aw = a[:, 0], a[:, 1], a[:, 2], a[:, 3]
    bx, by, bz, bw = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    return np.stack([aw * bx + ax * bw + ay * bz - az * by,
                     aw * by + ay * bw + az * bx - ax * bz,
                     aw * bz + az * bw + ax * by - ay * bx,
                     aw * bw - ax * bx - ay * by - az * bz], -1)


def _vq_rot(q, p):
    uv = 2.0 * np.cross(q[:, :3], p)
    return p + q[:, 3:4] * uv + np.cross(q[:, :3], uv)


def _reproject_centres(poses, centres, depth, intr, ii, jj, kk):
    """float64 pinhole reprojection of patch centres (data generation only)."""
    fx, fy, cx, cy = intr
    ti, qi = poses[ii, :3], poses[ii, 3:]
    tj, qj = poses[jj, :3], poses[jj, 3:]
    qic = qi * np.array([-1.0, -1.0, -1.0, 1.0])
    qij = _vq_mul(qj, qic)
    tij = tj - _vq_rot(qij, ti)
    X = np.stack([(centres[kk, 0] - cx) / fx, (centres[kk, 1] - cy) / fy, np.ones(len(kk))], -1)
    Y = _vq_rot(qij, X) + depth[kk, None] * tij
    z = np.maximum(Y[:, 2], 0.1)
    return np.stack([fx * Y[:, 0] / z + cx, fy * Y[:, 1] / z + cy], -1)


@dataclass
class SynthState:
    cfg: GraphConfig
    n: int
    t0: int
    ii: np.ndarray
    jj: np.ndarray
    kk: np.ndarray
    poses: np.ndarray        # [buffer_size, 7] f32  (tx,ty,tz,qx,qy,qz,qw)
    patches: np.ndarray      # [buffer_size*M, 3, 3, 3] f32
    intrinsics: np.ndarray   # [buffer_size, 4] f32 (already / RES)
    target: np.ndarray       # [E, 2] f32
    weight: np.ndarray       # [E, 2] f32
    lmbda: float = 1e-4
    fmap1: np.ndarray = None  # [mem, C, h, w] f16
    fmap2: np.ndarray = None  # [mem, C, h/4, w/4] f16
    gmap: np.ndarray = None   # [pmem*M, C, 3, 3] f16
    extra: dict = field(default_factory=dict)

    @property
    def E(self):
        return len(self.ii)

    @property
    def ii1(self):
        """patch ring index used by SLAM.corr (slam.py:319)"""
        return self.kk % (self.cfg.M * self.cfg.pmem)

    @property
    def jj1(self):
        return self.jj % self.cfg.mem


def make_state(cfg="default", features=True, **overrides) -> SynthState:
    """Build the seeded synthetic state for a named config."""
    if isinstance(cfg, str):
        cfg = CONFIGS[cfg]
    if overrides:
        cfg = GraphConfig(**{**cfg.__dict__, **overrides})
    rng = np.random.default_rng(cfg.seed)
    M, n = cfg.M, cfg.frames
    h, w = cfg.ht // cfg.res, cfg.wd // cfg.res
    ii, jj, kk = replay_edges(cfg)

    # poses: T_0 = I, T_t = Exp(xi_t) T_{t-1}
    poses = np.zeros((cfg.buffer_size, 7))
    poses[:, 6] = 1.0
    t, q = np.zeros(3), np.array([0.0, 0.0, 0.0, 1.0])
    for f in range(1, n):
        xi = np.concatenate([rng.normal(0, 0.03, 3) + np.array([0.05, 0, 0]), rng.normal(0, 0.01, 3)])
        dt, dq = _se3_exp(xi)
        t = _quat_rot(dq, t) + dt
        q = _quat_mul(dq, q)
        q /= np.linalg.norm(q)
        poses[f, :3], poses[f, 3:] = t, q

    intr = np.array([cfg.wd / 2.0, cfg.wd / 2.0, cfg.wd / 2.0, cfg.ht / 2.0]) / cfg.res
    intrinsics = np.tile(intr, (cfg.buffer_size, 1))

    centres = np.stack([rng.uniform(8, w - 8, cfg.buffer_size * M), rng.uniform(8, h - 8, cfg.buffer_size * M)], -1)
    depth = rng.uniform(0.25, 1.0, cfg.buffer_size * M)
    off = np.array([-1.0, 0.0, 1.0])
    patches = np.empty((cfg.buffer_size * M, 3, 3, 3))
    patches[:, 0] = centres[:, 0, None, None] + off[None, None, :]
    patches[:, 1] = centres[:, 1, None, None] + off[None, :, None]
    patches[:, 2] = depth[:, None, None]

    t0 = max(n - cfg.opt_window, 1)
    if cfg.name == "init":
        t0 = 1
    if cfg.fully_connected:
        t0 = 1  # fixedp = 1

    centre_proj = _reproject_centres(poses, centres, depth, intr, ii, jj, kk)
    target = centre_proj + rng.normal(0, 1.0, centre_proj.shape)
    weight = rng.uniform(0, 1, centre_proj.shape)

    st = SynthState(cfg=cfg, n=n, t0=t0, ii=ii, jj=jj, kk=kk,
                    poses=poses.astype(np.float32), patches=patches.astype(np.float32),
                    intrinsics=intrinsics.astype(np.float32), target=target.astype(np.float32),
                    weight=weight.astype(np.float32))
    if features:
        st.fmap1 = (rng.standard_normal((cfg.mem, cfg.C, h, w), dtype=np.float32) / 4).astype(np.float16)
        f32 = st.fmap1.astype(np.float32)
        st.fmap2 = f32.reshape(cfg.mem, cfg.C, h // 4, 4, w // 4, 4).mean(axis=(3, 5)).astype(np.float16)
        # gmap: 3x3 feature tiles of each patch's own frame around the patch centre (nearest sample)
        gm = np.zeros((cfg.pmem * M, cfg.C, 3, 3), np.float16)
        first = max(0, n - cfg.pmem)
        for f in range(first, n):
            ks = np.arange(f * M, (f + 1) * M)
            cx = np.clip(np.floor(centres[ks, 0]).astype(int), 1, w - 2)
            cy = np.clip(np.floor(centres[ks, 1]).astype(int), 1, h - 2)
            for a in range(3):
                for b in range(3):
                    gm[ks % (M * cfg.pmem), :, a, b] = st.fmap1[f % cfg.mem][:, cy + a - 1, cx + b - 1].T
        st.gmap = gm
    return st
