"""oracle/block_e_py.py -- TEST INFRASTRUCTURE ONLY (CPU oracle; nothing under cdv_slam_amd/ may import it).

A numpy restatement of the reference's block-sparse storage of E for the global bundle adjustment,
`EfficentE` (cdvslam/fastba/block_e.cuh:9-25, block_e.cu:38-300), and of the three products the eff_impl branch of
cuda_ba.forward takes from it (ba_cuda.cu:567-580).  The product library does NOT reproduce these index structures (DESIGN.md
section 7: with 288 GB of HBM the dense E stays); this file exists so that "the dense branch and the block-sparse branch give
the same S, y, dZ" is CHECKED on a graph once (tests/test_oracle_golden.py) instead of assumed.  Parity unpinned by the
reference's own tests (there are none for this path; CUDA, unbuildable here): a restatement of the source text.

What the reference assumes and this restatement keeps:
  * patch id k belongs to frame k // ppf, and that frame is the source frame ii of every edge of k (block_e.cu:56-66,
    ba_cuda.cu:382-383 index E_lookup by kx % ppf inside the blocks of frame i = ii);
  * poses with index < t0 are fixed: their rows are dropped when a product is formed, not when E_lookup is filled
    (block_e.cu:171-184, 227-234, 273-282 test j - t0 >= 0; ba_cuda.cu:381-383 fills unconditionally);
  * the row order of index_tensor follows std::unordered_set iteration (block_e.cu:86-124): only the SUM over its rows is
    defined, so the rows are walked in sorted order here.
"""
import numpy as np


class EfficentE:
    def __init__(self, ii, jj, kx, ppf, t0):
        """block_e.cu:45-145.  ii, jj [E] int64; kx = sorted unique patch ids (ba_cuda.cu:476-478)"""
        ii, jj, kx = (np.asarray(a, np.int64) for a in (ii, jj, kx))
        self.ppf, self.t0 = int(ppf), int(t0)
        E = len(ii)
        n_frames = int(max(ii.max(), jj.max())) + 1                                          # :45
        ij_uniq, inv = np.unique(np.concatenate([ii * n_frames + jj, ii * n_frames + ii]), return_inverse=True)   # :46-47
        self.ij_xself = inv.reshape(2, E)                                                    # :50   [0]: block (i, j), [1]: block (i, i)
        self.E_lookup = np.zeros((len(ij_uniq), self.ppf, 6))                                # :51
        self.patch_to_ku = np.full((n_frames, self.ppf), -1, np.int64)                       # :54-63
        self.patch_to_ku[kx // self.ppf, kx % self.ppf] = np.arange(len(kx))
        frame_to_idx = np.full((n_frames, n_frames), -1, np.int64)                           # :68-84
        frame_to_idx[ii, jj] = self.ij_xself[0]
        frame_to_idx[ii, ii] = self.ij_xself[1]
        rows = []                                                                            # :86-124 (sorted instead of hash order)
        for i in range(n_frames):
            conn = np.unique(np.concatenate([jj[ii == i], [i] if (ii == i).any() else []]).astype(np.int64))
            for j1 in conn:
                for j2 in conn:
                    rows.append((i, j1, j2, frame_to_idx[i, j1], frame_to_idx[i, j2]))
        self.index_tensor = np.asarray(rows, np.int64).reshape(-1, 5)
        self.block_index_tensor = np.stack([ij_uniq // n_frames, ij_uniq % n_frames], 1)     # :128-144

    def fill(self, kk, w, Jz, Ji, Jj):
        """ba_cuda.cu:380-383 (eff_impl): E_lookup[ijs][kx % ppf] -= w Jz Ji, E_lookup[ijx][kx % ppf] += w Jz Jj, both
        residual rows of every edge; kk = the edges' patch ids (kx[k] of the kernel)"""
        self.E_lookup[:] = 0                                                                  # ba_cuda.cu:519
        kk = np.asarray(kk, np.int64)
        wz = w * Jz                                                                           # [E, 2]
        np.add.at(self.E_lookup, (self.ij_xself[1], kk % self.ppf), -(wz[:, :, None] * Ji).sum(1))
        np.add.at(self.E_lookup, (self.ij_xself[0], kk % self.ppf), (wz[:, :, None] * Jj).sum(1))

    def _q(self, vec, i):
        """vec[patch_to_ku[i][k]] for the ppf patch slots of frame i; a slot without a patch (index -1: the reference reads out
        of bounds there, its E_lookup slice is zero) contributes nothing"""
        idx = self.patch_to_ku[i]
        return np.where(idx >= 0, np.asarray(vec).reshape(-1)[np.maximum(idx, 0)], 0.0)

    def computeEQEt(self, N, Q):
        """block_e.cu:147-202"""
        out = np.zeros((6 * N, 6 * N))
        for i, j1, j2, a, b in self.index_tensor:
            r1, r2 = j1 - self.t0, j2 - self.t0
            if r1 < 0 or r2 < 0:
                continue
            q = self._q(Q, i)
            out[6 * r1:6 * r1 + 6, 6 * r2:6 * r2 + 6] += np.einsum("kx,ky,k->xy", self.E_lookup[a], self.E_lookup[b], q)
        return out

    def computeEv(self, N, vec):
        """block_e.cu:204-252"""
        out = np.zeros(6 * N)
        for idx, (i, j) in enumerate(self.block_index_tensor):
            r = j - self.t0
            if r >= 0:
                out[6 * r:6 * r + 6] += (self.E_lookup[idx] * self._q(vec, i)[:, None]).sum(0)
        return out

    def computeEtv(self, M, vec):
        """block_e.cu:254-299"""
        out = np.zeros(M)
        vec = np.asarray(vec).reshape(-1)
        for idx, (i, j) in enumerate(self.block_index_tensor):
            r = j - self.t0
            if r < 0:
                continue
            ku = self.patch_to_ku[i]
            dp = self.E_lookup[idx] @ vec[6 * r:6 * r + 6]
            np.add.at(out, ku[ku >= 0], dp[ku >= 0])
        return out


def solve_eff_impl(B, v, C, u, lmbda, blockE, N):
    """the eff_impl branch of the Schur solve, ba_cuda.cu:548,567-580: returns (S damped, y, dX, dZ)"""
    Q = 1.0 / (C + lmbda)
    S = B - blockE.computeEQEt(N, Q)
    y = v - blockE.computeEv(N, Q * u)
    S = S + np.eye(6 * N) * (1e-4 * S + 1.0)
    L = np.linalg.cholesky(S)
    dX = np.linalg.solve(L.T, np.linalg.solve(L, y))
    dZ = Q * (u - blockE.computeEtv(len(C), dX))
    return S, y, dX, dZ
