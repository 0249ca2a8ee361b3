"""oracle/edges_py.py -- TEST INFRASTRUCTURE ONLY.

numpy restatement of the reference's edge bookkeeping, line by line: append_factors (cdvslam/slam.py:331-337),
remove_factors (:339-354), keyframe() (:408-458, edge part), __edges_forw / __edges_back (:528-541).
parity unpinned by the reference (it has no test for these); for a stream without keyframe drops it reproduces
cdv_slam_amd.synth.replay_edges, the replay SURVEY.md 8 quotes E = 47,712 from."""
import numpy as np


def flatmeshgrid(a, b):
    A, B = np.meshgrid(a, b, indexing="ij")
    return A.reshape(-1), B.reshape(-1)


class EdgesPy:
    def __init__(self):
        self.ii = np.zeros(0, np.int64); self.jj = np.zeros(0, np.int64); self.kk = np.zeros(0, np.int64)
        self.target = np.zeros((0, 2), np.float32); self.weight = np.zeros((0, 2), np.float32)
        self.ii_inac = np.zeros(0, np.int64); self.jj_inac = np.zeros(0, np.int64); self.kk_inac = np.zeros(0, np.int64)
        self.target_inac = np.zeros((0, 2), np.float32); self.weight_inac = np.zeros((0, 2), np.float32)

    def append_factors(self, ii, jj, ix):                       # slam.py:331-337
        self.jj = np.concatenate([self.jj, jj]); self.kk = np.concatenate([self.kk, ii])
        self.ii = np.concatenate([self.ii, ix[ii]])
        z = np.zeros((len(ii), 2), np.float32)
        self.target = np.concatenate([self.target, z]); self.weight = np.concatenate([self.weight, z])

    def edges_forw(self, n, M, r):                              # slam.py:528-534
        return flatmeshgrid(np.arange(M * max(n - r, 0), M * max(n - 1, 0)), np.arange(n - 1, n))

    def edges_back(self, n, M, r):                              # slam.py:536-541
        return flatmeshgrid(np.arange(M * max(n - 1, 0), M * max(n, 0)), np.arange(max(n - r, 0), n))

    def remove_factors(self, m, store):                         # slam.py:339-354
        if store:
            self.ii_inac = np.concatenate([self.ii_inac, self.ii[m]]); self.jj_inac = np.concatenate([self.jj_inac, self.jj[m]])
            self.kk_inac = np.concatenate([self.kk_inac, self.kk[m]])
            self.weight_inac = np.concatenate([self.weight_inac, self.weight[m]])
            self.target_inac = np.concatenate([self.target_inac, self.target[m]])
        self.weight, self.target = self.weight[~m], self.target[~m]
        self.ii, self.jj, self.kk = self.ii[~m], self.jj[~m], self.kk[~m]

    def keyframe(self, k, n, M, ix, removal_window, drop, loop_closure=False, opt_window=10):   # slam.py:408-458
        if drop:
            self.remove_factors((self.ii == k) | (self.jj == k), store=False)
            self.kk[self.ii > k] -= M
            self.ii[self.ii > k] -= 1
            self.jj[self.jj > k] -= 1
            n -= 1
        to_remove = ix[self.kk] < n - removal_window
        if loop_closure:
            lc = ((self.jj - self.ii) > 30) & (self.jj > (n - opt_window))
            to_remove = to_remove & ~lc
        self.remove_factors(to_remove, store=True)
        return n
