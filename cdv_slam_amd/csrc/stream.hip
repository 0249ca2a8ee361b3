// stream.hip -- a frame stream whose sizes live on the device, gfx950.
//
// What SLAM.__call__ does around the update for an initialised system (cdvslam/slam.py:697-720) -- the new frame's state
// write and edges (:676-709), the keyframe test (:399-413), keyframe()'s removals, index shift and buffer shift
// (:415-458), the point cloud of the updated patches (:524-526) -- WITHOUT the reference's host round trips: its keyframe
// decision is two .item() read-backs, and every size after it (n, the number of edges left by the mask indexing of
// remove_factors, :339-354) is a host integer again.  Here the decision is a word on the device, n and E live in a
// "dynamic block" (include/cdvslam_hip.h CDV_DYN_*), the kernels that change a size read one block and write the next,
// and every launch is dimensioned by an upper bound: a frame is a fixed sequence of launches with no synchronisation.
//
// Edge order is the reference's, bit for bit: appending writes at E, removing is a stable compaction with the removal
// predicate evaluated in the kernels (no mask tensor), the index shift of a dropped keyframe rides the first compaction.
#include "cdv_common.h"
#include "cdv_se3.h"

namespace {

typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));

__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }

// ------------------------------------------------------------------------------------------------------------------
// 1. a frame arrives: n <- n + 1, its edges (slam.py:707-709, 528-541), its state
// ------------------------------------------------------------------------------------------------------------------
struct BeginArgs {
  const int32_t* dyn_in;
  int32_t* dyn_out;
  int64_t *ii, *jj, *kk;           // active edge lists (capacity `cap`)
  float *target, *weight;          // [cap][2]: the new edges' rows are zeroed (slam.py:336-337 appends zeros)
  const int64_t* ix;               // frame of every patch (pg.index_)
  int64_t cap;
  int M, r, opt_window, frames_cap;
  // the (stubbed) network outputs of the new frame: patch centres + inverse depths, the frame's feature map, and where
  // they go -- patches_[n], gmap_[n % pmem] (net_cdv.py:355-374, slam.py:676-696); all optional (cx == NULL: none)
  const float *cx, *cy, *d;
  const _Float16* fmap;            // [C][h][w]
  _Float16* gmap;                  // [pmem * M][C][3][3]
  float *poses, *patches;
  int C, h, w, pmem;
  float pose_step;                 // initial guess of the new pose: the previous one moved by this much along x
  float* flow_buf;                 // [4][M]: sums and counts of the keyframe test, zeroed here for this frame
  int n_edge_blocks, n_tile_blocks;
};

__global__ __launch_bounds__(256) void stream_begin_kernel(const BeginArgs A) {
  const int tid = threadIdx.x, b = blockIdx.x;
  const int n0 = A.dyn_in[CDV_DYN_N], E0 = A.dyn_in[CDV_DYN_E];
  const int err0 = A.dyn_in[CDV_DYN_ERR];
  const int n = n0 + 1, M = A.M, r = A.r;
  const int64_t f0 = (int64_t)M * imax(n - r, 0), f1 = (int64_t)M * imax(n - 1, 0);
  const int64_t nf = f1 > f0 ? f1 - f0 : 0;
  const int jb0 = imax(n - r, 0), nbj = n - jb0;
  const int64_t nb = (int64_t)M * nbj, cnt = nf + nb;
  const bool full = err0 != 0 || E0 + cnt > A.cap || n >= A.frames_cap;
  if (b < A.n_edge_blocks) {
    if (full) return;
    for (int64_t t = (int64_t)b * 256 + tid; t < cnt; t += (int64_t)A.n_edge_blocks * 256) {
      int64_t k, j;
      if (t < nf) { k = f0 + t; j = n - 1; }
      else { const int64_t u = t - nf; k = (int64_t)M * (n - 1) + u / nbj; j = jb0 + (int)(u % nbj); }
      A.kk[E0 + t] = k;
      A.jj[E0 + t] = j;
      A.ii[E0 + t] = A.ix[k];
      A.target[2 * (E0 + t)] = 0.f; A.target[2 * (E0 + t) + 1] = 0.f;
      A.weight[2 * (E0 + t)] = 0.f; A.weight[2 * (E0 + t) + 1] = 0.f;
    }
    return;
  }
  if (b == A.n_edge_blocks) {
    // ---- the block of sizes after this frame's arrival, the patch grid, the pose guess ----
    if (tid == 0) {
      for (int i = 0; i < CDV_DYN_WORDS; i++) A.dyn_out[i] = A.dyn_in[i];
      if (!full) {
        const int t0 = imax(1, n - A.opt_window);
        A.dyn_out[CDV_DYN_N] = n;
        A.dyn_out[CDV_DYN_E] = E0 + (int)cnt;
        A.dyn_out[CDV_DYN_T0] = t0;
        A.dyn_out[CDV_DYN_NFREE] = n - t0;
        A.dyn_out[CDV_DYN_FRAME] = A.dyn_in[CDV_DYN_FRAME] + 1;
      } else {
        A.dyn_out[CDV_DYN_ERR] = err0 ? err0 : 1;
      }
    }
    if (A.flow_buf) for (int i = tid; i < 4 * M; i += 256) A.flow_buf[i] = 0.f;
    if (full || A.cx == nullptr) return;
    for (int m = tid; m < M; m += 256) {
      float* pk = A.patches + ((int64_t)n0 * M + m) * 27;
      const float x = A.cx[m], y = A.cy[m], dd = A.d[m];
#pragma unroll
      for (int a = 0; a < 9; a++) {
        pk[a] = x + (float)(a % 3 - 1);
        pk[9 + a] = y + (float)(a / 3 - 1);
        pk[18 + a] = dd;
      }
    }
    if (tid == 0 && n0 > 0) {
      const float* ps = A.poses + 7 * (int64_t)(n0 - 1);
      float* pd = A.poses + 7 * (int64_t)n0;
      pd[0] = __fadd_rn(ps[0], A.pose_step);
#pragma unroll
      for (int c = 1; c < 7; c++) pd[c] = ps[c];
    }
    return;
  }
  // ---- the frame's patch tiles: altcorr.patchify(fmap, centres, 1, 'bilinear') as the reference composes it
  // (correlation_kernel.cu:16-47 gather with zeros outside, correlation.py:55-66: x00 + x01 + x10 + x11, float32 weights)
  // rounded to half into gmap_[n % pmem] ----
  if (full || A.cx == nullptr) return;
  const int C = A.C, H = A.h, W = A.w;
  const int64_t total = (int64_t)M * C * 9;
  const int64_t tile0 = (int64_t)(n0 % A.pmem) * M;
  for (int64_t t = (int64_t)(b - A.n_edge_blocks - 1) * 256 + tid; t < total; t += (int64_t)A.n_tile_blocks * 256) {
    const int px = (int)(t % 9);
    const int c = (int)((t / 9) % C);
    const int m = (int)(t / (9 * C));
    const float x = A.cx[m], y = A.cy[m];
    const float fx = floorf(x), fy = floorf(y);
    const float dx = x - fx, dy = y - fy;
    const int i0 = (int)fy + (px / 3 - 1), j0 = (int)fx + (px % 3 - 1);
    const _Float16* src = A.fmap + (int64_t)c * H * W;
    const auto at = [&](int i, int j) -> float { return (i >= 0 && i < H && j >= 0 && j < W) ? (float)src[(int64_t)i * W + j] : 0.f; };
    const float omx = 1.0f - dx, omy = 1.0f - dy;
    const float x00 = __fmul_rn(__fmul_rn(omy, omx), at(i0, j0));
    const float x01 = __fmul_rn(__fmul_rn(omy, dx), at(i0, j0 + 1));
    const float x10 = __fmul_rn(__fmul_rn(dy, omx), at(i0 + 1, j0));
    const float x11 = __fmul_rn(__fmul_rn(dy, dx), at(i0 + 1, j0 + 1));
    const float v = __fadd_rn(__fadd_rn(__fadd_rn(x00, x01), x10), x11);
    A.gmap[((tile0 + m) * C + c) * 9 + px] = (_Float16)v;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// 2. the stub of the update operator (cdv_slam_amd/stream.py): target = reprojected centre + gain tanh(corr[:2]),
//    weight = sigmoid(corr[2:4]) -- stands where net_cdv.py's Update runs; only here because its size lives on the device
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void stream_operator_stub_kernel(const int32_t* __restrict__ dyn, const float* __restrict__ coords,
                                                                   const _Float16* __restrict__ corr, int corr_pitch,
                                                                   float* __restrict__ target, float* __restrict__ weight,
                                                                   float gain) {
  const int E = dyn[CDV_DYN_E];
  for (int e = (int)blockIdx.x * 256 + threadIdx.x; e < E; e += (int)gridDim.x * 256) {
    const _Float16* c = corr + (size_t)e * corr_pitch;
    const float c0 = (float)c[0], c1 = (float)c[1], c2 = (float)c[2], c3 = (float)c[3];
    target[2 * e] = __fadd_rn(coords[(size_t)e * 18 + 4], __fmul_rn(gain, tanhf(c0)));
    target[2 * e + 1] = __fadd_rn(coords[(size_t)e * 18 + 13], __fmul_rn(gain, tanhf(c1)));
    weight[2 * e] = 1.0f / (1.0f + expf(-c2));
    weight[2 * e + 1] = 1.0f / (1.0f + expf(-c3));
  }
}

// ------------------------------------------------------------------------------------------------------------------
// 3. after the bundle adjustment, one launch, two independent jobs:
//    (a) the world points of the patches the update touched (slam.py:524-526 computes ALL m patches every frame; only those
//        of the frames inside the removal window can have moved): points_[m] = (P^-1 iproj(centre))[:3] / [3];
//    (b) the keyframe test's statistic (slam.py:399-413): mean flow_mag (projective_ops.py:120-130, beta = 0.5) over the
//        edges i -> j and j -> i with i = n - KI - 1, j = n - KI + 1.  A workgroup scans its share of the edge list, collects
//        the (few) matching edges in LDS and works them off lane-dense; every matching edge puts the sum over its nine
//        pixels into the slot of ITS patch (kk - M ii: one edge per patch and direction in a patch graph), so the total is
//        summed later in a fixed order -- the decision does not depend on which workgroup got where first.
// ------------------------------------------------------------------------------------------------------------------
struct AfterArgs {
  const int32_t* dyn;
  const float *poses, *patches, *intr;
  const int64_t* ix;
  const int64_t *ii, *jj, *kk;
  int M, window_frames, ki;
  float beta;
  float* points;
  float* flow_buf;
  int n_motion_blocks, n_point_blocks;
};

// one matching edge per lane: the three reprojections of flow_mag for its nine pixels.  (A group of 16 lanes per edge, one
// pixel per lane, was measured: 27 against 10.6 us for the launch -- the matches are neighbours in the edge list (the
// forward edges of a frame are appended together, slam.py:528-534), so ONE workgroup holds ~M of them and then needs six
// trips of the pose algebra and its three dependent load levels instead of one.)
__device__ __forceinline__ void motion_edge(const AfterArgs& A, int e, int dir) {
  const float* __restrict__ poses = A.poses;
  const float* __restrict__ intr = A.intr;
  const int64_t ix = A.ii[e], jx = A.jj[e], kx = A.kk[e];
  float Pi[7], Pj[7], Pinv[7], G[3][7];
#pragma unroll
  for (int a = 0; a < 7; a++) { Pi[a] = poses[7 * ix + a]; Pj[a] = poses[7 * jx + a]; }
  cdv::lt_se3_inv(Pi, Pinv);
  cdv::lt_se3_mul(Pi, Pinv, G[0]);
  cdv::lt_se3_mul(Pj, Pinv, G[1]);
#pragma unroll
  for (int a = 0; a < 3; a++) G[2][a] = G[1][a];
  G[2][3] = 0.f; G[2][4] = 0.f; G[2][5] = 0.f; G[2][6] = 1.f;   // tonly (projective_ops.py:62)
  float t[3][3], q[3][4];
#pragma unroll
  for (int v = 0; v < 3; v++) cdv::lt_se3_load(G[v], t[v], q[v]);
  const float fxi = intr[4 * ix + 0], fyi = intr[4 * ix + 1], cxi = intr[4 * ix + 2], cyi = intr[4 * ix + 3];
  const float fxj = intr[4 * jx + 0], fyj = intr[4 * jx + 1], cxj = intr[4 * jx + 2], cyj = intr[4 * jx + 3];
  const float* pk = A.patches + kx * 27;
  float tot = 0.f;
#pragma unroll
  for (int a = 0; a < 9; a++) {
    float X0[4], X1[4], xy[3][2];
    X0[0] = (pk[a] - cxi) / fxi;
    X0[1] = (pk[9 + a] - cyi) / fyi;
    X0[2] = 1.f;
    X0[3] = pk[18 + a];
#pragma unroll
    for (int v = 0; v < 3; v++) {
      cdv::lt_act4_loaded(t[v], q[v], X0, X1);
      const float d = 1.0f / fmaxf(X1[2], 0.1f);
      const float fx = v == 0 ? fxi : fxj, fy = v == 0 ? fyi : fyj, cx = v == 0 ? cxi : cxj, cy = v == 0 ? cyi : cyj;
      xy[v][0] = fx * (d * X1[0]) + cx;
      xy[v][1] = fy * (d * X1[1]) + cy;
    }
    const float ax = xy[1][0] - xy[0][0], ay = xy[1][1] - xy[0][1];
    const float bx = xy[2][0] - xy[0][0], by = xy[2][1] - xy[0][1];
    tot += A.beta * sqrtf(ax * ax + ay * ay) + (1.0f - A.beta) * sqrtf(bx * bx + by * by);
  }
  const int slot = (int)(kx - (int64_t)A.M * ix);
  if (slot >= 0 && slot < A.M) {
    A.flow_buf[dir * A.M + slot] = tot;          // one edge per (direction, patch): a plain store
    A.flow_buf[(2 + dir) * A.M + slot] = 9.0f;   // pixels counted
  }
}

__global__ __launch_bounds__(256) void stream_after_kernel(const AfterArgs A) {
  const int tid = threadIdx.x, b = blockIdx.x;
  const int n = A.dyn[CDV_DYN_N];
  if (b >= A.n_motion_blocks) {
    // ---- (a) point cloud of the removal window ----
    const int M = A.M;
    const int64_t m0 = (int64_t)imax(n - A.window_frames, 0) * M, m1 = (int64_t)n * M;
    for (int64_t m = m0 + (int64_t)(b - A.n_motion_blocks) * 256 + tid; m < m1; m += (int64_t)A.n_point_blocks * 256) {
      const int64_t f = A.ix[m];
      float Pi[7], Pinv[7], t[3], q[4];
#pragma unroll
      for (int a = 0; a < 7; a++) Pi[a] = A.poses[7 * f + a];
      cdv::lt_se3_inv(Pi, Pinv);
      cdv::lt_se3_load(Pinv, t, q);
      const float* pk = A.patches + m * 27;
      float X0[4], X1[4];
      X0[0] = (pk[4] - A.intr[4 * f + 2]) / A.intr[4 * f + 0];
      X0[1] = (pk[13] - A.intr[4 * f + 3]) / A.intr[4 * f + 1];
      X0[2] = 1.f;
      X0[3] = pk[22];
      cdv::lt_act4_loaded(t, q, X0, X1);
      A.points[3 * m] = X1[0] / X1[3];
      A.points[3 * m + 1] = X1[1] / X1[3];
      A.points[3 * m + 2] = X1[2] / X1[3];
    }
    return;
  }
  // ---- (b) the flow statistic: scan, collect, then work the matches off with dense lanes ----
  __shared__ int s_n;
  __shared__ int s_list[1024];   // (edge << 1 | direction) of this workgroup's matches
  if (tid == 0) s_n = 0;
  __syncthreads();
  const int E = A.dyn[CDV_DYN_E];
  const int64_t fi = n - A.ki - 1, fj = n - A.ki + 1;
  // (edges dealt to the workgroups in pieces of 64 -- wave w of workgroup b takes piece p0 + w nb + b --: the ~M matching
  // forward edges are neighbours in the list and would otherwise all be one workgroup's)
  const int nbk = A.n_motion_blocks;
  for (int p0 = 0; p0 * 64 < E; p0 += 16 * nbk) {   // workgroup-uniform trips: ONE when the launch was sized for E (1024 edges per workgroup)
    int64_t vi[4], vj[4];
    int ve[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {     // four pieces per thread, their loads in flight together
      ve[u] = (p0 + (4 * u + (tid >> 6)) * nbk + b) * 64 + (tid & 63);
      const bool in = ve[u] < E;
      vi[u] = in ? A.ii[ve[u]] : -1;
      vj[u] = in ? A.jj[ve[u]] : -1;
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int dir = (vi[u] == fi && vj[u] == fj) ? 0 : ((vi[u] == fj && vj[u] == fi) ? 1 : -1);
      if (dir >= 0 && vi[u] >= 0) {
        const int at = atomicAdd(&s_n, 1);
        if (at < 1024) s_list[at] = (ve[u] << 1) | dir;
      }
    }
    __syncthreads();
    const int cnt = min(s_n, 1024);
    for (int i = tid; i < cnt; i += 256) motion_edge(A, s_list[i] >> 1, s_list[i] & 1);
    __syncthreads();
    if (tid == 0) s_n = 0;
    __syncthreads();
  }
}

// the decision, by one wave, in a fixed order: (mean_ij + mean_ji) / 2 < thresh (slam.py:413); an empty selection gives
// NaN and no drop, like the reference's mean() of nothing.  force: -1 the test decides, 0 / 1 the caller does.
__device__ __forceinline__ int keyframe_decision(const float* __restrict__ flow_buf, int M, int n, int ki, float thresh,
                                                 int force, int lane, float* motion_out) {
  float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int v = 0; v < 4; v++)
    for (int i = lane; i < M; i += 64) s[v] += flow_buf[v * M + i];
#pragma unroll
  for (int v = 0; v < 4; v++)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s[v] += __shfl_xor(s[v], o);
  const float motion = 0.5f * (s[0] / s[2] + s[1] / s[3]);
  if (motion_out) *motion_out = motion;
  const bool guard = n > ki + 2;
  if (force >= 0) return (force != 0 && guard) ? 1 : 0;
  return (motion < thresh && guard) ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------------------------
// 4. keyframe()'s two removals (slam.py:415-427 and :453-458) as ONE stable compaction.  The reference removes the edges of
//    the dropped frame k (not stored), shifts the indices above k, then removes -- and stores as inactive edges -- those
//    whose source frame left the removal window, judged on the SHIFTED indices with n already decremented.  Per edge that is
//       gone   = drop and (ii == k or jj == k)
//       (i', j', k') = drop ? shifted : as they are;     pruned = not gone and ix[k'] < (n - drop) - REMOVAL_WINDOW
//    kept edges go to the twin buffer, pruned ones to the inactive lists, both in list order: two ranks from one pass.
// ------------------------------------------------------------------------------------------------------------------
struct RemoveArgs {
  const int32_t* dyn_in;
  int32_t* dyn_out;
  int M, ki, removal_window;
  float thresh;
  int force;
  const float* flow_buf;
  float* motion_out;              // [2]: the statistic and the decision, for whoever wants to look (tests)
  const int64_t* ix;
  const int64_t *ii, *jj, *kk;    // source lists
  const float *target, *weight;
  int64_t *ii_o, *jj_o, *kk_o;    // kept edges, compacted (the twin buffers)
  float *target_o, *weight_o;
  int64_t *ii_r, *jj_r, *kk_r;    // pruned edges, appended at dyn[EINAC]
  float *target_r, *weight_r;
  int64_t inac_cap;
  int32_t *counts, *meta;         // [2 nb] per-workgroup (kept, pruned) counts -> exclusive offsets; meta[2]: arrival counter
  int nb;
  int64_t* mirror;                // pinned host word: (frames << 32 | edges) afterwards (sizes the next launches without a sync)
  int n_compact_blocks;           // compact launch: workgroups beyond these shift the frame buffers
};

// 0: kept, 1: pruned (stored), 2: gone; i, j, k come back shifted.  f_same / f_down: ix[k] and ix[k - M] (the patch's frame if
// it stays / if it moves down), requested by the caller before the decision is known
__device__ __forceinline__ int edge_fate(const RemoveArgs& A, int drop, int n_after, int kf, int64_t& i, int64_t& j, int64_t& k,
                                         int64_t f_same, int64_t f_down) {
  bool moved = false;
  if (drop) {
    if (i == kf || j == kf) return 2;
    if (i > kf) { k -= A.M; i -= 1; moved = true; }
    if (j > kf) j -= 1;
  }
  return ((moved ? f_down : f_same) < n_after - A.removal_window) ? 1 : 0;
}

__global__ __launch_bounds__(256) void stream_count_kernel(const RemoveArgs A) {
  __shared__ int s_w[2][4];
  __shared__ int s_drop;
  const int t = threadIdx.x, b = blockIdx.x;
  const int n = A.dyn_in[CDV_DYN_N], E = A.dyn_in[CDV_DYN_E];
  const int kf = n - A.ki;
  // the edges first (their loads do not depend on the decision), then the decision, then their fates
  const int64_t base = (int64_t)b * 1024;
  int64_t vi[4], vj[4], vk[4], fs[4], fd[4];
#pragma unroll
  for (int u = 0; u < 4; u++) {
    const int64_t e = base + u * 256 + t;
    const bool in = e < E;
    vi[u] = in ? A.ii[e] : -1; vj[u] = in ? A.jj[e] : -1; vk[u] = in ? A.kk[e] : 0;
  }
#pragma unroll
  for (int u = 0; u < 4; u++) { fs[u] = A.ix[vk[u]]; fd[u] = A.ix[vk[u] >= A.M ? vk[u] - A.M : 0]; }
  if (t < 64) {
    float motion;
    const int d = keyframe_decision(A.flow_buf, A.M, n, A.ki, A.thresh, A.force, t, &motion);
    if (t == 0) {
      s_drop = d;
      if (b == 0 && A.motion_out) { A.motion_out[0] = motion; A.motion_out[1] = (float)d; }
    }
  }
  __syncthreads();
  const int drop = s_drop;
  int c0 = 0, c1 = 0;
#pragma unroll
  for (int u = 0; u < 4; u++) {
    if (vi[u] >= 0) {
      const int fate = edge_fate(A, drop, n - drop, kf, vi[u], vj[u], vk[u], fs[u], fd[u]);
      c0 += fate == 0; c1 += fate == 1;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { c0 += __shfl_xor(c0, o); c1 += __shfl_xor(c1, o); }
  if ((t & 63) == 0) { s_w[0][t >> 6] = c0; s_w[1][t >> 6] = c1; }
  __syncthreads();
  // plain stores, read by the compact LAUNCH that follows (every workgroup of it sums the counts in front of its own: a
  // shared arrival counter here serialised the workgroups' atomics, ~90 ns each, and one workgroup then scanned for all)
  if (t == 0) {
    A.counts[2 * b] = s_w[0][0] + s_w[0][1] + s_w[0][2] + s_w[0][3];
    A.counts[2 * b + 1] = s_w[1][0] + s_w[1][1] + s_w[1][2] + s_w[1][3];
  }
}

// keyframe(): every per-frame buffer moves frames k + 1 .. n - 1 down by one (slam.py:431-441) -- if the test dropped k
struct ShiftBufs {
  cdv_frame_buf b[CDV_MAX_FRAME_BUFS];
  int64_t first[CDV_MAX_FRAME_BUFS + 1];
  int32_t gran[CDV_MAX_FRAME_BUFS];
  int n_bufs;
  int n_blocks;
};

__device__ __forceinline__ void shift_buffers(const ShiftBufs& F, int bid, int tid, int k, int n) {
  const int64_t total = F.first[F.n_bufs];
  for (int64_t t = (int64_t)bid * 256 + tid; t < total; t += (int64_t)F.n_blocks * 256) {
    int bi = 0;
    while (bi + 1 < F.n_bufs && t >= F.first[bi + 1]) bi++;
    const int64_t piece = t - F.first[bi];
    char* base = reinterpret_cast<char*>(F.b[bi].base);
    const int64_t sb = F.b[bi].slot_bytes;
    const int m = F.b[bi].modulus;
    if (F.gran[bi] == 16) {
      typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
      for (int i = k; i < n - 1; i++) {
        const int64_t d = m > 0 ? i % m : i, s2 = m > 0 ? (i + 1) % m : i + 1;
        *reinterpret_cast<u32x4*>(base + d * sb + 16 * piece) = *reinterpret_cast<const u32x4*>(base + s2 * sb + 16 * piece);
      }
    } else {
      for (int i = k; i < n - 1; i++) {
        const int64_t d = m > 0 ? i % m : i, s2 = m > 0 ? (i + 1) % m : i + 1;
        *reinterpret_cast<uint32_t*>(base + d * sb + 4 * piece) = *reinterpret_cast<const uint32_t*>(base + s2 * sb + 4 * piece);
      }
    }
  }
}

__global__ __launch_bounds__(256) void stream_compact_kernel(const RemoveArgs A, const ShiftBufs F) {
  __shared__ int s_pre[2][4][4];
  __shared__ int s_drop, s_base[4];
  const int t = threadIdx.x, b = blockIdx.x, lane = t & 63, wave = t >> 6;
  const int n = A.dyn_in[CDV_DYN_N], E = A.dyn_in[CDV_DYN_E];
  const int kf = n - A.ki;
  // this thread's edges and their patches' candidate frames first: nothing of it depends on the decision
  int64_t vi[4], vj[4], vk[4], fs[4], fd[4];
  const bool edge_wg = b < A.n_compact_blocks;
#pragma unroll
  for (int u = 0; u < 4; u++) {
    const int64_t e = (int64_t)b * 1024 + u * 256 + t;
    const bool in = edge_wg && e < E;
    vi[u] = in ? A.ii[e] : -1; vj[u] = in ? A.jj[e] : -1; vk[u] = in ? A.kk[e] : 0;
  }
#pragma unroll
  for (int u = 0; u < 4; u++) { fs[u] = A.ix[vk[u]]; fd[u] = A.ix[vk[u] >= A.M ? vk[u] - A.M : 0]; }
  // the decision again (the same 4 M words summed in the same order: the same answer in every workgroup of both launches),
  // and -- by wave 1 -- the counts of the workgroups in front of this one and of all of them
  if (t < 64) {
    const int d = keyframe_decision(A.flow_buf, A.M, n, A.ki, A.thresh, A.force, t, nullptr);
    if (t == 0) s_drop = d;
  } else if (t < 128) {
    int k0 = 0, p0 = 0, kt = 0, pt = 0;
    for (int i = t - 64; i < A.nb; i += 64) {
      const int c0 = A.counts[2 * i], c1 = A.counts[2 * i + 1];
      kt += c0; pt += c1;
      if (i < b) { k0 += c0; p0 += c1; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      k0 += __shfl_xor(k0, o); p0 += __shfl_xor(p0, o); kt += __shfl_xor(kt, o); pt += __shfl_xor(pt, o);
    }
    if (t == 64) { s_base[0] = k0; s_base[1] = p0; s_base[2] = kt; s_base[3] = pt; }
  }
  __syncthreads();
  const int drop = s_drop;
  const int kept_total = s_base[2], pruned_total = s_base[3];
  const int inac = A.dyn_in[CDV_DYN_EINAC];
  const bool inac_fits = (int64_t)inac + pruned_total <= A.inac_cap;
  if (b == 0 && t == 0) {   // the sizes after keyframe()
    for (int i = 0; i < CDV_DYN_WORDS; i++) A.dyn_out[i] = A.dyn_in[i];
    A.dyn_out[CDV_DYN_E] = kept_total;
    A.dyn_out[CDV_DYN_DROP] = drop;
    A.dyn_out[CDV_DYN_N] = n - drop;
    if (!inac_fits) A.dyn_out[CDV_DYN_ERR] = 2;   // inactive edges beyond their capacity: not stored
    else A.dyn_out[CDV_DYN_EINAC] = inac + pruned_total;
    if (A.mirror)
      __hip_atomic_store(A.mirror, ((int64_t)A.dyn_in[CDV_DYN_FRAME] << 32) | (int64_t)(uint32_t)kept_total, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (b >= A.n_compact_blocks) {                 // the frame buffers: independent of the edge lists
    if (drop) shift_buffers(F, b - A.n_compact_blocks, t, kf, n);
    return;
  }
  const int64_t r0 = inac;
  const bool store = A.ii_r != nullptr && inac_fits && A.dyn_in[CDV_DYN_ERR] == 0;
  const int64_t base = (int64_t)b * 1024;
  int fate[4];
#pragma unroll
  for (int u = 0; u < 4; u++) {
    fate[u] = 2;
    if (vi[u] >= 0) fate[u] = edge_fate(A, drop, n - drop, kf, vi[u], vj[u], vk[u], fs[u], fd[u]);
    const int w0 = __popcll(__ballot(fate[u] == 0)), w1 = __popcll(__ballot(fate[u] == 1));
    if (lane == 0) { s_pre[0][u][wave] = w0; s_pre[1][u][wave] = w1; }
  }
  __syncthreads();
  const int kbase = s_base[0], pbase = s_base[1];
  int before0 = 0, before1 = 0;
#pragma unroll
  for (int u = 0; u < 4; u++) {
    int pre0 = before0, pre1 = before1;
    for (int w = 0; w < wave; w++) { pre0 += s_pre[0][u][w]; pre1 += s_pre[1][u][w]; }
    const unsigned long long below = (1ull << lane) - 1ull;
    const int64_t e = base + u * 256 + t;
    const unsigned long long bal0 = __ballot(fate[u] == 0);
    if (fate[u] == 0) {
      const int64_t d = (int64_t)kbase + pre0 + __popcll(bal0 & below);
      A.ii_o[d] = vi[u]; A.jj_o[d] = vj[u]; A.kk_o[d] = vk[u];
      *reinterpret_cast<float2*>(A.target_o + 2 * d) = *reinterpret_cast<const float2*>(A.target + 2 * e);
      *reinterpret_cast<float2*>(A.weight_o + 2 * d) = *reinterpret_cast<const float2*>(A.weight + 2 * e);
    }
    {
      const unsigned long long bal1 = __ballot(fate[u] == 1);
      if (fate[u] == 1 && store) {
        const int64_t d = r0 + pbase + pre1 + __popcll(bal1 & below);
        A.ii_r[d] = vi[u]; A.jj_r[d] = vj[u]; A.kk_r[d] = vk[u];
        *reinterpret_cast<float2*>(A.target_r + 2 * d) = *reinterpret_cast<const float2*>(A.target + 2 * e);
        *reinterpret_cast<float2*>(A.weight_r + 2 * d) = *reinterpret_cast<const float2*>(A.weight + 2 * e);
      }
    }
    before0 += s_pre[0][u][0] + s_pre[0][u][1] + s_pre[0][u][2] + s_pre[0][u][3];
    before1 += s_pre[1][u][0] + s_pre[1][u][1] + s_pre[1][u][2] + s_pre[1][u][3];
  }
}

inline int grid_of(int64_t n, int per, int cap) {
  const int64_t b = (n + per - 1) / per;
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace

extern "C" size_t cdv_stream_workspace_bytes(int64_t edge_capacity, int M) {
  // [16] meta words | per-workgroup (kept, pruned) counts | flow_buf [4][M] | motion [2]
  return sizeof(int32_t) * (size_t)(16 + 2 * ((edge_capacity + 1023) / 1024 + 16)) + sizeof(float) * (size_t)(4 * M + 8) + 256;
}

namespace {
struct StreamWs {
  int32_t *meta, *counts;
  float *flow_buf, *motion;
};
inline StreamWs stream_ws(void* ws, int64_t cap, int M) {
  StreamWs w;
  w.meta = (int32_t*)ws;
  w.counts = w.meta + 16;
  const size_t nc = 2 * (size_t)((cap + 1023) / 1024 + 16);
  w.flow_buf = reinterpret_cast<float*>(w.counts + nc);
  w.motion = w.flow_buf + 4 * (size_t)M;
  return w;
}
}  // namespace

// A frame arrives (slam.py:676-709 without the networks): reads dyn_in, writes dyn_out (n + 1, E + the frame's forward and
// backward edges, the window of the coming update); appends the edges; with cx != NULL also writes the frame's patches,
// pose guess and patch tiles (the stubbed network outputs).  ws: cdv_stream_workspace_bytes, zero-initialised once.
extern "C" int cdv_stream_frame_begin(const int32_t* dyn_in, int32_t* dyn_out, int64_t* ii, int64_t* jj, int64_t* kk, float* target,
                                      float* weight, const int64_t* ix, int64_t edge_capacity, int M, int patch_lifetime,
                                      int opt_window, int frames_capacity, const float* cx, const float* cy, const float* d,
                                      const void* fmap_chw, void* gmap_planar, float* poses, float* patches, int C, int H, int W,
                                      int pmem, float pose_step, void* ws, void* stream) {
  CDV_REQUIRE(dyn_in && dyn_out && dyn_in != dyn_out && ii && jj && kk && target && weight && ix && ws, CDV_ERR_ARG,
              "cdv_stream_frame_begin: NULL buffer (or dyn_in == dyn_out)");
  CDV_REQUIRE(M >= 1 && patch_lifetime >= 1 && opt_window >= 1 && frames_capacity >= 2, CDV_ERR_ARG, "cdv_stream_frame_begin: sizes");
  CDV_REQUIRE(cx == nullptr || (cy && d && fmap_chw && gmap_planar && poses && patches && C >= 1 && pmem >= 1), CDV_ERR_ARG,
              "cdv_stream_frame_begin: frame inputs");
  const StreamWs w = stream_ws(ws, edge_capacity, M);
  BeginArgs A;
  A.dyn_in = dyn_in; A.dyn_out = dyn_out; A.ii = ii; A.jj = jj; A.kk = kk; A.target = target; A.weight = weight; A.ix = ix;
  A.cap = edge_capacity; A.M = M; A.r = patch_lifetime; A.opt_window = opt_window; A.frames_cap = frames_capacity;
  A.cx = cx; A.cy = cy; A.d = d; A.fmap = (const _Float16*)fmap_chw; A.gmap = (_Float16*)gmap_planar; A.poses = poses;
  A.patches = patches; A.C = C; A.h = H; A.w = W; A.pmem = pmem; A.pose_step = pose_step; A.flow_buf = w.flow_buf;
  A.n_edge_blocks = grid_of((int64_t)2 * patch_lifetime * M, 256, 1024);
  A.n_tile_blocks = cx ? grid_of((int64_t)M * C * 9, 256, 1024) : 0;
  hipLaunchKernelGGL(stream_begin_kernel, dim3(A.n_edge_blocks + 1 + A.n_tile_blocks), dim3(256), 0, (hipStream_t)stream, A);
  CDV_LAUNCH_CHECK();
  return CDV_OK;
}

extern "C" int cdv_stream_operator_stub(const int32_t* dyn, const float* coords, const void* corr, int corr_pitch, float* target,
                                        float* weight, float gain, int64_t E_bound, void* stream) {
  CDV_REQUIRE(dyn && coords && corr && target && weight && corr_pitch >= 4, CDV_ERR_ARG, "cdv_stream_operator_stub: arguments");
  hipLaunchKernelGGL(stream_operator_stub_kernel, dim3(grid_of(E_bound, 256, 4096)), dim3(256), 0, (hipStream_t)stream, dyn, coords,
                     (const _Float16*)corr, corr_pitch, target, weight, gain);
  CDV_LAUNCH_CHECK();
  return CDV_OK;
}

// SLAM.keyframe() (slam.py:408-458) and the point cloud of slam.py:524-526 for a 3 x 3 patch graph, three launches, no read-back:
//   1. point cloud of the removal window | flow statistic of the frames around k = n - KEYFRAME_INDEX
//   2. the decision (force: -1 the reference's test with `thresh`, 0 / 1 the caller's) and the counts of ONE compaction that
//      does both removals of keyframe(); the sizes after it -> dyn_out
//   3. the compaction (kept edges -> the twin buffers `dst`, pruned ones -> the inactive lists, indices shifted when k was
//      dropped) | the shift of the frame buffers
// points may be NULL (no point cloud).
extern "C" int cdv_stream_keyframe(const int32_t* dyn_in, int32_t* dyn_out, const float* poses, const float* patches,
                                   const float* intrinsics, const int64_t* ix, const int64_t* ii_src, const int64_t* jj_src,
                                   const int64_t* kk_src, const float* target_src, const float* weight_src, int64_t* ii_dst,
                                   int64_t* jj_dst, int64_t* kk_dst, float* target_dst, float* weight_dst, int64_t* ii_inac,
                                   int64_t* jj_inac, int64_t* kk_inac, float* target_inac, float* weight_inac,
                                   int64_t inactive_capacity, int64_t edge_capacity, int64_t E_bound, int M, int keyframe_index,
                                   int removal_window, float keyframe_thresh, int force, const cdv_frame_buf* bufs, int n_bufs,
                                   float* points, int64_t* mirror_host, void* ws, void* stream) {
  CDV_REQUIRE(dyn_in && dyn_out && dyn_in != dyn_out, CDV_ERR_ARG, "cdv_stream_keyframe: two distinct dynamic blocks");
  CDV_REQUIRE(poses && patches && intrinsics && ix && ii_src && jj_src && kk_src && target_src && weight_src && ii_dst && jj_dst &&
                  kk_dst && target_dst && weight_dst && ws, CDV_ERR_ARG, "cdv_stream_keyframe: NULL buffer");
  CDV_REQUIRE(ii_src != ii_dst, CDV_ERR_ARG, "cdv_stream_keyframe: the compaction is not in place (src == dst)");
  CDV_REQUIRE(n_bufs >= 0 && n_bufs <= CDV_MAX_FRAME_BUFS && (n_bufs == 0 || bufs), CDV_ERR_ARG, "cdv_stream_keyframe: frame buffers");
  CDV_REQUIRE(E_bound >= 1 && E_bound <= edge_capacity, CDV_ERR_ARG, "cdv_stream_keyframe: E_bound");
  hipStream_t s = (hipStream_t)stream;
  const StreamWs w = stream_ws(ws, edge_capacity, M);
  AfterArgs P;
  P.dyn = dyn_in; P.poses = poses; P.patches = patches; P.intr = intrinsics; P.ix = ix; P.ii = ii_src; P.jj = jj_src; P.kk = kk_src;
  P.M = M; P.window_frames = removal_window + 2; P.ki = keyframe_index; P.beta = 0.5f; P.points = points; P.flow_buf = w.flow_buf;
  P.n_motion_blocks = grid_of(E_bound, 1024, 256);
  P.n_point_blocks = points ? grid_of((int64_t)(removal_window + 2) * M, 256, 256) : 0;
  hipLaunchKernelGGL(stream_after_kernel, dim3(P.n_motion_blocks + P.n_point_blocks), dim3(256), 0, s, P);
  RemoveArgs A;
  A.dyn_in = dyn_in; A.dyn_out = dyn_out; A.M = M; A.ki = keyframe_index; A.removal_window = removal_window;
  A.thresh = keyframe_thresh; A.force = force; A.flow_buf = w.flow_buf; A.motion_out = w.motion; A.ix = ix;
  A.ii = ii_src; A.jj = jj_src; A.kk = kk_src; A.target = target_src; A.weight = weight_src;
  A.ii_o = ii_dst; A.jj_o = jj_dst; A.kk_o = kk_dst; A.target_o = target_dst; A.weight_o = weight_dst;
  A.ii_r = ii_inac; A.jj_r = jj_inac; A.kk_r = kk_inac; A.target_r = target_inac; A.weight_r = weight_inac;
  A.inac_cap = inactive_capacity; A.counts = w.counts; A.meta = w.meta; A.mirror = mirror_host;
  const int nb = grid_of(E_bound, 1024, 1 << 22);
  A.nb = nb; A.n_compact_blocks = nb;
  hipLaunchKernelGGL(stream_count_kernel, dim3(nb), dim3(256), 0, s, A);
  ShiftBufs F;
  F.n_bufs = n_bufs;
  F.first[0] = 0;
  for (int i = 0; i < n_bufs; i++) {
    const cdv_frame_buf& b = bufs[i];
    CDV_REQUIRE(b.base != nullptr && b.slot_bytes > 0 && b.slot_bytes % 4 == 0 && b.modulus >= 0 && ((uintptr_t)b.base & 3) == 0,
                CDV_ERR_ARG, "cdv_stream_keyframe: a buffer needs a 4-byte aligned base, slot_bytes % 4 == 0, modulus >= 0");
    F.b[i] = b;
    F.gran[i] = (b.slot_bytes % 16 == 0 && ((uintptr_t)b.base & 15) == 0) ? 16 : 4;
    F.first[i + 1] = F.first[i] + b.slot_bytes / F.gran[i];
  }
  F.n_blocks = n_bufs > 0 ? grid_of(F.first[n_bufs], 256, 2048) : 0;
  hipLaunchKernelGGL(stream_compact_kernel, dim3(nb + F.n_blocks), dim3(256), 0, s, A, F);
  CDV_LAUNCH_CHECK();
  return CDV_OK;
}

// slam.py:524-526 on its own (cdv_stream_keyframe does it next to the flow statistic)
extern "C" int cdv_stream_points(const int32_t* dyn, const float* poses, const float* patches, const float* intrinsics,
                                 const int64_t* ix, int M, int window_frames, float* points, void* stream) {
  CDV_REQUIRE(dyn && poses && patches && intrinsics && ix && points && M >= 1 && window_frames >= 1, CDV_ERR_ARG,
              "cdv_stream_points: arguments");
  AfterArgs P;
  P.dyn = dyn; P.poses = poses; P.patches = patches; P.intr = intrinsics; P.ix = ix; P.ii = nullptr; P.jj = nullptr; P.kk = nullptr;
  P.M = M; P.window_frames = window_frames; P.ki = 0; P.beta = 0.5f; P.points = points; P.flow_buf = nullptr;
  P.n_motion_blocks = 0;
  P.n_point_blocks = grid_of((int64_t)window_frames * M, 256, 256);
  hipLaunchKernelGGL(stream_after_kernel, dim3(P.n_point_blocks), dim3(256), 0, (hipStream_t)stream, P);
  CDV_LAUNCH_CHECK();
  return CDV_OK;
}

// (motion, decision) of the last cdv_stream_keyframe on this workspace: two floats on the device (tests read them back)
extern "C" const float* cdv_stream_motion(void* ws, int64_t edge_capacity, int M) {
  return ws ? stream_ws(ws, edge_capacity, M).motion : nullptr;
}

extern "C" int cdv_frame_ingest(const void* fmap_chw, void* fmap1_nhwc, void* fmap2_nhwc, void* fmap1_nchw, void* fmap2_nchw, int slot,
                                int C, int H, int W, const void* gmap_planar, void* gmap_pm, int64_t Ng, int64_t gmap_first,
                                int64_t gmap_count, void* stream);

extern "C" int cdv_stream_frame(cdv_stream_desc* D, const void* fmap_chw, const float* cx, const float* cy, const float* depth,
                                int force, void* stream) {
  CDV_REQUIRE(D != nullptr && fmap_chw && cx && cy && depth, CDV_ERR_ARG, "cdv_stream_frame: NULL argument");
  // (no host-side capacity check: what is bounded is the number of KEYFRAMES n, which only the device knows -- frames the
  // keyframe test drops do not use up the buffers (slam.py bounds n the same way); the begin launch refuses a frame that does
  // not fit and raises CDV_DYN_ERR, and a replayed hipGraph goes through exactly the same check)
  CDV_REQUIRE(D->opt_window >= 1 && D->opt_window <= 32, CDV_ERR_UNSUPPORTED, "cdv_stream_frame: OPTIMIZATION_WINDOW 1 .. 32");
  const int M = D->M;
  const int ring = D->ring_blocks > 0 ? D->ring_blocks : 8;
  CDV_REQUIRE(ring == 2 || ring == 4 || ring == 8, CDV_ERR_ARG, "cdv_stream_frame: ring_blocks must be 0 (8), 2, 4 or 8");
  auto blk = [&](int i) { return D->dyn + CDV_DYN_WORDS * (i % ring); };
  const int a = D->slot, b = a + 1, e = a + 2;
  const int cur = D->cur & 1, oth = cur ^ 1;
  // an upper bound of the number of edges once this frame has arrived, without asking the device: what the last finished
  // keyframe() left (pinned word: frames << 32 | edges) plus 2 r M per frame begun since
  const int64_t seen = D->mirror_host ? *reinterpret_cast<volatile int64_t*>(D->mirror_host) : 0;
  int64_t Eb = (seen & 0xFFFFFFFFll) + ((int64_t)D->frames + 1 - (seen >> 32)) * 2 * D->patch_lifetime * M;
  if (Eb > D->edge_capacity || D->fixed_bound) Eb = D->edge_capacity;
  if (Eb < 1) Eb = 1;
  int rc = cdv_stream_frame_begin(blk(a), blk(b), D->ii[cur], D->jj[cur], D->kk[cur], D->target[cur], D->weight[cur], D->ix, D->edge_capacity, M,
                                  D->patch_lifetime, D->opt_window, D->frames_capacity, cx, cy, depth, fmap_chw, D->gmap_planar,
                                  D->poses, D->patches, D->C, D->H, D->W, D->pmem, D->pose_step, D->ws, stream);
  if (rc != CDV_OK) return rc;
  D->frames += 1;
  if (D->frames < 8) {   // before initialisation only the rings are filled (n == frames: no keyframe test has run yet)
    D->slot = b % ring;
    return cdv_frame_ingest(fmap_chw, D->fmap1_nhwc, D->fmap2_nhwc, nullptr, nullptr, (D->frames - 1) % D->mem, D->C, D->H, D->W,
                            D->gmap_planar, D->gmap_pm, (int64_t)D->pmem * M, (int64_t)((D->frames - 1) % D->pmem) * M, M, stream);
  }
  rc = cdv_update_prologue_table_dyn(fmap_chw, D->fmap1_nhwc, D->fmap2_nhwc, D->mem, D->pmem, D->C, D->H, D->W, D->gmap_planar,
                                     D->gmap_pm, (int64_t)D->pmem * M, M, D->poses, D->patches, D->intrinsics, D->ii[cur], D->jj[cur],
                                     D->kk[cur], Eb, blk(b), D->coords, D->graph_ws, D->graph_ws_bytes, D->graph_E_max,
                                     D->graph_k_range, D->table_capacity, stream);
  if (rc != CDV_OK) return rc;
  rc = cdv_corr_fused_stream_dyn(D->gmap_pm, D->fmap1_nhwc, D->fmap2_nhwc, cdv_graph_corr_records(D->graph_ws), D->corr_out, Eb,
                                 blk(b), (int64_t)D->pmem * M, D->mem, D->C, D->H, D->W, D->H / 4, D->W / 4, 1.0f, 4.0f, 1, stream);
  if (rc != CDV_OK) return rc;
  rc = cdv_stream_operator_stub(blk(b), D->coords, D->corr_out, 882, D->target[cur], D->weight[cur], D->gain, Eb, stream);
  if (rc != CDV_OK) return rc;
  rc = cdv_ba_forward_dyn(D->poses, D->patches, D->intrinsics, D->target[cur], D->weight[cur], D->lmbda, D->ii[cur], D->jj[cur], D->kk[cur], Eb,
                          3, D->opt_window, blk(b), 2, D->graph_ws, D->ba_ws, D->ba_ws_bytes, D->table_capacity, stream);
  if (rc != CDV_OK) return rc;
  rc = cdv_stream_keyframe(blk(b), blk(e), D->poses, D->patches, D->intrinsics, D->ix, D->ii[cur], D->jj[cur], D->kk[cur],
                           D->target[cur], D->weight[cur], D->ii[oth], D->jj[oth], D->kk[oth], D->target[oth], D->weight[oth], D->ii_inac,
                           D->jj_inac, D->kk_inac, D->target_inac, D->weight_inac, D->inactive_capacity, D->edge_capacity, Eb, M,
                           D->keyframe_index, D->removal_window, D->keyframe_thresh, force, D->bufs, D->n_bufs, D->points,
                           D->mirror_host, D->ws, stream);
  if (rc != CDV_OK) return rc;
  D->cur = oth;
  D->slot = e % ring;
  return CDV_OK;
}
