// cdv_parts.h -- bodies of the three independent per-frame kernels that open an update (feature / tile ingest,
// reprojection, patch-id histogram), as device functions taking an explicit (block, number of blocks): each is
// launched on its own behind its C entry point (corr.hip, reproject.hip, graph.hip) and all three side by side in
// ONE launch by cdv_update_prologue (prologue.hip) -- they are latency-bound (a few microseconds of work each), so
// sharing a launch costs the longest of them instead of their sum.
#pragma once
#include "cdv_common.h"
#include "cdv_graph.h"
#include "cdv_se3.h"

namespace cdv {

// ---- padded channels-last feature rings (corr.hip) --------------------------------------------------------
constexpr int PART_PADX = CDV_FMAP_PADX, PART_PADY = CDV_FMAP_PADY;

struct IngestArgs {
  const _Float16* src;     // new frame [C][H][W]
  _Float16 *f1_nhwc, *f2_nhwc, *f1_nchw, *f2_nchw;
  int slot, C, H, W;
  const _Float16* gsrc;    // planar patch tiles [Ng][C][3][3] (may be null)
  _Float16* gdst;          // pixel-major tiles [Ng][9][C]
  int64_t gfirst, gcount;
  int fblocks, gblocks;    // workgroups for the feature maps / for the tiles
  // sizes on the device (include/cdvslam_hip.h CDV_DYN_*): the newest keyframe is n - 1 with n = dyn[CDV_DYN_N]; its ring slot
  // is (n - 1) % dyn_mem and its tiles are [((n - 1) % dyn_pmem) * gcount, + gcount) -- `slot` and `gfirst` are then unused
  const int32_t* dyn = nullptr;
  int dyn_mem = 0, dyn_pmem = 0;
};

// patch tiles planar [C][3][3] -> pixel-major [9][C]: one thread per (tile, pixel, 8-channel group)
__device__ __forceinline__ void gmap_pm_convert(const _Float16* __restrict__ src, _Float16* __restrict__ dst,
                                                int64_t first, int64_t count, int C, int64_t tid, int64_t nthreads) {
  const int G = C / 8;
  const int64_t total = count * 9 * G;
  for (int64_t idx = tid; idx < total; idx += nthreads) {
    int64_t t = idx;
    const int gq = (int)(t % G); t /= G;
    const int px = (int)(t % 9); t /= 9;
    const int64_t tile = first + t;
    cdv_half8 v;
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = src[(tile * C + 8 * gq + j) * 9 + px];
    *reinterpret_cast<cdv_half8*>(dst + (tile * 9 + px) * C + 8 * gq) = v;
  }
}

// one frame [C][H][W] -> ring slot of the level-0 NHWC ring and its 4x4 average pool into level 1
// (F.avg_pool2d(fmap, 4, 4), slam.py:682: f16 in, f32 sum of 16, * 1/16, rounded to f16); blocks
// [fblocks, fblocks + gblocks) convert the frame's patch tiles to the pixel-major layout instead
__device__ __forceinline__ void ingest_body(const IngestArgs& a, int bid, int nthreads_per_block, int tid) {
  int slot_dyn = a.slot;
  int64_t gfirst = a.gfirst;
  if (a.dyn) {
    const int newest = max(a.dyn[CDV_DYN_N] - 1, 0);
    slot_dyn = newest % a.dyn_mem;
    gfirst = (int64_t)(newest % a.dyn_pmem) * a.gcount;
  }
  if (bid >= a.fblocks) {
    gmap_pm_convert(a.gsrc, a.gdst, gfirst, a.gcount, a.C, (int64_t)(bid - a.fblocks) * nthreads_per_block + tid,
                    (int64_t)a.gblocks * nthreads_per_block);
    return;
  }
  // one thread per (full-resolution pixel, 8-channel group); the 16 pixels of a 4x4 pooling block are the 16 lanes of a
  // DPP row, so the pooled sum is four row rotations (no LDS, no 128-gather serial loop per thread)
  const int C = a.C, H = a.H, W = a.W, slot = slot_dyn;
  const int G = C / 8, H4 = H / 4, W4 = W / 4;
  const int64_t total = (int64_t)H4 * W4 * G * 16;   // a multiple of 64: whole rows of 16 lanes stay together
  for (int64_t idx = (int64_t)bid * nthreads_per_block + tid; idx < total; idx += (int64_t)a.fblocks * nthreads_per_block) {
    int64_t t = idx;
    const int sp = (int)(t & 15); t >>= 4;    // sub-pixel of the pooling block: row sp >> 2, column sp & 3
    const int xq = (int)(t % W4); t /= W4;
    const int gq = (int)(t % G); t /= G;
    const int yq = (int)t;
    const int yh = 4 * yq + (sp >> 2), xw = 4 * xq + (sp & 3);
    cdv_half8 v;
    float sum[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const _Float16 sv = a.src[((int64_t)(8 * gq + j) * H + yh) * W + xw];
      v[j] = sv;
      sum[j] = (float)sv;
      if (a.f1_nchw) a.f1_nchw[(((int64_t)slot * C + 8 * gq + j) * H + yh) * W + xw] = sv;
    }
    *reinterpret_cast<cdv_half8*>(a.f1_nhwc + (((int64_t)slot * (H + 2 * PART_PADY) + yh + PART_PADY) *
                                                   (W + 2 * PART_PADX) + xw + PART_PADX) * C + 8 * gq) = v;
#pragma unroll
    for (int j = 0; j < 8; j++) {
#define CDV_ROR_ADD(n) sum[j] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sum[j]), 0x120 + (n), 0xf, 0xf, false));
      CDV_ROR_ADD(8) CDV_ROR_ADD(4) CDV_ROR_ADD(2) CDV_ROR_ADD(1)
#undef CDV_ROR_ADD
    }
    if (sp == 0) {
      cdv_half8 pv;
#pragma unroll
      for (int j = 0; j < 8; j++) {
        pv[j] = (_Float16)(sum[j] * (1.0f / 16.0f));
        if (a.f2_nchw) a.f2_nchw[(((int64_t)slot * C + 8 * gq + j) * H4 + yq) * W4 + xq] = pv[j];
      }
      *reinterpret_cast<cdv_half8*>(a.f2_nhwc + (((int64_t)slot * (H4 + 2 * PART_PADY) + yq + PART_PADY) *
                                                     (W4 + 2 * PART_PADX) + xq + PART_PADX) * C + 8 * gq) = pv;
    }
  }
}

// ---- pops.transform (reproject.hip) ----------------------------------------------------------------------
struct TfArgs {
  const float *poses, *patches, *intr;
  const int64_t *ii, *jj, *kk;
  int64_t E;
  int flags;
  float *coords, *validpx, *valid, *Ji, *Jj, *Jz;
};

// one patch pixel through iproj -> Act4 -> proj (projective_ops.py:19-50): shared by every kernel that reprojects, so
// that they agree bit for bit (the same expression tree gets the same multiply-add contractions)
__device__ __forceinline__ void tf_pixel(const float* t, const float* q, float px, float py, float pd, float fxi, float fyi,
                                         float cxi, float cyi, float fxj, float fyj, float cxj, float cyj, float& x, float& y,
                                         float (&X1)[4]) {
  CDV_NOCONTRACT
  float X0[4];
  X0[0] = (px - cxi) / fxi;              // iproj, projective_ops.py:19-29
  X0[1] = (py - cyi) / fyi;
  X0[2] = 1.f;
  X0[3] = pd;
  lt_act4_loaded(t, q, X0, X1);
  const float d = 1.0f / fmaxf(X1[2], 0.1f); // proj, projective_ops.py:43
  x = fxj * (d * X1[0]) + cxj;
  y = fyj * (d * X1[1]) + cyj;
}

// relative transform of an edge as Act4 uses it: poses[jj] * poses[ii].inv(), re-normalised on load (so3.h:30-37)
__device__ __forceinline__ void tf_relative(const float* __restrict__ poses, int64_t ix, int64_t jx, bool tonly, float* G,
                                            float* t, float* q) {
  float Pi[7], Pj[7], Pinv[7];
#pragma unroll
  for (int a = 0; a < 7; a++) { Pi[a] = poses[7 * ix + a]; Pj[a] = poses[7 * jx + a]; }
  lt_se3_inv(Pi, Pinv);          // poses[:, ii].inv()          projective_ops.py:60
  lt_se3_mul(Pj, Pinv, G);       // poses[:, jj] * ...
  if (tonly) { G[3] = 0.f; G[4] = 0.f; G[5] = 0.f; G[6] = 1.f; }
  lt_se3_load(G, t, q);          // Act4 reloads (re-normalises) Gij  so3.h:30-37
}

template <int P>
__device__ __forceinline__ void transform_body(const TfArgs& A, int64_t n) {
  if (n >= A.E) return;
  const float* __restrict__ poses = A.poses;
  const float* __restrict__ intr = A.intr;
  float* __restrict__ coords = A.coords;
  const int64_t ix = A.ii[n], jx = A.jj[n], kx = A.kk[n];
  constexpr int PP = P * P;

  float G[7], t[3], q[4];
  tf_relative(poses, ix, jx, (A.flags & CDV_TF_TONLY) != 0, G, t, q);

  const float fxi = intr[4 * ix + 0], fyi = intr[4 * ix + 1], cxi = intr[4 * ix + 2], cyi = intr[4 * ix + 3];
  const float fxj = intr[4 * jx + 0], fyj = intr[4 * jx + 1], cxj = intr[4 * jx + 2], cyj = intr[4 * jx + 3];
  const float* pk = A.patches + kx * 3 * PP;
  const bool e2pp = (A.flags & CDV_TF_LAYOUT_E2PP) != 0;

  float Xc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int a = 0; a < PP; a++) {
    float X1[4], x, y;
    tf_pixel(t, q, pk[a], pk[PP + a], pk[2 * PP + a], fxi, fyi, cxi, cyi, fxj, fyj, cxj, cyj, x, y, X1);
    if (e2pp) {
      coords[(n * 2 + 0) * PP + a] = x;
      coords[(n * 2 + 1) * PP + a] = y;
    } else {
      coords[(n * PP + a) * 2 + 0] = x;
      coords[(n * PP + a) * 2 + 1] = y;
    }
    if (A.validpx) A.validpx[n * PP + a] = (X1[2] > 0.2f) ? 1.f : 0.f;
    if (a == (P / 2) * P + P / 2) { Xc[0] = X1[0]; Xc[1] = X1[1]; Xc[2] = X1[2]; Xc[3] = X1[3]; }
  }

  if (A.Ji) {  // projective_ops.py:71-108
    const float X = Xc[0], Y = Xc[1], Z = Xc[2], H = Xc[3];
    const float d = (fabsf(Z) > 0.2f) ? 1.0f / Z : 0.f;
    float R[9];
    lt_quat_to_R(q, R);
    // rows of Jp*Ja: Jp = [fx d, 0, -fx X d^2, 0 ; 0, fy d, -fy Y d^2, 0], Ja = [H I | -[X]x ; 0]
    float row[2][6];
    const float a0 = fxj * d, a2 = -fxj * X * d * d;
    row[0][0] = a0 * H; row[0][1] = 0.f;    row[0][2] = a2 * H;
    row[0][3] = a2 * Y; row[0][4] = a0 * Z - a2 * X; row[0][5] = -a0 * Y;
    const float b1 = fyj * d, b2 = -fyj * Y * d * d;
    row[1][0] = 0.f;    row[1][1] = b1 * H; row[1][2] = b2 * H;
    row[1][3] = -b1 * Z + b2 * Y; row[1][4] = -b2 * X; row[1][5] = b1 * X;
#pragma unroll
    for (int r = 0; r < 2; r++) {
      float o[6];
      lt_se3_adjT_loaded(t, R, row[r], o);   // Ji = -Gij.adjT(Jj)
#pragma unroll
      for (int c = 0; c < 6; c++) {
        A.Jj[(n * 2 + r) * 6 + c] = row[r][c];
        A.Ji[(n * 2 + r) * 6 + c] = -o[c];
      }
    }
    // Jz = Jp * Gij.matrix()[:, 3]  (column (t, 1); Jp's 4th column is zero)
    A.Jz[n * 2 + 0] = a0 * t[0] + a2 * t[2];
    A.Jz[n * 2 + 1] = b1 * t[1] + b2 * t[2];
    A.valid[n] = (Z > 0.2f) ? 1.f : 0.f;
  }
}

// ---- patch-id histogram over (id mod R) + min / max of kk, jj (graph.hip, launch 1 of the index build) ----
// The same scan for a launch of its own (1024 threads, AFTER the histogram launch: plain loads see its atomics): wave w takes
// the w-th sixteenth of the id range, 64 consecutive bins at a time -- coalesced, where graph_scan_body gives every thread a
// contiguous run of bins (fine for the few thousand ids of a frame-to-frame graph, 250 us for the 10^5 frame-pair keys of
// a global bundle adjustment).
__device__ __forceinline__ void graph_scan_wide_body(int32_t* meta, const int32_t* __restrict__ stage, int nstage,
                                                     int32_t* khist, int32_t* kcount, int32_t* krank, int32_t E,
                                                     int64_t k_cap, int T, int t) {
  constexpr int IMAX = 0x7fffffff, IMIN = (int)0x80000000;
  __shared__ int32_t w_sum[16], w_cnt[16];
  __shared__ int32_t w_mm[16][4];
  int kmin = IMAX, kmax = IMIN, jmin = IMAX, jmax = IMIN;
  for (int i = t; i < nstage; i += T) {
    kmin = min(kmin, stage[4 * i]); kmax = max(kmax, stage[4 * i + 1]);
    jmin = min(jmin, stage[4 * i + 2]); jmax = max(jmax, stage[4 * i + 3]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    kmin = min(kmin, __shfl_xor(kmin, o)); kmax = max(kmax, __shfl_xor(kmax, o));
    jmin = min(jmin, __shfl_xor(jmin, o)); jmax = max(jmax, __shfl_xor(jmax, o));
  }
  const int lane = t & 63, wv = t >> 6, nw = T >> 6;
  if (lane == 0) { w_mm[wv][0] = kmin; w_mm[wv][1] = kmax; w_mm[wv][2] = jmin; w_mm[wv][3] = jmax; }
  __syncthreads();
  for (int w = 0; w < nw; w++) {
    kmin = min(kmin, w_mm[w][0]); kmax = max(kmax, w_mm[w][1]);
    jmin = min(jmin, w_mm[w][2]); jmax = max(jmax, w_mm[w][3]);
  }
  const int64_t krange = (E > 0) ? (int64_t)kmax - kmin + 1 : 0;
  const bool bad = E > 0 && (kmin < 0 || krange > k_cap);
  if (t == 0) {
    meta[GM_KMIN] = kmin; meta[GM_KMAX] = kmax; meta[GM_JMIN] = jmin; meta[GM_JMAX] = jmax;
    meta[GM_E] = E;
    meta[GM_ERROR] = bad ? 1 : 0;
    meta[GM_KRANGE] = bad ? 0 : (int32_t)krange;
    if (bad || E == 0) meta[GM_U] = 0;
  }
  const int R = (int)k_cap;
  if (bad || E == 0) {
    for (int i = t; i < R; i += T) khist[i] = 0;   // a failed build leaves a clean histogram too
    return;
  }
  const int b0 = kmin % R;
  const int n = (int)krange;
  const int seg = ((n + nw - 1) / nw + 63) / 64 * 64;   // bins per wave, whole groups of 64
  const int lo = min(wv * seg, n), hi = min(lo + seg, n);
  int32_t sum = 0, cnt = 0;
  for (int i0 = lo; i0 < hi; i0 += 4 * 64) {   // four coalesced loads in flight
    int32_t v[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int i = i0 + 64 * u + lane;
      int bin = b0 + min(i, hi - 1); bin = (bin >= R) ? bin - R : bin;
      v[u] = khist[bin];
    }
#pragma unroll
    for (int u = 0; u < 4; u++)
      if (i0 + 64 * u + lane < hi) { sum += v[u]; cnt += (v[u] > 0); }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { sum += __shfl_xor(sum, o); cnt += __shfl_xor(cnt, o); }
  if (lane == 0) { w_sum[wv] = sum; w_cnt[wv] = cnt; }
  __syncthreads();
  int32_t run = 0, rk = 0, tot = 0, totc = 0;
  for (int w = 0; w < nw; w++) {
    if (w < wv) { run += w_sum[w]; rk += w_cnt[w]; }
    tot += w_sum[w]; totc += w_cnt[w];
  }
  for (int i0 = lo; i0 < hi; i0 += 64) {
    const int i = i0 + lane;
    const bool in = i < hi;
    int bin = b0 + min(i, hi - 1); bin = (bin >= R) ? bin - R : bin;
    const int32_t v = in ? khist[bin] : 0;
    int32_t is = v, ic = (v > 0);
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int32_t a1 = __shfl_up(is, o), c1 = __shfl_up(ic, o);
      if (lane >= o) { is += a1; ic += c1; }
    }
    if (in) {
      khist[bin] = 0;
      kcount[i] = run + is - v;
      krank[i] = rk + ic - (v > 0);
    }
    run += __shfl(is, 63);
    rk += __shfl(ic, 63);
  }
  if (t == 0) { kcount[n] = tot; meta[GM_U] = totc; }
}

// An index build over more than GRAPH_WIDE_EDGES edges (a global bundle adjustment: 10^6 edges, an id range of 10^4 .. 10^5
// bins) leaves the scan of the histogram to a launch of its own with four times the threads (cdv_graph_finish) instead of
// the last workgroup of the histogram launch: hundreds of bins per thread of ONE 256-thread workgroup are hundreds of
// microseconds, and one more launch is nothing there.  The per-frame builds stay at one launch.
constexpr int GRAPH_WIDE_EDGES = 200000;

struct HistArgs {
  const int64_t *jj, *kk;
  int32_t E;
  int32_t* stage;   // [blocks][4] per-workgroup (kmin, kmax, jmin, jmax)
  int32_t* khist;
  int32_t R;
  int32_t *meta, *kcount, *krank;   // for the scan the last workgroup to finish performs
  int32_t* ocnt;    // [blocks][ORD_BINS] this workgroup's edges per target bin (the correlation's processing order)
};

// One workgroup: publish meta, exclusive scan of the histogram in id order (bin of id kmin + i is (kmin + i) mod R)
// -> kcount[i] = dense CSR offset, krank[i] = number of non-empty bins before i; clears the histogram.
__device__ __forceinline__ void graph_scan_body(int32_t* meta, const int32_t* __restrict__ stage, int nstage,
                                                int32_t* khist, int32_t* kcount, int32_t* krank, int32_t E,
                                                int64_t k_cap, int T, int t) {
  constexpr int IMAX = 0x7fffffff, IMIN = (int)0x80000000;
  __shared__ int32_t s_sum[1024];
  __shared__ int32_t s_cnt[1024];
  __shared__ int32_t s_mm[16][4];
  // min / max over the per-workgroup slots of the histogram launch
  int kmin = IMAX, kmax = IMIN, jmin = IMAX, jmax = IMIN;
  for (int i = t; i < nstage; i += T) {   // written through by the other workgroups of this launch: read past the caches
    kmin = min(kmin, __hip_atomic_load(&stage[4 * i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    kmax = max(kmax, __hip_atomic_load(&stage[4 * i + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    jmin = min(jmin, __hip_atomic_load(&stage[4 * i + 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    jmax = max(jmax, __hip_atomic_load(&stage[4 * i + 3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    kmin = min(kmin, __shfl_xor(kmin, o)); kmax = max(kmax, __shfl_xor(kmax, o));
    jmin = min(jmin, __shfl_xor(jmin, o)); jmax = max(jmax, __shfl_xor(jmax, o));
  }
  if ((t & 63) == 0) { s_mm[t >> 6][0] = kmin; s_mm[t >> 6][1] = kmax; s_mm[t >> 6][2] = jmin; s_mm[t >> 6][3] = jmax; }
  __syncthreads();
  for (int w = 0; w < T / 64; w++) {
    kmin = min(kmin, s_mm[w][0]); kmax = max(kmax, s_mm[w][1]);
    jmin = min(jmin, s_mm[w][2]); jmax = max(jmax, s_mm[w][3]);
  }
  const int64_t krange = (E > 0) ? (int64_t)kmax - kmin + 1 : 0;
  const bool bad = E > 0 && (kmin < 0 || krange > k_cap);
  if (t == 0) {
    meta[GM_KMIN] = kmin; meta[GM_KMAX] = kmax; meta[GM_JMIN] = jmin; meta[GM_JMAX] = jmax;
    meta[GM_E] = E;
    meta[GM_ERROR] = bad ? 1 : 0;
    meta[GM_KRANGE] = bad ? 0 : (int32_t)krange;
    if (bad || E == 0) meta[GM_U] = 0;
  }
  const int R = (int)k_cap;
  if (bad || E == 0) {
    for (int i = t; i < R; i += T) khist[i] = 0;   // a failed build leaves a clean histogram too
    return;
  }
  const int b0 = kmin % R;
  const int64_t n = krange;
  const int64_t per = (n + T - 1) / T;
  const int64_t lo = min((int64_t)t * per, n), hi = min(lo + per, n);
  // this thread's bins: loaded once, all loads in flight together (they bypass the caches: ~1 us each), kept in
  // registers for the second sweep when they fit
  constexpr int PERMAX = 24;
  int32_t vals[PERMAX];
  const bool cached = per <= PERMAX;
  int32_t sum = 0, cnt = 0;
  if (cached) {
#pragma unroll
    for (int u = 0; u < PERMAX; u++) {
      const int64_t i = lo + u;
      int bin = b0 + (int)i; bin = (bin >= R) ? bin - R : bin;
      vals[u] = (i < hi) ? __hip_atomic_load(&khist[bin], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    }
#pragma unroll
    for (int u = 0; u < PERMAX; u++) { sum += vals[u]; cnt += (vals[u] > 0); }
  } else {
    // a wide id range (the frame-pair index of a global bundle adjustment: ~10^5 bins, hundreds per thread): sixteen loads in
    // flight at a time -- one by one they were 350 dependent microseconds
    for (int64_t i0 = lo; i0 < hi; i0 += 16) {
      int32_t v[16];
#pragma unroll
      for (int u = 0; u < 16; u++) {
        int bin = b0 + (int)min(i0 + u, hi - 1); bin = (bin >= R) ? bin - R : bin;
        v[u] = __hip_atomic_load(&khist[bin], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
#pragma unroll
      for (int u = 0; u < 16; u++)
        if (i0 + u < hi) { sum += v[u]; cnt += (v[u] > 0); }
    }
  }
  // inclusive scan of the 1024 per-thread partials: shuffle scan inside each wave, then the 16 wave totals
  // (three barriers instead of the twenty of a Hillis-Steele sweep over LDS)
  int32_t isum = sum, icnt = cnt;
  const int lane = t & 63, wv = t >> 6;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int32_t a1 = __shfl_up(isum, o), c1 = __shfl_up(icnt, o);
    if (lane >= o) { isum += a1; icnt += c1; }
  }
  if (lane == 63) { s_sum[wv] = isum; s_cnt[wv] = icnt; }
  __syncthreads();
  if (t < 64) {
    int32_t ws = (t < T / 64) ? s_sum[t] : 0, wc = (t < T / 64) ? s_cnt[t] : 0;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
      const int32_t a1 = __shfl_up(ws, o), c1 = __shfl_up(wc, o);
      if (t >= o) { ws += a1; wc += c1; }
    }
    if (t < T / 64) { s_sum[64 + t] = ws; s_cnt[64 + t] = wc; }   // inclusive totals of waves 0..t
  }
  __syncthreads();
  if (wv > 0) { isum += s_sum[64 + wv - 1]; icnt += s_cnt[64 + wv - 1]; }
  __syncthreads();
  s_sum[t] = isum; s_cnt[t] = icnt;
  __syncthreads();
  int32_t run = s_sum[t] - sum, rk = s_cnt[t] - cnt;
  if (cached) {
#pragma unroll
    for (int u = 0; u < PERMAX; u++) {
      const int64_t i = lo + u;
      if (i < hi) {
        int bin = b0 + (int)i; bin = (bin >= R) ? bin - R : bin;
        khist[bin] = 0;
        kcount[i] = run;
        krank[i] = rk;
        run += vals[u]; rk += (vals[u] > 0);
      }
    }
  } else {
    for (int64_t i0 = lo; i0 < hi; i0 += 16) {
      int32_t v[16];
      int bins[16];
#pragma unroll
      for (int u = 0; u < 16; u++) {
        int bin = b0 + (int)min(i0 + u, hi - 1); bin = (bin >= R) ? bin - R : bin;
        bins[u] = bin;
        v[u] = __hip_atomic_load(&khist[bin], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
#pragma unroll
      for (int u = 0; u < 16; u++) {
        if (i0 + u < hi) {
          khist[bins[u]] = 0;
          kcount[i0 + u] = run;
          krank[i0 + u] = rk;
          run += v[u]; rk += (v[u] > 0);
        }
      }
    }
  }
  if (t == T - 1) { kcount[n] = s_sum[t]; meta[GM_U] = s_cnt[t]; }
}



// One atomic per RUN of equal bins among the consecutive lanes of a wave (consecutive lanes hold consecutive edges): a list
// that keeps the edges of a frame pair together -- the key of the bundle adjustment's pair index -- has runs of 64, and
// same-address atomics serialise at ~90 ns each.  Every lane of the wave must call; `in` false: the lane has no edge.
// Returns the value the run's counter had before (+ the lane's offset in the run): the lane's own slot.
__device__ __forceinline__ int run_atomic_add(int32_t* counters, int bin, bool in, int lane) {
  const int prev = __shfl_up(in ? bin : -1, 1);
  const bool head = in && (lane == 0 || prev != bin);
  const unsigned long long bound = __ballot(head || !in);
  int base = 0;
  if (head) {
    const unsigned long long rest = lane == 63 ? 0ull : bound >> (lane + 1);
    const int len = rest ? __ffsll((long long)rest) : 64 - lane;
    base = atomicAdd(&counters[bin], len);
  }
  const unsigned long long heads_le = __ballot(head) & (~0ull >> (63 - lane));
  const int hl = heads_le ? 63 - __clzll((long long)heads_le) : lane;   // lane of this run's head
  return __shfl(base, in ? hl : lane) + (lane - hl);
}

// The same for ids WITHOUT runs (the frame-pair keys of a global bundle adjustment: consecutive edges go from one patch to a dozen
// target frames, the same dozen for the next patch -- every key of a wave appears several times, never twice in a row): one atomic
// per DISTINCT id of the wave.  First every lane finds its group (the lanes with its id: one ballot per distinct id), its rank in
// it and the group's size; then the groups' first lanes add in ONE atomic instruction; then every lane picks up its group's base.
// (705 k keys over 90 k bins: 64 us of atomics for the histogram launch, 80 for the fill launch, one atomic per edge.)
__device__ __forceinline__ int group_atomic_add(int32_t* counters, int bin, bool in, int lane) {
  unsigned long long todo = __ballot(in);
  int leader = lane, rank = 0, size = 1;
  while (todo) {   // wave-uniform
    const int l = __ffsll((long long)todo) - 1;
    const int b = __shfl(bin, l);
    const unsigned long long same = __ballot(in && bin == b);
    if (in && bin == b) {
      leader = l;
      rank = __popcll(same & ((1ull << lane) - 1ull));
      size = __popcll(same);
    }
    todo &= ~same;
  }
  int base = 0;
  if (in && leader == lane) base = atomicAdd(&counters[bin], size);
  return __shfl(base, leader) + rank;
}

__device__ __forceinline__ void graph_hist_body(const HistArgs& a, int bid, int nblocks, int nthreads_per_block, int tid) {
  constexpr int IMAXV = 0x7fffffff, IMINV = (int)0x80000000;
  int kmin = IMAXV, kmax = IMINV, jmin = IMAXV, jmax = IMINV;
  const int R = a.R;
  const float rinv = 1.0f / (float)R;
  __shared__ int s_obin[ORD_BINS];
  if (tid < ORD_BINS) s_obin[tid] = 0;
  __syncthreads();
  const int hl = tid & 63;
  for (int e = bid * nthreads_per_block + tid; e - hl < a.E; e += nblocks * nthreads_per_block) {   // wave-uniform trip count
    const bool in = e < a.E;
    const int k = in ? (int)a.kk[e] : -1, j = in ? (int)a.jj[e] : 0;
    int m = 0;
    if (in) {
      atomicAdd(&s_obin[j & (ORD_BINS - 1)], 1);
      kmin = min(kmin, k); kmax = max(kmax, k);
      jmin = min(jmin, j); jmax = max(jmax, j);
      // k mod R without an integer division: float quotient estimate, then one correction step each way
      const int q = (int)((float)k * rinv);
      m = k - q * R;
      m = (m < 0) ? m + R : m;
      m = (m >= R) ? m - R : m;
    }
    if (a.E > GRAPH_WIDE_EDGES) group_atomic_add(a.khist, m, in && k >= 0, hl);
    else run_atomic_add(a.khist, m, in && k >= 0, hl);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    kmin = min(kmin, __shfl_xor(kmin, o)); kmax = max(kmax, __shfl_xor(kmax, o));
    jmin = min(jmin, __shfl_xor(jmin, o)); jmax = max(jmax, __shfl_xor(jmax, o));
  }
  __shared__ int s_mm[16][4];
  const int lane = tid & 63, wave = tid >> 6, nw = nthreads_per_block >> 6;
  if (lane == 0) { s_mm[wave][0] = kmin; s_mm[wave][1] = kmax; s_mm[wave][2] = jmin; s_mm[wave][3] = jmax; }
  __syncthreads();
  if (tid < ORD_BINS) a.ocnt[bid * ORD_BINS + tid] = s_obin[tid];   // read by the fill launch
  if (tid == 0) {
    for (int w = 1; w < nw; w++) {
      kmin = min(kmin, s_mm[w][0]); kmax = max(kmax, s_mm[w][1]);
      jmin = min(jmin, s_mm[w][2]); jmax = max(jmax, s_mm[w][3]);
    }
    // one private slot per workgroup: no contended atomics (~90 ns each on one word); written through, because the
    // last workgroup of THIS launch reads them
    int32_t* st = a.stage + 4 * bid;
    __hip_atomic_store(&st[0], kmin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&st[1], kmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&st[2], jmin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&st[3], jmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // ---- last workgroup done -> scan (saves a launch).  Every wave drains its histogram atomics, the workgroup meets,
  // one lane counts the workgroup in (after its own slot stores are acknowledged); whoever sees nblocks - 1 is last.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  __shared__ int s_last;
  if (tid == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int prev = __hip_atomic_fetch_add(&a.meta[GM_STAGE], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = (prev == nblocks - 1) ? 1 : 0;
    if (s_last) __hip_atomic_store(&a.meta[GM_STAGE], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // next build
  }
  __syncthreads();
  if (!s_last) return;
  if (a.E > GRAPH_WIDE_EDGES) return;   // scanned by a launch of its own (see GRAPH_WIDE_EDGES)
  graph_scan_body(a.meta, a.stage, nblocks, a.khist, a.kcount, a.krank, a.E, (int64_t)a.R, nthreads_per_block, tid);
}



// ---- patch TABLE, launch 1 of the two-launch index build (graph.hip: cdv_graph_build_table) -----------------------------
// One pass over the edges, no scan and no second pass behind it: slot = patch id mod R (R = the table's capacity: at least
// the number of ids between the oldest and the newest patch with an edge, so that no two of them share a slot -- checked by
// the sort launch), the t-th arriving edge of a patch takes record t of its slot (chunk-slot layout of `pell`,
// cdv_graph.h), later arrivals go to the overflow list.  Order inside a slot is arrival order here; the sort launch puts
// it into (jj, edge id) order.  Rides cdv_update_prologue_table next to the ring ingest.
struct TFillArgs {
  const int64_t *ii, *jj, *kk;   // ii may be NULL (records then carry -1: the bundle adjustment reads ii itself)
  int32_t E, R;                  // R = capacity (slots) of this build
  int32_t* meta;
  int32_t* tcur;
  unsigned long long* town;      // per slot (1 << 32 | id) of the patch that owns it in the build in flight; zero between builds
  int32_t *ttab, *tovf, *tprec;
  int32_t* ocnt;                 // [blocks][ORD_BINS]
  const int32_t* dyn = nullptr;  // != NULL: the number of edges is dyn[CDV_DYN_E]; E above is an upper bound (launch sizes)
};

__device__ __forceinline__ void graph_tfill_body(const TFillArgs& a_in, int bid, int nblocks, int nthreads_per_block, int tid) {
  TFillArgs a = a_in;
  if (a.dyn) a.E = min(a.dyn[CDV_DYN_E], a_in.E);
  // this build's generation: a word the sort launch of the build before left on the device (nobody writes it during this
  // launch).  Only its parity is used -- which of the two overflow counters / error words belongs to this build.
  const int par = a.meta[GM_GENNEXT] & 1;
  __shared__ int s_obin[ORD_BINS];
  if (tid < ORD_BINS) s_obin[tid] = 0;
  if (bid == 0 && tid == 0) {     // words the sort launch accumulates into; nobody reads them between the two launches
    a.meta[GM_PRECN] = 1;
    a.meta[GM_MODE] = 1; a.meta[GM_GEN] = a.meta[GM_GENNEXT]; a.meta[GM_E] = a.E; a.meta[GM_HAS_II] = a.ii ? 1 : 0; a.meta[GM_TCAP] = a.R;
  }
  __syncthreads();
  typedef int cdv_i4 __attribute__((ext_vector_type(4)));
  const int R = a.R;
  const float rinv = 1.0f / (float)R;
  const int lane = tid & 63;
  for (int e0 = bid * nthreads_per_block; e0 < a.E; e0 += nblocks * nthreads_per_block) {   // workgroup-uniform trips
    const int e = e0 + tid;
    const bool in = e < a.E;
    const int64_t k64 = in ? a.kk[e] : -1;
    const int j = in ? (int)a.jj[e] : 0;
    const int i = (in && a.ii) ? (int)a.ii[e] : -1;
    if (in) atomicAdd(&s_obin[j & (ORD_BINS - 1)], 1);
    const cdv_i4 rec = {e, i, j, (int)k64};
    if (e == 0) *reinterpret_cast<cdv_i4*>(a.tprec) = rec;        // overflow-CSR record 0: the "always valid" record
    const bool ok = in && k64 >= 0 && k64 < ((int64_t)1 << 31);
    if (in && !ok) a.meta[GM_TERR + par] = 1;                     // every writer stores the same value
    // id mod R without an integer division: float quotient estimate, then one correction step each way
    const int k = ok ? (int)k64 : -1 - lane;                      // invalid lanes: ids no neighbour shares
    int q = (int)((float)k * rinv);
    int slot = k - q * R;
    slot = (slot < 0) ? slot + R : slot;
    slot = (slot >= R) ? slot - R : slot;
    // Neighbouring lanes that hold edges of ONE patch (the 13 backward edges of a new patch sit next to each other in the
    // list, slam.py:536-541) take their table positions with ONE atomic of the run's first lane instead of one each on
    // the same address: a run = consecutive lanes with equal id.
    const int kprev = __shfl_up(k, 1);
    const bool head = lane == 0 || k != kprev;
    const unsigned long long heads = __ballot(head);
    const unsigned long long upto = heads & ((2ull << lane) - 1ull);          // heads at or below this lane (lane 63: all)
    const int start = 63 - __clzll((long long)(lane == 63 ? heads : upto));
    const unsigned long long above = lane == 63 ? 0ull : (heads >> (lane + 1)) << (lane + 1);
    const int nexth = above ? __ffsll((long long)above) - 1 : 64;
    int base = 0;
    if (ok && head) {
      // one patch per slot: the slot's owner word takes (1, id) by an atomic max -- whoever finds another id there has met a
      // second patch in the slot.  The sort launch zeroes the words of the slots it finds in use, next to their cursors.
      const unsigned long long mine = (1ull << 32) | (uint32_t)k;
      const unsigned long long was = atomicMax(&a.town[slot], mine);
      if (was != 0ull && (uint32_t)was != (uint32_t)k) a.meta[GM_TERR + par] = 1;
      base = atomicAdd(&a.tcur[slot], nexth - start);
    }
    base = __shfl(base, start);
    if (!ok) continue;
    const int t = base + (lane - start);
    if (t >= TAB_MAX_DEG) a.meta[GM_TERR + par] = 1;             // more edges than the sort launch serves: known before it starts
    if (t < ELL_SLOTS) {
      *reinterpret_cast<cdv_i4*>(a.ttab + 4 * ((size_t)((slot >> 4) * ELL_SLOTS + t) * 16 + (slot & 15))) = rec;
    } else {
      const int p = atomicAdd(&a.meta[GM_OVFN + par], 1);
      *reinterpret_cast<cdv_i4*>(a.tovf + 4 * (size_t)p) = rec;
    }
  }
  __syncthreads();
  if (tid < ORD_BINS) a.ocnt[bid * ORD_BINS + tid] = s_obin[tid];   // read by the sort launch (the correlation's order)
}

}  // namespace cdv
