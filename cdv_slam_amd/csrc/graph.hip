// graph.hip -- device-side patch-graph index, gfx950.
//
// Replaces, without any host round trip:
//   torch::_unique(kk, sorted, inverse)          cdvslam/fastba/ba_cuda.cu:476-478, ba.cpp:62
//   the CPU bucket + std::stable_sort loops of   cdvslam/fastba/ba.cpp:59-97 (neighbors)
// All integer work -- a counting sort over the dense patch-id range -- bit-exact and deterministic:
//   kx [U]   sorted unique patch ids           ku [E] inverse index
//   patch CSR (koff_u [U+1], pcsr [E]): the edges of every unique patch ordered by (jj, edge id),
//   i.e. the order std::stable_sort by jj gives on an ascending index list.
// (An edge order grouped by target frame was tried as the correlation kernel's processing order, with and
// without giving every XCD a contiguous share: no measurable gain -- that kernel is not bound by L2 misses --
// so it is not built.)
//
// Pipeline, 3 small launches, no host sync (every launch costs ~5 us of latency on an otherwise idle GPU, the
// work itself is a few hundred KB):
//   1 hist     : count[k mod R]++ (R = workspace capacity; the ids of one build span less than R, checked) and
//                min/max of kk, jj (one atomic per block).  Indexing by k mod R needs no kmin, so the min/max pass
//                and the histogram are ONE launch; the histogram is zero between builds (the scan clears it).
//     scan     : by the LAST workgroup of launch 1 to finish (arrival counter; the others' histogram atomics and
//                min/max slots are drained and written through before they count themselves in): validate the
//                range, publish meta, exclusive scan in id order (starting at bin kmin mod R) -> dense offsets by
//                id - kmin and unique ranks; clears the histogram
//   2 fill     : kx / koff_u from the dense bins, ku and the (unordered) CSR slots per edge
//   3 segsort  : rank every edge inside its patch segment by (jj, edge id); the same sweep finds the edge's
//                predecessor / successor in time (fastba.neighbors); clears the fill cursors
#include <mutex>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "cdv_common.h"
#include "cdv_graph.h"
#include "cdv_parts.h"

using namespace cdv;

CDV_STAMP_TU(graph)

namespace {

struct RegEntry {
  GraphLayout L;
  bool initialised;
  bool has_ii;
  CorrStream cs;      // coords == nullptr: no packed correlation stream is written
  bool table;         // the index in the workspace is a patch table
  int32_t tab_cap;    // ... of this capacity (slots)
};
std::mutex g_reg_mutex;
std::unordered_map<const void*, RegEntry> g_registry;
std::unordered_set<const void*> g_no_order;   // workspaces whose builds skip the correlation's processing order (cdv_graph_no_corr_order)


__global__ __launch_bounds__(256) void graph_init_kernel(int32_t* meta, int32_t* khist, int32_t* kcursor, int32_t* tcur,
                                                         unsigned long long* town, int64_t k_cap) {
  const int64_t n = k_cap + 1 > GM_WORDS ? k_cap + 1 : GM_WORDS;   // the meta words too when the id range is tiny
  const int64_t tcap = (k_cap < TAB_CAP_MAX ? k_cap : TAB_CAP_MAX) + 16;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
    if (t <= k_cap) {
      khist[t] = 0;
      kcursor[t] = 0;
      if (t < tcap) { tcur[t] = 0; town[t] = 0ull; }
    }
    if (t < GM_WORDS) {
      meta[t] = 0;   // incl. the arrival counter of the histogram launch (GM_STAGE)
    }
  }
}

// histogram over (id mod R) + min / max of kk, jj: body in cdv_parts.h
__global__ __launch_bounds__(256) void graph_hist_kernel(cdv::HistArgs a) {
  cdv::graph_hist_body(a, (int)blockIdx.x, (int)gridDim.x, (int)blockDim.x, (int)threadIdx.x);
}

// exclusive scan of the histogram as a launch of its own: when there are no edges (no histogram launch whose last
// workgroup would do it), and for the wide builds (GRAPH_WIDE_EDGES) -- body in cdv_parts.h
__global__ __launch_bounds__(1024) void graph_scan_kernel(int32_t* meta, const int32_t* __restrict__ stage,
                                                          int nstage, int32_t* khist, int32_t* kcount,
                                                          int32_t* krank, int32_t E, int64_t k_cap) {
  cdv::graph_scan_wide_body(meta, stage, nstage, khist, kcount, krank, E, k_cap, (int)blockDim.x, (int)threadIdx.x);
}

// The scan of a WIDE build (GRAPH_WIDE_EDGES) over several workgroups, two launches: (1) every workgroup sums its slice of the
// histogram (edges, non-empty bins) into part[2 w], part[2 w + 1]; (2) every workgroup adds up the slices before its own, scans
// its slice from there (CSR offsets by id - kmin, unique ranks), re-zeroes the histogram; workgroup 0 writes the meta words.  One
// workgroup of 1024 threads walking the 90 k frame-pair keys of a global bundle adjustment was 69 us of dependent memory round trips.
constexpr int SCAN_WG = 64, SCAN_T = 256;
__device__ __forceinline__ void scan_wide_range(const int32_t* __restrict__ stage, int nstage, int32_t E, int64_t k_cap, int t,
                                                int (&mm)[4], bool& bad, int& n) {
  constexpr int IMAX = 0x7fffffff, IMIN = (int)0x80000000;
  __shared__ int s_mm[SCAN_T / 64][4];
  int kmin = IMAX, kmax = IMIN, jmin = IMAX, jmax = IMIN;
  for (int i = t; i < nstage; i += SCAN_T) {
    kmin = min(kmin, stage[4 * i]); kmax = max(kmax, stage[4 * i + 1]);
    jmin = min(jmin, stage[4 * i + 2]); jmax = max(jmax, stage[4 * i + 3]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    kmin = min(kmin, __shfl_xor(kmin, o)); kmax = max(kmax, __shfl_xor(kmax, o));
    jmin = min(jmin, __shfl_xor(jmin, o)); jmax = max(jmax, __shfl_xor(jmax, o));
  }
  if ((t & 63) == 0) { s_mm[t >> 6][0] = kmin; s_mm[t >> 6][1] = kmax; s_mm[t >> 6][2] = jmin; s_mm[t >> 6][3] = jmax; }
  __syncthreads();
  for (int w = 0; w < SCAN_T / 64; w++) {
    kmin = min(kmin, s_mm[w][0]); kmax = max(kmax, s_mm[w][1]);
    jmin = min(jmin, s_mm[w][2]); jmax = max(jmax, s_mm[w][3]);
  }
  mm[0] = kmin; mm[1] = kmax; mm[2] = jmin; mm[3] = jmax;
  const int64_t krange = (E > 0) ? (int64_t)kmax - kmin + 1 : 0;
  bad = E > 0 && (kmin < 0 || krange > k_cap);
  n = (int)krange;
}
__device__ __forceinline__ int scan_wide_slice(int n) { return ((n + SCAN_WG - 1) / SCAN_WG + SCAN_T - 1) / SCAN_T * SCAN_T; }

__global__ __launch_bounds__(SCAN_T) void graph_scan_part_kernel(const int32_t* __restrict__ stage, int nstage, const int32_t* __restrict__ khist,
                                                                  int32_t* __restrict__ part, int32_t E, int64_t k_cap) {
  const int t = threadIdx.x;
  int mm[4], n; bool bad;
  scan_wide_range(stage, nstage, E, k_cap, t, mm, bad, n);
  __shared__ int s_sum[SCAN_T / 64], s_cnt[SCAN_T / 64];
  int32_t sum = 0, cnt = 0;
  if (!bad && E > 0) {
    const int R = (int)k_cap, b0 = mm[0] % R, S = scan_wide_slice(n);
    const int lo = min((int)blockIdx.x * S, n), hi = min(lo + S, n);
    for (int i = lo + t; i < hi; i += SCAN_T) {
      int bin = b0 + i; bin = (bin >= R) ? bin - R : bin;
      const int32_t v = khist[bin];
      sum += v; cnt += (v > 0);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { sum += __shfl_xor(sum, o); cnt += __shfl_xor(cnt, o); }
  if ((t & 63) == 0) { s_sum[t >> 6] = sum; s_cnt[t >> 6] = cnt; }
  __syncthreads();
  if (t == 0) {
    int32_t a = 0, c = 0;
    for (int w = 0; w < SCAN_T / 64; w++) { a += s_sum[w]; c += s_cnt[w]; }
    part[2 * blockIdx.x] = a; part[2 * blockIdx.x + 1] = c;
  }
}

__global__ __launch_bounds__(SCAN_T) void graph_scan_apply_kernel(int32_t* meta, const int32_t* __restrict__ stage, int nstage, int32_t* khist,
                                                                   int32_t* kcount, int32_t* krank, const int32_t* __restrict__ part,
                                                                   int32_t E, int64_t k_cap) {
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  int mm[4], n; bool bad;
  scan_wide_range(stage, nstage, E, k_cap, t, mm, bad, n);
  const int R = (int)k_cap;
  if (blockIdx.x == 0 && t == 0) {
    meta[GM_KMIN] = mm[0]; meta[GM_KMAX] = mm[1]; meta[GM_JMIN] = mm[2]; meta[GM_JMAX] = mm[3];
    meta[GM_E] = E;
    meta[GM_ERROR] = bad ? 1 : 0;
    meta[GM_KRANGE] = bad ? 0 : n;
    if (bad || E == 0) meta[GM_U] = 0;
  }
  if (bad || E == 0) {   // a failed build leaves a clean histogram too
    for (int i = blockIdx.x * SCAN_T + t; i < R; i += gridDim.x * SCAN_T) khist[i] = 0;
    return;
  }
  // what lies before this workgroup's slice, and the totals
  __shared__ int s_a[SCAN_T / 64][4];
  int32_t run = 0, rk = 0, tot = 0, totc = 0;
  for (int w = t; w < SCAN_WG; w += SCAN_T) {
    const int32_t a = part[2 * w], c = part[2 * w + 1];
    tot += a; totc += c;
    if (w < (int)blockIdx.x) { run += a; rk += c; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    run += __shfl_xor(run, o); rk += __shfl_xor(rk, o); tot += __shfl_xor(tot, o); totc += __shfl_xor(totc, o);
  }
  if (lane == 0) { s_a[wv][0] = run; s_a[wv][1] = rk; s_a[wv][2] = tot; s_a[wv][3] = totc; }
  __syncthreads();
  run = rk = tot = totc = 0;
  for (int w = 0; w < SCAN_T / 64; w++) { run += s_a[w][0]; rk += s_a[w][1]; tot += s_a[w][2]; totc += s_a[w][3]; }
  if (blockIdx.x == 0 && t == 0) { kcount[n] = tot; meta[GM_U] = totc; }
  const int b0 = mm[0] % R, S = scan_wide_slice(n);
  const int lo = min((int)blockIdx.x * S, n), hi = min(lo + S, n);
  __shared__ int s_ws[SCAN_T / 64], s_wc[SCAN_T / 64];
  for (int i0 = lo; i0 < hi; i0 += SCAN_T) {   // workgroup-uniform
    const int i = i0 + t;
    const bool in = i < hi;
    int bin = b0 + min(i, hi - 1); bin = (bin >= R) ? bin - R : bin;
    const int32_t v = in ? khist[bin] : 0;
    int32_t is = v, ic = (v > 0);
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int32_t a1 = __shfl_up(is, o), c1 = __shfl_up(ic, o);
      if (lane >= o) { is += a1; ic += c1; }
    }
    __syncthreads();   // (the slots below are read by the previous round until here)
    if (lane == 63) { s_ws[wv] = is; s_wc[wv] = ic; }
    __syncthreads();
    int32_t wb = 0, wc = 0, rt = 0, rc = 0;
    for (int w = 0; w < SCAN_T / 64; w++) {
      if (w < wv) { wb += s_ws[w]; wc += s_wc[w]; }
      rt += s_ws[w]; rc += s_wc[w];
    }
    if (in) {
      khist[bin] = 0;
      kcount[i] = run + wb + is - v;
      krank[i] = rk + wc + ic - (v > 0);
    }
    run += rt; rk += rc;
  }
}

// index % modulus with the host's reciprocal (as the correlation kernel reduces kk / jj itself when it reads them)
__device__ __forceinline__ bool corr_ring_index(int64_t v64, uint32_t mod, uint32_t magic, uint32_t limit, uint32_t& q) {
  q = (uint32_t)v64;
  const bool small = (uint32_t)(v64 >> 32) == 0u && q < 0x80000000u;
  if (mod > 1u) {
    const int32_t r = (int32_t)(q - __umulhi(q, magic) * mod);
    q = (uint32_t)(r < 0 ? r + (int32_t)mod : r);
  } else if (mod == 1u) {
    q = 0u;
  }
  return small && q < limit;
}

__global__ __launch_bounds__(256) void graph_fill_kernel(const int64_t* __restrict__ kk, int32_t E,
                                                         const int32_t* __restrict__ meta,
                                                         const int32_t* __restrict__ kcount, int32_t* kcursor,
                                                         const int32_t* __restrict__ krank,
                                                         int32_t* __restrict__ koff_u, int64_t* __restrict__ kx,
                                                         int32_t* __restrict__ ku, int32_t* __restrict__ pcsr_tmp,
                                                         const int64_t* __restrict__ jj,
                                                         const int32_t* __restrict__ ocnt, int nblk_hist,
                                                         int32_t* __restrict__ order, const CorrStream cs,
                                                         uint32_t* __restrict__ crec) {
  // ---- the correlation's processing order (independent of the patch index and of its error state): a counting sort
  // of the edges by target bin.  The histogram launch left every workgroup's count per bin; this workgroup handles the
  // same edges, so its first position in bin b is (edges of the bins before b) + (edges of bin b in the workgroups
  // before it) -- summed here from the whole table (6 K words, L2-resident: cheaper than a scan pass of its own), the
  // rank inside the workgroup from an LDS counter.
  const bool do_order = order && nblk_hist == (int)gridDim.x;
  __shared__ int s_pos[ORD_BINS];
  // this thread's first edge: requested before the order computation (whose barriers the loads would not cross)
  const int t_first = blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t kk_first = t_first < E ? kk[t_first] : 0;
  const int64_t jj_first = t_first < E ? jj[t_first] : 0;
  // ... and its slot in the patch's list: the cursor's round trip runs under the order computation as well
  const int err = meta[GM_ERROR];
  const int krange = meta[GM_KRANGE], kmin = meta[GM_KMIN];
  int d_first = 0, p_first = 0, kc_first = 0, kr_first = 0;
  const int flane = threadIdx.x & 63;
  if (!err) {   // (workgroup-uniform; one cursor atomic per run of equal ids among the wave's consecutive edges)
    d_first = t_first < E ? (int)kk_first - kmin : 0;
    p_first = (E > cdv::GRAPH_WIDE_EDGES) ? cdv::group_atomic_add(kcursor, d_first, t_first < E, flane)
                                          : cdv::run_atomic_add(kcursor, d_first, t_first < E, flane);
    if (t_first < E) {
      kc_first = kcount[d_first];
      kr_first = krank[d_first];
    }
  }
  if (do_order) {
    __shared__ int s_tot[32][ORD_BINS + 1], s_pre[32][ORD_BINS + 1];
    const int tid = threadIdx.x, bq = tid & 7, part = tid >> 3;   // 256 threads: 8 groups of four bins x 32 parts
    int tot[4] = {0, 0, 0, 0}, pre[4] = {0, 0, 0, 0};
    const int4* tab = reinterpret_cast<const int4*>(ocnt);
    for (int b0 = part; b0 < nblk_hist; b0 += 32 * 4) {   // up to four 16-byte loads in flight
      int4 c[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int b = b0 + 32 * u;
        c[u] = (b < nblk_hist) ? tab[b * (ORD_BINS / 4) + bq] : int4{0, 0, 0, 0};
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const bool before = b0 + 32 * u < (int)blockIdx.x;
        tot[0] += c[u].x; tot[1] += c[u].y; tot[2] += c[u].z; tot[3] += c[u].w;
        pre[0] += before ? c[u].x : 0; pre[1] += before ? c[u].y : 0; pre[2] += before ? c[u].z : 0; pre[3] += before ? c[u].w : 0;
      }
    }
#pragma unroll
    for (int q = 0; q < 4; q++) { s_tot[part][4 * bq + q] = tot[q]; s_pre[part][4 * bq + q] = pre[q]; }
    __syncthreads();
    if (tid < ORD_BINS) {
      int t = 0, p = 0;
#pragma unroll
      for (int q = 0; q < 32; q++) { t += s_tot[q][tid]; p += s_pre[q][tid]; }
      int incl = t;   // inclusive scan of the bin totals over the 32 lanes
#pragma unroll
      for (int o = 1; o < ORD_BINS; o <<= 1) {
        const int up = __shfl_up(incl, o);
        if (tid >= o) incl += up;
      }
      s_pos[tid] = incl - t + p;
    }
    __syncthreads();
    for (int e = t_first; e < E; e += gridDim.x * blockDim.x) {
      const int64_t j64 = e == t_first ? jj_first : jj[e];
      const int b = (int)j64 & (ORD_BINS - 1);
      const int pos = atomicAdd(&s_pos[b], 1);
      order[pos] = e;
      if (cs.coords) {
        // the correlation's packed input record of this edge, at its processing position: coordinates + reduced ring
        // indices (slam.py:319-320), so that the correlation wave needs ONE memory round trip before its window loads
        const int64_t k64 = e == t_first ? kk_first : kk[e];
        uint32_t kq, jq;
        const bool ok = corr_ring_index(k64, cs.kmod, cs.kmagic, cs.Ng, kq) & corr_ring_index(j64, cs.jmod, cs.jmagic, cs.slots, jq);
        const float2* c2 = reinterpret_cast<const float2*>(cs.coords + (size_t)e * 18);   // 72 e: 8-byte aligned
        uint32_t* r = crec + (size_t)pos * CORR_REC_WORDS;
        float2 c[9];
#pragma unroll
        for (int i = 0; i < 9; i++) c[i] = c2[i];
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const auto bits = [](float f) { return (uint32_t)__float_as_int(f); };
        u32x4* r4 = reinterpret_cast<u32x4*>(r);
        r4[0] = u32x4{bits(c[0].x), bits(c[0].y), bits(c[1].x), bits(c[1].y)};
        r4[1] = u32x4{bits(c[2].x), bits(c[2].y), bits(c[3].x), bits(c[3].y)};
        r4[2] = u32x4{bits(c[4].x), bits(c[4].y), bits(c[5].x), bits(c[5].y)};
        r4[3] = u32x4{bits(c[6].x), bits(c[6].y), bits(c[7].x), bits(c[7].y)};
        r4[4] = u32x4{bits(c[8].x), bits(c[8].y), (uint32_t)e, ok ? kq : 0xFFFFFFFFu};
        // extremes of floor(coordinate / scale of level 0) over the nine pixels, clamped to 16 bits (a box that far out
        // is off the map either way): the correlation's window boxes without a cross-lane reduction
        const auto fl = [&](float v) { return (int)fminf(fmaxf(floorf(v * cs.inv_scale0), -30000.f), 30000.f); };
        int xlo = 30000, xhi = -30000, ylo = 30000, yhi = -30000;      // x = values 0..8, y = values 9..17
#pragma unroll
        for (int i = 0; i < 18; i++) {
          const int v = fl((i & 1) ? c[i >> 1].y : c[i >> 1].x);       // per value, like the kernel's own reduction (NaN -> far out)
          if (i < 9) { xlo = min(xlo, v); xhi = max(xhi, v); } else { ylo = min(ylo, v); yhi = max(yhi, v); }
        }
        const uint32_t bxw = ((uint32_t)xhi << 16) | ((uint32_t)xlo & 0xffffu);
        const uint32_t byw = ((uint32_t)yhi << 16) | ((uint32_t)ylo & 0xffffu);
        r4[5] = u32x4{ok ? jq : 0xFFFFFFFFu, bxw, byw, 0u};
      }
    }
  }
  if (err) return;
  const int n = max(E, krange + 1);
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t - flane < n; t += gridDim.x * blockDim.x) {   // wave-uniform trip count
    const bool first = t == t_first;   // (the whole wave's first trip, or nobody's)
    int p_run = 0;
    if (!first) p_run = (E > cdv::GRAPH_WIDE_EDGES) ? cdv::group_atomic_add(kcursor, t < E ? (int)kk[t] - kmin : 0, t < E, flane)
                                                    : cdv::run_atomic_add(kcursor, t < E ? (int)kk[t] - kmin : 0, t < E, flane);
    if (t <= krange) {  // dense bins -> unique ranks
      if (t == krange) {
        koff_u[meta[GM_U]] = kcount[krange];
      } else if (kcount[t + 1] > kcount[t]) {
        const int r = krank[t];
        kx[r] = (int64_t)(kmin + t);
        koff_u[r] = kcount[t];
      }
    }
    if (t < E) {
      const int d = first ? d_first : (int)kk[t] - kmin;
      const int jt = first ? (int)jj_first : (int)jj[t];
      const int p = first ? p_first : p_run;
      // what the segment sort needs of the edge, in one 16-byte entry: it then walks the lists without going back to
      // kk / jj (three dependent load levels instead of five)
      *reinterpret_cast<int4*>(pcsr_tmp + 4 * (size_t)((first ? kc_first : kcount[d]) + p)) = make_int4(t, jt, d, 0);
      ku[t] = first ? kr_first : krank[d];
    }
  }
}

// Deterministic order inside every patch segment: rank by (jj, edge id) == std::stable_sort by jj over an
// ascending index list (ba.cpp:84-86).  O(d^2) per segment, d ~ 25 in a SLAM graph.  The same sweep yields the
// edge's neighbours in time (ba.cpp:88-94): the largest key below its own and the smallest key above.
__global__ __launch_bounds__(256) void graph_segsort_kernel(const int64_t* __restrict__ ii,
                                                            const int64_t* __restrict__ jj,
                                                            const int64_t* __restrict__ kk, int32_t E,
                                                            int32_t* __restrict__ meta,
                                                            const int32_t* __restrict__ kcount,
                                                            const int32_t* __restrict__ pcsr_tmp,
                                                            int32_t* __restrict__ pcsr, int32_t* __restrict__ prec,
                                                            int32_t* __restrict__ pell, int ell_chunks,
                                                            const int32_t* __restrict__ krank,
                                                            int32_t* __restrict__ nprev,
                                                            int32_t* __restrict__ nnext, int32_t* __restrict__ kcursor,
                                                            int64_t* __restrict__ ix, int64_t* __restrict__ jx) {
  if (meta[GM_ERROR]) {   // no index: the neighbor lists say "none" instead of staying uninitialised
    if (ix)
      for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < E; e += gridDim.x * blockDim.x) { ix[e] = -1; jx[e] = -1; }
    return;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) { meta[GM_HAS_II] = ii ? 1 : 0; meta[GM_MODE] = 0; }
  const int krange = meta[GM_KRANGE];
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t <= krange; t += gridDim.x * blockDim.x)
    kcursor[t] = 0;   // the fill cursors of this build: zero again for the next one
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < E; p += gridDim.x * blockDim.x) {
    const int4 me = *reinterpret_cast<const int4*>(pcsr_tmp + 4 * (size_t)p);    // {edge, jj, id - kmin, 0} from the fill
    const int e = me.x, d = me.z;
    const int lo = kcount[d], hi = kcount[d + 1];
    // one 64-bit key per edge: (jj, edge id), both below 2^31
    const uint64_t ke = ((uint64_t)(uint32_t)me.y << 32) | (uint32_t)e;
    int r = 0;
    uint64_t pk = 0, nk = ~(uint64_t)0;   // best predecessor / successor key so far (sentinels: none)
    // (sixteen entries of the list requested together: a frame pair of a global bundle adjustment has ~200 edges, and four at a
    // time were 48 dependent memory round trips per thread -- 81 us for this launch at 705 k edges)
    constexpr int SU = 16;
    for (int s0 = lo; s0 < hi; s0 += SU) {
      uint64_t ko[SU];
#pragma unroll
      for (int u = 0; u < SU; u++) {
        const int2 o = *reinterpret_cast<const int2*>(pcsr_tmp + 4 * (size_t)min(s0 + u, hi - 1));
        ko[u] = ((uint64_t)(uint32_t)o.y << 32) | (uint32_t)o.x;
      }
#pragma unroll
      for (int u = 0; u < SU; u++) {
        const bool in = s0 + u < hi;
        const bool below = in && ko[u] < ke, above = in && ko[u] > ke;
        r += below;
        pk = (below && (ko[u] >= pk)) ? ko[u] : pk;
        nk = (above && (ko[u] <= nk)) ? ko[u] : nk;
      }
    }
    const bool hasp = r > 0, hasn = lo + r + 1 < hi;
    const int pe = hasp ? (int)(uint32_t)pk : -1, ne = hasn ? (int)(uint32_t)nk : -1;
    pcsr[lo + r] = e;
    // the record the bundle adjustment walks: edge id, source frame, target frame in ONE 16-byte load
    const int4 record = make_int4(e, ii ? (int)ii[e] : -1, (int)(ke >> 32), 0);
    *reinterpret_cast<int4*>(prec + 4 * (size_t)(lo + r)) = record;
    const int u = krank[d];                    // unique rank of the patch
    if (r < ELL_SLOTS && (u >> 4) < ell_chunks)
      *reinterpret_cast<int4*>(pell + 4 * ((size_t)((u >> 4) * ELL_SLOTS + r) * 16 + (u & 15))) = record;
    nprev[e] = pe;             // kept in the workspace for a later cdv_neighbors
    nnext[e] = ne;
    if (ix) {
      ix[e] = (int64_t)pe;     // previous edge in time (-1: none)   ba.cpp:88-94
      jx[e] = (int64_t)ne;     // next edge in time
    }
  }
}

__global__ __launch_bounds__(256) void graph_neighbors_kernel(int32_t E, const int32_t* __restrict__ meta,
                                                              const int32_t* __restrict__ nprev,
                                                              const int32_t* __restrict__ nnext,
                                                              int64_t* __restrict__ ix, int64_t* __restrict__ jx) {
  const bool bad = graph_error(meta) != 0;   // no index: "none" everywhere instead of uninitialised memory
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < E; e += gridDim.x * blockDim.x) {
    if (bad) { ix[e] = -1; jx[e] = -1; continue; }
    ix[e] = (int64_t)nprev[e];   // previous edge in time (-1: none)
    jx[e] = (int64_t)nnext[e];   // next edge in time
  }
}

__global__ __launch_bounds__(256) void graph_copy_unique_kernel(const int32_t* __restrict__ meta,
                                                                const int64_t* __restrict__ kx_src,
                                                                const int32_t* __restrict__ ku_src,
                                                                int64_t* __restrict__ kx, int64_t kx_cap,
                                                                int64_t* __restrict__ ku, int32_t E) {
  if (meta[GM_ERROR]) return;
  const int U = meta[GM_U];
  const int n = max(U, E);
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
    if (kx && t < U && t < kx_cap) kx[t] = kx_src[t];
    if (ku && t < E) ku[t] = (int64_t)ku_src[t];
  }
}

// ---- patch TABLE, launch 2: sort every slot into (jj, edge id) order (== std::stable_sort by jj over an ascending index
// list, ba.cpp:84-86), neighbors, live range; and, by further workgroups of the same launch, everything that is per EDGE
// and wants the processing order: the correlation's order and packed input stream, with the edge's reprojection
// (projective_ops.py:53-113) computed right there when the caller asks for it (cdv_update_prologue_table) ------------
struct TSortArgs {
  int32_t* meta;
  int32_t R, E;
  int32_t *tcur, *tdeg, *tplo, *tkid, *tlive, *ttab, *tovf, *tprec;
  unsigned long long* town;
  int32_t *nprev, *nnext;
  int64_t *ix, *jx;
  int n_patch_wg;                 // workgroups [0, n_patch_wg) take 8 slots each, the rest are edge workgroups
  const int32_t* ocnt;
  int nblk_edges;
  int32_t* order;
  const int64_t *ii, *jj, *kk;
  CorrStream cs;                  // ring sizes of the packed stream; cs.coords: where finished coordinates are read from
  uint32_t* crec;                 // NULL: no packed stream
  const float *poses, *patches, *intr;   // poses != NULL: reproject here and write coords_out [E][2][3][3]
  float* coords_out;
  const int32_t* dyn;             // != NULL: the number of edges is dyn[CDV_DYN_E], E is an upper bound
};

constexpr int TS_OVF_MAX = TAB_MAX_DEG;   // edges of one patch the sort launch handles (beyond: range-error state)

typedef int cdv_i4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ size_t tab_index(int slot, int t) {   // int32 index of record t of a slot (chunk-slot layout)
  return 4 * ((size_t)((slot >> 4) * ELL_SLOTS + t) * 16 + (slot & 15));
}

__device__ __forceinline__ uint64_t rec_key(const cdv_i4& r) { return ((uint64_t)(uint32_t)r.z << 32) | (uint32_t)r.x; }

// reprojection of one edge: the arithmetic of transform_body<3> (cdv_parts.h), coordinates only
__device__ __forceinline__ void reproject_edge(const TSortArgs& A, int64_t ix, int64_t jx, int64_t kx, float (&cx)[9],
                                               float (&cy)[9]) {
  const float* __restrict__ poses = A.poses;
  const float* __restrict__ intr = A.intr;
  float G[7], t[3], q[4];
  tf_relative(poses, ix, jx, false, G, t, q);
  const cdv_float4 Ki = *reinterpret_cast<const cdv_float4*>(intr + 4 * ix);
  const cdv_float4 Kj = *reinterpret_cast<const cdv_float4*>(intr + 4 * jx);
  // the patch: 27 consecutive floats, 4-byte aligned -- seven wide loads instead of 27 scalar ones
  typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
  const float* pk = A.patches + kx * 27;
  float pv[28];
#pragma unroll
  for (int v = 0; v < 6; v++) {
    const f4u x4 = *reinterpret_cast<const f4u*>(pk + 4 * v);
    pv[4 * v] = x4[0]; pv[4 * v + 1] = x4[1]; pv[4 * v + 2] = x4[2]; pv[4 * v + 3] = x4[3];
  }
  pv[24] = pk[24]; pv[25] = pk[25]; pv[26] = pk[26];
#pragma unroll
  for (int a = 0; a < 9; a++) {
    float X1[4];
    tf_pixel(t, q, pv[a], pv[9 + a], pv[18 + a], Ki[0], Ki[1], Ki[2], Ki[3], Kj[0], Kj[1], Kj[2], Kj[3], cx[a], cy[a], X1);
  }
}

__global__ __launch_bounds__(256) void graph_tsort_kernel(const TSortArgs A_in) {
  TSortArgs A = A_in;
  if (A.dyn) A.E = min(A.dyn[CDV_DYN_E], A_in.E);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  CDV_IF_STAMPS(const int sslot = (int)blockIdx.x;)
  if (tid < 64) { CDV_STAMP_RT(graph, sslot, 0); CDV_STAMP(graph, sslot, 1); }
  // this build's generation and its error word, as the fill launch left them (nothing in this launch depends on host state
  // that a captured hipGraph would freeze)
  const int gen = A.meta[GM_GEN], par = gen & 1;
  const bool terr = A.meta[GM_TERR + par] != 0;
  if ((int)blockIdx.x >= A.n_patch_wg) {
    // =================================== edge workgroups ===================================
    const int bid = (int)blockIdx.x - A.n_patch_wg;
    const int nblk = A.nblk_edges;
    __shared__ int s_pos[ORD_BINS];
    __shared__ int s_tot[32][ORD_BINS + 1], s_pre[32][ORD_BINS + 1];
    __shared__ __attribute__((aligned(16))) float s_xy[256 * 18];
    // Three things this workgroup needs before it can write anything, none of which depends on another: the count table of
    // the fill launch (-> its first position per target bin), its first tile's edges (kk, jj, ii), and -- behind those --
    // the poses / intrinsics / patch of every edge.  They are requested in THAT order and waited for in the order of use:
    // the table's 16-byte loads and the edge loads go out together, the reprojection (loads of its own, ~2.5k cycles of
    // arithmetic) runs while the table is in flight, the table is summed afterwards.  (Round 5, stamps: the edge workgroups
    // are this launch's long pole -- they ended at 8.4 us, the per-slot workgroups at 2.9 -- and spent 4.3k cycles on the
    // table before their first edge load went out.)
    const int bq = tid & 7, part = tid >> 3;
    const cdv_i4* tab = reinterpret_cast<const cdv_i4*>(A.ocnt);
    cdv_i4 c0[8];      // the first 256 rows of the table (all of it up to 256 fill workgroups)
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int b = part + 32 * u;
      c0[u] = (b < nblk) ? tab[b * (ORD_BINS / 4) + bq] : cdv_i4{0, 0, 0, 0};
    }
    // the first tile
    const auto load_edge = [&](int e, bool in, float (&cx)[9], float (&cy)[9], int64_t& k64, int64_t& j64) {
      k64 = 0; j64 = 0;
      if (in) {
        k64 = A.kk[e]; j64 = A.jj[e];
        if (A.poses) {
          reproject_edge(A, A.ii[e], j64, k64, cx, cy);
#pragma unroll
          for (int a = 0; a < 9; a++) { s_xy[tid * 18 + a] = cx[a]; s_xy[tid * 18 + 9 + a] = cy[a]; }
        } else if (A.cs.coords && A.crec) {
          const float2* c2 = reinterpret_cast<const float2*>(A.cs.coords + (size_t)e * 18);
#pragma unroll
          for (int a = 0; a < 9; a++) {
            const float2 v = c2[a];
            if (2 * a < 9) cx[2 * a] = v.x; else cy[2 * a - 9] = v.x;
            if (2 * a + 1 < 9) cx[2 * a + 1] = v.y; else cy[2 * a + 1 - 9] = v.y;
          }
        }
        if (terr && A.ix) { A.ix[e] = -1; A.jx[e] = -1; }         // no index: "none", not uninitialised memory
      }
    };
    float cx[9], cy[9];
    int64_t k64 = 0, j64 = 0;
    const int e_first = bid * 256;
    load_edge(e_first + tid, e_first + tid < A.E, cx, cy, k64, j64);
    // this workgroup's first position per target bin: (edges of the bins before) + (edges of this bin in the workgroups
    // before it), summed from the per-workgroup counts the fill launch left
    {
      int tot[4] = {0, 0, 0, 0}, pre[4] = {0, 0, 0, 0};
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const bool before = part + 32 * u < bid;
#pragma unroll
        for (int q = 0; q < 4; q++) { tot[q] += c0[u][q]; pre[q] += before ? c0[u][q] : 0; }
      }
      for (int b0 = part + 32 * 8; b0 < nblk; b0 += 32 * 8) {   // beyond 256 fill workgroups: eight 16-byte loads in flight per trip
        cdv_i4 c[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const int b = b0 + 32 * u;
          c[u] = (b < nblk) ? tab[b * (ORD_BINS / 4) + bq] : cdv_i4{0, 0, 0, 0};
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const bool before = b0 + 32 * u < bid;
#pragma unroll
          for (int q = 0; q < 4; q++) { tot[q] += c[u][q]; pre[q] += before ? c[u][q] : 0; }
        }
      }
#pragma unroll
      for (int q = 0; q < 4; q++) { s_tot[part][4 * bq + q] = tot[q]; s_pre[part][4 * bq + q] = pre[q]; }
      __syncthreads();
      if (tid < ORD_BINS) {
        int t = 0, p = 0;
#pragma unroll
        for (int q = 0; q < 32; q++) { t += s_tot[q][tid]; p += s_pre[q][tid]; }
        int incl = t;
#pragma unroll
        for (int o = 1; o < ORD_BINS; o <<= 1) {
          const int up = __shfl_up(incl, o);
          if (tid >= o) incl += up;
        }
        s_pos[tid] = incl - t + p;
      }
      __syncthreads();
    }
    if (tid < 64) { CDV_STAMP(graph, sslot, 2); }
    for (int e0 = e_first; e0 < A.E; e0 += nblk * 256) {      // workgroup-uniform trip count
      const int e = e0 + tid;
      const bool in = e < A.E;
      if (e0 != e_first) load_edge(e, in, cx, cy, k64, j64);   // (the first tile is in hand; further trips only beyond 262 k edges)
      if (A.poses) {
        // the tile's coordinates leave as one contiguous block: 256 edges x 72 bytes, 16 bytes per lane
        __syncthreads();
        const int n_e = min(256, A.E - e0);
        cdv_float4* dst = reinterpret_cast<cdv_float4*>(A.coords_out + (size_t)e0 * 18);   // 72 e0 bytes: e0 % 256 == 0, 16-byte aligned
        const cdv_float4* src = reinterpret_cast<const cdv_float4*>(s_xy);
        const int n4 = (n_e * 18) >> 2;                          // n_e * 18 is a multiple of 2; a last half vector below
        for (int v = tid; v < n4; v += 256) dst[v] = src[v];
        if (tid == 0 && ((n_e * 18) & 3)) {
          A.coords_out[(size_t)e0 * 18 + 4 * n4] = s_xy[4 * n4];
          A.coords_out[(size_t)e0 * 18 + 4 * n4 + 1] = s_xy[4 * n4 + 1];
        }
      }
      if (in) {
        const int b = (int)j64 & (ORD_BINS - 1);
        const int pos = atomicAdd(&s_pos[b], 1);
        A.order[pos] = e;
        if (A.crec) {
          uint32_t kq, jq;
          const bool ok = corr_ring_index(k64, A.cs.kmod, A.cs.kmagic, A.cs.Ng, kq) &
                          corr_ring_index(j64, A.cs.jmod, A.cs.jmagic, A.cs.slots, jq);
          const auto bits = [](float f) { return (uint32_t)__float_as_int(f); };
          const auto fl = [&](float v) { return (int)fminf(fmaxf(floorf(v * A.cs.inv_scale0), -30000.f), 30000.f); };
          int xlo = 30000, xhi = -30000, ylo = 30000, yhi = -30000;
#pragma unroll
          for (int a = 0; a < 9; a++) {
            const int vx = fl(cx[a]), vy = fl(cy[a]);
            xlo = min(xlo, vx); xhi = max(xhi, vx); ylo = min(ylo, vy); yhi = max(yhi, vy);
          }
          typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
          u32x4* r4 = reinterpret_cast<u32x4*>(A.crec + (size_t)pos * CORR_REC_WORDS);
          r4[0] = u32x4{bits(cx[0]), bits(cx[1]), bits(cx[2]), bits(cx[3])};
          r4[1] = u32x4{bits(cx[4]), bits(cx[5]), bits(cx[6]), bits(cx[7])};
          r4[2] = u32x4{bits(cx[8]), bits(cy[0]), bits(cy[1]), bits(cy[2])};
          r4[3] = u32x4{bits(cy[3]), bits(cy[4]), bits(cy[5]), bits(cy[6])};
          r4[4] = u32x4{bits(cy[7]), bits(cy[8]), (uint32_t)e, ok ? kq : 0xFFFFFFFFu};
          r4[5] = u32x4{ok ? jq : 0xFFFFFFFFu, ((uint32_t)xhi << 16) | ((uint32_t)xlo & 0xffffu),
                        ((uint32_t)yhi << 16) | ((uint32_t)ylo & 0xffffu), 0u};
        }
      }
      if (A.poses) __syncthreads();                              // the tile is reused by the next trip
    }
    if (tid < 64) { CDV_STAMP(graph, sslot, 3); CDV_STAMP_RT(graph, sslot, 15); }
    return;
  }
  // =================================== patch workgroups: 8 slots each, a half-wave per slot ===================================
  // everything a slot needs is requested at once, before its degree is known: cursor and all ELL_SLOTS records (a slot's
  // unused records hold old data: masked by the degree when it arrives)
  const int h = tid >> 5, hl = tid & 31;
  const int slot = 8 * (int)blockIdx.x + h;
  const bool in_tab = slot < A.R;
  const int cs = in_tab ? slot : 0;
  const int deg_raw = A.tcur[cs];
  cdv_i4 rec = *reinterpret_cast<const cdv_i4*>(A.ttab + tab_index(cs, hl));
  __shared__ uint64_t s_key[8][32];
  __shared__ int s_e[8][32];
  __shared__ int s_flag[8];
  __shared__ cdv_i4 s_orec[TS_OVF_MAX];
  __shared__ uint64_t s_okey[TS_OVF_MAX];
  __shared__ int s_osort[TS_OVF_MAX], s_ocnt;
  if (blockIdx.x == 0 && tid == 0) {   // the next build's overflow counter, error word and generation
    A.meta[GM_OVFN + (par ^ 1)] = 0;
    A.meta[GM_TERR + (par ^ 1)] = 0;
    A.meta[GM_GENNEXT] = (int)((unsigned)gen + 1u);
  }
  const int deg = in_tab ? deg_raw : 0;
  const int d32 = min(deg, ELL_SLOTS);
  // two patch ids in one slot (the table's capacity is smaller than the live id range): error state
  const int k0 = __shfl(rec.w, lane & 32);                      // record 0 of this half-wave's slot
  const bool clash = hl < d32 && rec.w != k0;
  if (__ballot(clash) != 0ull && lane == 0) A.meta[GM_TERR + par] = 1;
  if (in_tab && hl == 0) {
    A.tdeg[slot] = min(deg, TS_OVF_MAX);
    A.tplo[slot] = 0;
    A.tkid[slot] = deg > 0 ? k0 : -1;
    if (deg > 0) { A.tcur[slot] = 0; A.town[slot] = 0ull; }     // cursor and owner word: zero again for the next build
    s_flag[h] = deg > 0 ? 1 : 0;
  } else if (hl == 0) {
    s_flag[h] = 0;
  }
  if (tid < 64) { CDV_STAMP(graph, sslot, 2); }
  if (deg > 0 && deg <= ELL_SLOTS) {
    const uint64_t key = hl < deg ? rec_key(rec) : ~0ull;
    s_key[h][hl] = key;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    int rank = 0;
    for (int u = 0; u < deg; u++) rank += (s_key[h][u] < key) ? 1 : 0;
    if (hl < deg) s_e[h][rank] = rec.x;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    if (hl < deg) {
      *reinterpret_cast<cdv_i4*>(A.ttab + tab_index(slot, rank)) = rec;   // every lane holds its record: in place
      const int pe = rank > 0 ? s_e[h][rank - 1] : -1, ne = rank + 1 < deg ? s_e[h][rank + 1] : -1;
      A.nprev[rec.x] = pe;
      A.nnext[rec.x] = ne;
      if (A.ix && !terr) { A.ix[rec.x] = (int64_t)pe; A.jx[rec.x] = (int64_t)ne; }
    }
  }
  __syncthreads();
  if (tid == 0) {
    int n = 0;
#pragma unroll
    for (int u = 0; u < 8; u++) n += s_flag[u];
    A.tlive[blockIdx.x] = n;                                     // live patches of this workgroup (summed by whoever asks)
  }
  // ---- patches with more than ELL_SLOTS edges (loop-closure graphs): wave 0, one after the other.  The first ELL_SLOTS
  // records are in the table, the rest in the overflow list; all of them are sorted into an overflow CSR segment
  // (tprec[tplo + t], t = 0 .. deg - 1) whose first ELL_SLOTS records also go back into the table ----
  __shared__ int s_deg8[8];
  if (hl == 0) s_deg8[h] = deg;
  __syncthreads();
  if (tid < 64) { CDV_STAMP(graph, sslot, 3); CDV_STAMP_RT(graph, sslot, 15); }
  if (wave != 0) return;
  bool any_ovf = false;
#pragma unroll
  for (int u = 0; u < 8; u++) any_ovf |= s_deg8[u] > ELL_SLOTS;
  if (!any_ovf) return;
  const int n_ovf = A.meta[GM_OVFN + par];
  for (int hs = 0; hs < 8; hs++) {
    const int sdeg = s_deg8[hs], oslot = 8 * (int)blockIdx.x + hs;
    if (sdeg <= ELL_SLOTS) continue;
    if (sdeg > TS_OVF_MAX) {                                      // not served (flagged by the fill launch already)
      if (lane == 0) A.meta[GM_TERR + par] = 1;
      continue;
    }
    if (lane < ELL_SLOTS) s_orec[lane] = *reinterpret_cast<const cdv_i4*>(A.ttab + tab_index(oslot, lane));
    if (lane == 0) s_ocnt = ELL_SLOTS;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    const int okid = s_orec[0].w;
    const float rinv = 1.0f / (float)A.R;
    for (int u = lane; u < n_ovf; u += 64) {
      const cdv_i4 r = *reinterpret_cast<const cdv_i4*>(A.tovf + 4 * (size_t)u);
      int q = (int)((float)r.w * rinv);
      int rs = r.w - q * A.R;
      rs = (rs < 0) ? rs + A.R : rs;
      rs = (rs >= A.R) ? rs - A.R : rs;
      if (rs == oslot) {
        if (r.w != okid) A.meta[GM_TERR + par] = 1;               // another id in this slot
        const int idx = atomicAdd(&s_ocnt, 1);
        if (idx < TS_OVF_MAX) s_orec[idx] = r;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    const int total = min(s_ocnt, TS_OVF_MAX);                   // == deg
    for (int x = lane; x < total; x += 64) s_okey[x] = rec_key(s_orec[x]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    for (int x = lane; x < total; x += 64) {
      const uint64_t key = s_okey[x];
      int rank = 0;
      for (int y = 0; y < total; y++) rank += (s_okey[y] < key) ? 1 : 0;
      s_osort[rank] = x;
    }
    int plo = 0;
    if (lane == 0) plo = atomicAdd(&A.meta[GM_PRECN], total);
    plo = __shfl(plo, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    for (int x = lane; x < total; x += 64) {
      const cdv_i4 r = s_orec[s_osort[x]];
      *reinterpret_cast<cdv_i4*>(A.tprec + 4 * (size_t)(plo + x)) = r;
      if (x < ELL_SLOTS) *reinterpret_cast<cdv_i4*>(A.ttab + tab_index(oslot, x)) = r;
      const int pe = x > 0 ? s_orec[s_osort[x - 1]].x : -1, ne = x + 1 < total ? s_orec[s_osort[x + 1]].x : -1;
      A.nprev[r.x] = pe;
      A.nnext[r.x] = ne;
      if (A.ix && !terr) { A.ix[r.x] = (int64_t)pe; A.jx[r.x] = (int64_t)ne; }
    }
    if (lane == 0) A.tplo[oslot] = plo;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  }
}

__global__ __launch_bounds__(256) void graph_tfill_kernel(cdv::TFillArgs a) {
  cdv::graph_tfill_body(a, (int)blockIdx.x, (int)gridDim.x, (int)blockDim.x, (int)threadIdx.x);
}

inline int grid_for(int64_t n, int threads, int cap) {
  const int b = cdv_div_up(n > 0 ? n : 1, threads);
  return b < cap ? b : cap;
}

}  // namespace

bool cdv_graph_lookup(const void* ws, GraphLayout* out) {
  std::lock_guard<std::mutex> lk(g_reg_mutex);
  auto it = g_registry.find(ws);
  if (it == g_registry.end()) return false;
  *out = it->second.L;
  return true;
}

bool cdv_graph_has_ii(const void* ws) {
  std::lock_guard<std::mutex> lk(g_reg_mutex);
  auto it = g_registry.find(ws);
  return it != g_registry.end() && it->second.has_ii;
}

// forget what is known about a workspace address (see cdv_workspace_forget in the header)
void cdv_graph_forget(const void* ws) {
  std::lock_guard<std::mutex> lk(g_reg_mutex);
  g_registry.erase(ws);
  g_no_order.erase(ws);
}

// Explicit initialisation of an index workspace: zeroes what the builds keep zero between calls (histogram, cursors,
// table owner words, meta words) NOW, on `stream`, and records the layout -- so that nothing depends on whether the library
// has seen this address before (an allocator may hand out the address of a freed workspace: whoever allocates calls this).
// A bound correlation stream is dropped.
extern "C" int cdv_graph_workspace_init(void* ws, size_t ws_bytes, int64_t E_max, int64_t k_range, void* stream) {
  CDV_REQUIRE(ws != nullptr, CDV_ERR_ARG, "cdv_graph_workspace_init: workspace is NULL");
  CDV_REQUIRE(k_range >= 1 && k_range < ((int64_t)1 << 31) - 64 && E_max >= 1, CDV_ERR_ARG, "cdv_graph_workspace_init: bad sizes");
  const GraphLayout L = graph_layout(E_max, k_range);
  CDV_REQUIRE(L.total <= ws_bytes, CDV_ERR_WORKSPACE, "cdv_graph_workspace_init: workspace too small for (E_max, k_range)");
  const GraphView v = graph_view(ws, L);
  hipLaunchKernelGGL(graph_init_kernel, dim3(grid_for(k_range + 1 + GM_WORDS, 256, 2048)), dim3(256), 0, (hipStream_t)stream,
                     v.meta, v.khist, v.kcursor, v.tcur, v.town, k_range);
  CDV_LAUNCH_CHECK();
  std::lock_guard<std::mutex> lk(g_reg_mutex);
  g_registry[ws] = RegEntry{L, true, false, CorrStream{nullptr, 0, 0, 0, 0, 0, 0, 1.0f}, false, 0};
  return CDV_OK;
}

extern "C" size_t cdv_graph_workspace_bytes(int64_t E_max, int64_t k_range) {
  if (E_max < 1) E_max = 1;
  if (k_range < 1) k_range = 1;
  return graph_layout(E_max, k_range).total;
}

extern "C" int cdv_graph_build(const int64_t* jj, const int64_t* kk, int64_t E, void* ws, size_t ws_bytes,
                               int64_t E_max, int64_t k_range, void* stream) {
  return cdv_graph_build_neighbors(jj, kk, E, ws, ws_bytes, E_max, k_range, nullptr, nullptr, stream);
}

// Index build in two halves, so that cdv_update_prologue (prologue.hip) can run the histogram launch fused with the
// other independent per-frame kernels: prepare = argument checks, registry, one-time initialisation;
// finish = scan, fill, segment sort (+ neighbors).
int cdv_graph_prepare(const int64_t* jj, const int64_t* kk, int64_t E, void* ws, size_t ws_bytes, int64_t E_max,
                      int64_t k_range, int64_t* ix, int64_t* jx, void* stream, cdv::HistArgs* hist, int* hist_blocks) {
  (void)jj; (void)kk;
  CDV_REQUIRE((ix == nullptr) == (jx == nullptr), CDV_ERR_ARG, "cdv_graph_build_neighbors: give both ix and jx or neither");
  CDV_REQUIRE(ws != nullptr, CDV_ERR_ARG, "cdv_graph_build: workspace is NULL");
  CDV_REQUIRE(E >= 0 && E < (int64_t)1 << 31, CDV_ERR_ARG, "cdv_graph_build: E out of range");
  CDV_REQUIRE(k_range >= 1 && E_max >= 1 && E <= E_max, CDV_ERR_ARG,
              "cdv_graph_build: need 1 <= E <= E_max, k_range >= 1");
  const GraphLayout L = graph_layout(E_max, k_range);
  CDV_REQUIRE(L.total <= ws_bytes, CDV_ERR_WORKSPACE, "cdv_graph_build: workspace too small for (E_max, k_range)");
  bool need_init;
  {
    std::lock_guard<std::mutex> lk(g_reg_mutex);
    auto it = g_registry.find(ws);
    need_init = it == g_registry.end() || !it->second.initialised || it->second.L.E_max != E_max ||
                it->second.L.k_range != k_range;
    const CorrStream keep = it != g_registry.end() ? it->second.cs : CorrStream{nullptr, 0, 0, 0, 0, 0, 0, 1.0f};
    g_registry[ws] = RegEntry{L, true, false, keep, false, 0};
  }
  const GraphView v = graph_view(ws, L);
  if (need_init)
    hipLaunchKernelGGL(graph_init_kernel, dim3(grid_for(k_range + 1 + GM_WORDS, 256, 2048)), dim3(256), 0,
                       (hipStream_t)stream, v.meta, v.khist, v.kcursor, v.tcur, v.town, k_range);
  *hist = cdv::HistArgs{jj, kk, (int32_t)E, v.stage, v.khist, (int32_t)k_range, v.meta, v.kcount, v.krank, v.ocnt};
  *hist_blocks = E > 0 ? grid_for(E, 256, GRAPH_MAX_BLOCKS) : 0;
  return CDV_OK;
}

int cdv_graph_finish(const int64_t* ii, const int64_t* jj, const int64_t* kk, int64_t E, void* ws, int64_t E_max,
                     int64_t k_range, int hist_blocks, int64_t* ix, int64_t* jx, void* stream) {
  const GraphLayout L = graph_layout(E_max, k_range);
  const GraphView v = graph_view(ws, L);
  CorrStream cs{nullptr, 0, 0, 0, 0, 0, 0, 1.0f};
  bool want_order = true;
  {
    std::lock_guard<std::mutex> lk(g_reg_mutex);
    auto it = g_registry.find(ws);
    if (it != g_registry.end()) {
      it->second.has_ii = ii != nullptr && E > 0;
      cs = it->second.cs;
    }
    want_order = g_no_order.find(ws) == g_no_order.end();
  }
  hipStream_t s = (hipStream_t)stream;
  const int32_t En = (int32_t)E;
  const int tb = 256;
  const int fb = grid_for(E, tb, GRAPH_MAX_BLOCKS);
  if (E > cdv::GRAPH_WIDE_EDGES) {   // the scan of a wide build: two launches over SCAN_WG workgroups (partial sums in ku, which the fill writes later)
    hipLaunchKernelGGL(graph_scan_part_kernel, dim3(SCAN_WG), dim3(SCAN_T), 0, s, v.stage, hist_blocks, v.khist, v.ku, En, k_range);
    hipLaunchKernelGGL(graph_scan_apply_kernel, dim3(SCAN_WG), dim3(SCAN_T), 0, s, v.meta, v.stage, hist_blocks, v.khist, v.kcount,
                       v.krank, v.ku, En, k_range);
  } else if (hist_blocks == 0) {     // otherwise the last workgroup of the histogram launch has done the scan
    hipLaunchKernelGGL(graph_scan_kernel, dim3(1), dim3(1024), 0, s, v.meta, v.stage, hist_blocks, v.khist, v.kcount,
                       v.krank, En, k_range);
  }
  if (E > 0) {
    hipLaunchKernelGGL(graph_fill_kernel, dim3(fb), dim3(tb), 0, s, kk, En, v.meta, v.kcount, v.kcursor, v.krank,
                       v.koff_u, v.kx, v.ku, v.pcsr_tmp, jj, v.ocnt, hist_blocks, want_order ? v.order : nullptr, cs, v.crec);
    hipLaunchKernelGGL(graph_segsort_kernel, dim3(fb), dim3(tb), 0, s, ii, jj, kk, En, v.meta, v.kcount, v.pcsr_tmp,
                       v.pcsr, v.prec, v.pell, (int)L.ell_chunks, v.krank, v.nprev, v.nnext, v.kcursor, ix, jx);
  }
  CDV_LAUNCH_CHECK();
  return CDV_OK;
}

// An index that no correlation will ever walk (the frame-pair index of a global bundle adjustment: ba.hip): its builds skip the
// correlation's processing order -- every workgroup of the fill launch sums the whole per-block count table for it, which is
// nothing at the ~190 blocks of a frame-to-frame graph and 130 MB of reads at the 1024 blocks of a 700 k-edge one.
void cdv_graph_no_corr_order(const void* ws) {
  std::lock_guard<std::mutex> lk(g_reg_mutex);
  g_no_order.insert(ws);
}

bool cdv_graph_is_table(const void* ws) {
  std::lock_guard<std::mutex> lk(g_reg_mutex);
  auto it = g_registry.find(ws);
  return it != g_registry.end() && it->second.table;
}

int64_t cdv_graph_table_capacity(const void* ws) {
  std::lock_guard<std::mutex> lk(g_reg_mutex);
  auto it = g_registry.find(ws);
  return (it != g_registry.end() && it->second.table) ? it->second.tab_cap : 0;
}

// The table build in two halves (like cdv_graph_prepare / cdv_graph_finish), so that cdv_update_prologue_table can run the
// fill pass inside its own first launch: prepare = checks, registry, one-time initialisation, arguments of the fill pass;
// finish = the sort launch (patch workgroups + edge workgroups).
int cdv_graph_table_prepare(const int64_t* ii, const int64_t* jj, const int64_t* kk, int64_t E, void* ws, size_t ws_bytes,
                            int64_t E_max, int64_t k_range, int64_t tab_cap, int64_t* ix, int64_t* jx, void* stream,
                            cdv::TFillArgs* fill, int* fill_blocks, const int32_t* dyn) {
  CDV_REQUIRE((ix == nullptr) == (jx == nullptr), CDV_ERR_ARG, "cdv_graph_build_table: give both ix and jx or neither");
  CDV_REQUIRE(ws != nullptr, CDV_ERR_ARG, "cdv_graph_build_table: workspace is NULL");
  CDV_REQUIRE(E >= 0 && E < (int64_t)1 << 31, CDV_ERR_ARG, "cdv_graph_build_table: E out of range");
  CDV_REQUIRE(k_range >= 1 && k_range < ((int64_t)1 << 31) - 64 && E_max >= 1 && E <= E_max, CDV_ERR_ARG,
              "cdv_graph_build_table: need 1 <= E <= E_max, 1 <= k_range < 2^31");
  CDV_REQUIRE(tab_cap >= 1 && tab_cap <= k_range && tab_cap <= TAB_CAP_MAX, CDV_ERR_ARG,
              "cdv_graph_build_table: table capacity must lie in [1, min(k_range, 65536)]");
  const GraphLayout L = graph_layout(E_max, k_range);
  CDV_REQUIRE(L.total <= ws_bytes, CDV_ERR_WORKSPACE, "cdv_graph_build_table: workspace too small for (E_max, k_range)");
  bool need_init;
  {
    std::lock_guard<std::mutex> lk(g_reg_mutex);
    auto it = g_registry.find(ws);
    need_init = it == g_registry.end() || !it->second.initialised || it->second.L.E_max != E_max ||
                it->second.L.k_range != k_range;
    const CorrStream keep = it != g_registry.end() ? it->second.cs : CorrStream{nullptr, 0, 0, 0, 0, 0, 0, 1.0f};
    g_registry[ws] = RegEntry{L, true, ii != nullptr && E > 0, keep, true, (int32_t)tab_cap};
  }
  const GraphView v = graph_view(ws, L);
  if (need_init)
    hipLaunchKernelGGL(graph_init_kernel, dim3(grid_for(k_range + 1 + GM_WORDS, 256, 2048)), dim3(256), 0,
                       (hipStream_t)stream, v.meta, v.khist, v.kcursor, v.tcur, v.town, k_range);
  // (the build's generation is a device word, GM_GENNEXT: nothing of this call's host state reaches the launches by value
  // except sizes and pointers, so the two launches may be captured into a hipGraph and replayed any number of times)
  *fill = cdv::TFillArgs{ii, jj, kk, (int32_t)E, (int32_t)tab_cap, v.meta, v.tcur, v.town, v.ttab, v.tovf, v.tprec, v.ocnt, dyn};
  *fill_blocks = grid_for(E, 256, GRAPH_MAX_BLOCKS);   // >= 1: the first workgroup also resets the words of this build
  return CDV_OK;
}

int cdv_graph_table_finish(const cdv::TFillArgs& fill, int fill_blocks, void* ws, int64_t E_max, int64_t k_range, int64_t* ix,
                           int64_t* jx, const float* poses, const float* patches, const float* intr, float* coords_out,
                           bool with_stream, void* stream) {
  const GraphLayout L = graph_layout(E_max, k_range);
  const GraphView v = graph_view(ws, L);
  CorrStream cs{nullptr, 0, 0, 0, 0, 0, 0, 1.0f};
  {
    std::lock_guard<std::mutex> lk(g_reg_mutex);
    auto it = g_registry.find(ws);
    if (it != g_registry.end()) cs = it->second.cs;
  }
  const bool stream_ok = with_stream && (poses != nullptr || cs.coords != nullptr);
  TSortArgs A;
  A.meta = v.meta; A.R = fill.R; A.E = fill.E; A.town = v.town;
  A.tcur = v.tcur; A.tdeg = v.tdeg; A.tplo = v.tplo; A.tkid = v.tkid; A.tlive = v.tlive; A.ttab = v.ttab; A.tovf = v.tovf;
  A.tprec = v.tprec;
  A.nprev = v.nprev; A.nnext = v.nnext; A.ix = ix; A.jx = jx;
  A.n_patch_wg = (fill.R + 7) / 8;
  A.ocnt = v.ocnt; A.nblk_edges = fill_blocks; A.order = v.order;
  A.ii = fill.ii; A.jj = fill.jj; A.kk = fill.kk;
  A.cs = cs; A.crec = stream_ok ? v.crec : nullptr;
  A.poses = poses; A.patches = patches; A.intr = intr; A.coords_out = coords_out;
  A.dyn = fill.dyn;
  const int n_edge_wg = fill.E > 0 ? fill_blocks : 0;
  hipLaunchKernelGGL(graph_tsort_kernel, dim3(A.n_patch_wg + n_edge_wg), dim3(256), 0, (hipStream_t)stream, A);
  CDV_LAUNCH_CHECK();
  return CDV_OK;
}

extern "C" int cdv_graph_build_table(const int64_t* ii, const int64_t* jj, const int64_t* kk, int64_t E, void* ws,
                                     size_t ws_bytes, int64_t E_max, int64_t k_range, int64_t table_capacity, int64_t* ix,
                                     int64_t* jx, void* stream) {
  cdv::TFillArgs f;
  int fb = 0;
  const int rc = cdv_graph_table_prepare(ii, jj, kk, E, ws, ws_bytes, E_max, k_range, table_capacity, ix, jx, stream, &f, &fb, nullptr);
  if (rc != CDV_OK) return rc;
  hipLaunchKernelGGL(graph_tfill_kernel, dim3(fb), dim3(256), 0, (hipStream_t)stream, f);
  return cdv_graph_table_finish(f, fb, ws, E_max, k_range, ix, jx, nullptr, nullptr, nullptr, nullptr, true, stream);
}

// byte offsets (into the workspace) of the table's arrays, for tools and tests that want to look at it:
// out[0..6] = degree per slot (int32 [capacity]), overflow-CSR offset per slot (int32 [capacity]), records (int32 x 4, chunk-
// slot layout), overflow-CSR records (int32 x 4), the correlation's order (int32 [E]), its packed stream (24 x uint32 per edge),
// patch id per slot (int32 [capacity], -1: none)
extern "C" int cdv_graph_table_offsets(int64_t E_max, int64_t k_range, int64_t* out) {
  CDV_REQUIRE(out != nullptr && E_max >= 1 && k_range >= 1, CDV_ERR_ARG, "cdv_graph_table_offsets: bad argument");
  const GraphLayout L = graph_layout(E_max, k_range);
  out[0] = (int64_t)L.tdeg; out[1] = (int64_t)L.tplo; out[2] = (int64_t)L.ttab; out[3] = (int64_t)L.tprec;
  out[4] = (int64_t)L.order; out[5] = (int64_t)L.crec; out[6] = (int64_t)L.tkid;
  return CDV_OK;
}

extern "C" int cdv_graph_build_edges(const int64_t* ii, const int64_t* jj, const int64_t* kk, int64_t E, void* ws,
                                     size_t ws_bytes, int64_t E_max, int64_t k_range, int64_t* ix, int64_t* jx,
                                     void* stream) {
  cdv::HistArgs h;
  int hb = 0;
  const int rc = cdv_graph_prepare(jj, kk, E, ws, ws_bytes, E_max, k_range, ix, jx, stream, &h, &hb);
  if (rc != CDV_OK) return rc;
  if (hb > 0) hipLaunchKernelGGL(graph_hist_kernel, dim3(hb), dim3(256), 0, (hipStream_t)stream, h);
  return cdv_graph_finish(ii, jj, kk, E, ws, E_max, k_range, hb, ix, jx, stream);
}

extern "C" int cdv_graph_build_neighbors(const int64_t* jj, const int64_t* kk, int64_t E, void* ws, size_t ws_bytes,
                                         int64_t E_max, int64_t k_range, int64_t* ix, int64_t* jx, void* stream) {
  return cdv_graph_build_edges(nullptr, jj, kk, E, ws, ws_bytes, E_max, k_range, ix, jx, stream);
}

extern "C" const int32_t* cdv_graph_corr_order(const void* ws) {
  GraphLayout L;
  if (!cdv_graph_lookup(ws, &L)) return nullptr;
  return graph_view((void*)ws, L).order;
}

extern "C" int cdv_graph_bind_corr_stream(void* ws, const float* coords, int64_t kmod, int64_t jmod, int64_t Ng,
                                          int64_t slots, float scale0) {
  CDV_REQUIRE(scale0 > 0.f, CDV_ERR_ARG, "cdv_graph_bind_corr_stream: scale of level 0 must be positive");
  CDV_REQUIRE(ws != nullptr, CDV_ERR_ARG, "cdv_graph_bind_corr_stream: workspace is NULL");
  CDV_REQUIRE(kmod >= 0 && jmod >= 0 && kmod < ((int64_t)1 << 31) && jmod < ((int64_t)1 << 31) && Ng >= 0 &&
                  Ng < ((int64_t)1 << 31) && slots >= 0 && slots < ((int64_t)1 << 31),
              CDV_ERR_ARG, "cdv_graph_bind_corr_stream: moduli / ring sizes out of range");
  CorrStream cs{coords, (uint32_t)kmod, (uint32_t)jmod, 0u, 0u, (uint32_t)Ng, (uint32_t)slots, 1.0f / scale0};
  cs.kmagic = kmod > 1 ? (uint32_t)((((uint64_t)1 << 32) + (uint64_t)kmod - 1) / (uint64_t)kmod) : 0u;
  cs.jmagic = jmod > 1 ? (uint32_t)((((uint64_t)1 << 32) + (uint64_t)jmod - 1) / (uint64_t)jmod) : 0u;
  std::lock_guard<std::mutex> lk(g_reg_mutex);
  auto it = g_registry.find(ws);
  if (it == g_registry.end()) g_registry[ws] = RegEntry{GraphLayout{}, false, false, cs, false, 0};
  else it->second.cs = cs;
  return CDV_OK;
}

extern "C" const uint32_t* cdv_graph_corr_records(const void* ws) {
  GraphLayout L;
  if (!cdv_graph_lookup(ws, &L)) return nullptr;
  return graph_view((void*)ws, L).crec;
}

extern "C" int cdv_graph_read_meta_host(const void* ws, int64_t* meta_host, void* stream) {
  GraphLayout L;
  CDV_REQUIRE(cdv_graph_lookup(ws, &L), CDV_ERR_ARG, "cdv_graph_read_meta_host: workspace has no built graph");
  int32_t m[GM_WORDS];
  CDV_HIP_CHECK(hipMemcpyAsync(m, (const char*)ws + L.meta, sizeof(m), hipMemcpyDeviceToHost, (hipStream_t)stream));
  CDV_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
  if (cdv_graph_is_table(ws)) {   // patch table: live patches and their id range from the per-slot ids; no frame range is kept
    const int err = m[GM_TERR + (m[GM_GEN] & 1)] != 0;
    const int cap = (int)cdv_graph_table_capacity(ws);
    std::vector<int32_t> kid((size_t)(cap > 0 ? cap : 0));
    if (cap > 0) {
      CDV_HIP_CHECK(hipMemcpyAsync(kid.data(), (const char*)ws + L.tkid, sizeof(int32_t) * (size_t)cap, hipMemcpyDeviceToHost,
                                   (hipStream_t)stream));
      CDV_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    }
    int64_t U = 0, lo = 0x7fffffff, hi = -1;
    for (int32_t k : kid)
      if (k >= 0) { U++; lo = k < lo ? k : lo; hi = k > hi ? k : hi; }
    meta_host[0] = err ? 0 : U; meta_host[1] = 1; meta_host[2] = lo; meta_host[3] = hi;
    meta_host[4] = 0; meta_host[5] = 0; meta_host[6] = err; meta_host[7] = m[GM_E];
    return CDV_OK;
  }
  meta_host[0] = m[GM_U]; meta_host[1] = 0; meta_host[2] = m[GM_KMIN]; meta_host[3] = m[GM_KMAX];
  meta_host[4] = m[GM_JMIN]; meta_host[5] = m[GM_JMAX]; meta_host[6] = m[GM_ERROR]; meta_host[7] = m[GM_E];
  return CDV_OK;
}

extern "C" int cdv_graph_get_unique(const void* ws, int64_t* kx, int64_t kx_capacity, int64_t* ku, int64_t E,
                                    void* stream) {
  GraphLayout L;
  CDV_REQUIRE(cdv_graph_lookup(ws, &L), CDV_ERR_ARG, "cdv_graph_get_unique: workspace has no built graph");
  CDV_REQUIRE(!cdv_graph_is_table(ws), CDV_ERR_UNSUPPORTED,
              "cdv_graph_get_unique: the workspace holds a patch table (no ranks); build the ranked index (cdv_graph_build*)");
  if (E == 0) return CDV_OK;
  const GraphView v = graph_view((void*)ws, L);
  const int64_t n = E > kx_capacity ? E : kx_capacity;
  hipLaunchKernelGGL(graph_copy_unique_kernel, dim3(grid_for(n, 256, 1024)), dim3(256), 0, (hipStream_t)stream, v.meta,
                     v.kx, v.ku, kx, kx_capacity, ku, (int32_t)E);
  CDV_LAUNCH_CHECK();
  return CDV_OK;
}

extern "C" int cdv_neighbors(const void* ws, int64_t E, int64_t* ix, int64_t* jx, void* stream) {
  GraphLayout L;
  CDV_REQUIRE(cdv_graph_lookup(ws, &L), CDV_ERR_ARG, "cdv_neighbors: workspace has no built graph");
  if (E == 0) return CDV_OK;
  const GraphView v = graph_view((void*)ws, L);
  hipLaunchKernelGGL(graph_neighbors_kernel, dim3(grid_for(E, 256, 1024)), dim3(256), 0, (hipStream_t)stream,
                     (int32_t)E, v.meta, v.nprev, v.nnext, ix, jx);
  CDV_LAUNCH_CHECK();
  return CDV_OK;
}
