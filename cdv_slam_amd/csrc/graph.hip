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
// Pipeline, 4 small launches, no host sync (every launch costs ~5 us of latency on an otherwise idle GPU, the
// work itself is a few hundred KB):
//   1 hist     : count[k mod R]++ (R = workspace capacity; the ids of one build span less than R, checked) and
//                min/max of kk, jj (one atomic per block).  Indexing by k mod R needs no kmin, so the min/max pass
//                and the histogram are ONE launch; the histogram is zero between builds (the scan clears it).
//   2 scan     : (1 WG) validate the range, publish meta, exclusive scan in id order (starting at bin kmin mod R)
//                -> dense offsets by id - kmin and unique ranks; clears the histogram
//   3 fill     : kx / koff_u from the dense bins, ku and the (unordered) CSR slots per edge
//   4 segsort  : rank every edge inside its patch segment by (jj, edge id); the same sweep finds the edge's
//                predecessor / successor in time (fastba.neighbors); clears the fill cursors
#include <mutex>
#include <unordered_map>

#include "cdv_common.h"
#include "cdv_graph.h"
#include "cdv_parts.h"

using namespace cdv;

namespace {

struct RegEntry {
  GraphLayout L;
  bool initialised;
};
std::mutex g_reg_mutex;
std::unordered_map<const void*, RegEntry> g_registry;

constexpr int IMAX = 0x7fffffff;
constexpr int IMIN = (int)0x80000000;

__global__ __launch_bounds__(256) void graph_init_kernel(int32_t* meta, int32_t* khist, int32_t* kcursor,
                                                         int64_t k_cap) {
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t <= k_cap; t += (int64_t)gridDim.x * blockDim.x) {
    khist[t] = 0;
    kcursor[t] = 0;
    if (t < GM_WORDS) {
      int32_t v = 0;
      if (t == GM_STAGE + 0 || t == GM_STAGE + 2) v = IMAX;
      if (t == GM_STAGE + 1 || t == GM_STAGE + 3) v = IMIN;
      meta[t] = v;
    }
  }
}

// histogram over (id mod R) + min / max of kk, jj: body in cdv_parts.h
__global__ __launch_bounds__(256) void graph_hist_kernel(cdv::HistArgs a) {
  cdv::graph_hist_body(a, (int)blockIdx.x, (int)gridDim.x, (int)blockDim.x, (int)threadIdx.x);
}

// One workgroup of 1024 threads: publish meta, exclusive scan of the histogram in id order (bin of id kmin + i is
// (kmin + i) mod R) -> kcount[i] = dense CSR offset, krank[i] = number of non-empty bins before i; clears the histogram.
__global__ __launch_bounds__(1024) void graph_scan_kernel(int32_t* meta, const int32_t* __restrict__ stage,
                                                          int nstage, int32_t* khist, int32_t* kcount,
                                                          int32_t* krank, int32_t E, int64_t k_cap) {
  __shared__ int32_t s_sum[1024];
  __shared__ int32_t s_cnt[1024];
  __shared__ int32_t s_mm[16][4];
  const int T = blockDim.x, t = threadIdx.x;
  // min / max over the per-workgroup slots of the histogram launch
  int kmin = IMAX, kmax = IMIN, jmin = IMAX, jmax = IMIN;
  if (t < nstage) { kmin = stage[4 * t]; kmax = stage[4 * t + 1]; jmin = stage[4 * t + 2]; jmax = stage[4 * t + 3]; }
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
  int32_t sum = 0, cnt = 0;
  for (int64_t i = lo; i < hi; i++) {
    int bin = b0 + (int)i; bin = (bin >= R) ? bin - R : bin;
    const int32_t v = khist[bin];
    sum += v; cnt += (v > 0);
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
  for (int64_t i = lo; i < hi; i++) {
    int bin = b0 + (int)i; bin = (bin >= R) ? bin - R : bin;
    const int32_t v = khist[bin];
    khist[bin] = 0;
    kcount[i] = run;
    krank[i] = rk;
    run += v; rk += (v > 0);
  }
  if (t == T - 1) { kcount[n] = s_sum[t]; meta[GM_U] = s_cnt[t]; }
}

__global__ __launch_bounds__(256) void graph_fill_kernel(const int64_t* __restrict__ kk, int32_t E,
                                                         const int32_t* __restrict__ meta,
                                                         const int32_t* __restrict__ kcount, int32_t* kcursor,
                                                         const int32_t* __restrict__ krank,
                                                         int32_t* __restrict__ koff_u, int64_t* __restrict__ kx,
                                                         int32_t* __restrict__ ku, int32_t* __restrict__ pcsr_tmp) {
  if (meta[GM_ERROR]) return;
  const int krange = meta[GM_KRANGE], kmin = meta[GM_KMIN];
  const int n = max(E, krange + 1);
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
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
      const int d = (int)kk[t] - kmin;
      const int p = atomicAdd(&kcursor[d], 1);
      pcsr_tmp[kcount[d] + p] = t;
      ku[t] = krank[d];
    }
  }
}

// Deterministic order inside every patch segment: rank by (jj, edge id) == std::stable_sort by jj over an
// ascending index list (ba.cpp:84-86).  O(d^2) per segment, d ~ 25 in a SLAM graph.  The same sweep yields the
// edge's neighbours in time (ba.cpp:88-94): the largest key below its own and the smallest key above.
__global__ __launch_bounds__(256) void graph_segsort_kernel(const int64_t* __restrict__ jj,
                                                            const int64_t* __restrict__ kk, int32_t E,
                                                            const int32_t* __restrict__ meta,
                                                            const int32_t* __restrict__ kcount,
                                                            const int32_t* __restrict__ pcsr_tmp,
                                                            int32_t* __restrict__ pcsr, int32_t* __restrict__ nprev,
                                                            int32_t* __restrict__ nnext, int32_t* __restrict__ kcursor,
                                                            int64_t* __restrict__ ix, int64_t* __restrict__ jx) {
  if (meta[GM_ERROR]) return;
  const int kmin = meta[GM_KMIN], krange = meta[GM_KRANGE];
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t <= krange; t += gridDim.x * blockDim.x)
    kcursor[t] = 0;   // the fill cursors of this build: zero again for the next one
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < E; p += gridDim.x * blockDim.x) {
    const int e = pcsr_tmp[p];
    const int d = (int)kk[e] - kmin;
    const int lo = kcount[d], hi = kcount[d + 1];
    // one 64-bit key per edge: (jj, edge id), both below 2^31
    const uint64_t ke = ((uint64_t)jj[e] << 32) | (uint32_t)e;
    int r = 0;
    uint64_t pk = 0, nk = ~(uint64_t)0;   // best predecessor / successor key so far (sentinels: none)
    for (int s0 = lo; s0 < hi; s0 += 4) {
      int o[4];
      uint64_t ko[4];
#pragma unroll
      for (int u = 0; u < 4; u++) o[u] = pcsr_tmp[min(s0 + u, hi - 1)];
#pragma unroll
      for (int u = 0; u < 4; u++) ko[u] = ((uint64_t)jj[o[u]] << 32) | (uint32_t)o[u];
#pragma unroll
      for (int u = 0; u < 4; u++) {
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
  if (meta[GM_ERROR]) return;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < E; e += gridDim.x * blockDim.x) {
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
    g_registry[ws] = RegEntry{L, true};
  }
  const GraphView v = graph_view(ws, L);
  if (need_init)
    hipLaunchKernelGGL(graph_init_kernel, dim3(grid_for(k_range + 1, 256, 2048)), dim3(256), 0, (hipStream_t)stream,
                       v.meta, v.khist, v.kcursor, k_range);
  *hist = cdv::HistArgs{jj, kk, (int32_t)E, v.stage, v.khist, (int32_t)k_range};
  *hist_blocks = E > 0 ? grid_for(E, 256, GRAPH_MAX_BLOCKS) : 0;
  return CDV_OK;
}

int cdv_graph_finish(const int64_t* jj, const int64_t* kk, int64_t E, void* ws, int64_t E_max, int64_t k_range,
                     int hist_blocks, int64_t* ix, int64_t* jx, void* stream) {
  const GraphLayout L = graph_layout(E_max, k_range);
  const GraphView v = graph_view(ws, L);
  hipStream_t s = (hipStream_t)stream;
  const int32_t En = (int32_t)E;
  const int tb = 256;
  const int fb = grid_for(E, tb, GRAPH_MAX_BLOCKS);
  hipLaunchKernelGGL(graph_scan_kernel, dim3(1), dim3(1024), 0, s, v.meta, v.stage, hist_blocks, v.khist, v.kcount, v.krank,
                     En, k_range);
  if (E > 0) {
    hipLaunchKernelGGL(graph_fill_kernel, dim3(fb), dim3(tb), 0, s, kk, En, v.meta, v.kcount, v.kcursor, v.krank,
                       v.koff_u, v.kx, v.ku, v.pcsr_tmp);
    hipLaunchKernelGGL(graph_segsort_kernel, dim3(fb), dim3(tb), 0, s, jj, kk, En, v.meta, v.kcount, v.pcsr_tmp,
                       v.pcsr, v.nprev, v.nnext, v.kcursor, ix, jx);
  }
  CDV_LAUNCH_CHECK();
  return CDV_OK;
}

extern "C" int cdv_graph_build_neighbors(const int64_t* jj, const int64_t* kk, int64_t E, void* ws, size_t ws_bytes,
                                         int64_t E_max, int64_t k_range, int64_t* ix, int64_t* jx, void* stream) {
  cdv::HistArgs h;
  int hb = 0;
  const int rc = cdv_graph_prepare(jj, kk, E, ws, ws_bytes, E_max, k_range, ix, jx, stream, &h, &hb);
  if (rc != CDV_OK) return rc;
  if (hb > 0) hipLaunchKernelGGL(graph_hist_kernel, dim3(hb), dim3(256), 0, (hipStream_t)stream, h);
  return cdv_graph_finish(jj, kk, E, ws, E_max, k_range, hb, ix, jx, stream);
}

extern "C" int cdv_graph_read_meta_host(const void* ws, int64_t* meta_host, void* stream) {
  GraphLayout L;
  CDV_REQUIRE(cdv_graph_lookup(ws, &L), CDV_ERR_ARG, "cdv_graph_read_meta_host: workspace has no built graph");
  int32_t m[GM_WORDS];
  CDV_HIP_CHECK(hipMemcpyAsync(m, (const char*)ws + L.meta, sizeof(m), hipMemcpyDeviceToHost, (hipStream_t)stream));
  CDV_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
  meta_host[0] = m[GM_U]; meta_host[1] = 0; meta_host[2] = m[GM_KMIN]; meta_host[3] = m[GM_KMAX];
  meta_host[4] = m[GM_JMIN]; meta_host[5] = m[GM_JMAX]; meta_host[6] = m[GM_ERROR]; meta_host[7] = m[GM_E];
  return CDV_OK;
}

extern "C" int cdv_graph_get_unique(const void* ws, int64_t* kx, int64_t kx_capacity, int64_t* ku, int64_t E,
                                    void* stream) {
  GraphLayout L;
  CDV_REQUIRE(cdv_graph_lookup(ws, &L), CDV_ERR_ARG, "cdv_graph_get_unique: workspace has no built graph");
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
