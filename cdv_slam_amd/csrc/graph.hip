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
// Pipeline, 5 small launches, no host sync:
//   1 minmax+clear : min/max of kk, jj (one atomic per block) and re-zeroing of the histogram range the
//                    PREVIOUS build used (so no separate memset pass)
//   2 hist         : count[k - kmin]++           (about E/U adders per address: no hot spot)
//   3 scan (1 WG)  : validate range, publish meta, exclusive scan -> dense offsets + unique ranks
//   4 fill         : kx / koff_u from the dense bins, ku and the (unordered) CSR slots per edge
//   5 segsort      : rank every edge inside its patch segment by (jj, edge id)
#include <mutex>
#include <unordered_map>

#include "cdv_common.h"
#include "cdv_graph.h"

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

__global__ __launch_bounds__(256) void graph_init_kernel(int32_t* meta, int32_t* kcount, int32_t* kcursor,
                                                         int64_t k_cap) {
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t <= k_cap; t += (int64_t)gridDim.x * blockDim.x) {
    kcount[t] = 0;
    kcursor[t] = 0;
    if (t < GM_WORDS) {
      int32_t v = 0;
      if (t == GM_STAGE + 0 || t == GM_STAGE + 2) v = IMAX;
      if (t == GM_STAGE + 1 || t == GM_STAGE + 3) v = IMIN;
      meta[t] = v;
    }
  }
}

__global__ __launch_bounds__(256) void graph_minmax_clear_kernel(const int64_t* __restrict__ jj,
                                                                 const int64_t* __restrict__ kk, int32_t E,
                                                                 int32_t* meta, int32_t* kcount, int32_t* kcursor) {
  // re-zero what the previous build left in the histogram / cursor arrays
  const int old = meta[GM_KRANGE];
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t <= old; t += gridDim.x * blockDim.x) {
    kcount[t] = 0;
    kcursor[t] = 0;
  }
  int kmin = IMAX, kmax = IMIN, jmin = IMAX, jmax = IMIN;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < E; e += gridDim.x * blockDim.x) {
    const int k = (int)kk[e], j = (int)jj[e];
    kmin = min(kmin, k); kmax = max(kmax, k);
    jmin = min(jmin, j); jmax = max(jmax, j);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    kmin = min(kmin, __shfl_xor(kmin, o)); kmax = max(kmax, __shfl_xor(kmax, o));
    jmin = min(jmin, __shfl_xor(jmin, o)); jmax = max(jmax, __shfl_xor(jmax, o));
  }
  __shared__ int s[4][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { s[wave][0] = kmin; s[wave][1] = kmax; s[wave][2] = jmin; s[wave][3] = jmax; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; w++) {
      kmin = min(kmin, s[w][0]); kmax = max(kmax, s[w][1]);
      jmin = min(jmin, s[w][2]); jmax = max(jmax, s[w][3]);
    }
    atomicMin(&meta[GM_STAGE + 0], kmin);
    atomicMax(&meta[GM_STAGE + 1], kmax);
    atomicMin(&meta[GM_STAGE + 2], jmin);
    atomicMax(&meta[GM_STAGE + 3], jmax);
  }
}

__global__ __launch_bounds__(256) void graph_hist_kernel(const int64_t* __restrict__ kk, int32_t E,
                                                         const int32_t* __restrict__ meta, int32_t* kcount,
                                                         int64_t k_cap) {
  const int kmin = meta[GM_STAGE + 0], kmax = meta[GM_STAGE + 1];
  if (kmin < 0 || (int64_t)kmax - kmin + 1 > k_cap) return;  // reported by the scan kernel
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < E; e += gridDim.x * blockDim.x)
    atomicAdd(&kcount[(int)kk[e] - kmin], 1);
}

// One workgroup of 1024 threads: publish meta, exclusive scan of the histogram in place
// (kcount -> dense CSR offsets), krank[d] = number of non-empty bins before d.
__global__ __launch_bounds__(1024) void graph_scan_kernel(int32_t* meta, int32_t* kcount, int32_t* krank,
                                                          int32_t E, int64_t k_cap) {
  __shared__ int32_t s_sum[1024];
  __shared__ int32_t s_cnt[1024];
  const int T = blockDim.x, t = threadIdx.x;
  const int kmin = meta[GM_STAGE + 0], kmax = meta[GM_STAGE + 1];
  const int jmin = meta[GM_STAGE + 2], jmax = meta[GM_STAGE + 3];
  const int64_t krange = (E > 0) ? (int64_t)kmax - kmin + 1 : 0;
  const bool bad = E > 0 && (kmin < 0 || krange > k_cap);
  __syncthreads();  // every thread has read the staging words
  if (t == 0) {
    meta[GM_STAGE + 0] = IMAX; meta[GM_STAGE + 1] = IMIN;  // ready for the next build
    meta[GM_STAGE + 2] = IMAX; meta[GM_STAGE + 3] = IMIN;
    meta[GM_KMIN] = kmin; meta[GM_KMAX] = kmax; meta[GM_JMIN] = jmin; meta[GM_JMAX] = jmax;
    meta[GM_E] = E;
    meta[GM_ERROR] = bad ? 1 : 0;
    meta[GM_KRANGE] = bad ? 0 : (int32_t)krange;  // hist was skipped when bad: the arrays are still zero
    if (bad || E == 0) meta[GM_U] = 0;
  }
  if (bad || E == 0) return;
  const int64_t n = krange;
  const int64_t per = (n + T - 1) / T;
  const int64_t lo = min((int64_t)t * per, n), hi = min(lo + per, n);
  int32_t sum = 0, cnt = 0;
  for (int64_t i = lo; i < hi; i++) { const int32_t v = kcount[i]; sum += v; cnt += (v > 0); }
  s_sum[t] = sum; s_cnt[t] = cnt;
  __syncthreads();
  for (int o = 1; o < T; o <<= 1) {  // Hillis-Steele inclusive scan over the per-thread partials
    int32_t a1 = 0, c1 = 0;
    if (t >= o) { a1 = s_sum[t - o]; c1 = s_cnt[t - o]; }
    __syncthreads();
    s_sum[t] += a1; s_cnt[t] += c1;
    __syncthreads();
  }
  int32_t run = s_sum[t] - sum, rk = s_cnt[t] - cnt;
  for (int64_t i = lo; i < hi; i++) {
    const int32_t v = kcount[i];
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
// ascending index list (ba.cpp:84-86).  O(d^2) per segment, d ~ 25 in a SLAM graph.
__global__ __launch_bounds__(256) void graph_segsort_kernel(const int64_t* __restrict__ jj,
                                                            const int64_t* __restrict__ kk, int32_t E,
                                                            const int32_t* __restrict__ meta,
                                                            const int32_t* __restrict__ kcount,
                                                            const int32_t* __restrict__ pcsr_tmp,
                                                            int32_t* __restrict__ pcsr) {
  if (meta[GM_ERROR]) return;
  const int kmin = meta[GM_KMIN];
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < E; p += gridDim.x * blockDim.x) {
    const int e = pcsr_tmp[p];
    const int d = (int)kk[e] - kmin;
    const int lo = kcount[d], hi = kcount[d + 1];
    const int64_t je = jj[e];
    int r = 0;
    for (int s = lo; s < hi; s++) {
      const int o = pcsr_tmp[s];
      const int64_t jo = jj[o];
      r += (jo < je) || (jo == je && o < e);
    }
    pcsr[lo + r] = e;
  }
}

__global__ __launch_bounds__(256) void graph_neighbors_kernel(int32_t E, const int32_t* __restrict__ meta,
                                                              const int32_t* __restrict__ koff_u,
                                                              const int32_t* __restrict__ ku,
                                                              const int32_t* __restrict__ pcsr,
                                                              int64_t* __restrict__ ix, int64_t* __restrict__ jx) {
  if (meta[GM_ERROR]) return;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < E; p += gridDim.x * blockDim.x) {
    const int e = pcsr[p];
    const int r = ku[e];
    const int lo = koff_u[r], hi = koff_u[r + 1];
    ix[e] = (p > lo) ? (int64_t)pcsr[p - 1] : -1;      // previous edge in time
    jx[e] = (p + 1 < hi) ? (int64_t)pcsr[p + 1] : -1;  // next edge in time
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
  hipStream_t s = (hipStream_t)stream;
  const int32_t En = (int32_t)E;
  const int tb = 256;
  if (need_init)
    hipLaunchKernelGGL(graph_init_kernel, dim3(grid_for(k_range + 1, tb, 2048)), dim3(tb), 0, s, v.meta, v.kcount,
                       v.kcursor, k_range);
  // ~47 blocks for E = 47,712: few enough that one min/max atomic per block is free, enough to stream kk/jj
  const int eb = grid_for(E, 1024, 256);
  hipLaunchKernelGGL(graph_minmax_clear_kernel, dim3(eb), dim3(tb), 0, s, jj, kk, En, v.meta, v.kcount, v.kcursor);
  const int fb = grid_for(E, tb, 1024);
  if (E > 0) hipLaunchKernelGGL(graph_hist_kernel, dim3(fb), dim3(tb), 0, s, kk, En, v.meta, v.kcount, k_range);
  hipLaunchKernelGGL(graph_scan_kernel, dim3(1), dim3(1024), 0, s, v.meta, v.kcount, v.krank, En, k_range);
  if (E > 0) {
    hipLaunchKernelGGL(graph_fill_kernel, dim3(fb), dim3(tb), 0, s, kk, En, v.meta, v.kcount, v.kcursor, v.krank,
                       v.koff_u, v.kx, v.ku, v.pcsr_tmp);
    hipLaunchKernelGGL(graph_segsort_kernel, dim3(fb), dim3(tb), 0, s, jj, kk, En, v.meta, v.kcount, v.pcsr_tmp,
                       v.pcsr);
  }
  CDV_LAUNCH_CHECK();
  return CDV_OK;
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
                     (int32_t)E, v.meta, v.koff_u, v.ku, v.pcsr, ix, jx);
  CDV_LAUNCH_CHECK();
  return CDV_OK;
}
