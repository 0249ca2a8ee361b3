// graph.hip -- device-side patch-graph index, gfx950.
//
// Replaces, without any host round trip:
//   torch::_unique(kk, sorted, inverse)          cdvslam/fastba/ba_cuda.cu:476-478, ba.cpp:62
//   the CPU bucket + std::stable_sort loops of   cdvslam/fastba/ba.cpp:59-97 (neighbors)
// and adds a grouping of the edges by target frame jj that the correlation kernel can use as its
// processing order (all edges reading one feature-map slot run together -> L2 locality).
// All integer work: counting sorts over dense id ranges; kx, ku, the patch CSR and neighbors are bit-exact
// and deterministic (the order INSIDE a target group is not, and nothing depends on it).
//
// Pipeline (7 small launches, no sync): reset -> min/max -> clear -> histogram -> scan (1 WG) ->
// fill -> per-patch rank sort by (jj, edge id).
#include <mutex>
#include <unordered_map>

#include "cdv_common.h"
#include "cdv_graph.h"

using namespace cdv;

namespace {

std::mutex g_reg_mutex;
std::unordered_map<const void*, GraphLayout> g_registry;

__global__ void graph_reset_kernel(int32_t* meta, int32_t E) {
  const int t = threadIdx.x;
  if (t < GM_WORDS) {
    int32_t v = 0;
    if (t == GM_KMIN || t == GM_FMIN) v = 0x7fffffff;
    if (t == GM_KMAX || t == GM_FMAX) v = (int32_t)0x80000000;
    if (t == GM_E) v = E;
    meta[t] = v;
  }
}

__device__ __forceinline__ int wave_min(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
  return v;
}

__global__ __launch_bounds__(256) void graph_minmax_kernel(const int64_t* __restrict__ jj,
                                                           const int64_t* __restrict__ kk, int32_t E,
                                                           int32_t* meta) {
  int kmin = 0x7fffffff, kmax = (int)0x80000000, fmin = 0x7fffffff, fmax = (int)0x80000000;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < E; e += gridDim.x * blockDim.x) {
    const int k = (int)kk[e], j = (int)jj[e];
    kmin = min(kmin, k); kmax = max(kmax, k);
    fmin = min(fmin, j); fmax = max(fmax, j);
  }
  kmin = wave_min(kmin); kmax = wave_max(kmax); fmin = wave_min(fmin); fmax = wave_max(fmax);
  if ((threadIdx.x & 63) == 0) {
    atomicMin(&meta[GM_KMIN], kmin);
    atomicMax(&meta[GM_KMAX], kmax);
    atomicMin(&meta[GM_FMIN], fmin);
    atomicMax(&meta[GM_FMAX], fmax);
  }
}

// zero the histogram / cursor ranges actually used; also validates the ranges against capacity
__global__ __launch_bounds__(256) void graph_clear_kernel(int32_t* meta, int32_t* kcount, int32_t* kcursor,
                                                          int32_t* pcount, int32_t* pcursor, int64_t k_cap,
                                                          int64_t f_cap) {
  const int64_t krange = (int64_t)meta[GM_KMAX] - meta[GM_KMIN] + 1;
  const int64_t frange = (int64_t)meta[GM_FMAX] - meta[GM_FMIN] + 1;
  const bool bad = krange > k_cap || frange > f_cap || meta[GM_KMIN] < 0 || meta[GM_FMIN] < 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    meta[GM_KRANGE] = (int32_t)krange;
    meta[GM_FRANGE] = (int32_t)frange;
    if (bad) meta[GM_ERROR] = 1;
  }
  if (bad) return;
  const int64_t nb = frange;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t <= max(krange, nb);
       t += (int64_t)gridDim.x * blockDim.x) {
    if (t <= krange) kcount[t] = 0;
    if (t < krange) kcursor[t] = 0;
    if (t <= nb) pcount[t] = 0;
    if (t < nb) pcursor[t] = 0;
  }
}

__global__ __launch_bounds__(256) void graph_hist_kernel(const int64_t* __restrict__ jj,
                                                         const int64_t* __restrict__ kk, int32_t E,
                                                         const int32_t* __restrict__ meta, int32_t* kcount,
                                                         int32_t* pcount) {
  // NOTE: meta is read after graph_clear_kernel completed (stream order)
  if (meta[GM_ERROR]) return;
  const int kmin = meta[GM_KMIN], fmin = meta[GM_FMIN];
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < E; e += gridDim.x * blockDim.x) {
    atomicAdd(&kcount[(int)kk[e] - kmin], 1);
    atomicAdd(&pcount[(int)jj[e] - fmin], 1);
  }
}

// block-wide exclusive scan helper over `n` ints in global memory, in place, 1024 threads.
// `flag_rank` (optional): rank[d] = number of non-empty bins before d; returns totals via shared.
__device__ void block_scan_inplace(int32_t* a, int64_t n, int32_t* rank, int32_t* out_total,
                                   int32_t* out_nonempty) {
  __shared__ int32_t s_sum[1024];
  __shared__ int32_t s_cnt[1024];
  const int T = blockDim.x, t = threadIdx.x;
  const int64_t per = (n + T - 1) / T;
  const int64_t lo = min((int64_t)t * per, n), hi = min(lo + per, n);
  int32_t sum = 0, cnt = 0;
  for (int64_t i = lo; i < hi; i++) { const int32_t v = a[i]; sum += v; cnt += (v > 0); }
  s_sum[t] = sum; s_cnt[t] = cnt;
  __syncthreads();
  // Hillis-Steele inclusive scan over the 1024 partials
  for (int o = 1; o < T; o <<= 1) {
    int32_t a1 = 0, c1 = 0;
    if (t >= o) { a1 = s_sum[t - o]; c1 = s_cnt[t - o]; }
    __syncthreads();
    s_sum[t] += a1; s_cnt[t] += c1;
    __syncthreads();
  }
  int32_t run = s_sum[t] - sum, rk = s_cnt[t] - cnt;
  for (int64_t i = lo; i < hi; i++) {
    const int32_t v = a[i];
    a[i] = run;
    if (rank) rank[i] = rk;
    run += v; rk += (v > 0);
  }
  if (t == T - 1) { *out_total = s_sum[t]; *out_nonempty = s_cnt[t]; }
  __syncthreads();
}

__global__ __launch_bounds__(1024) void graph_scan_kernel(int32_t* meta, int32_t* kcount, int32_t* krank,
                                                          int32_t* pcount) {
  if (meta[GM_ERROR]) return;
  __shared__ int32_t tot, nz;
  const int64_t krange = meta[GM_KRANGE];
  const int64_t nb = (int64_t)meta[GM_FRANGE];
  block_scan_inplace(kcount, krange, krank, &tot, &nz);
  if (threadIdx.x == 0) { kcount[krange] = tot; meta[GM_U] = nz; }
  __syncthreads();
  block_scan_inplace(pcount, nb, nullptr, &tot, &nz);
  if (threadIdx.x == 0) { pcount[nb] = tot; meta[GM_NPAIRS] = nz; }
}

// kx[rank] = patch id, koff_u[rank] = first CSR slot   (dense bins -> unique ranks)
__global__ __launch_bounds__(256) void graph_unique_kernel(const int32_t* __restrict__ meta,
                                                           const int32_t* __restrict__ kcount,
                                                           const int32_t* __restrict__ krank,
                                                           int32_t* __restrict__ koff_u, int64_t* __restrict__ kx) {
  if (meta[GM_ERROR]) return;
  const int krange = meta[GM_KRANGE], kmin = meta[GM_KMIN];
  for (int d = blockIdx.x * blockDim.x + threadIdx.x; d <= krange; d += gridDim.x * blockDim.x) {
    if (d == krange) {
      koff_u[meta[GM_U]] = kcount[krange];
    } else if (kcount[d + 1] > kcount[d]) {
      const int r = krank[d];
      kx[r] = (int64_t)(kmin + d);
      koff_u[r] = kcount[d];
    }
  }
}

__global__ __launch_bounds__(256) void graph_fill_kernel(const int64_t* __restrict__ jj,
                                                         const int64_t* __restrict__ kk, int32_t E,
                                                         const int32_t* __restrict__ meta,
                                                         const int32_t* __restrict__ kcount, int32_t* kcursor,
                                                         const int32_t* __restrict__ krank,
                                                         const int32_t* __restrict__ pcount, int32_t* pcursor,
                                                         int32_t* __restrict__ ku, int32_t* __restrict__ pcsr_tmp,
                                                         int32_t* __restrict__ pperm) {
  if (meta[GM_ERROR]) return;
  const int kmin = meta[GM_KMIN], fmin = meta[GM_FMIN];
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < E; e += gridDim.x * blockDim.x) {
    const int d = (int)kk[e] - kmin;
    const int p = atomicAdd(&kcursor[d], 1);
    pcsr_tmp[kcount[d] + p] = e;
    ku[e] = krank[d];
    const int b = (int)jj[e] - fmin;
    const int q = atomicAdd(&pcursor[b], 1);
    pperm[pcount[b] + q] = e;
  }
}

// Deterministic order inside every patch segment: rank by (jj, edge id)  == std::stable_sort by jj
// over an ascending index list (ba.cpp:84-86).  O(d^2) per segment, d ~ 25.
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

__global__ __launch_bounds__(256) void graph_neighbors_kernel(const int64_t* __restrict__ kk_unused, int32_t E,
                                                              const int32_t* __restrict__ meta,
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

}  // namespace

bool cdv_graph_lookup(const void* ws, GraphLayout* out) {
  std::lock_guard<std::mutex> lk(g_reg_mutex);
  auto it = g_registry.find(ws);
  if (it == g_registry.end()) return false;
  *out = it->second;
  return true;
}

extern "C" size_t cdv_graph_workspace_bytes(int64_t E_max, int64_t k_range, int64_t f_range) {
  if (E_max < 1) E_max = 1;
  if (k_range < 1) k_range = 1;
  if (f_range < 1) f_range = 1;
  return graph_layout(E_max, k_range, f_range).total;
}

extern "C" int cdv_graph_build(const int64_t* jj, const int64_t* kk, int64_t E, void* ws, size_t ws_bytes,
                               int64_t k_range, int64_t f_range, void* stream) {
  CDV_REQUIRE(ws != nullptr, CDV_ERR_ARG, "cdv_graph_build: workspace is NULL");
  CDV_REQUIRE(E >= 0 && E < (int64_t)1 << 31, CDV_ERR_ARG, "cdv_graph_build: E out of range");
  CDV_REQUIRE(k_range >= 1 && f_range >= 1, CDV_ERR_ARG, "cdv_graph_build: ranges must be >= 1");
  const int64_t E_cap = E > 0 ? E : 1;
  const GraphLayout L = graph_layout(E_cap, k_range, f_range);
  CDV_REQUIRE(L.total <= ws_bytes, CDV_ERR_WORKSPACE, "cdv_graph_build: workspace too small for (E, k_range, f_range)");
  {
    std::lock_guard<std::mutex> lk(g_reg_mutex);
    g_registry[ws] = L;
  }
  const GraphView v = graph_view(ws, L);
  hipStream_t s = (hipStream_t)stream;
  const int32_t En = (int32_t)E;
  hipLaunchKernelGGL(graph_reset_kernel, dim3(1), dim3(64), 0, s, v.meta, En);
  CDV_LAUNCH_CHECK();
  if (E == 0) return CDV_OK;
  const int tb = 256;
  const int eb = cdv_div_up(E, tb) < 1024 ? cdv_div_up(E, tb) : 1024;
  hipLaunchKernelGGL(graph_minmax_kernel, dim3(eb), dim3(tb), 0, s, jj, kk, En, v.meta);
  const int64_t clear_n = (k_range > f_range ? k_range : f_range) + 1;
  const int cb = cdv_div_up(clear_n, tb) < 1024 ? cdv_div_up(clear_n, tb) : 1024;
  hipLaunchKernelGGL(graph_clear_kernel, dim3(cb), dim3(tb), 0, s, v.meta, v.kcount, v.kcursor, v.pcount, v.pcursor,
                     k_range, f_range);
  hipLaunchKernelGGL(graph_hist_kernel, dim3(eb), dim3(tb), 0, s, jj, kk, En, v.meta, v.kcount, v.pcount);
  hipLaunchKernelGGL(graph_scan_kernel, dim3(1), dim3(1024), 0, s, v.meta, v.kcount, v.krank, v.pcount);
  const int ub = cdv_div_up(k_range + 1, tb) < 1024 ? cdv_div_up(k_range + 1, tb) : 1024;
  hipLaunchKernelGGL(graph_unique_kernel, dim3(ub), dim3(tb), 0, s, v.meta, v.kcount, v.krank, v.koff_u, v.kx);
  hipLaunchKernelGGL(graph_fill_kernel, dim3(eb), dim3(tb), 0, s, jj, kk, En, v.meta, v.kcount, v.kcursor,
                     v.krank, v.pcount, v.pcursor, v.ku, v.pcsr_tmp, v.pperm);
  hipLaunchKernelGGL(graph_segsort_kernel, dim3(eb), dim3(tb), 0, s, jj, kk, En, v.meta, v.kcount, v.pcsr_tmp,
                     v.pcsr);
  CDV_LAUNCH_CHECK();
  return CDV_OK;
}

extern "C" int cdv_graph_read_meta_host(const void* ws, int64_t* meta_host, void* stream) {
  GraphLayout L;
  CDV_REQUIRE(cdv_graph_lookup(ws, &L), CDV_ERR_ARG, "cdv_graph_read_meta_host: workspace has no built graph");
  int32_t m[GM_WORDS];
  CDV_HIP_CHECK(hipMemcpyAsync(m, (const char*)ws + L.meta, sizeof(m), hipMemcpyDeviceToHost, (hipStream_t)stream));
  CDV_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
  meta_host[0] = m[GM_U]; meta_host[1] = m[GM_NPAIRS]; meta_host[2] = m[GM_KMIN]; meta_host[3] = m[GM_KMAX];
  meta_host[4] = m[GM_FMIN]; meta_host[5] = m[GM_FMAX]; meta_host[6] = m[GM_ERROR]; meta_host[7] = m[GM_E];
  return CDV_OK;
}

extern "C" int cdv_graph_get_unique(const void* ws, int64_t* kx, int64_t kx_capacity, int64_t* ku, int64_t E,
                                    void* stream) {
  GraphLayout L;
  CDV_REQUIRE(cdv_graph_lookup(ws, &L), CDV_ERR_ARG, "cdv_graph_get_unique: workspace has no built graph");
  if (E == 0) return CDV_OK;
  const GraphView v = graph_view((void*)ws, L);
  const int64_t n = E > kx_capacity ? E : kx_capacity;
  const int blocks = cdv_div_up(n, 256) < 1024 ? cdv_div_up(n, 256) : 1024;
  hipLaunchKernelGGL(graph_copy_unique_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, v.meta, v.kx, v.ku, kx,
                     kx_capacity, ku, (int32_t)E);
  CDV_LAUNCH_CHECK();
  return CDV_OK;
}

extern "C" const int32_t* cdv_graph_pair_order(const void* ws) {
  GraphLayout L;
  if (!cdv_graph_lookup(ws, &L)) return nullptr;
  return graph_view((void*)ws, L).pperm;
}

extern "C" int cdv_neighbors(const void* ws, int64_t E, int64_t* ix, int64_t* jx, void* stream) {
  GraphLayout L;
  CDV_REQUIRE(cdv_graph_lookup(ws, &L), CDV_ERR_ARG, "cdv_neighbors: workspace has no built graph");
  if (E == 0) return CDV_OK;
  const GraphView v = graph_view((void*)ws, L);
  const int blocks = cdv_div_up(E, 256) < 1024 ? cdv_div_up(E, 256) : 1024;
  hipLaunchKernelGGL(graph_neighbors_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, nullptr, (int32_t)E,
                     v.meta, v.koff_u, v.ku, v.pcsr, ix, jx);
  CDV_LAUNCH_CHECK();
  return CDV_OK;
}
