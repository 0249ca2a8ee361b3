// edges.hip -- device-resident edge bookkeeping of the patch graph, gfx950.
//
// The steps either side of the update path that the reference does with torch.cat / boolean-mask indexing (a
// reallocation and a copy of every edge array per call) and Python loops:
//   append_factors (slam.py:331-337) with the frame's forward / backward edge lists (slam.py:528-541)
//   remove_factors (slam.py:339-354): drop edges by mask, optionally keeping them as inactive edges
//   keyframe()'s index shift after a frame is dropped (slam.py:425-427)
// Here the edge arrays live in fixed-capacity device buffers; appending writes the new edges in place, removing is a
// stable stream compaction (same order as boolean-mask indexing: bit-exact bookkeeping) into the twin buffer.
#include "cdv_common.h"

namespace {

// forward edges: patches of frames [n - r, n - 1) -> frame n - 1, patch-major (flatmeshgrid 'ij', slam.py:528-534);
// backward edges: patches of frame n - 1 -> frames [n - r, n), patch-outer / frame-inner (slam.py:536-541).
// Written at ii/jj/kk[E0 ...]; ii = ix[kk] (append_factors, slam.py:334).
__global__ __launch_bounds__(256) void edges_frame_kernel(int64_t* __restrict__ ii, int64_t* __restrict__ jj,
                                                          int64_t* __restrict__ kk, const int64_t* __restrict__ ix,
                                                          int64_t E0, int n, int M, int r) {
  const int64_t f0 = (int64_t)M * max(n - r, 0), f1 = (int64_t)M * max(n - 1, 0);   // forward patch range
  const int64_t nf = max(f1 - f0, (int64_t)0);
  const int jb0 = max(n - r, 0), nbj = n - jb0;                                       // backward target range
  const int64_t nb = (int64_t)M * nbj;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nf + nb; t += (int64_t)gridDim.x * blockDim.x) {
    int64_t k, j;
    if (t < nf) {
      k = f0 + t; j = n - 1;
    } else {
      const int64_t u = t - nf;
      k = (int64_t)M * (n - 1) + u / nbj;
      j = jb0 + (int)(u % nbj);
    }
    kk[E0 + t] = k;
    jj[E0 + t] = j;
    ii[E0 + t] = ix[k];
  }
}

// generic append_factors(ii = patch ids, jj = frames): kk <- patch ids, ii <- ix[patch ids]
__global__ __launch_bounds__(256) void edges_append_kernel(int64_t* __restrict__ ii, int64_t* __restrict__ jj,
                                                           int64_t* __restrict__ kk, const int64_t* __restrict__ ix,
                                                           const int64_t* __restrict__ new_k,
                                                           const int64_t* __restrict__ new_j, int64_t E0, int64_t cnt) {
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < cnt; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t k = new_k[t];
    kk[E0 + t] = k;
    jj[E0 + t] = new_j[t];
    ii[E0 + t] = ix[k];
  }
}

// ---- stable compaction, pass 1: keep counts per workgroup of 1024 edges; the last workgroup to arrive scans them --
// counts[b] -> exclusive offset of workgroup b among the kept edges (roff[b] among the removed ones); meta[0] = kept,
// meta[1] = removed, meta[2] = arrival counter (self-resetting)
__global__ __launch_bounds__(256) void edges_count_kernel(const uint8_t* __restrict__ remove, int64_t E,
                                                          int32_t* __restrict__ counts, int32_t* __restrict__ meta) {
  __shared__ int s_w[4];
  __shared__ int s_last;
  const int t = threadIdx.x, b = blockIdx.x;
  const int64_t base = (int64_t)b * 1024;
  int c = 0;
#pragma unroll
  for (int u = 0; u < 4; u++) {
    const int64_t e = base + u * 256 + t;
    c += (e < E && !remove[e]) ? 1 : 0;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
  if ((t & 63) == 0) s_w[t >> 6] = c;
  __syncthreads();
  if (t == 0) {
    const int tot = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    __hip_atomic_store(&counts[b], tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int prev = __hip_atomic_fetch_add(&meta[2], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = prev == (int)gridDim.x - 1;
    if (s_last) __hip_atomic_store(&meta[2], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (!s_last) return;
  // exclusive scan of the per-workgroup keep counts (<= a few thousand workgroups): one wave, serial chunks
  if (t < 64) {
    const int nb = (int)gridDim.x;
    int run = 0;
    for (int c0 = 0; c0 < nb; c0 += 64) {
      const int i = c0 + t;
      const int v = (i < nb) ? __hip_atomic_load(&counts[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
      int inc = v;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int a1 = __shfl_up(inc, o);
        if (t >= o) inc += a1;
      }
      if (i < nb) counts[i] = run + inc - v;   // plain store: read by the NEXT launch
      run += __shfl(inc, 63);
    }
    if (t == 0) { meta[0] = run; meta[1] = (int32_t)(E - run); }
  }
}

struct CompactArgs {
  const uint8_t* remove;
  int64_t E;
  const int32_t* counts;      // exclusive kept-offsets per workgroup of 1024 edges
  const int64_t *ii, *jj, *kk;
  const float *target, *weight;   // [E][2]
  const void* net;                // [E][net_bytes] (may be null)
  int net_bytes;
  int64_t *ii_o, *jj_o, *kk_o;    // kept edges, compacted
  float *target_o, *weight_o;
  void* net_o;
  int64_t *ii_r, *jj_r, *kk_r;    // removed edges appended at r0 (store = true), else null
  float *target_r, *weight_r;
  int64_t r0;
};

// pass 2: every edge finds its rank among the kept (or removed) edges of its workgroup and moves
__global__ __launch_bounds__(256) void edges_compact_kernel(CompactArgs A) {
  __shared__ int s_pre[4][4];   // [sub-tile u][wave]: kept in the waves before
  const int t = threadIdx.x, b = blockIdx.x, lane = t & 63, wave = t >> 6;
  const int64_t base = (int64_t)b * 1024;
  bool keep[4], in[4];
  int wcount[4];
#pragma unroll
  for (int u = 0; u < 4; u++) {
    const int64_t e = base + u * 256 + t;
    in[u] = e < A.E;
    keep[u] = in[u] && !A.remove[e];
    wcount[u] = __popcll(__ballot(keep[u]));
    if (lane == 0) s_pre[u][wave] = wcount[u];
  }
  __syncthreads();
  const int kbase = A.counts[b];
  const int64_t rbase = base - kbase;   // removed edges before this workgroup = edges before - kept before
  int kept_before_tile = 0;
#pragma unroll
  for (int u = 0; u < 4; u++) {
    int pre = kept_before_tile;
    for (int w = 0; w < wave; w++) pre += s_pre[u][w];
    const unsigned long long bal = __ballot(keep[u]);
    const int rank_in_wave = __popcll(bal & ((1ull << lane) - 1ull));
    const int64_t e = base + u * 256 + t;
    if (keep[u]) {
      const int64_t d = (int64_t)kbase + pre + rank_in_wave;
      A.ii_o[d] = A.ii[e]; A.jj_o[d] = A.jj[e]; A.kk_o[d] = A.kk[e];
      if (A.target) { A.target_o[2 * d] = A.target[2 * e]; A.target_o[2 * d + 1] = A.target[2 * e + 1]; }
      if (A.weight) { A.weight_o[2 * d] = A.weight[2 * e]; A.weight_o[2 * d + 1] = A.weight[2 * e + 1]; }
      if (A.net) {
        const uint32_t* s = reinterpret_cast<const uint32_t*>((const char*)A.net + (size_t)e * A.net_bytes);
        uint32_t* o = reinterpret_cast<uint32_t*>((char*)A.net_o + (size_t)d * A.net_bytes);
        for (int i = 0; i < A.net_bytes / 4; i++) o[i] = s[i];
      }
    } else if (in[u] && A.ii_r) {
      // removed rank = (lanes before in this tile that are in range) - (kept before in this tile)
      const int before_in_tile = u * 256 + t;
      const int64_t d = A.r0 + rbase + (before_in_tile - (pre + rank_in_wave));
      A.ii_r[d] = A.ii[e]; A.jj_r[d] = A.jj[e]; A.kk_r[d] = A.kk[e];
      if (A.target_r) { A.target_r[2 * d] = A.target[2 * e]; A.target_r[2 * d + 1] = A.target[2 * e + 1]; }
      if (A.weight_r) { A.weight_r[2 * d] = A.weight[2 * e]; A.weight_r[2 * d + 1] = A.weight[2 * e + 1]; }
    }
    kept_before_tile += s_pre[u][0] + s_pre[u][1] + s_pre[u][2] + s_pre[u][3];
  }
}

// keyframe(): after frame k is dropped, patches and frames above it move down (slam.py:425-427)
__global__ __launch_bounds__(256) void edges_shift_kernel(int64_t* __restrict__ ii, int64_t* __restrict__ jj,
                                                          int64_t* __restrict__ kk, int64_t E, int k, int M) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = ii[e], j = jj[e];
    if (i > k) { kk[e] -= M; ii[e] = i - 1; }
    if (j > k) jj[e] = j - 1;
  }
}

inline int grid_of(int64_t n, int per, int cap) {
  const int64_t b = (n + per - 1) / per;
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace

extern "C" int cdv_edges_frame(int64_t* ii, int64_t* jj, int64_t* kk, const int64_t* ix, int64_t E0, int64_t capacity,
                               int n, int M, int r, int64_t* added_host, void* stream) {
  CDV_REQUIRE(n >= 1 && M >= 1 && r >= 1, CDV_ERR_ARG, "cdv_edges_frame: n, M, r must be >= 1");
  const int64_t nf = (int64_t)M * ((n - 1 > 0 ? n - 1 : 0) - (n - r > 0 ? n - r : 0));
  const int64_t nb = (int64_t)M * (n - (n - r > 0 ? n - r : 0));
  const int64_t cnt = (nf > 0 ? nf : 0) + nb;
  CDV_REQUIRE(E0 >= 0 && E0 + cnt <= capacity, CDV_ERR_WORKSPACE, "cdv_edges_frame: edge capacity exceeded");
  if (added_host) *added_host = cnt;
  if (cnt == 0) return CDV_OK;
  hipLaunchKernelGGL(edges_frame_kernel, dim3(grid_of(cnt, 256, 4096)), dim3(256), 0, (hipStream_t)stream, ii, jj, kk, ix,
                     E0, n, M, r);
  CDV_LAUNCH_CHECK();
  return CDV_OK;
}

extern "C" int cdv_edges_append(int64_t* ii, int64_t* jj, int64_t* kk, const int64_t* ix, const int64_t* new_k,
                                const int64_t* new_j, int64_t E0, int64_t count, int64_t capacity, void* stream) {
  CDV_REQUIRE(E0 >= 0 && count >= 0 && E0 + count <= capacity, CDV_ERR_WORKSPACE, "cdv_edges_append: edge capacity exceeded");
  if (count == 0) return CDV_OK;
  hipLaunchKernelGGL(edges_append_kernel, dim3(grid_of(count, 256, 4096)), dim3(256), 0, (hipStream_t)stream, ii, jj, kk,
                     ix, new_k, new_j, E0, count);
  CDV_LAUNCH_CHECK();
  return CDV_OK;
}

extern "C" size_t cdv_edges_workspace_bytes(int64_t capacity) {
  return sizeof(int32_t) * (size_t)((capacity + 1023) / 1024 + 16);
}

// remove [E] uint8 (1 = drop).  ws: cdv_edges_workspace_bytes(capacity) bytes, zero-initialised ONCE by the caller.
// Outputs go to the *_out buffers (the twin buffers); removed edges are appended at inactive index r0 when ii_r != NULL.
// counts_host[0..1] (optional, pinned or pageable): kept / removed counts, valid after the stream is synchronised.
extern "C" int cdv_edges_remove(const uint8_t* remove, int64_t E, void* ws, const int64_t* ii, const int64_t* jj,
                                const int64_t* kk, const float* target, const float* weight, const void* net,
                                int net_bytes, int64_t* ii_out, int64_t* jj_out, int64_t* kk_out, float* target_out,
                                float* weight_out, void* net_out, int64_t* ii_r, int64_t* jj_r, int64_t* kk_r,
                                float* target_r, float* weight_r, int64_t r0, int32_t* counts_host, void* stream) {
  CDV_REQUIRE(E >= 0 && E < ((int64_t)1 << 31), CDV_ERR_ARG, "cdv_edges_remove: E out of range");
  CDV_REQUIRE(net == nullptr || net_bytes % 4 == 0, CDV_ERR_ARG, "cdv_edges_remove: net row size must be a multiple of 4");
  hipStream_t s = (hipStream_t)stream;
  int32_t* meta = (int32_t*)ws;          // [16]: kept, removed, arrival counter
  int32_t* counts = meta + 16;
  const int nb = grid_of(E, 1024, 1 << 22);
  hipLaunchKernelGGL(edges_count_kernel, dim3(nb), dim3(256), 0, s, remove, E, counts, meta);
  if (E > 0) {
    const CompactArgs A{remove, E, counts, ii, jj, kk, target, weight, net, net_bytes, ii_out, jj_out, kk_out, target_out,
                        weight_out, net_out, ii_r, jj_r, kk_r, target_r, weight_r, r0};
    hipLaunchKernelGGL(edges_compact_kernel, dim3(nb), dim3(256), 0, s, A);
  }
  CDV_LAUNCH_CHECK();
  if (counts_host) CDV_HIP_CHECK(hipMemcpyAsync(counts_host, meta, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  return CDV_OK;
}

// keyframe(): frame k leaves and every per-frame buffer moves frames k + 1 .. n - 1 down by one (slam.py:431-441: a
// Python loop of nine tensor copies per frame).  One launch: a thread owns one 16-byte (or 4-byte) piece of a slot and
// walks the frames in the reference's order, buf[i % m] = buf[(i + 1) % m] for i = k .. n - 2 -- the chains of different
// pieces are independent, so ring buffers that wrap (fmap / gmap rings, m < n) come out exactly as the sequential loop
// leaves them.
struct FrameBufs {
  cdv_frame_buf b[CDV_MAX_FRAME_BUFS];
  int64_t first[CDV_MAX_FRAME_BUFS + 1];   // prefix sums of the pieces per slot
  int32_t gran[CDV_MAX_FRAME_BUFS];        // bytes per piece: 16 or 4
  int n_bufs;
};

__global__ __launch_bounds__(256) void frames_shift_kernel(const FrameBufs F, int k, int n) {
  const int64_t total = F.first[F.n_bufs];
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
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

extern "C" int cdv_frames_keyframe_shift(const cdv_frame_buf* bufs, int n_bufs, int k, int n, void* stream) {
  CDV_REQUIRE(n_bufs >= 0 && n_bufs <= CDV_MAX_FRAME_BUFS, CDV_ERR_ARG, "cdv_frames_keyframe_shift: too many buffers");
  CDV_REQUIRE(k >= 0 && n >= 0, CDV_ERR_ARG, "cdv_frames_keyframe_shift: negative frame index");
  if (n_bufs == 0 || k >= n - 1) return CDV_OK;
  CDV_REQUIRE(bufs != nullptr, CDV_ERR_ARG, "cdv_frames_keyframe_shift: NULL descriptor array");
  FrameBufs F;
  F.n_bufs = n_bufs;
  F.first[0] = 0;
  for (int i = 0; i < n_bufs; i++) {
    const cdv_frame_buf& b = bufs[i];
    CDV_REQUIRE(b.base != nullptr && b.slot_bytes > 0 && b.slot_bytes % 4 == 0 && b.modulus >= 0 &&
                    ((uintptr_t)b.base & 3) == 0,
                CDV_ERR_ARG, "cdv_frames_keyframe_shift: a buffer needs a 4-byte aligned base, slot_bytes % 4 == 0, modulus >= 0");
    F.b[i] = b;
    F.gran[i] = (b.slot_bytes % 16 == 0 && ((uintptr_t)b.base & 15) == 0) ? 16 : 4;
    F.first[i + 1] = F.first[i] + b.slot_bytes / F.gran[i];
  }
  hipLaunchKernelGGL(frames_shift_kernel, dim3(grid_of(F.first[n_bufs], 256, 4096)), dim3(256), 0, (hipStream_t)stream, F,
                     k, n);
  CDV_LAUNCH_CHECK();
  return CDV_OK;
}

extern "C" int cdv_edges_keyframe_shift(int64_t* ii, int64_t* jj, int64_t* kk, int64_t E, int k, int M, void* stream) {
  if (E <= 0) return CDV_OK;
  hipLaunchKernelGGL(edges_shift_kernel, dim3(grid_of(E, 256, 4096)), dim3(256), 0, (hipStream_t)stream, ii, jj, kk, E, k, M);
  CDV_LAUNCH_CHECK();
  return CDV_OK;
}
