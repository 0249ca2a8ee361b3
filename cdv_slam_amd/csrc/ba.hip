// ba.hip -- fastba: Schur-reduced Gauss-Newton bundle adjustment over the patch graph, gfx950: the entry point
// cdv_ba_forward (replaces cuda_ba.forward, cdvslam/fastba/ba.cpp:31-45, ba_cuda.cu:462-611), the status words, and the
// path for more than 32 free poses (the global optimisation, slam.py:460-478).
//
// Dispatch on the number of free poses N:
//   1 <= N <= 10    ba_win.hip   two launches per iteration, no float atomics, bitwise reproducible
//   10 < N <= 32    ba_mid.hip   three launches per iteration, the same properties
//   N > 32 (<=1024) this file    dense E in HBM; per iteration:
//     1. ba_assemble_kernel: workgroup = (chunk of 64 unique patches, group of target slots) through the patch CSR.  A wave
//        owns ONE "target slot" t: lane = patch, edge = t-th edge of that patch in (jj, edge id) order.  Patches of one
//        source frame share their target list, so the 64 edges of a wave belong to (almost always) ONE frame pair (i, j):
//          - B blocks and v: the 13x13 Gram matrix of the wave's 128 residual rows [Ji | Jj | r], weighted by w, is ONE
//            16x16 f32 MFMA tile with K = 128; one atomic per entry per wave adds it into one of four copies of [S | y];
//          - E, C, u: lanes are consecutive unique patches, so every atomic wave-instruction is one contiguous 256-byte
//            row segment of E (the shape the memory-side atomic units run at full rate).
//     2. ba_big_schur_kernel: panel-sparse Schur products on the matrix cores; 3. fold + blocked multi-workgroup Cholesky +
//        back substitution; 4. ba_retract_kernel: dZ = Q (u - E^T dX), depth and pose update (ba_cuda.cu:178-229, 592).
//     (Float atomics: results agree with the oracle to the stated tolerances, not bit for bit between runs.)
//   N = 0           assemble + ba_schur_kernel (q only) + retract: depths alone.
#include <stdlib.h>

#include <mutex>
#include <unordered_map>

#include "cdv_ba.h"
#include "cdv_se3.h"

using namespace cdv;

CDV_STAMP_TU(ba)

namespace {

constexpr int BIG_PB = 32;        // poses per panel of the panel-sparse Schur products (192 rows; <= 32 panels: one mask word)
constexpr int BIG_PR = 6 * BIG_PB;
constexpr int XLD = 17;           // floats per residual row in the Gram staging buffer (16 + 1 pad)
constexpr int PAIR_LDS_FLOATS = 128 * XLD + 64;  // per wave: [128][XLD] rows + 64 per-edge pair keys
constexpr int ELD = BA_CHUNK + 4; // row stride of the chunk's E block in LDS (2-way bank conflicts at most)
constexpr int ASM_WAVES = 1;      // waves per assemble workgroup, one target slot each (the waves are independent;
                                  // single-wave workgroups spread the atomics of the busy chunks over all CUs)
constexpr int ASM_SG = 32;        // slot groups (workgroups) per chunk of 64 patches: 32 slots per pass (48, one pass for every patch of the
                                  // steady-state graph, measured: no change)
constexpr int ASM_THREADS = 64 * ASM_WAVES;

struct WsState {
  bool valid;
  int64_t U_max;
  int N;
  size_t bytes;
  int32_t token;      // last hand-off tag handed to a launch on this workspace (window path)
};
std::mutex g_ws_mutex;
std::unordered_map<const void*, WsState> g_ws_state;
std::unordered_map<const void*, int32_t*> g_ws_counters;   // cdv_ba_bind_status_counters

typedef EdgeFactor EdgeJ;   // residual, weights and Jacobian rows of one edge (cdv_se3.h: fastba_factor)

// Inputs of one edge, fetched ahead of use (the slot loop is software-pipelined: indices two slots
// ahead, inputs one slot ahead, so the global-load round trips overlap the Gram / E work).
struct EdgeIdx {
  int e;
  int ix, jx;
  int64_t kx;
};
struct EdgeIn {
  float pi[7], pj[7], px, py, pd, tx, ty, wx, wy;
};

__device__ __forceinline__ EdgeIdx load_idx(const int32_t* __restrict__ pcsr, const int64_t* __restrict__ ii,
                                            const int64_t* __restrict__ jj, const int64_t* __restrict__ kk, int p) {
  EdgeIdx o;
  o.e = pcsr[p];
  o.ix = (int)ii[o.e];
  o.jx = (int)jj[o.e];
  o.kx = kk[o.e];
  return o;
}

__device__ __forceinline__ EdgeIn load_in(const float* __restrict__ poses, const float* __restrict__ patches,
                                          const float* __restrict__ target, const float* __restrict__ weight,
                                          const EdgeIdx& x, int PP, int centre) {
  EdgeIn o;
#pragma unroll
  for (int a = 0; a < 7; a++) { o.pi[a] = poses[7 * (int64_t)x.ix + a]; o.pj[a] = poses[7 * (int64_t)x.jx + a]; }
  const float* pk = patches + x.kx * 3 * PP;
  o.px = pk[centre];
  o.py = pk[PP + centre];
  o.pd = pk[2 * PP + centre];
  o.tx = target[2 * (int64_t)x.e + 0];
  o.ty = target[2 * (int64_t)x.e + 1];
  o.wx = weight[2 * (int64_t)x.e + 0];
  o.wy = weight[2 * (int64_t)x.e + 1];
  return o;
}

__device__ __forceinline__ void ba_edge(const EdgeIn& in, float fx, float fy, float cx, float cy, EdgeJ& o) {
  fastba_factor(in.pi, in.pj, in.px, in.py, in.pd, in.tx, in.ty, in.wx, in.wy, fx, fy, cx, cy, o);
}

// one entry (row, col) of the 13x13 Gram matrix G = sum_k w_k X[k] X[k]^T, X[k] = [Ji | Jj | r]
__device__ __forceinline__ void pair_emit(float val, int row, int col, int ixf, int jxf, int n6,
                                          float* __restrict__ S, float* __restrict__ y) {
  if (row >= 12 || col >= 13 || val == 0.0f) return;  // row 12 duplicates column 12; (12,12) = sum w r^2
  const bool ri = row < 6;                             // row block: i (Ji) or j (Jj)
  const int rb = ri ? ixf : jxf;
  if (rb < 0) return;
  const int r = 6 * rb + (ri ? row : row - 6);
  // column 12: v[i] -= w r Ji ; v[j] += w r Jj  (ba_cuda.cu:393-398); y follows S in memory, so one atomic
  // instruction serves both.  Other columns: B[ii] += w Ji Ji^T, B[jj] += w Jj Jj^T, B[ij] -= w Ji Jj^T,
  // B[ji] -= (w Ji Jj^T)^T   (ba_cuda.cu:364-377)
  const bool isv = col == 12;
  const bool ci = col < 6;
  const int cb = isv ? 0 : (ci ? ixf : jxf);
  if (cb < 0) return;
  const int c = 6 * cb + (ci ? col : col - 6);
  const bool neg = isv ? ri : (ri != ci);
  float* dst = isv ? &y[r] : &S[r * n6 + c];
  atomicAdd(dst, neg ? -val : val);
}

struct AsmArgs {
  const float *poses, *patches, *intr, *target, *weight;
  const int64_t *ii, *jj, *kk;
  int P, t0, N;
  const int32_t *gmeta, *pcsr, *koff_u;
  float* sy;
  int sy_stride;
  float *Cg, *ug, *Edg;
  int U_stride, U_max;
  int32_t* info;
  uint32_t* cmask;
  int32_t* counters;   // optional host-visible event counters of the workspace (may be NULL)
  int first;           // first iteration of a call
};

__device__ __forceinline__ void assemble_body(const AsmArgs& A, int bid, float* smem) {
  const float* __restrict__ poses = A.poses;
  const float* __restrict__ patches = A.patches;
  const float* __restrict__ intr = A.intr;
  const float* __restrict__ target = A.target;
  const float* __restrict__ weight = A.weight;
  const int64_t* __restrict__ ii = A.ii;
  const int64_t* __restrict__ jj = A.jj;
  const int64_t* __restrict__ kk = A.kk;
  const int P = A.P, t0 = A.t0, N = A.N;
  const int32_t* __restrict__ gmeta = A.gmeta;
  const int32_t* __restrict__ pcsr = A.pcsr;
  const int32_t* __restrict__ koff_u = A.koff_u;
  float* __restrict__ sy = A.sy;
  const int sy_stride = A.sy_stride;
  float* __restrict__ Cg = A.Cg;
  float* __restrict__ ug = A.ug;
  float* __restrict__ Edg = A.Edg;
  const int U_stride = A.U_stride, U_max = A.U_max;
  int32_t* __restrict__ info = A.info;
  uint32_t* __restrict__ cmask = A.cmask;
  const int gerr = gmeta[GM_ERROR];
  const int U = gmeta[GM_U];
  if (threadIdx.x == 0 && bid == 0) ba_begin_status(info, A.counters, A.first, gerr, U > U_max);
  if (gerr || U > U_max) return;   // no index / workspace too small: BA is skipped, the status words say so
  // workgroup = (chunk of 64 unique patches, slot group): wave w takes target slot t = 8 sg + w (+ 32 per pass)
  const int chunk = bid / ASM_SG, sg = bid - chunk * ASM_SG;
  const int r0 = chunk * BA_CHUNK;
  if (r0 >= U) return;
  const int n6 = 6 * N;
  const int PP = P * P;
  const int centre = (P > 1) ? (P + 1) : 0;
  const float fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];  // ba_cuda.cu:253-259
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* S = sy + (size_t)(bid % BA_REPL) * sy_stride;
  float* y = S + (size_t)n6 * n6;
  CDV_IF_STAMPS(const int sslot = bid * ASM_WAVES + wave;)
  CDV_STAMP(ba, sslot, 0);
  CDV_STAMP_RT(ba, sslot, 8);
  float* X = smem + (size_t)wave * PAIR_LDS_FLOATS;    // per wave [128][XLD]
  int* keys = reinterpret_cast<int*>(X + 128 * XLD);   // per wave [64]

  const int r = r0 + lane;
  const int plo = (r < U) ? koff_u[r] : 0;
  const int deg = (r < U) ? koff_u[r + 1] - plo : 0;
  int maxdeg = deg;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) maxdeg = max(maxdeg, __shfl_xor(maxdeg, o));
  maxdeg = __builtin_amdgcn_readfirstlane(maxdeg);
  const int c16 = lane & 15, g4 = lane >> 4;
  const int pdef = (deg > 0) ? plo : 0;
  float Cacc = 0.f, uacc = 0.f;
  for (int t = sg * ASM_WAVES + wave; t < maxdeg; t += ASM_SG * ASM_WAVES) {
    const bool active = t < deg;
    const EdgeIdx idx = load_idx(pcsr, ii, jj, kk, active ? plo + t : pdef);
    const EdgeIn in = load_in(poses, patches, target, weight, idx, PP, centre);
    EdgeJ J;
    ba_edge(in, fx, fy, cx, cy, J);
    int ixf = -1, jxf = -1;
    if (active) {
      const int a = idx.ix - t0, b = idx.jx - t0;
      ixf = (a >= 0 && a < N) ? a : -1;
      jxf = (b >= 0 && b < N) ? b : -1;

      // E, C, u of this lane's patch (ba_cuda.cu:380-390, 401-402).  Lanes are consecutive unique patches, so
      // every atomic wave-instruction below is one contiguous 256-byte row segment of E.
      float ei[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, ej[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int row = 0; row < 2; row++) {
        const float w = J.w[row];
        const float wr = w * J.r[row], wz = w * J.Jz[row];
        Cacc += wz * J.Jz[row];
        uacc += wr * J.Jz[row];
#pragma unroll
        for (int c = 0; c < 6; c++) { ei[c] -= wz * J.Ji[6 * row + c]; ej[c] += wz * J.Jj[6 * row + c]; }
      }
      if (ixf >= 0) {
#pragma unroll
        for (int c = 0; c < 6; c++) atomicAdd(&Edg[(size_t)(6 * ixf + c) * U_stride + r], ei[c]);
      }
      if (jxf >= 0) {
#pragma unroll
        for (int c = 0; c < 6; c++) atomicAdd(&Edg[(size_t)(6 * jxf + c) * U_stride + r], ej[c]);
      }
    }
    if (cmask) {   // global-BA path: which 32-pose panels have a non-zero E block in this chunk (wave-uniform branch)
      unsigned pm = (ixf >= 0 ? 1u << (ixf >> 5) : 0u) | (jxf >= 0 ? 1u << (jxf >> 5) : 0u);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) pm |= __shfl_xor(pm, o);
      if (pm && lane == 0) atomicOr(&cmask[chunk], pm);
    }
    CDV_STAMP(ba, sslot, 3);
    if (N > 0) {
      // ---- B and v of this wave's frame pair(s): Gram matrix on the matrix cores --------------------
      const int key = (ixf + 1) * (N + 1) + (jxf + 1);
      keys[lane] = active ? key : 0;
#pragma unroll
      for (int row = 0; row < 2; row++) {
        float* xr = X + (2 * lane + row) * XLD;
#pragma unroll
        for (int c = 0; c < 6; c++) {
          xr[c] = active ? J.Ji[6 * row + c] : 0.f;
          xr[6 + c] = active ? J.Jj[6 * row + c] : 0.f;
        }
        xr[12] = active ? J.r[row] : 0.f;
        xr[13] = 0.f;
        xr[14] = 0.f;
        xr[15] = active ? J.w[row] : 0.f;
      }
      wave_lds_sync();
      CDV_STAMP(ba, sslot, 4);
      unsigned long long todo = __ballot(active && key != 0);
      CDV_IF_STAMPS(unsigned long long npass = 0, t_rd = 0, t_mf = 0, t_em = 0, t_x;)
      if (todo) {
        CDV_IF_STAMPS(t_x = cdv_now();)
        // this lane's MFMA operands of all 32 k-steps, read once (unconditional, batched LDS reads): column c16 of
        // row k = 4 st + g4, with the row's weight and its edge's pair key
        float xa[32], xw[32];
        int xk[32];
#pragma unroll
        for (int st = 0; st < 32; st++) {
          const int k = 4 * st + g4;
          xa[st] = X[k * XLD + c16];
          xw[st] = X[k * XLD + 15];
          xk[st] = keys[k >> 1];
        }
        CDV_IF_STAMPS(asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); t_rd = cdv_now() - t_x;)
        while (todo) {
          CDV_IF_STAMPS(npass++; t_x = cdv_now();)
          const int leader = __ffsll((long long)todo) - 1;
          const int kcur = __shfl(key, leader);
          const int ci = __shfl(ixf, leader), cj = __shfl(jxf, leader);
          // two accumulators: consecutive f32 MFMAs do not wait on each other's result
          cdv_float4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
          const unsigned long long match = __ballot(active && key == kcur);
#pragma unroll
          for (int st = 0; st < 32; st += 2) {
            // k-steps st, st + 1 hold the rows of edges (lanes) 2 st .. 2 st + 3: skipped when none is of this pair
            // (wave-uniform test; a chunk that straddles two source frames then costs one pass in total, not two)
            if (((match >> (2 * st)) & 15ull) == 0) continue;
            const float w0 = (xk[st] == kcur) ? xw[st] : 0.f;
            const float w1 = (xk[st + 1] == kcur) ? xw[st + 1] : 0.f;
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[st], w0 * xa[st], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[st + 1], w1 * xa[st + 1], acc1, 0, 0, 0);
          }
          CDV_IF_STAMPS(asm volatile("v_nop" :: "v"(acc0[0] + acc1[0])); t_mf += cdv_now() - t_x; t_x = cdv_now();)
          // D layout: col = lane & 15, row = 4 * (lane >> 4) + reg
#pragma unroll
          for (int q = 0; q < 4; q++) pair_emit(acc0[q] + acc1[q], 4 * g4 + q, c16, ci, cj, n6, S, y);
          CDV_IF_STAMPS(t_em += cdv_now() - t_x;)
          todo &= ~match;
        }
      }
      CDV_STAMP(ba, sslot, 5);
      CDV_STAMP_VAL(ba, sslot, 6, npass);
      CDV_STAMP_VAL(ba, sslot, 10, t_rd);
      CDV_STAMP_VAL(ba, sslot, 11, t_mf);
      CDV_STAMP_VAL(ba, sslot, 12, t_em);
      wave_lds_sync();  // the next slot overwrites X
    }
  }
  if (r < U && (Cacc != 0.f || uacc != 0.f)) {
    atomicAdd(&Cg[r], Cacc);
    atomicAdd(&ug[r], uacc);
  }
  CDV_STAMP(ba, sslot, 2);
  CDV_STAMP_RT(ba, sslot, 9);
}

__global__ __launch_bounds__(ASM_THREADS) void ba_assemble_kernel(AsmArgs A) {
  extern __shared__ float smem[];
  assemble_body(A, (int)blockIdx.x, smem);
}

// Schur products of one chunk of 64 patches, after E, C, u are complete in global memory:
//   q = 1 / (C + lambda);  S -= Ed diag(q) Ed^T;  y -= Ed (q .* u)      (ba_cuda.cu:548, 583-587)
// as [Ed; u] diag(q) [Ed; u]^T on the matrix cores (K = 64 patches, v_mfma_f32_16x16x4_f32).
struct SchurArgs {
  const float* lmbda;
  int N;
  const int32_t* gmeta;
  float* sy;
  int sy_stride;
  const float *Cg, *ug;
  float* qg;
  const float* Edg;
  int U_stride;
  int32_t* info;
};

// chunk: the 64-patch chunk; nthreads: workgroup size (256 as a launch of its own, 64 as a rider of the assemble launch)
__device__ __forceinline__ void schur_body(const SchurArgs& A, int chunk, int nthreads, float* smem) {
  const float* __restrict__ lmbda = A.lmbda;
  const int N = A.N;
  const int32_t* __restrict__ gmeta = A.gmeta;
  float* __restrict__ sy = A.sy;
  const int sy_stride = A.sy_stride;
  const float* __restrict__ Cg = A.Cg;
  const float* __restrict__ ug = A.ug;
  float* __restrict__ qg = A.qg;
  const float* __restrict__ Edg = A.Edg;
  const int U_stride = A.U_stride;
  const int U = gmeta[GM_U];
  const int r0 = chunk * BA_CHUNK;
  if (r0 >= U) return;
  const int n6 = 6 * N, nrow = n6 + 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = nthreads >> 6;
  const int c16 = lane & 15, g4 = lane >> 4;
  CDV_IF_STAMPS(const int sslot = 5000 + chunk * 4 + wave;)
  CDV_STAMP(ba, sslot, 0);
  CDV_STAMP_RT(ba, sslot, 8);
  const int T16 = (nrow + 15) / 16;
  float* Ed = smem;                          // [16 T16][ELD], row n6 = u, rows beyond it zero
  float* qs = Ed + (size_t)16 * T16 * ELD;   // [64]
  const float lm = lmbda[0];
  if (threadIdx.x < BA_CHUNK) {
    const int rr = r0 + threadIdx.x;
    const float q = (rr < U) ? 1.0f / (Cg[rr] + lm) : 0.f;
    qs[threadIdx.x] = q;
    if (rr < U) qg[rr] = q;
    Ed[n6 * ELD + threadIdx.x] = (rr < U) ? ug[rr] : 0.f;
  }
  // E rows of the chunk -> LDS: float4 loads, all of a thread's loads in flight together; rows nrow .. 16 T16 - 1
  // are zero so that the tile reads below need no bounds checks
  {
    const int tot4 = n6 * (BA_CHUNK / 4), pad4 = 16 * T16 * (BA_CHUNK / 4);
    const cdv_float4 z4 = {0.f, 0.f, 0.f, 0.f};
    for (int base = 0; base < pad4; base += 4 * nthreads) {
      cdv_float4 v[4];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int i4 = base + i * nthreads + (int)threadIdx.x;
        const int row = i4 >> 4, k4 = (i4 & 15) * 4;
        // U_stride is a multiple of 64 and the columns beyond U are kept zero by the retract kernel
        v[i] = (i4 < tot4) ? *reinterpret_cast<const cdv_float4*>(Edg + (size_t)row * U_stride + r0 + k4) : z4;
      }
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int i4 = base + i * nthreads + (int)threadIdx.x;
        const int row = i4 >> 4, k4 = (i4 & 15) * 4;
        if (i4 < pad4 && row != n6) *reinterpret_cast<cdv_float4*>(Ed + row * ELD + k4) = v[i];
      }
    }
  }
  __syncthreads();
  CDV_STAMP(ba, sslot, 1);
  if (N == 0) return;
  float* S = sy + (size_t)(chunk % BA_REPL) * sy_stride;
  float* y = S + (size_t)n6 * n6;
  const int npairs = T16 * (T16 + 1) / 2;
  for (int pidx = wave; pidx < npairs; pidx += nwaves) {
    int ti = 0, acc_rows = 0;  // lower-triangular tile pair (ti >= tj)
    while (acc_rows + ti + 1 <= pidx) { acc_rows += ti + 1; ti++; }
    const int tj = pidx - acc_rows;
    const float* pa = Ed + (size_t)(16 * ti + c16) * ELD;
    const float* pb = Ed + (size_t)(16 * tj + c16) * ELD;
    float a[16], bq[16];
#pragma unroll
    for (int st = 0; st < 16; st++) {
      const int k = 4 * st + g4;
      a[st] = pa[k];
      bq[st] = qs[k] * pb[k];
    }
    cdv_float4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int st = 0; st < 16; st += 2) {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[st], bq[st], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[st + 1], bq[st + 1], acc1, 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int R = 16 * ti + 4 * g4 + q, Cc = 16 * tj + c16;
      const float v = acc0[q] + acc1[q];
      if (v == 0.f || R >= nrow || Cc >= n6) continue;  // column n6 (u) only duplicates row n6
      if (R == n6) {
        atomicAdd(&y[Cc], -v);
      } else {
        atomicAdd(&S[R * n6 + Cc], -v);
        if (ti != tj) atomicAdd(&S[Cc * n6 + R], -v);
      }
    }
  }
  CDV_STAMP(ba, sslot, 2);
  CDV_STAMP_RT(ba, sslot, 9);
}

__global__ __launch_bounds__(1024) void ba_schur_kernel(SchurArgs A) {
  if (A.gmeta[GM_ERROR] || A.info[1]) return;
  // the dX granules {tag, value} of the following solve + retract launch (its retract workgroups poll the tags)
  if (blockIdx.x == 0 && threadIdx.x < 64) reinterpret_cast<uint64_t*>(A.info + 16)[threadIdx.x] = 0ull;
  extern __shared__ float smem[];
  schur_body(A, (int)blockIdx.x, (int)blockDim.x, smem);
}

// =========================================================================================================
// Global bundle adjustment (N > 32 free poses; slam.py:460-478 calls fastba.BA(..., eff_impl=True) over the active
// and the inactive edges).  The reference switches to a block-sparse E (block_e.cu) because a dense [6N x U] E does
// not fit its GPUs' budget; the numbers it computes -- S = B - E Q E^T, y = v - E Q u, dX, dZ -- are those of the
// dense path (ba_cuda.cu:567-580 vs :583-592).  On a 288 GB part the dense E simply stays in HBM (assemble and
// retract are unchanged); what changes is where zeros would be multiplied:
//   * the Schur products run per (chunk of 64 patches, pair of 32-pose panels), and only for panels in which the chunk
//     has a non-zero E block (one mask word per chunk, set by the assemble kernel) -- a patch is seen from ~25-40 of
//     the hundreds of free poses;
//   * the 6N x 6N system is factored by a right-looking blocked Cholesky over many workgroups (two launches per
//     64-column block: diagonal block + panel, trailing update on the matrix cores), the right-hand side carried as
//     an extra row, then one back-substitution launch.
// =========================================================================================================

// S -= E_pa diag(q) E_pb^T (lower part), y -= E_pa (q .* u) for one chunk and one pair of pose panels.
__global__ __launch_bounds__(256) void ba_big_schur_kernel(const float* __restrict__ lmbda, int N,
                                                           const int32_t* __restrict__ gmeta, float* __restrict__ sy,
                                                           int sy_stride, const float* __restrict__ Cg,
                                                           const float* __restrict__ ug, const float* __restrict__ Edg,
                                                           int U_stride, const uint32_t* __restrict__ cmask, int npair,
                                                           const int32_t* __restrict__ info) {
  if (gmeta[GM_ERROR] || info[1]) return;
  const int U = gmeta[GM_U];
  // workgroup (chunk, slot): the chunk's ACTIVE panel pairs (pa >= pb, both bits set) are dealt round-robin to
  // BIG_SLOTS workgroups -- a chunk touches 2-3 of the up to 32 panels, so almost every one of the 528 possible pairs
  // would be an empty workgroup if each had its own
  const int chunk = blockIdx.x / npair, slot = blockIdx.x - chunk * npair;   // npair == BIG_SLOTS here
  const int r0 = chunk * BA_CHUNK;
  if (r0 >= U) return;
  const uint32_t mask = cmask[chunk];
  const int n6 = 6 * N;
  int pair_idx = 0;
  for (int pa = 0; pa < 32; pa++) {
    if (!((mask >> pa) & 1u)) continue;
    for (int pb = 0; pb <= pa; pb++) {
      if (!((mask >> pb) & 1u)) continue;
      const bool mine = (pair_idx % npair) == slot;
      pair_idx++;
      if (!mine) continue;
      __syncthreads();   // the previous pair's tiles are done with the LDS panels
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ea = smem;                          // [BIG_PR][ELD]
  float* Eb = Ea + (size_t)BIG_PR * ELD;     // [BIG_PR][ELD]  (aliases Ea when pa == pb)
  float* qs = Eb + (size_t)BIG_PR * ELD;     // [64]
  float* qu = qs + BA_CHUNK;                 // [64] q .* u
  if (pa == pb) Eb = Ea;
  const float lm = lmbda[0];
  if (threadIdx.x < BA_CHUNK) {
    const int rr = r0 + threadIdx.x;
    const float q = (rr < U) ? 1.0f / (Cg[rr] + lm) : 0.f;
    qs[threadIdx.x] = q;
    qu[threadIdx.x] = (rr < U) ? q * ug[rr] : 0.f;
  }
  const cdv_float4 z4 = {0.f, 0.f, 0.f, 0.f};
  for (int half = 0; half < (pa == pb ? 1 : 2); half++) {
    float* dstE = half == 0 ? Ea : Eb;
    const int rowbase = BIG_PR * (half == 0 ? pa : pb);
    for (int i4 = threadIdx.x; i4 < BIG_PR * (BA_CHUNK / 4); i4 += 256) {
      const int row = i4 >> 4, k4 = (i4 & 15) * 4;
      const int grow = rowbase + row;
      const cdv_float4 v = (grow < n6) ? *reinterpret_cast<const cdv_float4*>(Edg + (size_t)grow * U_stride + r0 + k4) : z4;
      *reinterpret_cast<cdv_float4*>(dstE + row * ELD + k4) = v;
    }
  }
  __syncthreads();
  float* S = sy + (size_t)(blockIdx.x % BA_REPL) * sy_stride;
  float* y = S + (size_t)n6 * n6;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c16 = lane & 15, g4 = lane >> 4;
  constexpr int TT = BIG_PR / 16;   // 12 tiles per panel side
  const int ntile = (pa == pb) ? TT * (TT + 1) / 2 : TT * TT;
  for (int tix = wave; tix < ntile; tix += 4) {
    int ti, tj;
    if (pa == pb) {
      ti = 0; int ar = 0;
      while (ar + ti + 1 <= tix) { ar += ti + 1; ti++; }
      tj = tix - ar;
    } else {
      ti = tix / TT; tj = tix - ti * TT;
    }
    const float* pra = Ea + (size_t)(16 * ti + c16) * ELD;
    const float* prb = Eb + (size_t)(16 * tj + c16) * ELD;
    float a[16], bq[16];
#pragma unroll
    for (int st = 0; st < 16; st++) {
      const int k = 4 * st + g4;
      a[st] = pra[k];
      bq[st] = qs[k] * prb[k];
    }
    cdv_float4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int st = 0; st < 16; st += 2) {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[st], bq[st], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[st + 1], bq[st + 1], acc1, 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int R = BIG_PR * pa + 16 * ti + 4 * g4 + q, Cc = BIG_PR * pb + 16 * tj + c16;
      const float v = acc0[q] + acc1[q];
      if (v == 0.f || R >= n6 || Cc > R) continue;   // lower triangle only: the blocked Cholesky reads nothing else
      atomicAdd(&S[(size_t)R * n6 + Cc], -v);
    }
  }
  if (pa == pb && threadIdx.x < BIG_PR) {
    const int R = BIG_PR * pa + threadIdx.x;
    if (R < n6) {
      float sacc = 0.f;
      const float* pr = Ea + (size_t)threadIdx.x * ELD;
#pragma unroll 8
      for (int k = 0; k < BA_CHUNK; k++) sacc += pr[k] * qu[k];
      if (sacc != 0.f) atomicAdd(&y[R], -sacc);
    }
  }
    }
  }
}

// copies of [S | y] -> working matrix A [(npad + 1)][npad]: rows 0..n-1 = S with the damping of ba_cuda.cu:589, identity
// on the padded diagonal, row npad = y^T; re-zeroes the copies.
__global__ __launch_bounds__(256) void ba_big_fold_kernel(float* __restrict__ sy, int sy_stride, int n, int npad,
                                                          float* __restrict__ A, const int32_t* __restrict__ gmeta,
                                                          float* __restrict__ dbg, const int32_t* __restrict__ info) {
  if (gmeta[GM_ERROR] || info[1]) return;
  const int64_t total = (int64_t)(npad + 1) * npad;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int a = (int)(idx / npad), b = (int)(idx - (int64_t)a * npad);
    float v = 0.f;
    if (b < n && (a < n || a == npad)) {
      const size_t src = (a < n) ? (size_t)a * n + b : (size_t)n * n + b;
#pragma unroll
      for (int rep = 0; rep < BA_REPL; rep++) {
        float* p = sy + (size_t)rep * sy_stride + src;
        v += *p;
        *p = 0.f;
      }
      if (a == b) v += 1e-4f * v + 1.0f;
      if (dbg) dbg[src] = v;
    } else if (a == b) {
      v = 1.0f;
    }
    A[idx] = v;
  }
}

// 16x16 tiles of X Y^T for two 64x64 blocks in LDS (row stride CLD), K = 64: tile (ti, tj) -> lane (c16, g4) holds
// rows 16 ti + 4 g4 + q, column 16 tj + c16
constexpr int CLD = CNB + 4;
__device__ __forceinline__ cdv_float4 tile64_xyt(const float* X, const float* Y, int ti, int tj, int c16, int g4) {
  const float* pa = X + (size_t)(16 * ti + c16) * CLD;
  const float* pb = Y + (size_t)(16 * tj + c16) * CLD;
  cdv_float4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int st = 0; st < 16; st += 2) {
    const int k0 = 4 * st + g4, k1 = k0 + 4;
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[k0], pb[k0], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[k1], pb[k1], acc1, 0, 0, 0);
  }
  return acc0 + acc1;
}

// Block step kb, part 1.  One wave per workgroup, lane = matrix row.  Every workgroup factors the 64x64 diagonal block
// in its registers (redundantly; the single-wave scheme of ba_win.hip's solver: columns broadcast through LDS one ahead,
// v_pk_fma_f32 rank-1 updates, no barrier) and parks L_kk in LDS; workgroup 0 writes it back; workgroup b > 0 solves its 64 rows of
// the panel, X L_kk^T = A, by forward substitution along the row (L entries as LDS broadcast reads, 16 bytes at a
// time); the last workgroup does the same for the right-hand-side row.
__global__ __launch_bounds__(64) void ba_big_panel_kernel(float* __restrict__ A, int npad, int kb,
                                                          const int32_t* __restrict__ gmeta, int32_t* __restrict__ info) {
  if (gmeta[GM_ERROR] || info[1]) return;
  __shared__ __attribute__((aligned(16))) float Ls[CNB * CLD];   // L_kk, row stride CLD (16-byte aligned rows)
  typedef float cdv_float2 __attribute__((ext_vector_type(2)));
  const int lane = threadIdx.x;
  const int nb = npad / CNB;
  const int rb = kb + blockIdx.x;               // block row handled here; rb == nb: the right-hand-side row
  const size_t lda = (size_t)npad;
  const int c0 = CNB * kb;
  // ---- diagonal block: row `lane`, columns 0..63 -------------------------------------------------------
  cdv_float2 a2[CNB / 2];
  {
    const float* src = A + (size_t)(c0 + lane) * lda + c0;
#pragma unroll
    for (int c4 = 0; c4 < CNB / 4; c4++) {
      const cdv_float4 q = *reinterpret_cast<const cdv_float4*>(src + 4 * c4);
      a2[2 * c4] = cdv_float2{q[0], q[1]};
      a2[2 * c4 + 1] = cdv_float2{q[2], q[3]};
    }
  }
  // this workgroup's own row of the panel (unused by workgroup 0), requested now
  const bool rhs = rb == nb;
  float* rowp = A + (size_t)(rhs ? npad : CNB * min(rb, nb - 1) + lane) * lda + c0;
  float x[CNB];
#pragma unroll
  for (int c4 = 0; c4 < CNB / 4; c4++) {
    const cdv_float4 q = *reinterpret_cast<const cdv_float4*>(rowp + 4 * c4);
    x[4 * c4] = q[0]; x[4 * c4 + 1] = q[1]; x[4 * c4 + 2] = q[2]; x[4 * c4 + 3] = q[3];
  }
  // right-looking Cholesky in the wave's registers, column k broadcast through LDS ONE COLUMN AHEAD of its rank-1 update
  // (the scheme of ba_win.hip's solver: a v_readlane costs ~16 cycles of issue, so only the chain -- L[k+1][k] and the next
  // pivot -- travels that way; 2 x 2,016 of them made this kernel 21.7 us)
  __shared__ __attribute__((aligned(16))) float colb[CNB];
  bool bad = false;
  float Lk;
  {
    const float piv = readlane_f(a2[0][0], 0);
    bad = !(piv > 0.f);
    Lk = a2[0][0] * __builtin_amdgcn_rsqf(piv);
    a2[0][0] = Lk;
    colb[lane] = Lk;
  }
  cdv_float2 bcur[CNB / 2], bnxt[CNB / 2];   // column k / column k + 1 of L, the same in every lane (pairs of columns)
#pragma unroll
  for (int c4 = 0; c4 < CNB / 4; c4++) {
    const cdv_float4 v = *reinterpret_cast<const cdv_float4*>(&colb[4 * c4]);
    bcur[2 * c4] = cdv_float2{v[0], v[1]};
    bcur[2 * c4 + 1] = cdv_float2{v[2], v[3]};
  }
#pragma unroll
  for (int k = 0; k < CNB; k++) {
    float Ln = 0.f;
    if (k + 1 < CNB) {
      // column k + 1 first: its one update from column k, pivot, scale, broadcast request
      const float an = fmaf(-Lk, readlane_f(Lk, k + 1), a2[(k + 1) >> 1][(k + 1) & 1]);
      const float piv = readlane_f(an, k + 1);
      bad = bad || !(piv > 0.f);                          // wave-uniform
      Ln = an * __builtin_amdgcn_rsqf(piv);
      a2[(k + 1) >> 1][(k + 1) & 1] = Ln;
      colb[lane] = Ln;      // in-order LDS: the reads of column k were issued before this write
#pragma unroll
      for (int c4 = (k + 2) / 4; c4 < CNB / 4; c4++) {
        const cdv_float4 v = *reinterpret_cast<const cdv_float4*>(&colb[4 * c4]);
        bnxt[2 * c4] = cdv_float2{v[0], v[1]};
        bnxt[2 * c4 + 1] = cdv_float2{v[2], v[3]};
      }
    }
    // the rest of column k's rank-1 update (columns k + 2 ..) runs while column k + 1 travels through LDS
    if (((k + 2) & 1) && k + 2 < CNB)
      a2[(k + 2) >> 1][1] = fmaf(-Lk, bcur[(k + 2) >> 1][1], a2[(k + 2) >> 1][1]);
    const cdv_float2 nLk = {-Lk, -Lk};
#pragma unroll
    for (int pp = (k + 3) >> 1; pp < CNB / 2; pp++) a2[pp] = __builtin_elementwise_fma(nLk, bcur[pp], a2[pp]);
    Lk = Ln;
#pragma unroll
    for (int pp = (k + 2) >> 1; pp < CNB / 2; pp++) bcur[pp] = bnxt[pp];
  }
  // L_kk -> LDS (zeros above the diagonal), and back to the matrix from workgroup 0
#pragma unroll
  for (int c4 = 0; c4 < CNB / 4; c4++) {
    cdv_float4 q;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int c = 4 * c4 + j;
      q[j] = (c <= lane) ? a2[c >> 1][c & 1] : 0.f;
    }
    *reinterpret_cast<cdv_float4*>(&Ls[lane * CLD + 4 * c4]) = q;
    if (blockIdx.x == 0) {
      float* dst = A + (size_t)(c0 + lane) * lda + c0 + 4 * c4;
#pragma unroll
      for (int j = 0; j < 4; j++)
        if (4 * c4 + j <= lane) dst[j] = q[j];
    }
  }
  if (blockIdx.x == 0) {
    if (lane == 0 && bad && info[BI_CHOL] == 0) ba_flag(info, BI_CHOL, kb + 1);
    return;
  }
  wave_lds_sync();
  // ---- this workgroup's rows of the panel: x L^T = a, c = 0..63 in turn (the row was requested before the
  // factorisation: its memory round trip ran underneath) -----------------------------------------------------
  if (rhs && lane > 0) return;                    // the right-hand side is one row
#pragma unroll
  for (int c = 0; c < CNB; c++) {
    float sacc = x[c];
#pragma unroll
    for (int j4 = 0; j4 < (c + 3) / 4; j4++) {      // L[c][4 j4 .. 4 j4 + 3]: one broadcast read of 16 bytes
      const cdv_float4 l = *reinterpret_cast<const cdv_float4*>(&Ls[c * CLD + 4 * j4]);
#pragma unroll
      for (int j = 0; j < 4; j++)
        if (4 * j4 + j < c) sacc = fmaf(-x[4 * j4 + j], l[j], sacc);
    }
    x[c] = sacc / Ls[c * CLD + c];
  }
#pragma unroll
  for (int c4 = 0; c4 < CNB / 4; c4++)
    *reinterpret_cast<cdv_float4*>(rowp + 4 * c4) = cdv_float4{x[4 * c4], x[4 * c4 + 1], x[4 * c4 + 2], x[4 * c4 + 3]};
}

// Block step kb, part 2: trailing update A[rb][cb] -= P_rb P_cb^T for kb < cb <= rb (and the right-hand-side row
// against every cb) on the matrix cores; every target block is owned by one workgroup (no atomics).
__global__ __launch_bounds__(256) void ba_big_update_kernel(float* __restrict__ A, int npad, int kb,
                                                            const int32_t* __restrict__ gmeta,
                                                            const int32_t* __restrict__ info) {
  if (gmeta[GM_ERROR] || info[1]) return;
  __shared__ float Pr[CNB * CLD];
  __shared__ float Pc[CNB * CLD];
  const int t = threadIdx.x;
  const int nb = npad / CNB;
  const int T = nb - kb - 1;                 // trailing block rows / columns
  const int ntri = T * (T + 1) / 2;
  int rb, cb;
  bool rhs = false;
  if ((int)blockIdx.x < ntri) {
    int ri = 0, ar = 0;
    while (ar + ri + 1 <= (int)blockIdx.x) { ar += ri + 1; ri++; }
    rb = kb + 1 + ri; cb = kb + 1 + ((int)blockIdx.x - ar);
  } else {
    rhs = true; rb = nb; cb = kb + 1 + ((int)blockIdx.x - ntri);
  }
  const size_t lda = (size_t)npad;
  const int c0 = CNB * kb;
  for (int i = t; i < CNB * CNB; i += 256) {
    const int r = i >> 6, c = i & 63;
    Pr[r * CLD + c] = rhs ? (r == 0 ? A[(size_t)npad * lda + c0 + c] : 0.f) : A[(size_t)(CNB * rb + r) * lda + c0 + c];
    Pc[r * CLD + c] = A[(size_t)(CNB * cb + r) * lda + c0 + c];
  }
  __syncthreads();
  const int lane = t & 63, wave = t >> 6, c16 = lane & 15, g4 = lane >> 4;
  for (int tix = wave; tix < 16; tix += 4) {
    const int ti = tix >> 2, tj = tix & 3;
    const cdv_float4 acc = tile64_xyt(Pr, Pc, ti, tj, c16, g4);
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int r = 16 * ti + 4 * g4 + q, c = 16 * tj + c16;
      if (rhs) {
        if (r == 0) A[(size_t)npad * lda + CNB * cb + c] -= acc[q];
      } else {
        A[(size_t)(CNB * rb + r) * lda + CNB * cb + c] -= acc[q];
      }
    }
  }
}

// L^T x = z (z = row npad of A after the factorisation), one launch per 64-block from the bottom.  Every workgroup of
// step kb loads L_kk and z_kb and solves the 64 unknowns in wave 0's registers (redundantly; lane k holds column k of
// L_kk: one v_readlane + FMA per unknown); workgroup 0 publishes x_kb, workgroup w folds it into the z of its 256
// columns to the left of the block (every column is owned by one workgroup: no atomics).
__global__ __launch_bounds__(256) void ba_big_backstep_kernel(float* __restrict__ A, int npad, int n, int kb,
                                                              float* __restrict__ dXg, const int32_t* __restrict__ gmeta,
                                                              float* __restrict__ dbg, const int32_t* __restrict__ info) {
  if (gmeta[GM_ERROR] || info[1]) return;
  __shared__ __attribute__((aligned(16))) float Lb[CNB * CLD];
  __shared__ float xs[CNB];
  const int t = threadIdx.x;
  const size_t lda = (size_t)npad;
  const int c0 = CNB * kb;
  float* zrow = A + (size_t)npad * lda;
  for (int i = t; i < CNB * CNB; i += 256) {
    const int r = i >> 6, c = i & 63;
    Lb[r * CLD + c] = (c <= r) ? A[(size_t)(c0 + r) * lda + c0 + c] : 0.f;
  }
  __syncthreads();
  if (t < 64) {   // x_r = z_r / L[r][r] once every x_j, j > r, has been folded into z (as in ba_solve60_kernel)
    float col[CNB];
#pragma unroll
    for (int r = 0; r < CNB; r++) col[r] = Lb[r * CLD + t];   // column t of L_kk: L[r][t], zero for r < t
    float z = zrow[c0 + t];
    const float invd = 1.0f / Lb[t * CLD + t];
    float x = 0.f;
#pragma unroll
    for (int r = CNB - 1; r >= 0; r--) {
      const float xr = readlane_f(z * invd, r);
      x = (t == r) ? xr : x;
      z = fmaf(-col[r], xr, z);
    }
    xs[t] = x;
    if (blockIdx.x == 0 && c0 + t < n) {
      dXg[c0 + t] = x;
      if (dbg) dbg[(size_t)n * n + n + c0 + t] = x;
    }
  }
  __syncthreads();
  // z[c'] -= sum_r L[c0 + r][c'] x_r for this workgroup's columns left of the block
  const int cc = (int)blockIdx.x * 256 + t;
  if (cc < c0) {
    float sacc = 0.f;
#pragma unroll 16
    for (int r = 0; r < CNB; r++) sacc += A[(size_t)(c0 + r) * lda + cc] * xs[r];
    zrow[cc] -= sacc;
  }
}

// dZ = Q (u - E^T dX), inverse-depth update, and re-zeroing of this patch's E column / C / u so that the
// next iteration (or call) accumulates into zeros.
constexpr int RET_RG = 4;   // waves per retract workgroup: each sweeps every fourth pose's rows of the chunk's E columns

// Workgroup = one chunk of 64 patches (lane = patch) x RET_RG waves; wave g sweeps the rows of poses b = g, g + RET_RG, ..
// of the chunk's E columns (a column is 6 N entries long: one wave alone walked it in 6 N / 6 dependent round trips --
// 22 at N = 22, 299 in a global BA), the partial sums meet in LDS and wave 0 finishes the patches.
__global__ __launch_bounds__(64 * RET_RG) void ba_retract_kernel(float* __restrict__ poses, int t0, int pose_retr,
                                                        float* __restrict__ patches, int P, int N,
                                                        const int32_t* __restrict__ gmeta,
                                                        const int64_t* __restrict__ kx, float* __restrict__ Cg,
                                                        float* __restrict__ ug, const float* __restrict__ qg,
                                                        float* __restrict__ Edg, int U_stride,
                                                        const float* __restrict__ dXg, float* __restrict__ dbgp,
                                                        const int32_t* __restrict__ info,
                                                        const float* __restrict__ lmbda_q, uint32_t* __restrict__ cmask,
                                                        int n_chunks) {
  if (gmeta[GM_ERROR] || info[1]) return;
  const int U = gmeta[GM_U];
  const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
  // global-BA path: a workgroup is one chunk of 64 patches; its panel mask says which 32-pose panels of E are non-zero
  // (the rest of the column is zero and stays zero: not read, not rewritten); the mask is consumed here
  uint32_t pmask = 0xffffffffu;
  if (cmask) {
    pmask = ((int)blockIdx.x < n_chunks) ? cmask[blockIdx.x] : 0u;
    __syncthreads();
    if (threadIdx.x == 0 && (int)blockIdx.x < n_chunks) cmask[blockIdx.x] = 0u;
  }
  // pose_retr_kernel (ba_cuda.cu:178-206): T <- Exp(dX_i) T, one lane per free pose, in wave 0 of the last workgroups
  // (the first ones carry the longest E-column sweeps)
  const int gid_rev = (int)(gridDim.x * 64) - 1 - (int)(blockIdx.x * 64 + lane);
  if (pose_retr && g == 0 && gid_rev < N) {
    const int pi = gid_rev;
    float* p = poses + 7 * (size_t)(t0 + pi);
    float pose[7], xi[6];
#pragma unroll
    for (int c = 0; c < 7; c++) pose[c] = p[c];
#pragma unroll
    for (int c = 0; c < 6; c++) xi[c] = dXg[6 * pi + c];
    se3_retract_raw(xi, pose);
#pragma unroll
    for (int c = 0; c < 7; c++) p[c] = pose[c];
  }
  const int PP = P * P;
  __shared__ float part[RET_RG][64];
  const int r = (int)blockIdx.x * 64 + lane;   // the launch has one workgroup per 64 patches
  // u - E^T dX  (ba_cuda.cu:592); six independent partial sums keep six loads in flight
  float s[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (r < U) {
    for (int b = g; b < N; b += RET_RG) {
      if (!((pmask >> (b >> 5)) & 1u)) continue;
#pragma unroll
      for (int c = 0; c < 6; c++) {
        float* ep = &Edg[(size_t)(6 * b + c) * U_stride + r];
        const float ev = *ep;
        if (dbgp) dbgp[3 * (size_t)U_stride + (size_t)(6 * b + c) * U_stride + r] = ev;
        *ep = 0.f;
        s[c] += ev * dXg[6 * b + c];
      }
    }
  }
  part[g][lane] = ((s[0] + s[1]) + (s[2] + s[3])) + (s[4] + s[5]);
  __syncthreads();
  if (g != 0 || r >= U) return;
  float tot = part[0][lane];
#pragma unroll
  for (int w = 1; w < RET_RG; w++) tot += part[w][lane];
  const float cv = Cg[r], uv = ug[r];
  const float qv = lmbda_q ? 1.0f / (cv + lmbda_q[0]) : qg[r];
  const float dz = qv * (uv - tot);
  if (dbgp) { dbgp[r] = dz; dbgp[U_stride + r] = cv; dbgp[2 * (size_t)U_stride + r] = uv; }
  Cg[r] = 0.f;
  ug[r] = 0.f;
  float* pk = patches + kx[r] * 3 * PP + 2 * PP;
  float d = pk[0];                 // patch_retr_kernel reads pixel [0][0]   ba_cuda.cu:218
  d = d + dz;
  d = (d > 20.f) ? 1.0f : d;
  d = fmaxf(d, 1e-4f);
  for (int a = 0; a < PP; a++) pk[a] = d;
}

}  // namespace

// The library remembers, per workspace ADDRESS, that it has initialised the accumulators in it.  A caller that frees a
// workspace and later gets the same address back from its allocator (torch's caching allocator does that) must say so,
// or stale bytes would be taken for zeroed accumulators.
extern "C" void cdv_workspace_forget(const void* ws) {
  {
    std::lock_guard<std::mutex> lk(g_ws_mutex);
    g_ws_state.erase(ws);
    g_ws_counters.erase(ws);
  }
  cdv_graph_forget(ws);
}

// Explicit initialisation (instead of "the first time the library sees this address"): whoever allocates a workspace says so.
// The bundle-adjustment workspace needs nothing but to be unknown to the library: the first cdv_ba_forward then zeroes what
// it keeps zero; the status counters stay bound across a re-initialisation.
extern "C" int cdv_ba_workspace_init(void* ba_ws, void* stream) {
  (void)stream;
  CDV_REQUIRE(ba_ws != nullptr, CDV_ERR_ARG, "cdv_ba_workspace_init: workspace is NULL");
  std::lock_guard<std::mutex> lk(g_ws_mutex);
  g_ws_state.erase(ba_ws);
  return CDV_OK;
}

extern "C" size_t cdv_ba_workspace_bytes(int64_t E_max, int64_t U_max, int N_max) {
  (void)E_max;
  if (U_max < 1) U_max = 1;
  if (N_max < 1) N_max = 1;
  if (N_max > BA_NBIG) N_max = BA_NBIG;
  return ba_layout(U_max, N_max).total;
}

extern "C" int cdv_ba_forward(float* poses, float* patches, const float* intrinsics, const float* target,
                              const float* weight, const float* lmbda, const int64_t* ii, const int64_t* jj,
                              const int64_t* kk, int64_t E, int P, int t0, int t1, int iterations,
                              const void* graph_ws, void* ba_ws, size_t ba_ws_bytes, int64_t U_max, float* dbg,
                              void* stream) {
  const int N = t1 - t0;
  CDV_REQUIRE(N >= 0, CDV_ERR_ARG, "cdv_ba_forward: t1 < t0");
  CDV_REQUIRE(N <= BA_NBIG, CDV_ERR_UNSUPPORTED, "cdv_ba_forward: more than 1024 free poses");
  const bool big = N > BA_NMAX;   // global BA: panel-sparse Schur products + blocked multi-workgroup Cholesky
  CDV_REQUIRE(P == 3 || P == 1, CDV_ERR_UNSUPPORTED, "cdv_ba_forward: patch size P must be 3 or 1");
  CDV_REQUIRE(E >= 0 && E < ((int64_t)1 << 31), CDV_ERR_ARG, "cdv_ba_forward: E out of range");
  if (E == 0 || iterations <= 0) return CDV_OK;
  GraphLayout GL;
  CDV_REQUIRE(cdv_graph_lookup(graph_ws, &GL), CDV_ERR_ARG, "cdv_ba_forward: graph_ws has no built graph");
  CDV_REQUIRE(GL.E_max >= E, CDV_ERR_ARG, "cdv_ba_forward: graph was built for fewer edges");
  const GraphView gv = graph_view((void*)graph_ws, GL);
  CDV_REQUIRE(U_max >= 1, CDV_ERR_ARG, "cdv_ba_forward: U_max must be >= 1");
  const BaLayout L = ba_layout(U_max, N > 0 ? N : 1);
  CDV_REQUIRE(L.total <= ba_ws_bytes, CDV_ERR_WORKSPACE, "cdv_ba_forward: workspace too small for (U_max, N)");
  char* b = (char*)ba_ws;
  float* sy = (float*)(b + L.sy);
  float* dXg = (float*)(b + L.dX);
  float* Cg = (float*)(b + L.C);
  float* ug = (float*)(b + L.u);
  float* qg = (float*)(b + L.q);
  float* Edg = (float*)(b + L.Ed);
  int32_t* info = (int32_t*)(b + L.info);
  uint32_t* cmask = big ? (uint32_t*)(b + L.cmask) : nullptr;
  float* Abig = (float*)(b + L.Abig);
  hipStream_t s = (hipStream_t)stream;

  const int n6i = 6 * N;
  // The accumulators are zeroed once per (workspace, U_max, N): afterwards the solve / retract kernels leave
  // them zero, so the steady-state call enqueues no memset.
  bool fresh;
  int32_t token_base = 0;
  int32_t* counters = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_ws_mutex);
    WsState& st = g_ws_state[ba_ws];
    fresh = !(st.valid && st.U_max == U_max && st.N == N && st.bytes == ba_ws_bytes);
    const int32_t tok0 = (fresh || st.token > 0x7ffffff0 - 4 * (iterations > 0 ? iterations : 1)) ? 0 : st.token;
    token_base = tok0;
    st = WsState{true, U_max, N, ba_ws_bytes, tok0 + (iterations > 0 ? iterations : 1)};
    auto it = g_ws_counters.find(ba_ws);
    if (it != g_ws_counters.end()) counters = it->second;
  }
  // the slab paths: two launches per iteration, no float atomics (ba_win.hip up to 10 free poses, ba_mid.hip up to 32)
  const bool window = N >= 1 && N <= MID_N;
  const bool table = cdv_graph_is_table(graph_ws);
  CDV_REQUIRE(!table || window, CDV_ERR_UNSUPPORTED,
              "cdv_ba_forward: graph_ws holds a patch table (cdv_graph_build_table), which serves 1 .. 32 free poses; build "
              "the ranked index (cdv_graph_build_edges) for the global bundle adjustment");
  if (fresh) {
    if (!window) CDV_HIP_CHECK(hipMemsetAsync(b + L.sy, 0, L.zero_bytes, s));   // the window path keeps no accumulators
    CDV_HIP_CHECK(hipMemsetAsync(info, 0, sizeof(int32_t) * 16 + sizeof(uint64_t) * MID_GRAN, s));
    if (window) CDV_HIP_CHECK(hipMemsetAsync(b + L.hand, 0, sizeof(int32_t) * HAND_WORDS, s));   // token 0, no flag set
  }
  if (window) {
    BaWinArgs wa;
    wa.poses = poses; wa.patches = patches; wa.intr = intrinsics; wa.target = target; wa.weight = weight; wa.lmbda = lmbda;
    wa.ii = ii; wa.P = P; wa.t0 = t0; wa.N = N;
    wa.gmeta = gv.meta; wa.koff_u = gv.koff_u; wa.kx = gv.kx;
    wa.tdeg = gv.tdeg; wa.tplo = gv.tplo; wa.tkid = gv.tkid; wa.tab_cap = 0;
    if (table) {   // patch table: rows are slots, the overflow CSR stands where the CSR records do
      wa.prec = gv.tprec; wa.pell = gv.ttab; wa.ell_chunks = 0x7fffffff;
      wa.tab_cap = (int)cdv_graph_table_capacity(graph_ws);
      CDV_REQUIRE(wa.tab_cap >= 1 && wa.tab_cap <= U_max, CDV_ERR_ARG,
                  "cdv_ba_forward: U_max must be at least the capacity of the patch table in graph_ws");
    } else {
      wa.prec = gv.prec; wa.pell = gv.pell; wa.ell_chunks = (int)GL.ell_chunks;
    }
    wa.has_ii = cdv_graph_has_ii(graph_ws) ? 1 : 0;
    wa.slabs = (float*)(b + L.slabs); wa.ared = (float*)(b + L.ared);
    wa.arrive = (int32_t*)(b + L.hand);
    wa.granX = reinterpret_cast<uint64_t*>(info + 16);
    wa.Cg = Cg; wa.ug = ug; wa.qg = qg; wa.Edg = Edg; wa.dXg = dXg;
    wa.U_stride = (int)L.U_stride; wa.U_max = (int)L.U_max; wa.n_ck_cap = (int)L.n_ck;
    wa.info = info; wa.counters = counters;
    wa.pose_next = (float*)(b + L.pnext); wa.pose_src = nullptr;
    wa.dbg = nullptr;
    wa.token = token_base + 1;
    // CDV_BA_FUSE=1: two iterations as THREE launches (ba_win.hip cdv_ba_window_two_iterations: the first solve and the
    // second chunk pass share a grid).  Measured on the default graph: 12.8 + 26.3 + 15.8 us against 2 x (12.7 + 14.9), the
    // same 9,880 updates/s -- the chunk pass cannot start before dX exists, so only its load levels overlap the solver and
    // the saved launch boundary is given back in polling.  Kept as an option (and tested against the plain sequence), off
    // by default.
    const char* fuse_env = getenv("CDV_BA_FUSE");
    const bool fuse = fuse_env != nullptr && atoi(fuse_env) == 1;
    if (N <= WIN_N && iterations == 2 && dbg == nullptr && fuse && cdv_ba_window_can_fuse(wa))
      return cdv_ba_window_two_iterations(wa, s);
    for (int itr = 0; itr < iterations; itr++) {
      wa.dbg = (dbg && itr == 0) ? dbg : nullptr;
      wa.first = itr == 0 ? 1 : 0;
      wa.token = token_base + 1 + itr;
      const int rc = N <= WIN_N ? cdv_ba_window_iteration(wa, s) : cdv_ba_mid_iteration(wa, s);
      if (rc != CDV_OK) return rc;
    }
    return CDV_OK;
  }

  const int n_chunks = cdv_div_up(L.U_max, BA_CHUNK);
  const size_t smem_asm = sizeof(float) * ASM_WAVES * (size_t)PAIR_LDS_FLOATS;
  const size_t smem_sch = sizeof(float) * ((size_t)((n6i + 1 + 15) / 16 * 16) * ELD + BA_CHUNK);
  // raise the dynamic-LDS limits once (not a stream operation: kept out of the per-call path so that the
  // call sequence can be captured into a hipGraph)
  static std::once_flag attr_once;
  static hipError_t attr_err = hipSuccess;
  std::call_once(attr_once, [] {
    hipError_t e4 = hipFuncSetAttribute((const void*)ba_big_schur_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        112 * 1024);
    if (e4 != hipSuccess) { attr_err = e4; return; }
    hipError_t e1 = hipFuncSetAttribute((const void*)ba_assemble_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        96 * 1024);
    hipError_t e2 = hipFuncSetAttribute((const void*)ba_schur_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        64 * 1024);
    attr_err = e1 != hipSuccess ? e1 : e2;
  });
  CDV_HIP_CHECK(attr_err);
  const int rb = cdv_div_up(L.U_max > N ? L.U_max : N, 64);
  const int npair = 6;   // workgroups per chunk that share its active panel pairs (ba_big_schur_kernel)
  const int npad = (int)L.npad, nbk = npad / CNB;
  const size_t smem_bsch = sizeof(float) * (2 * (size_t)BIG_PR * ELD + 2 * BA_CHUNK);
  for (int itr = 0; itr < iterations; itr++) {
    float* d = (dbg && itr == 0) ? dbg : nullptr;
    const AsmArgs aa{poses, patches, intrinsics, target, weight, ii, jj, kk, P, t0, N, gv.meta, gv.pcsr, gv.koff_u, sy,
                     (int)L.sy_stride, Cg, ug, Edg, (int)L.U_stride, (int)L.U_max, info, cmask, counters, itr == 0 ? 1 : 0};
    const SchurArgs sa{lmbda, N, gv.meta, sy, (int)L.sy_stride, Cg, ug, qg, Edg, (int)L.U_stride, info};
    hipLaunchKernelGGL(ba_assemble_kernel, dim3(n_chunks * ASM_SG), dim3(ASM_THREADS), smem_asm, s, aa);
    if (big) {
      hipLaunchKernelGGL(ba_big_schur_kernel, dim3(n_chunks * npair), dim3(256), smem_bsch, s, lmbda, N, gv.meta, sy,
                         (int)L.sy_stride, Cg, ug, Edg, (int)L.U_stride, cmask, npair, info);
      hipLaunchKernelGGL(ba_big_fold_kernel, dim3(1024), dim3(256), 0, s, sy, (int)L.sy_stride, n6i, npad, Abig, gv.meta,
                         d, info);
      for (int kb = 0; kb < nbk; kb++) {
        hipLaunchKernelGGL(ba_big_panel_kernel, dim3(nbk - kb + 1), dim3(64), 0, s, Abig, npad, kb, gv.meta, info);
        const int T = nbk - kb - 1;
        if (T > 0)
          hipLaunchKernelGGL(ba_big_update_kernel, dim3(T * (T + 1) / 2 + T), dim3(256), 0, s, Abig, npad, kb, gv.meta,
                             info);
      }
      for (int kb = nbk - 1; kb >= 0; kb--)
        hipLaunchKernelGGL(ba_big_backstep_kernel, dim3(kb > 0 ? cdv_div_up(CNB * kb, 256) : 1), dim3(256), 0, s, Abig, npad,
                           n6i, kb, dXg, gv.meta, d, info);
    } else {
      // only N = 0 gets here (no free pose: depths alone are refined): q = 1 / (C + lambda) per chunk
      hipLaunchKernelGGL(ba_schur_kernel, dim3(n_chunks), dim3(256), smem_sch, s, sa);
    }
    // dbg layout: [S n6^2 | y n6 | dX n6 | dZ U_stride | C U_stride | u U_stride | E n6*U_stride]
    float* dbgp = d ? d + (size_t)n6i * n6i + 2 * n6i : nullptr;
    const int pose_retr = big ? 1 : 0;   // N = 0: no pose to retract
    hipLaunchKernelGGL(ba_retract_kernel, dim3(rb), dim3(64 * RET_RG), 0, s, poses, t0, pose_retr, patches, P, N, gv.meta, gv.kx,
                       Cg, ug, qg, Edg, (int)L.U_stride, dXg, dbgp, info, big ? lmbda : (const float*)nullptr, cmask,
                       n_chunks);
    CDV_LAUNCH_CHECK();
  }
  return CDV_OK;
}

// ---------------------------------------------------------------------------------------------------------
// status of a workspace
// ---------------------------------------------------------------------------------------------------------

extern "C" int cdv_ba_bind_status_counters(void* ba_ws, int32_t* counters) {
  CDV_REQUIRE(ba_ws != nullptr, CDV_ERR_ARG, "cdv_ba_bind_status_counters: workspace is NULL");
  std::lock_guard<std::mutex> lk(g_ws_mutex);
  if (counters) g_ws_counters[ba_ws] = counters;
  else g_ws_counters.erase(ba_ws);
  return CDV_OK;
}

extern "C" int cdv_ba_status(const void* ba_ws, int32_t* info_host, void* stream) {
  CDV_REQUIRE(ba_ws != nullptr && info_host != nullptr, CDV_ERR_ARG, "cdv_ba_status: NULL argument");
  WsState st;
  {
    std::lock_guard<std::mutex> lk(g_ws_mutex);
    auto it = g_ws_state.find(ba_ws);
    CDV_REQUIRE(it != g_ws_state.end() && it->second.valid, CDV_ERR_ARG, "cdv_ba_status: no cdv_ba_forward has run on this workspace");
    st = it->second;
  }
  const BaLayout L = ba_layout(st.U_max, st.N > 0 ? st.N : 1);
  CDV_HIP_CHECK(hipMemcpyAsync(info_host, (const char*)ba_ws + L.info, 4 * sizeof(int32_t), hipMemcpyDeviceToHost,
                               (hipStream_t)stream));
  CDV_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
  if (info_host[BI_GRAPH]) {
    cdv_set_error(CDV_ERR_GRAPH_RANGE, "bundle adjustment skipped: the patch-graph index reported a patch-id range beyond its capacity");
    return CDV_ERR_GRAPH_RANGE;
  }
  if (info_host[BI_OVERFLOW]) {
    cdv_set_error(CDV_ERR_BA_OVERFLOW, "bundle adjustment skipped: more unique patches than U_max");
    return CDV_ERR_BA_OVERFLOW;
  }
  if (info_host[BI_HANDOFF]) {
    cdv_set_error(CDV_ERR_BA_HANDOFF, "bundle adjustment: an in-launch hand-off timed out, the update was not applied");
    return CDV_ERR_BA_HANDOFF;
  }
  if (info_host[BI_CHOL]) {
    cdv_set_error(CDV_ERR_BA_NOT_SPD, "bundle adjustment: the reduced system is not positive definite (Cholesky pivot <= 0)");
    return CDV_ERR_BA_NOT_SPD;
  }
  return CDV_OK;
}
