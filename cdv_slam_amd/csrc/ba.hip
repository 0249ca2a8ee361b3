// ba.hip -- fastba: Schur-reduced Gauss-Newton bundle adjustment over the patch graph, gfx950.
//
// Replaces cuda_ba.forward (cdvslam/fastba/ba.cpp:31-45, ba_cuda.cu:462-611, dense-E path) without
// the reference's ~16 M contended global float atomics per iteration (ba_cuda.cu:350-402) and without
// ATen's matmul / cholesky_ex / cholesky_solve launches (ba_cuda.cu:583-592).
//
// Per Gauss-Newton iteration, three launches:
//  1. ba_assemble_kernel, two roles in one grid:
//     - pair role   : wave = (64 unique patches, target slot t) through the patch CSR; every wave holds
//                     edges of (almost always) one frame pair, so the 6x6 blocks of B and the 6-vectors of v are
//                     ONE 13x13 Gram matrix of the wave's 128 residual rows: 32 f32 MFMAs
//                     (v_mfma_f32_16x16x4_f32), then one atomic per matrix entry per wave.
//     - patch role  : 32 unique patches per workgroup through the patch CSR; the E columns, C and u of
//                     those patches are complete inside the workgroup (LDS), so the Schur products
//                     E Q E^T and E Q u are formed in LDS and only the 6N x 6N partial leaves the CU.
//     Both add into R replicas of [S | y] (S = B - E Q E^T, y = v - E Q u) to keep the number of
//     same-address memory-side atomics per replica low.
//  2. ba_solve_kernel (one workgroup): sums the replicas into LDS, damping (ba_cuda.cu:589), Cholesky,
//     forward/back substitution, pose retraction (ba_cuda.cu:178-206), re-zeroes the replicas.
//  3. ba_retract_kernel: dZ = Q (u - E^T dX) and the inverse-depth update (ba_cuda.cu:209-229, 592).
#include "cdv_common.h"
#include "cdv_graph.h"
#include "cdv_se3.h"

using namespace cdv;

namespace {

constexpr int BA_REPLICAS = 8;
constexpr int BA_CHUNK = 32;      // unique patches per patch-role workgroup
constexpr int BA_NMAX = 32;       // free poses supported by the single-workgroup solver
constexpr int PAIR_TSY = 8;       // pair-role workgroups per 64-patch chunk (4 target slots each per pass)

struct BaLayout {
  size_t sy, dX, C, u, q, Ed, info, total;
  int64_t U_max, U_stride;
  int N_max;
};

inline BaLayout ba_layout(int64_t U_max, int N_max) {
  BaLayout L;
  L.U_max = U_max; L.N_max = N_max;
  L.U_stride = (U_max + 31) / 32 * 32;
  const size_t n6 = 6 * (size_t)N_max;
  size_t o = 0;
  L.sy = o;   o = align256(o + sizeof(float) * BA_REPLICAS * (n6 * n6 + n6));
  L.dX = o;   o = align256(o + sizeof(float) * (n6 + 8));
  L.C = o;    o = align256(o + sizeof(float) * (size_t)L.U_stride);
  L.u = o;    o = align256(o + sizeof(float) * (size_t)L.U_stride);
  L.q = o;    o = align256(o + sizeof(float) * (size_t)L.U_stride);
  L.Ed = o;   o = align256(o + sizeof(float) * n6 * (size_t)L.U_stride);
  L.info = o; o = align256(o + sizeof(int32_t) * 16);
  L.total = o;
  return L;
}

struct EdgeJ {
  float r[2], w[2], Jz[2], Ji[12], Jj[12];
};

// ba_cuda.cu:261-342: residual, validity mask and the two Jacobian rows of one edge
__device__ __forceinline__ void ba_edge(const float* __restrict__ poses, const float* __restrict__ patches,
                                        float fx, float fy, float cx, float cy, const float* __restrict__ target,
                                        const float* __restrict__ weight, int64_t ix, int64_t jx, int64_t kx, int PP,
                                        int centre, int64_t n, EdgeJ& o) {
  float ti[3], tj[3], qi[4], qj[4];
#pragma unroll
  for (int a = 0; a < 3; a++) { ti[a] = poses[7 * ix + a]; tj[a] = poses[7 * jx + a]; }
#pragma unroll
  for (int a = 0; a < 4; a++) { qi[a] = poses[7 * ix + 3 + a]; qj[a] = poses[7 * jx + 3 + a]; }
  const float* pk = patches + kx * 3 * PP;
  float Xi[4], Xj[4];
  Xi[0] = (pk[centre] - cx) / fx;
  Xi[1] = (pk[PP + centre] - cy) / fy;
  Xi[2] = 1.0f;
  Xi[3] = pk[2 * PP + centre];
  float tij[3], qij[4];
  fb_relSE3(ti, qi, tj, qj, tij, qij);
  fb_actSE3(tij, qij, Xi, Xj);
  const float X = Xj[0], Y = Xj[1], Z = Xj[2], W = Xj[3];
  const float d = (Z >= 0.2f) ? 1.0f / Z : 0.0f;
  const float d2 = d * d;
  const float x1 = fx * (X / Z) + cx;
  const float y1 = fy * (Y / Z) + cy;
  const float rx = target[2 * n + 0] - x1;
  const float ry = target[2 * n + 1] - y1;
  const bool in_bounds = (sqrtf(rx * rx + ry * ry) < 128.f) && (Z > 0.2f) && (x1 > -64.f) && (y1 > -64.f) &&
                         (x1 < 2 * cx + 64.f) && (y1 < 2 * cy + 64.f);
  const float mask = in_bounds ? 1.0f : 0.0f;
  o.r[0] = rx;
  o.w[0] = mask * weight[2 * n + 0];
  o.Jz[0] = fx * (tij[0] * d - tij[2] * X * d2);
  o.Jj[0] = fx * W * d;
  o.Jj[1] = 0.0f;
  o.Jj[2] = -fx * X * W * d2;
  o.Jj[3] = -fx * X * Y * d2;
  o.Jj[4] = fx * (1.0f + X * X * d2);
  o.Jj[5] = -fx * Y * d;
  o.r[1] = ry;
  o.w[1] = mask * weight[2 * n + 1];
  o.Jz[1] = fy * (tij[1] * d - tij[2] * Y * d2);
  o.Jj[6] = 0.0f;
  o.Jj[7] = fy * W * d;
  o.Jj[8] = -fy * Y * W * d2;
  o.Jj[9] = -fy * (1.0f + Y * Y * d2);
  o.Jj[10] = fy * X * Y * d2;
  o.Jj[11] = fy * X * d;
  fb_adjSE3(tij, qij, o.Jj, o.Ji);
  fb_adjSE3(tij, qij, o.Jj + 6, o.Ji + 6);
}

// Pair-role reduction on the matrix cores.  For the 64 edges (128 residual rows) of a wave, with
// X[k] = [Ji(6) | Jj(6) | r | 0 0 | w] per residual row k, the 13x13 Gram matrix
//      G = sum_k w_k X[k] X[k]^T  =  [ Aii  Aij  vi ]      Aii = sum w Ji Ji^T, Aij = sum w Ji Jj^T,
//                                    [ Aji  Ajj  vj ]      vi = sum w r Ji, vj = sum w r Jj
//                                    [ vi^T vj^T rr ]
// is one 16x16 f32 MFMA tile with K = 128: 32 x v_mfma_f32_16x16x4_f32 (exact f32 fma chains).
// The rows are transposed from lane-per-edge to the MFMA operand layout through LDS.
constexpr int XLD = 17;                       // floats per residual row in LDS (16 + 1 pad)
constexpr int PAIR_LDS_FLOATS = 128 * XLD + 64;  // + 64 ints of per-edge pair keys

__device__ __forceinline__ void pair_emit(float val, int row, int col, int ixf, int jxf, int n6,
                                          float* __restrict__ S, float* __restrict__ y) {
  if (row >= 12 || col >= 13 || val == 0.0f) return;  // row 12 duplicates column 12; (12,12) = sum w r^2
  const bool ri = row < 6;                             // row block: i (Ji) or j (Jj)
  const int rb = ri ? ixf : jxf;
  if (rb < 0) return;
  const int r = 6 * rb + (ri ? row : row - 6);
  if (col == 12) {
    // v[i] -= w r Ji ; v[j] += w r Jj      (ba_cuda.cu:393-398)
    atomicAdd(&y[r], ri ? -val : val);
    return;
  }
  const bool ci = col < 6;
  const int cb = ci ? ixf : jxf;
  if (cb < 0) return;
  const int c = 6 * cb + (ci ? col : col - 6);
  // B[ii] += w Ji Ji^T, B[jj] += w Jj Jj^T, B[ij] -= w Ji Jj^T, B[ji] -= (w Ji Jj^T)^T   (ba_cuda.cu:364-377)
  atomicAdd(&S[r * n6 + c], (ri == ci) ? val : -val);
}

__global__ __launch_bounds__(256) void ba_assemble_kernel(
    const float* __restrict__ poses, const float* __restrict__ patches, const float* __restrict__ intr,
    const float* __restrict__ target, const float* __restrict__ weight, const float* __restrict__ lmbda,
    const int64_t* __restrict__ ii, const int64_t* __restrict__ jj, const int64_t* __restrict__ kk, int E, int P,
    int t0, int N, const int32_t* __restrict__ gmeta, const int32_t* __restrict__ pperm,
    const int32_t* __restrict__ pcsr, const int32_t* __restrict__ koff_u, const int32_t* __restrict__ ku,
    float* __restrict__ sy, float* __restrict__ Cg, float* __restrict__ ug, float* __restrict__ qg,
    float* __restrict__ Edg, int U_stride, int U_max, int n_pair_blocks, int32_t* __restrict__ info) {
  if (gmeta[GM_ERROR]) return;
  const int n6 = 6 * N;
  const int PP = P * P;
  const int centre = (P > 1) ? (P + 1) : 0;
  const float fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];  // ba_cuda.cu:253-259
  const size_t rep_stride = (size_t)n6 * n6 + n6;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

  if ((int)blockIdx.x < n_pair_blocks) {
    // ------------------------------ pair role: B and v ------------------------------------------
    if (N == 0) return;
    // wave (chunk c of 64 unique patches, target slot t): lane = patch, edge = t-th edge of the patch
    // in (jj, edge id) order.  Patches of one source frame share their target list, so a wave holds
    // one (i, j) frame pair unless the chunk straddles frames or the graph is irregular (handled by
    // the distinct-key loop below).
    const int U = gmeta[GM_U];
    const int chunk64 = blockIdx.x / PAIR_TSY, ty = blockIdx.x % PAIR_TSY;
    const int r = chunk64 * 64 + lane;
    if (chunk64 * 64 >= U || U > U_max) return;
    const int plo = (r < U) ? koff_u[r] : 0;
    const int deg = (r < U) ? koff_u[r + 1] - plo : 0;
    int maxdeg = deg;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) maxdeg = max(maxdeg, __shfl_xor(maxdeg, o));
    maxdeg = __builtin_amdgcn_readfirstlane(maxdeg);
    float* S = sy + (size_t)(blockIdx.x % BA_REPLICAS) * rep_stride;
    float* y = S + (size_t)n6 * n6;
    extern __shared__ float smem_pair[];
    float* X = smem_pair + (size_t)wave * PAIR_LDS_FLOATS;  // [128][XLD]
    int* keys = reinterpret_cast<int*>(X + 128 * XLD);      // [64]
    const int c16 = lane & 15, g4 = lane >> 4;
    for (int t = ty * 4 + wave; t < maxdeg; t += 4 * PAIR_TSY) {
    const bool active = t < deg;
    EdgeJ J;
    int ixf = -1, jxf = -1;
    if (active) {
      const int e = pcsr[plo + t];
      const int64_t ix = ii[e], jx = jj[e];
      ba_edge(poses, patches, fx, fy, cx, cy, target, weight, ix, jx, kk[e], PP, centre, e, J);
      const int64_t a = ix - t0, b = jx - t0;
      ixf = (a >= 0 && a < N) ? (int)a : -1;
      jxf = (b >= 0 && b < N) ? (int)b : -1;
    }
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
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    unsigned long long todo = __ballot(active && key != 0);
    while (todo) {
      const int leader = __ffsll((long long)todo) - 1;
      const int kcur = __shfl(key, leader);
      const int ci = __shfl(ixf, leader), cj = __shfl(jxf, leader);
      cdv_float4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
      for (int st = 0; st < 32; st++) {
        const int k = 4 * st + g4;
        const float a = X[k * XLD + c16];
        const float wk = (keys[k >> 1] == kcur) ? X[k * XLD + 15] : 0.f;
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wk * a, acc, 0, 0, 0);
      }
      // D layout: col = lane & 15, row = 4 * (lane >> 4) + reg
#pragma unroll
      for (int q = 0; q < 4; q++) pair_emit(acc[q], 4 * g4 + q, c16, ci, cj, n6, S, y);
      todo &= ~__ballot(active && key == kcur);
    }
    // the next slot overwrites X: all lanes must be done reading it
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    return;
  }

  // -------------------------------- patch role: E, C, u, Schur ----------------------------------
  const int U = gmeta[GM_U];
  if (U > U_max) {
    if (threadIdx.x == 0 && blockIdx.x == (unsigned)n_pair_blocks) info[1] = 1;  // workspace too small
    return;
  }
  const int chunk = blockIdx.x - n_pair_blocks;
  const int r0 = chunk * BA_CHUNK;
  if (r0 >= U) return;
  const int r1 = min(r0 + BA_CHUNK, U);
  extern __shared__ float smem[];
  constexpr int LD = BA_CHUNK + 1;
  float* Ed = smem;                 // [n6][LD]
  float* Cs = Ed + (size_t)n6 * LD; // [32]
  float* us = Cs + BA_CHUNK;
  float* qs = us + BA_CHUNK;
  for (int t = threadIdx.x; t < n6 * LD + 3 * BA_CHUNK; t += blockDim.x) smem[t] = 0.f;
  __syncthreads();
  const int p0 = koff_u[r0], p1 = koff_u[r1];
  for (int p = p0 + threadIdx.x; p < p1; p += blockDim.x) {
    const int e = pcsr[p];
    const int kl = ku[e] - r0;
    const int64_t ix = ii[e], jx = jj[e];
    EdgeJ J;
    ba_edge(poses, patches, fx, fy, cx, cy, target, weight, ix, jx, kk[e], PP, centre, e, J);
    const int64_t a = ix - t0, b = jx - t0;
    const bool fi = (a >= 0 && a < N), fj = (b >= 0 && b < N);
#pragma unroll
    for (int row = 0; row < 2; row++) {
      const float w = J.w[row];
      const float wr = w * J.r[row], wz = w * J.Jz[row];
      if (w != 0.f) {
#pragma unroll
        for (int c = 0; c < 6; c++) {
          if (fi) atomicAdd(&Ed[(6 * (int)a + c) * LD + kl], -wz * J.Ji[6 * row + c]);  // ba_cuda.cu:380-390
          if (fj) atomicAdd(&Ed[(6 * (int)b + c) * LD + kl], wz * J.Jj[6 * row + c]);
        }
        atomicAdd(&Cs[kl], wz * J.Jz[row]);  // ba_cuda.cu:401-402
        atomicAdd(&us[kl], wr * J.Jz[row]);
      }
    }
  }
  __syncthreads();
  const float lm = lmbda[0];
  if (threadIdx.x < BA_CHUNK) {
    const int r = r0 + threadIdx.x;
    const float q = 1.0f / (Cs[threadIdx.x] + lm);  // ba_cuda.cu:548
    qs[threadIdx.x] = (r < r1) ? q : 0.f;
    if (r < r1) { Cg[r] = Cs[threadIdx.x]; ug[r] = us[threadIdx.x]; qg[r] = q; }
  }
  __syncthreads();
  // E columns of this chunk -> global (read back by ba_retract_kernel)
  for (int t = threadIdx.x; t < n6 * BA_CHUNK; t += blockDim.x) {
    const int row = t / BA_CHUNK, kl = t % BA_CHUNK;
    if (r0 + kl < r1) Edg[(size_t)row * U_stride + r0 + kl] = Ed[row * LD + kl];
  }
  if (N == 0) return;
  // Schur partial: S -= Ed diag(q) Ed^T, y -= Ed (q .* u)      (ba_cuda.cu:583-587)
  float* S = sy + (size_t)(chunk % BA_REPLICAS) * rep_stride;
  float* y = S + (size_t)n6 * n6;
  for (int idx = threadIdx.x; idx < n6 * n6 + n6; idx += blockDim.x) {
    float acc = 0.f;
    if (idx < n6 * n6) {
      const int a = idx / n6, b = idx - a * n6;
#pragma unroll 8
      for (int k = 0; k < BA_CHUNK; k++) acc += Ed[a * LD + k] * qs[k] * Ed[b * LD + k];
    } else {
      const int a = idx - n6 * n6;
#pragma unroll 8
      for (int k = 0; k < BA_CHUNK; k++) acc += Ed[a * LD + k] * qs[k] * us[k];
    }
    if (acc != 0.f) atomicAdd((idx < n6 * n6) ? &S[idx] : &y[idx - n6 * n6], -acc);
  }
}

// One workgroup: S = sum of replicas, damping, Cholesky, solve, pose retraction.
__global__ __launch_bounds__(256) void ba_solve_kernel(float* __restrict__ poses, float* __restrict__ sy,
                                                       float* __restrict__ dXg, int t0, int N,
                                                       const int32_t* __restrict__ gmeta, float* __restrict__ dbg,
                                                       int32_t* __restrict__ info) {
  if (gmeta[GM_ERROR] || info[1]) return;
  extern __shared__ float smem[];
  const int n = 6 * N;
  const int LD = n + 1;
  float* A = smem;            // [n][LD]
  float* yv = A + (size_t)n * LD;
  const size_t rep_stride = (size_t)n * n + n;
  const int T = blockDim.x, t = threadIdx.x;
  for (int idx = t; idx < n * n + n; idx += T) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < BA_REPLICAS; r++) {
      s += sy[r * rep_stride + idx];
      sy[r * rep_stride + idx] = 0.f;  // ready for the next iteration / call
    }
    if (idx < n * n) {
      const int a = idx / n, b = idx - a * n;
      if (a == b) s += 1e-4f * s + 1.0f;  // S += I * (1e-4 * S + 1.0)   ba_cuda.cu:589
      A[a * LD + b] = s;
      if (dbg) dbg[idx] = s;
    } else {
      yv[idx - n * n] = s;
      if (dbg) dbg[idx] = s;
    }
  }
  __syncthreads();
  // left-looking Cholesky, lower triangle in place (info ignored by the reference, ba_cuda.cu:590)
  __shared__ float s_diag;
  __shared__ int s_bad;
  if (t == 0) s_bad = 0;
  for (int j = 0; j < n; j++) {
    float s = 0.f;
    const int i = j + t;
    if (i < n) {
      s = A[i * LD + j];
      for (int k = 0; k < j; k++) s -= A[i * LD + k] * A[j * LD + k];
    }
    if (t == 0) {
      if (!(s > 0.f)) s_bad = j + 1;
      s_diag = sqrtf(s);
    }
    __syncthreads();
    if (i < n) A[i * LD + j] = (t == 0) ? s_diag : s / s_diag;
    __syncthreads();
  }
  // forward substitution L z = y, then L^T x = z (one wave is plenty: n <= 192)
  for (int j = 0; j < n; j++) {
    if (t == 0) yv[j] = yv[j] / A[j * LD + j];
    __syncthreads();
    const int i = j + 1 + t;
    if (i < n) yv[i] -= A[i * LD + j] * yv[j];
    __syncthreads();
  }
  for (int j = n - 1; j >= 0; j--) {
    if (t == 0) yv[j] = yv[j] / A[j * LD + j];
    __syncthreads();
    if (t < j) yv[t] -= A[j * LD + t] * yv[j];
    __syncthreads();
  }
  if (t < n) {
    dXg[t] = yv[t];
    if (dbg) dbg[n * n + n + t] = yv[t];
  }
  if (t == 0) info[0] = s_bad;
  // pose_retr_kernel (ba_cuda.cu:178-206)
  if (t < N) {
    float* p = poses + 7 * (size_t)(t0 + t);
    float tt[3] = {p[0], p[1], p[2]}, qq[4] = {p[3], p[4], p[5], p[6]}, tn[3], qn[4], xi[6];
#pragma unroll
    for (int c = 0; c < 6; c++) xi[c] = yv[6 * t + c];
    fb_retrSE3(xi, tt, qq, tn, qn);
    p[0] = tn[0]; p[1] = tn[1]; p[2] = tn[2];
    p[3] = qn[0]; p[4] = qn[1]; p[5] = qn[2]; p[6] = qn[3];
  }
}

__global__ __launch_bounds__(256) void ba_retract_kernel(float* __restrict__ patches, int P, int N,
                                                         const int32_t* __restrict__ gmeta,
                                                         const int64_t* __restrict__ kx,
                                                         const float* __restrict__ ug, const float* __restrict__ qg,
                                                         const float* __restrict__ Edg, int U_stride,
                                                         const float* __restrict__ dXg, float* __restrict__ dZdbg,
                                                         const int32_t* __restrict__ info) {
  if (gmeta[GM_ERROR] || info[1]) return;
  const int U = gmeta[GM_U];
  const int n6 = 6 * N, PP = P * P;
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < U; r += gridDim.x * blockDim.x) {
    float s = ug[r];
    for (int a = 0; a < n6; a++) s -= Edg[(size_t)a * U_stride + r] * dXg[a];  // u - E^T dX  (ba_cuda.cu:592)
    const float dz = qg[r] * s;
    if (dZdbg) dZdbg[r] = dz;
    float* pk = patches + kx[r] * 3 * PP + 2 * PP;
    float d = pk[0];                 // patch_retr_kernel reads pixel [0][0]   ba_cuda.cu:218
    d = d + dz;
    d = (d > 20.f) ? 1.0f : d;
    d = fmaxf(d, 1e-4f);
    for (int a = 0; a < PP; a++) pk[a] = d;
  }
}

}  // namespace

extern "C" size_t cdv_ba_workspace_bytes(int64_t E_max, int64_t U_max, int N_max) {
  (void)E_max;
  if (U_max < 1) U_max = 1;
  if (N_max < 1) N_max = 1;
  if (N_max > BA_NMAX) N_max = BA_NMAX;
  return ba_layout(U_max, N_max).total;
}

extern "C" int cdv_ba_forward(float* poses, float* patches, const float* intrinsics, const float* target,
                              const float* weight, const float* lmbda, const int64_t* ii, const int64_t* jj,
                              const int64_t* kk, int64_t E, int P, int t0, int t1, int iterations,
                              const void* graph_ws, void* ba_ws, size_t ba_ws_bytes, int64_t U_max, float* dbg,
                              void* stream) {
  const int N = t1 - t0;
  CDV_REQUIRE(N >= 0, CDV_ERR_ARG, "cdv_ba_forward: t1 < t0");
  CDV_REQUIRE(N <= BA_NMAX, CDV_ERR_UNSUPPORTED,
              "cdv_ba_forward: more than 32 free poses needs the block-sparse global-BA path (not built yet)");
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
  hipStream_t s = (hipStream_t)stream;

  const int n6i = 6 * N;
  const size_t sy_bytes = sizeof(float) * BA_REPLICAS * ((size_t)n6i * n6i + n6i);
  if (sy_bytes) CDV_HIP_CHECK(hipMemsetAsync(sy, 0, sy_bytes, s));
  CDV_HIP_CHECK(hipMemsetAsync(info, 0, sizeof(int32_t) * 16, s));

  const int n_pair_blocks = (N > 0) ? cdv_div_up(L.U_max, 64) * PAIR_TSY : 0;
  const int n_chunk_blocks = cdv_div_up(L.U_max, BA_CHUNK);
  size_t smem_asm = sizeof(float) * ((size_t)n6i * (BA_CHUNK + 1) + 3 * BA_CHUNK);
  if (N > 0 && smem_asm < sizeof(float) * 4 * PAIR_LDS_FLOATS) smem_asm = sizeof(float) * 4 * PAIR_LDS_FLOATS;
  const size_t smem_sol = sizeof(float) * ((size_t)n6i * (n6i + 1) + n6i + 8);
  const int rb = cdv_div_up(L.U_max, 256);
  if (smem_sol > 48 * 1024)
    CDV_HIP_CHECK(hipFuncSetAttribute((const void*)ba_solve_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)smem_sol));
  for (int itr = 0; itr < iterations; itr++) {
    float* d = (dbg && itr == 0) ? dbg : nullptr;
    hipLaunchKernelGGL(ba_assemble_kernel, dim3(n_pair_blocks + n_chunk_blocks), dim3(256), smem_asm, s, poses,
                       patches, intrinsics, target, weight, lmbda, ii, jj, kk, (int)E, P, t0, N, gv.meta, gv.pperm,
                       gv.pcsr, gv.koff_u, gv.ku, sy, Cg, ug, qg, Edg, (int)L.U_stride, (int)L.U_max, n_pair_blocks,
                       info);
    if (N > 0)
      hipLaunchKernelGGL(ba_solve_kernel, dim3(1), dim3(256), smem_sol, s, poses, sy, dXg, t0, N, gv.meta, d, info);
    // dbg layout: [S n6^2 | y n6 | dX n6 | dZ U_stride | C U_stride | u U_stride | E n6*U_stride]
    float* dZdbg = d ? d + (size_t)n6i * n6i + 2 * n6i : nullptr;
    if (d) {
      float* q = d + (size_t)n6i * n6i + 2 * n6i + L.U_stride;
      CDV_HIP_CHECK(hipMemcpyAsync(q, Cg, sizeof(float) * L.U_stride, hipMemcpyDeviceToDevice, s));
      CDV_HIP_CHECK(hipMemcpyAsync(q + L.U_stride, ug, sizeof(float) * L.U_stride, hipMemcpyDeviceToDevice, s));
      CDV_HIP_CHECK(hipMemcpyAsync(q + 2 * L.U_stride, Edg, sizeof(float) * (size_t)n6i * L.U_stride,
                                   hipMemcpyDeviceToDevice, s));
    }
    hipLaunchKernelGGL(ba_retract_kernel, dim3(rb), dim3(256), 0, s, patches, P, N, gv.meta, gv.kx, ug, qg, Edg,
                       (int)L.U_stride, dXg, dZdbg, info);
    CDV_LAUNCH_CHECK();
  }
  return CDV_OK;
}
