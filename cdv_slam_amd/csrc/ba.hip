// ba.hip -- fastba: Schur-reduced Gauss-Newton bundle adjustment over the patch graph, gfx950: the entry point
// cdv_ba_forward (replaces cuda_ba.forward, cdvslam/fastba/ba.cpp:31-45, ba_cuda.cu:462-611), the status words, and the
// path for more than 32 free poses (the global optimisation, slam.py:460-478).
//
// Dispatch on the number of free poses N:
//   1 <= N <= 10    ba_win.hip   two launches per iteration, no float atomics, bitwise reproducible
//   10 < N <= 32    ba_mid.hip   three launches per iteration, the same properties
//   N > 32 (<=1024) this file    dense E in HBM; per iteration: patch owners (E, C, u), frame-pair owners (B, v), pose owners
//                                (diagonal blocks), tile owners (Schur products on the matrix cores), blocked multi-workgroup
//                                Cholesky, retract -- one owner and a fixed order for every sum here too (see below)
//   N = 0           ba_patch_kernel + q + retract: depths alone.
#include <stdlib.h>

#include <atomic>
#include <mutex>
#include <unordered_map>

#include "cdv_ba.h"
#include "cdv_se3.h"

using namespace cdv;

CDV_STAMP_TU(ba)

namespace {

constexpr int XLD = 17;           // floats per residual row in the Gram staging buffer (16 + 1 pad)
constexpr int ELD = BA_CHUNK + 4; // row stride of a chunk's E panel in LDS (2-way bank conflicts at most)

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
std::unordered_map<const void*, int> g_ws_ppf;              // cdv_ba_set_patches_per_frame
std::atomic<int> g_handoff_test{0};                        // cdv_ba_test_handoff: fault injection for the in-launch hand-offs

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

// =========================================================================================================
// Systems with more than 32 free poses (the global bundle adjustment; slam.py:460-478 calls fastba.BA(..., eff_impl=True)
// over the active and the inactive edges), and the structure-only call (no free pose).  The reference switches to a
// block-sparse E (block_e.cu) because a dense [6N x U] E does not fit its GPUs' budget; the numbers it computes -- S = B - E Q E^T,
// y = v - E Q u, dX, dZ -- are those of the dense path (ba_cuda.cu:567-580 vs :583-592).  On a 288 GB part the dense E stays in
// HBM.  EVERY sum below has ONE owner and a fixed order -- no float atomic, results identical from run to run:
//   ba_patch_kernel   E, C, u: a wave owns 64 unique patches (lane = patch) and walks each patch's edge list in its order;
//   ba_pair_kernel    B, v: a wave owns a FRAME PAIR {a, b} and walks its edges in edge order (the pair index: an ordinary
//                     patch-graph index built once per call over the key (a, b)); the 13 x 13 Gram matrix of the pair's
//                     residual rows [Ja | Jb | r] is one 16 x 16 f32 MFMA tile accumulated over the edges; the off-diagonal
//                     block goes straight into S, the two diagonal parts into the pair's slots of a scratch array;
//   ba_diag_kernel    a wave owns a free pose: its diagonal block and v = the pair partials in pair order;
//   ba_schur_kernel   a workgroup owns a 48 x 48 tile of S (two panels of 8 poses) and walks the chunks of 64 patches in
//                     which both panels have a non-zero E block (mask words written by ba_patch_kernel), K = 64 per chunk
//                     on the matrix cores; the panel pairs nobody sees together cost one look at the mask words;
//   then fold (damping, padding), the blocked multi-workgroup Cholesky with the right-hand side as an extra row, back
//   substitution, and ba_retract_kernel: dZ = Q (u - E^T dX), depth and pose update (ba_cuda.cu:178-229, 592).
// =========================================================================================================

struct PatchArgs {
  const float *poses, *patches, *intr, *target, *weight;
  const int64_t* ii;
  int P, t0, N;
  const int32_t *gmeta, *prec, *koff_u;
  const int64_t* kx;
  float *Cg, *ug, *Edg;
  int U_stride, U_max;
  int32_t* info;
  uint32_t* cmask;     // [chunks][BIG_MW] panel bits, or NULL (no free pose)
  int32_t* counters;   // optional host-visible event counters of the workspace (may be NULL)
  int first;           // first iteration of a call
};

struct RecIn {
  int e, ix, jx;
  float pi[7], pj[7], tx, ty, wx, wy;
};

// record p of a CSR ({edge, ii, jj, 0}: one 16-byte load) and what the edge needs of the state
__device__ __forceinline__ int4 rec_load(const int32_t* __restrict__ prec, int p) {
  return *reinterpret_cast<const int4*>(prec + 4 * (size_t)p);
}
__device__ __forceinline__ RecIn rec_inputs(const int4 rec, const int64_t* __restrict__ ii, const float* __restrict__ poses,
                                            const float* __restrict__ target, const float* __restrict__ weight) {
  RecIn o;
  o.e = rec.x;
  o.ix = rec.y >= 0 ? rec.y : (int)ii[rec.x];   // an index built without source frames: one more dependent load
  o.jx = rec.z;
#pragma unroll
  for (int a = 0; a < 7; a++) { o.pi[a] = poses[7 * (int64_t)o.ix + a]; o.pj[a] = poses[7 * (int64_t)o.jx + a]; }
  const float2 t = *reinterpret_cast<const float2*>(target + 2 * (int64_t)o.e);
  const float2 w = *reinterpret_cast<const float2*>(weight + 2 * (int64_t)o.e);
  o.tx = t.x; o.ty = t.y; o.wx = w.x; o.wy = w.y;
  return o;
}

// six entries of one pose's rows of E for patch r: stored, or added onto what is there (see ba_patch_kernel)
__device__ __forceinline__ void e_rows_out(float* __restrict__ Edg, int U_stride, int r, int b, const float (&v)[6], bool add) {
#pragma unroll
  for (int c = 0; c < 6; c++) {
    float* p = &Edg[(size_t)(6 * b + c) * U_stride + r];
    *p = add ? *p + v[c] : v[c];
  }
}

// E, C, u (ba_cuda.cu:380-390, 401-402 semantics).  One wave per chunk of 64 unique patches, lane = patch, the patch's
// edges one after the other in the order of its list -- (target frame, edge id), so the edges to one target frame are
// neighbours.  A lane sums in registers: C, u, the six E entries of the patch's source frame, and the six of the target
// frame in hand, which leave as plain stores when the target changes (E is kept zero between iterations by the retract
// kernel: a first write needs no read).  An edge from the patch's frame to itself folds into the source-frame sum.  A patch
// whose edges name more than one source frame (slam.py builds none) switches its lane to read-modify-writes: correct
// whatever the list, one lane, program order.  Lanes are consecutive unique patches: every store instruction of the wave
// writes contiguous 256-byte row segments of E.  Loads run one edge ahead (records two ahead).
__global__ __launch_bounds__(64) void ba_patch_kernel(PatchArgs A) {
  const int32_t* __restrict__ gmeta = A.gmeta;
  const int gerr = gmeta[GM_ERROR];
  const int U = gmeta[GM_U];
  if (threadIdx.x == 0 && blockIdx.x == 0) ba_begin_status(A.info, A.counters, A.first, gerr, U > A.U_max);
  if (gerr || U > A.U_max) return;   // no index / workspace too small: BA is skipped, the status words say so
  const int chunk = (int)blockIdx.x, r0 = chunk * BA_CHUNK;
  if (r0 >= U) return;
  const int lane = threadIdx.x;
  const int N = A.N, t0 = A.t0, PP = A.P * A.P;
  const int centre = (A.P > 1) ? (A.P + 1) : 0;
  const float fx = A.intr[0], fy = A.intr[1], cx = A.intr[2], cy = A.intr[3];  // ba_cuda.cu:253-259
  const int r = r0 + lane;
  const bool live = r < U;
  const int plo = live ? A.koff_u[r] : 0;
  const int deg = live ? A.koff_u[r + 1] - plo : 0;
  int maxdeg = deg;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) maxdeg = max(maxdeg, __shfl_xor(maxdeg, o));
  maxdeg = __builtin_amdgcn_readfirstlane(maxdeg);
  const int pdef = (deg > 0) ? plo : 0;
  const float* pk = A.patches + (live ? A.kx[r] : 0) * 3 * PP;
  const float px = pk[centre], py = pk[PP + centre], pd = pk[2 * PP + centre];
  float Cacc = 0.f, uacc = 0.f;
  float eacc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, jacc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int icur = -1, jcur = -1;   // free-pose numbers of the rows being summed (-1: none)
  int ifirst = -2;            // source frame of the patch's first edge (-2: no edge yet)
  bool multi = false;         // this patch's edges name more than one source frame
  uint32_t pm[BIG_MW] = {0u, 0u, 0u, 0u};
  int4 rec1 = rec_load(A.prec, (1 < deg) ? plo + 1 : pdef);
  RecIn in0 = rec_inputs(rec_load(A.prec, pdef), A.ii, A.poses, A.target, A.weight);
  for (int t = 0; t < maxdeg; t++) {
    const bool active = t < deg;
    const int4 rec2 = rec_load(A.prec, (t + 2 < deg) ? plo + t + 2 : pdef);
    const RecIn in1 = rec_inputs(rec1, A.ii, A.poses, A.target, A.weight);
    EdgeJ J;
    fastba_factor(in0.pi, in0.pj, px, py, pd, in0.tx, in0.ty, in0.wx, in0.wy, fx, fy, cx, cy, J);
    if (active) {
      const int a = in0.ix - t0, b = in0.jx - t0;
      const int ixf = (a >= 0 && a < N) ? a : -1;
      const int jxf = (b >= 0 && b < N) ? b : -1;
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
      if (ifirst == -2) ifirst = ixf;
      else if (ixf != ifirst) multi = true;
      if (ixf >= 0) {
        if (ixf != icur) {   // (only a patch with several source frames gets here with a sum in hand)
          if (icur >= 0) e_rows_out(A.Edg, A.U_stride, r, icur, eacc, true);
          icur = ixf;
#pragma unroll
          for (int c = 0; c < 6; c++) eacc[c] = 0.f;
        }
#pragma unroll
        for (int c = 0; c < 6; c++) eacc[c] += ei[c];
      }
      if (jxf >= 0) {
        if (jxf == icur) {
#pragma unroll
          for (int c = 0; c < 6; c++) eacc[c] += ej[c];
        } else {
          if (jxf != jcur) {
            if (jcur >= 0) e_rows_out(A.Edg, A.U_stride, r, jcur, jacc, multi);
            jcur = jxf;
#pragma unroll
            for (int c = 0; c < 6; c++) jacc[c] = 0.f;
          }
#pragma unroll
          for (int c = 0; c < 6; c++) jacc[c] += ej[c];
        }
      }
#pragma unroll
      for (int wd = 0; wd < BIG_MW; wd++) {   // panel = pose / 8, word = panel / 32
        if (ixf >= 0 && (ixf >> 8) == wd) pm[wd] |= 1u << ((ixf >> 3) & 31);
        if (jxf >= 0 && (jxf >> 8) == wd) pm[wd] |= 1u << ((jxf >> 3) & 31);
      }
    }
    in0 = in1;
    rec1 = rec2;
  }
  if (jcur >= 0) e_rows_out(A.Edg, A.U_stride, r, jcur, jacc, multi);
  if (icur >= 0) e_rows_out(A.Edg, A.U_stride, r, icur, eacc, multi);
  if (live) {
    A.Cg[r] = Cacc;
    A.ug[r] = uacc;
  }
  if (A.cmask) {   // which 8-pose panels have a non-zero E block in this chunk
#pragma unroll
    for (int wd = 0; wd < BIG_MW; wd++) {
      uint32_t m = pm[wd];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) m |= __shfl_xor(m, o);
      if (lane == 0) A.cmask[(size_t)chunk * BIG_MW + wd] = m;
    }
  }
}

// what a workspace's first call zeroes (accumulators, status words, hand-off words), as a kernel
__global__ __launch_bounds__(256) void ba_zero_kernel(uint32_t* __restrict__ p, int64_t n4) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const int64_t nv = n4 >> 2;
  u32x4* p4 = reinterpret_cast<u32x4*>(p);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x) p4[i] = u32x4{0u, 0u, 0u, 0u};
  if (blockIdx.x == 0 && threadIdx.x < (n4 & 3)) p[4 * nv + threadIdx.x] = 0u;
}

// q = 1 / (C + lambda) of every patch (ba_cuda.cu:548): the structure-only call, whose retract kernel reads it
__global__ __launch_bounds__(256) void ba_q_kernel(const float* __restrict__ lmbda, const int32_t* __restrict__ gmeta,
                                                   const float* __restrict__ Cg, float* __restrict__ qg,
                                                   const int32_t* __restrict__ info) {
  if (gmeta[GM_ERROR] || info[1]) return;
  const int U = gmeta[GM_U];
  const float lm = lmbda[0];
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < U; r += gridDim.x * blockDim.x) qg[r] = 1.0f / (Cg[r] + lm);
}

// ---- the frame-pair index ------------------------------------------------------------------------------------------
// key of an edge: its two poses as free-pose numbers + 1 (0: a fixed pose), smaller first
// ... and the (a, b) -> pair table goes back to zero here, two launches ahead of ba_pair_table_kernel: the library enqueues
// kernels only -- no memset node ends up in a captured hipGraph
__global__ __launch_bounds__(256) void ba_pair_key_kernel(const int64_t* __restrict__ ii, const int64_t* __restrict__ jj,
                                                          int32_t E, int t0, int N, int64_t* __restrict__ keys,
                                                          int32_t* __restrict__ ptab, int64_t pair_range) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < pair_range; i += (int64_t)gridDim.x * blockDim.x) ptab[i] = 0;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < E; e += gridDim.x * blockDim.x) {
    const int a = (int)ii[e] - t0, b = (int)jj[e] - t0;
    const int ra = (a >= 0 && a < N) ? a + 1 : 0, rb = (b >= 0 && b < N) ? b + 1 : 0;
    keys[e] = (int64_t)min(ra, rb) * (N + 1) + max(ra, rb);
  }
}

// (a, b) -> pair number + 1 (the table was zeroed: 0 = no such pair)
__global__ __launch_bounds__(256) void ba_pair_table_kernel(const int32_t* __restrict__ pmeta, const int64_t* __restrict__ pkx,
                                                            int32_t* __restrict__ ptab) {
  if (pmeta[GM_ERROR]) return;
  const int Up = pmeta[GM_U];
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < Up; p += gridDim.x * blockDim.x) ptab[pkx[p]] = p + 1;
}

struct PairArgs {
  const float *poses, *patches, *intr, *target, *weight;
  const int64_t *ii, *kk;
  int P, t0, N;
  const int32_t *gmeta;                    // of the patch index (its error state gates the whole BA)
  const int32_t *pmeta, *pprec, *pkoff;    // the pair index
  const int64_t* pkx;
  float* sy;                               // [S | y]
  float* pdiag;                            // [pairs][2][PDIAG]
  int32_t pair_cap;
  int32_t* info;
};

// B and v (ba_cuda.cu:364-377, 393-398 semantics).  One wave per frame pair {a, b}, a <= b (pose numbers + 1, 0 = fixed):
// 64 of the pair's edges at a time (lane = edge, in edge order), each lane's two residual rows
//     X = [s_a J_a | s_b J_b | r],   s = -1 for the pose the edge starts from, +1 for the one it points to
// go to LDS, and the Gram matrix G = sum_k w_k X_k X_k^T is ONE 16 x 16 f32 MFMA tile (K = 128 rows per batch) that stays
// in the accumulators across the batches.  With those signs G holds everything at once, whichever way an edge runs:
//     G[0:6, 0:6] -> B_aa, G[6:12, 6:12] -> B_bb, G[0:6, 6:12] -> B_ab, G[0:6, 12] -> v_a, G[6:12, 12] -> v_b.
// B_ab is written to S by its only owner (this wave); the diagonal parts wait in the pair's scratch slots for ba_diag_kernel.
// a == b (an edge inside one frame): everything lands on the one diagonal block, folded here.
__global__ __launch_bounds__(64) void ba_pair_kernel(PairArgs A) {
  if (A.gmeta[GM_ERROR] || A.pmeta[GM_ERROR] || A.info[1]) return;
  __shared__ float X[128 * XLD];
  __shared__ float G[16 * XLD];
  const int lane = threadIdx.x;
  const int c16 = lane & 15, g4 = lane >> 4;
  const int N = A.N, t0 = A.t0, PP = A.P * A.P, n6 = 6 * N;
  const int centre = (A.P > 1) ? (A.P + 1) : 0;
  const float fx = A.intr[0], fy = A.intr[1], cx = A.intr[2], cy = A.intr[3];
  const int Up = min(A.pmeta[GM_U], A.pair_cap);
  float* S = A.sy;
  for (int p = (int)blockIdx.x; p < Up; p += (int)gridDim.x) {
    const int64_t key = A.pkx[p];
    const int pa = (int)(key / (N + 1)), pb = (int)(key - (int64_t)pa * (N + 1));
    if (pb == 0) continue;   // both poses fixed: nothing of B or v
    const int lo = A.pkoff[p], hi = A.pkoff[p + 1];
    cdv_float4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    for (int base = lo; base < hi; base += 64) {
      const bool active = base + lane < hi;
      const RecIn in = rec_inputs(rec_load(A.pprec, active ? base + lane : lo), A.ii, A.poses, A.target, A.weight);
      const float* pk = A.patches + A.kk[in.e] * 3 * PP;
      EdgeJ J;
      fastba_factor(in.pi, in.pj, pk[centre], pk[PP + centre], pk[2 * PP + centre], in.tx, in.ty, in.wx, in.wy, fx, fy, cx, cy, J);
      // which way the edge runs: forward = it starts from pose a (for a == b both ends are pose a: forward)
      const int ai = in.ix - t0;
      const int ri = (ai >= 0 && ai < N) ? ai + 1 : 0;
      const bool fwd = ri == pa;
#pragma unroll
      for (int row = 0; row < 2; row++) {
        float* xr = X + (2 * lane + row) * XLD;
#pragma unroll
        for (int c = 0; c < 6; c++) {
          const float vi = -J.Ji[6 * row + c], vj = J.Jj[6 * row + c];
          xr[c] = active ? (fwd ? vi : vj) : 0.f;
          xr[6 + c] = active ? (fwd ? vj : vi) : 0.f;
        }
        xr[12] = active ? J.r[row] : 0.f;
        xr[13] = 0.f;
        xr[14] = 0.f;
        xr[15] = active ? J.w[row] : 0.f;
      }
      wave_lds_sync();
      const int nrow = 2 * min(64, hi - base);   // rows of this batch that carry an edge (the rest are zero: skipped)
#pragma unroll
      for (int st = 0; st < 32; st += 2) {
        if (4 * st >= nrow) break;               // wave-uniform
        const int k0 = 4 * st + g4, k1 = k0 + 4;
        const float a0 = X[k0 * XLD + c16], w0 = X[k0 * XLD + 15];
        const float a1 = X[k1 * XLD + c16], w1 = X[k1 * XLD + 15];
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, w0 * a0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, w1 * a1, acc1, 0, 0, 0);
      }
      wave_lds_sync();   // the next batch overwrites X
    }
    // G: lane (c16, g4) holds rows 4 g4 + q, column c16
#pragma unroll
    for (int q = 0; q < 4; q++) G[(4 * g4 + q) * XLD + c16] = acc0[q] + acc1[q];
    wave_lds_sync();
    float* slot0 = A.pdiag + (size_t)p * 2 * PDIAG;
    float* slot1 = slot0 + PDIAG;
    if (lane < 42) {
      // entry `lane` of a diagonal part: 36 of the 6 x 6 block (row-major), then the 6 of v
      const int rr = lane < 36 ? lane / 6 : lane - 36, cc = lane < 36 ? lane - 6 * (lane / 6) : 12;
      const float da = G[rr * XLD + cc], db = G[(6 + rr) * XLD + (cc == 12 ? 12 : 6 + cc)];
      if (pa == pb) {
        const float cross = lane < 36 ? G[rr * XLD + 6 + cc] + G[cc * XLD + 6 + rr] : 0.f;
        slot0[lane] = (da + db) + cross;
        slot1[lane] = 0.f;
      } else {
        slot0[lane] = da;    // pose a's part (never read when a is the fixed pose 0)
        slot1[lane] = db;
      }
    }
    if (pa != pb && pa >= 1 && lane < 36) {
      // B_ab: rows of pose b, columns of pose a in the lower triangle (b > a)
      const int ra = lane / 6, cb = lane - 6 * ra;
      S[(size_t)(6 * (pb - 1) + cb) * n6 + 6 * (pa - 1) + ra] = G[ra * XLD + 6 + cb];
    }
    wave_lds_sync();   // the next pair overwrites G
  }
}

// Diagonal block and v of free pose x (block x of the grid): the parts of every pair that holds x, in a fixed order --
// lane l takes the partner poses m = l, l + 64, .. (the pair (m, x + 1) for m <= x, (x + 1, m) beyond; looked up in the pair
// table), sums them in increasing m, and the 64 lane sums meet in a fixed tree.
__global__ __launch_bounds__(64) void ba_diag_kernel(const int32_t* __restrict__ gmeta, const int32_t* __restrict__ pmeta,
                                                     const int32_t* __restrict__ ptab, const float* __restrict__ pdiag,
                                                     int N, float* __restrict__ sy, const int32_t* __restrict__ info) {
  if (gmeta[GM_ERROR] || pmeta[GM_ERROR] || info[1]) return;
  const int lane = threadIdx.x;
  const int x1 = (int)blockIdx.x + 1;   // pose number + 1
  const int n6 = 6 * N;
  float acc[42];
#pragma unroll
  for (int i = 0; i < 42; i++) acc[i] = 0.f;
  for (int m = lane; m <= N; m += 64) {
    const int a = min(m, x1), b = max(m, x1);
    const int p1 = ptab[(size_t)a * (N + 1) + b];
    if (p1 == 0) continue;
    // the pair's slot 0 belongs to its smaller pose, slot 1 to the larger (a pair (x, x) has everything in slot 0)
    const float* src = pdiag + ((size_t)(p1 - 1) * 2 + (m < x1 ? 1 : 0)) * PDIAG;
#pragma unroll
    for (int i4 = 0; i4 < 40; i4 += 4) {
      const cdv_float4 v = *reinterpret_cast<const cdv_float4*>(src + i4);
#pragma unroll
      for (int h = 0; h < 4; h++) acc[i4 + h] += v[h];
    }
    acc[40] += src[40];
    acc[41] += src[41];
  }
  float* S = sy;
  float* y = S + (size_t)n6 * n6;
#pragma unroll
  for (int i = 0; i < 42; i++) {
    float v = acc[i];
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o);   // butterfly: every lane ends with the same total, same tree
    if (lane == i) {
      if (i < 36) S[(size_t)(6 * (x1 - 1) + i / 6) * n6 + 6 * (x1 - 1) + (i % 6)] = v;
      else y[6 * (x1 - 1) + (i - 36)] = v;
    }
  }
}

// S -= E Q E^T, y -= E Q u (ba_cuda.cu:583-587), lower triangle.  Workgroup = a pair of 8-pose panels (pa >= pb): the 48 x 48
// tile of S it owns is nine 16 x 16 MFMA tiles dealt to four waves, accumulated over the chunks of 64 patches whose mask
// words have both panels, in chunk order (K = 64 per chunk); q = 1 / (C + lambda) per chunk (ba_cuda.cu:548).  A diagonal
// workgroup also owns its 48 entries of y.  At the end the tile is subtracted from S (B is there already), one owner per entry.
constexpr int SPR = 6 * BIG_PP;   // rows of a panel
__global__ __launch_bounds__(256) void ba_schur_kernel(const float* __restrict__ lmbda, int N, const int32_t* __restrict__ gmeta,
                                                       float* __restrict__ sy, const float* __restrict__ Cg,
                                                       const float* __restrict__ ug, const float* __restrict__ Edg,
                                                       int U_stride, const uint32_t* __restrict__ cmask, int n_chunks,
                                                       const int32_t* __restrict__ info) {
  if (gmeta[GM_ERROR] || info[1]) return;
  const int U = gmeta[GM_U];
  const int nck = min(n_chunks, (U + BA_CHUNK - 1) / BA_CHUNK);
  // (pa, pb) of this workgroup: lower-triangular pair number blockIdx.x
  int pa = (int)((sqrtf(8.0f * (float)blockIdx.x + 1.0f) - 1.0f) * 0.5f);
  if (((pa + 1) * (pa + 2)) >> 1 <= (int)blockIdx.x) pa++;
  if ((pa * (pa + 1)) >> 1 > (int)blockIdx.x) pa--;
  const int pb = (int)blockIdx.x - ((pa * (pa + 1)) >> 1);
  __shared__ __attribute__((aligned(16))) float Ea[SPR * ELD];
  __shared__ __attribute__((aligned(16))) float Eb[SPR * ELD];
  __shared__ __attribute__((aligned(16))) float qs[BA_CHUNK];
  __shared__ float qu[BA_CHUNK];
  __shared__ int lst[256];
  __shared__ int wcnt[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c16 = lane & 15, g4 = lane >> 4;
  const int n6 = 6 * N;
  const float lm = lmbda[0];
  const uint32_t bita = 1u << (pa & 31), bitb = 1u << (pb & 31);
  const int wa = pa >> 5, wb = pb >> 5;
  const float* Ebp = (pa == pb) ? Ea : Eb;
  cdv_float4 acc[3];
#pragma unroll
  for (int u = 0; u < 3; u++) acc[u] = cdv_float4{0.f, 0.f, 0.f, 0.f};
  float yacc = 0.f;
  const cdv_float4 z4 = {0.f, 0.f, 0.f, 0.f};
  for (int base = 0; base < nck; base += 256) {   // workgroup-uniform
    // the chunks of this batch that hold both panels, in chunk order
    const int c = base + tid;
    const bool hit = c < nck && (cmask[(size_t)c * BIG_MW + wa] & bita) && (cmask[(size_t)c * BIG_MW + wb] & bitb);
    const unsigned long long bal = __ballot(hit);
    if (lane == 0) wcnt[wave] = __popcll(bal);
    __syncthreads();
    int before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) { const int n = wcnt[w]; before += (w < wave) ? n : 0; total += n; }
    if (hit) lst[before + __popcll(bal & ((1ull << lane) - 1ull))] = c;
    __syncthreads();
    // the panels of a chunk travel memory -> registers -> LDS, one chunk ahead of the products (a chunk's MFMA work is
    // shorter than its memory round trip)
    constexpr int LPT = SPR * (BA_CHUNK / 4) / 256;   // 16-byte loads per thread and panel (3)
    cdv_float4 ra[LPT], rbv[LPT];
    float rq = 0.f, rqu = 0.f;
    const auto fetch = [&](int chunk) {
      const int r0 = chunk * BA_CHUNK;
      if (tid < BA_CHUNK) {
        const int rr = r0 + tid;
        rq = (rr < U) ? 1.0f / (Cg[rr] + lm) : 0.f;
        rqu = (rr < U) ? rq * ug[rr] : 0.f;
      }
#pragma unroll
      for (int l = 0; l < LPT; l++) {
        const int i4 = tid + 256 * l;
        const int row = i4 >> 4, k4 = (i4 & 15) * 4;
        const int ga = SPR * pa + row, gb = SPR * pb + row;
        ra[l] = (ga < n6) ? *reinterpret_cast<const cdv_float4*>(Edg + (size_t)ga * U_stride + r0 + k4) : z4;
        rbv[l] = (pa != pb && gb < n6) ? *reinterpret_cast<const cdv_float4*>(Edg + (size_t)gb * U_stride + r0 + k4) : z4;
      }
    };
    if (total > 0) fetch(lst[0]);
    for (int i = 0; i < total; i++) {
      if (tid < BA_CHUNK) { qs[tid] = rq; qu[tid] = rqu; }
#pragma unroll
      for (int l = 0; l < LPT; l++) {
        const int i4 = tid + 256 * l;
        const int row = i4 >> 4, k4 = (i4 & 15) * 4;
        *reinterpret_cast<cdv_float4*>(Ea + row * ELD + k4) = ra[l];
        if (pa != pb) *reinterpret_cast<cdv_float4*>(Eb + row * ELD + k4) = rbv[l];
      }
      __syncthreads();
      if (i + 1 < total) fetch(lst[i + 1]);   // in flight during the products
      // nine 16 x 16 tiles on four waves: tiles 0 .. 7 two per wave, the ninth -- (2, 2) -- cut along k, wave w taking the k
      // steps 4 w .. 4 w + 3 of every lane group (round 5: as a whole tile of wave 0 it made that wave's 48 MFMAs the chunk's
      // critical path, 36 now); its four parts meet once, behind the loop
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const int tix = wave + 4 * u;   // tiles 0 .. 7: (ti, tj) = (tix / 3, tix % 3)
        const int ti = tix / 3, tj = tix - 3 * ti;
        if (pa == pb && tj > ti) continue;
        const float* pra = Ea + (size_t)(16 * ti + c16) * ELD;
        const float* prb = Ebp + (size_t)(16 * tj + c16) * ELD;
        // the k index of an MFMA step is ours to choose (the same for both operands): lane group g4 takes the 16 patches
        // 16 g4 .. 16 g4 + 15 of the chunk, so every operand is four 16-byte reads instead of sixteen 4-byte ones
        cdv_float4 t0 = {0.f, 0.f, 0.f, 0.f}, t1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++) {
          const int k = 16 * g4 + 4 * s4;
          const cdv_float4 av = *reinterpret_cast<const cdv_float4*>(pra + k);
          const cdv_float4 bv = *reinterpret_cast<const cdv_float4*>(prb + k);
          const cdv_float4 qv = *reinterpret_cast<const cdv_float4*>(qs + k);
          t0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], qv[0] * bv[0], t0, 0, 0, 0);
          t1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1], qv[1] * bv[1], t1, 0, 0, 0);
          t0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2], qv[2] * bv[2], t0, 0, 0, 0);
          t1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[3], qv[3] * bv[3], t1, 0, 0, 0);
        }
        acc[u] += t0 + t1;
      }
      {
        const float* pra = Ea + (size_t)(32 + c16) * ELD;
        const float* prb = Ebp + (size_t)(32 + c16) * ELD;
        const int k = 16 * g4 + 4 * wave;
        const cdv_float4 av = *reinterpret_cast<const cdv_float4*>(pra + k);
        const cdv_float4 bv = *reinterpret_cast<const cdv_float4*>(prb + k);
        const cdv_float4 qv = *reinterpret_cast<const cdv_float4*>(qs + k);
        cdv_float4 t0 = {0.f, 0.f, 0.f, 0.f}, t1 = {0.f, 0.f, 0.f, 0.f};
        t0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], qv[0] * bv[0], t0, 0, 0, 0);
        t1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1], qv[1] * bv[1], t1, 0, 0, 0);
        t0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2], qv[2] * bv[2], t0, 0, 0, 0);
        t1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[3], qv[3] * bv[3], t1, 0, 0, 0);
        acc[2] += t0 + t1;
      }
      if (pa == pb && tid < SPR) {
        float sacc = 0.f;
        const float* pr = Ea + (size_t)tid * ELD;
#pragma unroll 8
        for (int k = 0; k < BA_CHUNK; k++) sacc += pr[k] * qu[k];
        yacc += sacc;
      }
      __syncthreads();   // the next chunk overwrites the panels
    }
  }
  // the four k parts of tile (2, 2), in wave order, into wave 0 (the panels' LDS is free now)
  {
    cdv_float4* part = reinterpret_cast<cdv_float4*>(Ea);
    __syncthreads();
    part[tid] = acc[2];
    __syncthreads();
    if (wave == 0) acc[2] = (part[lane] + part[64 + lane]) + (part[128 + lane] + part[192 + lane]);
  }
  float* S = sy;
  float* y = S + (size_t)n6 * n6;
#pragma unroll
  for (int u = 0; u < 3; u++) {
    const int tix = wave + 4 * u;
    if (tix >= 9) continue;
    const int ti = tix / 3, tj = tix - 3 * ti;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int R = SPR * pa + 16 * ti + 4 * g4 + q, Cc = SPR * pb + 16 * tj + c16;
      const float v = acc[u][q];
      if (v == 0.f || R >= n6 || Cc > R) continue;   // lower triangle only: the blocked Cholesky reads nothing else
      S[(size_t)R * n6 + Cc] -= v;
    }
  }
  if (pa == pb && tid < SPR && SPR * pa + tid < n6 && yacc != 0.f) y[SPR * pa + tid] -= yacc;
}

// [S | y] -> working matrix A [(npad + 1)][npad]: rows 0..n-1 = S with the damping of ba_cuda.cu:589, identity
// on the padded diagonal, row npad = y^T; re-zeroes [S | y] (its owners write only the blocks that exist).
__global__ __launch_bounds__(256) void ba_big_fold_kernel(float* __restrict__ sy, int sy_stride, int n, int npad,
                                                          float* __restrict__ A, const int32_t* __restrict__ gmeta,
                                                          float* __restrict__ dbg, const int32_t* __restrict__ info,
                                                          uint64_t* __restrict__ xg, int32_t* __restrict__ fctl, int n_fctl) {
  if (gmeta[GM_ERROR] || info[1]) return;
  // the granules of the back-substitution launch lose their tags (per-launch tokens from host state at enqueue time: a
  // captured hipGraph replays the same ones); the factorisation launch's ticket counter, abort word and block flags go down
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < npad; i += gridDim.x * blockDim.x) xg[i] = 0ull;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_fctl; i += gridDim.x * blockDim.x) fctl[i] = 0;
  const int64_t total = (int64_t)(npad + 1) * npad;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int a = (int)(idx / npad), b = (int)(idx - (int64_t)a * npad);
    float v = 0.f;
    if (b < n && (a < n || a == npad)) {
      const size_t src = (a < n) ? (size_t)a * n + b : (size_t)n * n + b;
      float* p = sy + src;
      v = *p;
      *p = 0.f;
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
  // (lane group g4 takes columns 16 g4 .. 16 g4 + 15: four 16-byte reads per operand; rows are 16-byte aligned, CLD = 68)
  cdv_float4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s4 = 0; s4 < 4; s4++) {
    const int k = 16 * g4 + 4 * s4;
    const cdv_float4 av = *reinterpret_cast<const cdv_float4*>(pa + k);
    const cdv_float4 bv = *reinterpret_cast<const cdv_float4*>(pb + k);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], bv[0], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1], bv[1], acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2], bv[2], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[3], bv[3], acc1, 0, 0, 0);
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

// Block step kb >= 1 as ONE launch, the trailing update delayed by a step: what step kb - 1 owes the matrix is applied
//   * to block column kb by the panel workgroups themselves, each to its own block and (redundantly) to the diagonal block,
//     on the matrix cores, straight into the LDS the panel wave then reads its rows from;
//   * to the blocks right of column kb by update workgroups, as in ba_big_update_kernel;
// both need only panel kb - 1 (the previous launch) and touch disjoint blocks, so the panel chain of step kb (one wave,
// ~18 us) no longer waits for a trailing-update launch of its own: 29 x (20 + 14.5 us + two launch gaps) became 29 x ~23.
// Workgroups 0 .. nb - kb: panel (0 = diagonal block, last = the right-hand-side row); the rest: updates.
constexpr int STEP_T = 512;   // threads of a step workgroup: eight waves share the 16 (update) or 32 (panel) tiles
__global__ __launch_bounds__(STEP_T) void ba_big_step_kernel(float* __restrict__ A, int npad, int kb,
                                                             const int32_t* __restrict__ gmeta, int32_t* __restrict__ info) {
  if (gmeta[GM_ERROR] || info[1]) return;
  __shared__ __attribute__((aligned(16))) float Pr[CNB * CLD];
  __shared__ __attribute__((aligned(16))) float Pc[CNB * CLD];
  __shared__ __attribute__((aligned(16))) float Ls[CNB * CLD];
  __shared__ __attribute__((aligned(16))) float colb[CNB];
  typedef float cdv_float2 __attribute__((ext_vector_type(2)));
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, c16 = lane & 15, g4 = lane >> 4;
  const int nb = npad / CNB;
  const size_t lda = (size_t)npad;
  const int c0 = CNB * kb, cp = CNB * (kb - 1);   // columns of this step's panel / of the panel whose update is due
  const int nA = nb - kb + 1;
  if ((int)blockIdx.x >= nA) {
    // ---- update role: block (rb, cb), cb > kb, or the right-hand-side row against column block cb ----
    const int idx = (int)blockIdx.x - nA;
    const int T = nb - kb - 1;
    const int ntri = T * (T + 1) / 2;
    int rb, cb;
    bool rhs = false;
    if (idx < ntri) {
      int ri = 0, ar = 0;
      while (ar + ri + 1 <= idx) { ar += ri + 1; ri++; }
      rb = kb + 1 + ri; cb = kb + 1 + (idx - ar);
    } else {
      rhs = true; rb = nb; cb = kb + 1 + (idx - ntri);
    }
    for (int i = t; i < CNB * CNB; i += STEP_T) {
      const int r = i >> 6, c = i & 63;
      Pr[r * CLD + c] = rhs ? (r == 0 ? A[(size_t)npad * lda + cp + c] : 0.f) : A[(size_t)(CNB * rb + r) * lda + cp + c];
      Pc[r * CLD + c] = A[(size_t)(CNB * cb + r) * lda + cp + c];
    }
    __syncthreads();
    for (int tix = wave; tix < 16; tix += STEP_T / 64) {
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
    return;
  }
  // ---- panel role ----
  const int rb = kb + (int)blockIdx.x;          // block row handled here; rb == nb: the right-hand-side row
  const bool rhs = rb == nb;
  const bool diag = blockIdx.x == 0;
  // panel kb - 1: this workgroup's rows (Pr) and the rows of block row kb (Pc); and, in the same round trip, the entries of
  // the own block and of the diagonal block this wave's tiles will be subtracted from
  constexpr int TU = 16 / (STEP_T / 64);   // tiles of each of the two blocks per wave
  float o[TU][4], d[TU][4];
#pragma unroll
  for (int u = 0; u < TU; u++) {
    const int tix = wave + (STEP_T / 64) * u, ti = tix >> 2, tj = tix & 3;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int r = 16 * ti + 4 * g4 + q, c = 16 * tj + c16;
      o[u][q] = rhs ? (r == 0 ? A[(size_t)npad * lda + c0 + c] : 0.f) : A[(size_t)(CNB * rb + r) * lda + c0 + c];
      d[u][q] = A[(size_t)(c0 + r) * lda + c0 + c];
    }
  }
  for (int i = t; i < CNB * CNB; i += STEP_T) {
    const int r = i >> 6, c = i & 63;
    Pr[r * CLD + c] = rhs ? (r == 0 ? A[(size_t)npad * lda + cp + c] : 0.f) : A[(size_t)(CNB * rb + r) * lda + cp + c];
    Pc[r * CLD + c] = A[(size_t)(CNB * kb + r) * lda + cp + c];
  }
  __syncthreads();
  // the 16 tiles of the own block and the 16 of the diagonal block, dealt to the waves: products in registers ...
  cdv_float4 own[TU], dia[TU];
#pragma unroll
  for (int u = 0; u < TU; u++) {
    const int tix = wave + (STEP_T / 64) * u, ti = tix >> 2, tj = tix & 3;
    own[u] = tile64_xyt(Pr, Pc, ti, tj, c16, g4);
    dia[u] = tile64_xyt(Pc, Pc, ti, tj, c16, g4);
  }
  __syncthreads();   // everybody is done reading the panels
  // ... subtracted from the blocks, the results parked where the panels were: Pr <- own block (the right-hand side: its
  // row 0), Pc <- diagonal block
#pragma unroll
  for (int u = 0; u < TU; u++) {
    const int tix = wave + (STEP_T / 64) * u, ti = tix >> 2, tj = tix & 3;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int r = 16 * ti + 4 * g4 + q, c = 16 * tj + c16;
      Pr[r * CLD + c] = o[u][q] - own[u][q];
      Pc[r * CLD + c] = d[u][q] - dia[u][q];
    }
  }
  __syncthreads();
  if (wave != 0) return;
  // ---- from here on: ba_big_panel_kernel's single wave, its rows read from LDS ----
  cdv_float2 a2[CNB / 2];
#pragma unroll
  for (int c4 = 0; c4 < CNB / 4; c4++) {
    const cdv_float4 q = *reinterpret_cast<const cdv_float4*>(&Pc[lane * CLD + 4 * c4]);
    a2[2 * c4] = cdv_float2{q[0], q[1]};
    a2[2 * c4 + 1] = cdv_float2{q[2], q[3]};
  }
  float* rowp = A + (size_t)(rhs ? npad : CNB * min(rb, nb - 1) + lane) * lda + c0;
  float x[CNB];
#pragma unroll
  for (int c4 = 0; c4 < CNB / 4; c4++) {
    const cdv_float4 q = *reinterpret_cast<const cdv_float4*>(&Pr[(rhs ? 0 : lane) * CLD + 4 * c4]);
    x[4 * c4] = q[0]; x[4 * c4 + 1] = q[1]; x[4 * c4 + 2] = q[2]; x[4 * c4 + 3] = q[3];
  }
  bool bad = false;
  float Lk;
  {
    const float piv = readlane_f(a2[0][0], 0);
    bad = !(piv > 0.f);
    Lk = a2[0][0] * __builtin_amdgcn_rsqf(piv);
    a2[0][0] = Lk;
    colb[lane] = Lk;
  }
  cdv_float2 bcur[CNB / 2], bnxt[CNB / 2];
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
      const float an = fmaf(-Lk, readlane_f(Lk, k + 1), a2[(k + 1) >> 1][(k + 1) & 1]);
      const float piv = readlane_f(an, k + 1);
      bad = bad || !(piv > 0.f);                          // wave-uniform
      Ln = an * __builtin_amdgcn_rsqf(piv);
      a2[(k + 1) >> 1][(k + 1) & 1] = Ln;
      colb[lane] = Ln;
#pragma unroll
      for (int c4 = (k + 2) / 4; c4 < CNB / 4; c4++) {
        const cdv_float4 v = *reinterpret_cast<const cdv_float4*>(&colb[4 * c4]);
        bnxt[2 * c4] = cdv_float2{v[0], v[1]};
        bnxt[2 * c4 + 1] = cdv_float2{v[2], v[3]};
      }
    }
    if (((k + 2) & 1) && k + 2 < CNB)
      a2[(k + 2) >> 1][1] = fmaf(-Lk, bcur[(k + 2) >> 1][1], a2[(k + 2) >> 1][1]);
    const cdv_float2 nLk = {-Lk, -Lk};
#pragma unroll
    for (int pp = (k + 3) >> 1; pp < CNB / 2; pp++) a2[pp] = __builtin_elementwise_fma(nLk, bcur[pp], a2[pp]);
    Lk = Ln;
#pragma unroll
    for (int pp = (k + 2) >> 1; pp < CNB / 2; pp++) bcur[pp] = bnxt[pp];
  }
#pragma unroll
  for (int c4 = 0; c4 < CNB / 4; c4++) {
    cdv_float4 q;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int c = 4 * c4 + j;
      q[j] = (c <= lane) ? a2[c >> 1][c & 1] : 0.f;
    }
    *reinterpret_cast<cdv_float4*>(&Ls[lane * CLD + 4 * c4]) = q;
    if (diag) {
      float* dst = A + (size_t)(c0 + lane) * lda + c0 + 4 * c4;
#pragma unroll
      for (int j = 0; j < 4; j++)
        if (4 * c4 + j <= lane) dst[j] = q[j];
    }
  }
  if (diag) {
    if (lane == 0 && bad && info[BI_CHOL] == 0) ba_flag(info, BI_CHOL, kb + 1);
    return;
  }
  wave_lds_sync();
  if (rhs && lane > 0) return;                    // the right-hand side is one row
#pragma unroll
  for (int c = 0; c < CNB; c++) {
    float sacc = x[c];
#pragma unroll
    for (int j4 = 0; j4 < (c + 3) / 4; j4++) {
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

// L^T x = z as ONE launch of one-wave workgroups, one per 64-column block (round 5; until then 256-column workgroups with four
// barriers and an LDS exchange per step: 12.8k cycles a step, of which 5.0k went into ISSUING 64 four-byte row loads per
// thread -- a wave cannot have more than 64 vector loads outstanding -- in front of the block solve instead of under it).
// The wave of block b keeps its 64 entries of z (lane = column) and the column `lane` of its own diagonal block L_bb, scaled by
// 1 / L[lane][lane], in registers for the whole sweep.  Step kb > b: it picks up x_kb -- published by block kb's wave as {launch
// token, value} granules, written through; the poll is the load -- and folds it into its z: the 64 x 64 tile L[kb rows][my
// columns] travels as sixteen 16-byte loads per lane (lane group g = rows 16 g .., lane q = columns 4 q ..), requested one step
// ahead; the four row groups' sums meet through lane shuffles, in a fixed order.  Step kb == b: the 64-step chain (one
// v_readlane + one FMA per unknown on the scaled columns), publish, done.  No barrier, no LDS hand-off between waves; a wave
// only ever waits for blocks to its RIGHT, whose waves wait for nobody to their left, so the launch cannot lock up; the polls
// are bounded all the same: a lost hand-off raises the hand-off word, dX is then incomplete and the retract launch that
// follows applies NOTHING of it (it only re-zeroes the accumulators): the update is all-or-nothing.
__global__ __launch_bounds__(64) void ba_big_backsolve_kernel(float* __restrict__ A, int npad, int n,
                                                              float* __restrict__ dXg, uint64_t* __restrict__ xg, int token,
                                                              const int32_t* __restrict__ gmeta, float* __restrict__ dbg,
                                                              int32_t* __restrict__ info, int test) {
  if (gmeta[GM_ERROR] || info[1]) return;
  if (info[BI_HANDOFF]) return;   // the factorisation in front gave up on a hand-off: nothing of this iteration is applied
  __shared__ __attribute__((aligned(16))) float xs[CNB];
  const int lane = threadIdx.x;
  const size_t lda = (size_t)npad;
  const int nb = npad / CNB;
  const int me = (int)blockIdx.x, c0m = CNB * me;     // my block and its first column
  CDV_IF_STAMPS(const int sbase = 2000 + 64 * me;)
  CDV_STAMP(ba, sbase + 63, 0);
  CDV_STAMP_RT(ba, sbase + 63, 5);
  float z = A[(size_t)npad * lda + c0m + lane];
  // my diagonal block: column `lane`, L[c0m + r][c0m + lane] (zero above the diagonal), 64 coalesced row loads -- in flight while
  // the blocks to my right are folded
  float col[CNB];
#pragma unroll
  for (int r = 0; r < CNB; r++) col[r] = A[(size_t)(c0m + r) * lda + c0m + lane];
  const int g = lane >> 4, q = lane & 15;
  const auto tile_rows = [&](int kb, cdv_float4 (&dst)[16]) {
#pragma unroll
    for (int u = 0; u < 16; u++)
      dst[u] = *reinterpret_cast<const cdv_float4*>(A + (size_t)(CNB * kb + 16 * g + u) * lda + c0m + 4 * q);
  };
  cdv_float4 cur[16], nxt[16];
  if (me < nb - 1) tile_rows(nb - 1, cur);
  for (int kb = nb - 1; kb > me; kb--) {
    CDV_IF_STAMPS(const int sslot = sbase + kb;)
    CDV_STAMP(ba, sslot, 0);
    if (kb - 1 > me) tile_rows(kb - 1, nxt);          // requested now, consumed a step later
    float xv = 0.f;
    bool ok = false;
    for (int spins = 0; spins < (test == HO_TEST_STALL_BEFORE ? (1 << 10) : (1 << 20)); spins++) {
      if (!ok) {
        const uint64_t gr = __hip_atomic_load(&xg[CNB * kb + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((uint32_t)(gr >> 32) == (uint32_t)token) { xv = __int_as_float((int)(uint32_t)gr); ok = true; }
      }
      if (__all(ok)) break;
      __builtin_amdgcn_s_sleep(1);
    }
    CDV_STAMP(ba, sslot, 1);
    CDV_STAMP_RT(ba, sslot, 6);
    if (!__all(ok)) {
      if (lane == 0) ba_flag(info, BI_HANDOFF, 1);
      return;
    }
    xs[lane] = xv;
    wave_lds_sync();
    // z[c] -= sum_r L[64 kb + r][c] x_r: my 16 rows of my four columns ...
    cdv_float4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u4 = 0; u4 < 4; u4++) {
      const cdv_float4 x4 = *reinterpret_cast<const cdv_float4*>(&xs[16 * g + 4 * u4]);
      s0 += cur[4 * u4] * x4[0];
      s1 += cur[4 * u4 + 1] * x4[1];
      s0 += cur[4 * u4 + 2] * x4[2];
      s1 += cur[4 * u4 + 3] * x4[3];
    }
    cdv_float4 sm = s0 + s1;
    // ... the four row groups together (lanes 16 apart), then column c's sum from lane c >> 2, element c & 3
    float pick = 0.f;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      float v = sm[e];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      const float w = __shfl(v, lane >> 2);
      pick = ((lane & 3) == e) ? w : pick;
    }
    z -= pick;
    wave_lds_sync();                                  // xs is rewritten by the next step
    CDV_STAMP(ba, sslot, 2);
#pragma unroll
    for (int u = 0; u < 16; u++) cur[u] = nxt[u];
  }
  // my own block: x_r = z_r / L[r][r] once every x_j, j > r, is folded in.  On columns scaled by the lane's own 1 / L[lane][lane]:
  // zs = (z - folded part) / L[lane][lane] is what lane r hands out as x_r, so a step of the chain is one v_readlane and one
  // FMA (as in the window solver, ba_win.hip solve_wave); x_r lands in lane r with a v_writelane, off the chain
  CDV_STAMP(ba, sbase + me, 0);
  float dg = 0.f;
#pragma unroll
  for (int r = 0; r < CNB; r++) dg = (lane == r) ? col[r] : dg;   // L[lane][lane]
  const float inv = 1.0f / dg;
#pragma unroll
  for (int r = 0; r < CNB; r++) col[r] = (r >= lane) ? col[r] * inv : 0.f;
  float zs = z * inv, x = 0.f;
#pragma unroll
  for (int r = CNB - 1; r >= 0; r--) {
    const float xr = readlane_f(zs, r);
    asm("v_writelane_b32 %0, %1, %2" : "+v"(x) : "s"(xr), "n"(r));
    zs = fmaf(-col[r], xr, zs);
  }
  if (!(test == HO_TEST_STALL_BEFORE && me == nb - 2))   // fault injection (tests, mode 1 only): the second block's solution never leaves its owner
    __hip_atomic_store(&xg[c0m + lane], ((uint64_t)(uint32_t)token << 32) | (uint64_t)(uint32_t)__float_as_int(x),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (c0m + lane < n) {
    dXg[c0m + lane] = x;
    if (dbg) dbg[(size_t)n * n + n + c0m + lane] = x;
  }
  CDV_STAMP(ba, sbase + me, 1);
  CDV_STAMP_RT(ba, sbase + me, 5);
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
  // a lost hand-off in the back-substitution launch (final by now: that launch is over): dX is incomplete, so NOTHING of this
  // iteration is applied -- poses and depths stay as they are, only the accumulators are re-zeroed for the next call
  const bool apply = info[BI_HANDOFF] == 0;
  const int U = gmeta[GM_U];
  const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
  // global-BA path: a workgroup is one chunk of 64 patches; its panel mask says which 8-pose panels of E are non-zero
  // (the rest of the column is zero and stays zero: not read, not rewritten); the mask is consumed here
  uint32_t pmask[BIG_MW];
#pragma unroll
  for (int wd = 0; wd < BIG_MW; wd++) pmask[wd] = 0xffffffffu;
  if (cmask) {
#pragma unroll
    for (int wd = 0; wd < BIG_MW; wd++) pmask[wd] = ((int)blockIdx.x < n_chunks) ? cmask[(size_t)blockIdx.x * BIG_MW + wd] : 0u;
    __syncthreads();
    if (threadIdx.x < BIG_MW && (int)blockIdx.x < n_chunks) cmask[(size_t)blockIdx.x * BIG_MW + threadIdx.x] = 0u;
  }
  // pose_retr_kernel (ba_cuda.cu:178-206): T <- Exp(dX_i) T, one lane per free pose, in wave 0 of the last workgroups
  // (the first ones carry the longest E-column sweeps)
  const int gid_rev = (int)(gridDim.x * 64) - 1 - (int)(blockIdx.x * 64 + lane);
  if (pose_retr && apply && g == 0 && gid_rev < N) {
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
      uint32_t word = pmask[0];
#pragma unroll
      for (int wd = 1; wd < BIG_MW; wd++) word = ((b >> 8) == wd) ? pmask[wd] : word;
      if (!((word >> ((b >> 3) & 31)) & 1u)) continue;
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
  if (!apply) return;
  float* pk = patches + kx[r] * 3 * PP + 2 * PP;
  float d = pk[0];                 // patch_retr_kernel reads pixel [0][0]   ba_cuda.cu:218
  d = d + dz;
  d = (d > 20.f) ? 1.0f : d;
  d = fmaxf(d, 1e-4f);
  store_depth(pk, PP, d);
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
    g_ws_ppf.erase(ws);
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
  if (E_max < 1) E_max = 1;
  if (U_max < 1) U_max = 1;
  if (N_max < 1) N_max = 1;
  if (N_max > BA_NBIG) N_max = BA_NBIG;
  return ba_layout(U_max, N_max, E_max).total;
}

static int ba_forward_impl(float* poses, float* patches, const float* intrinsics, const float* target,
                           const float* weight, const float* lmbda, const int64_t* ii, const int64_t* jj,
                           const int64_t* kk, int64_t E, int P, int t0, int t1, int iterations,
                           const void* graph_ws, void* ba_ws, size_t ba_ws_bytes, int64_t U_max, float* dbg,
                           void* stream, const int32_t* dyn);

extern "C" int cdv_ba_forward(float* poses, float* patches, const float* intrinsics, const float* target,
                              const float* weight, const float* lmbda, const int64_t* ii, const int64_t* jj,
                              const int64_t* kk, int64_t E, int P, int t0, int t1, int iterations,
                              const void* graph_ws, void* ba_ws, size_t ba_ws_bytes, int64_t U_max, float* dbg,
                              void* stream) {
  return ba_forward_impl(poses, patches, intrinsics, target, weight, lmbda, ii, jj, kk, E, P, t0, t1, iterations, graph_ws, ba_ws,
                         ba_ws_bytes, U_max, dbg, stream, nullptr);
}

// cdv_ba_forward with the window on the device: the free poses are [dyn[CDV_DYN_T0], + dyn[CDV_DYN_NFREE]) with
// dyn[CDV_DYN_NFREE] <= N_max <= 32 (the optimisation window of a frame stream, slam.py:512-513: known to the device only
// when the keyframe decision stays there; N_max picks the path -- <= 10 the window kernels, <= 32 ba_mid.hip's -- and the
// kernels then work on whatever the block says, e.g. the 7 free poses of a stream's first update inside launches laid out
// for 22); E_bound sizes the workspace, the index in graph_ws must be a patch table.
extern "C" int cdv_ba_forward_dyn(float* poses, float* patches, const float* intrinsics, const float* target, const float* weight,
                                  const float* lmbda, const int64_t* ii, const int64_t* jj, const int64_t* kk, int64_t E_bound,
                                  int P, int N_max, const int32_t* dyn, int iterations, const void* graph_ws, void* ba_ws,
                                  size_t ba_ws_bytes, int64_t U_max, void* stream) {
  CDV_REQUIRE(dyn != nullptr, CDV_ERR_ARG, "cdv_ba_forward_dyn: NULL dynamic block");
  CDV_REQUIRE(N_max >= 1 && N_max <= MID_N, CDV_ERR_UNSUPPORTED, "cdv_ba_forward_dyn: 1 <= N_max <= 32 free poses");
  CDV_REQUIRE(cdv_graph_is_table(graph_ws), CDV_ERR_UNSUPPORTED, "cdv_ba_forward_dyn: graph_ws must hold a patch table");
  return ba_forward_impl(poses, patches, intrinsics, target, weight, lmbda, ii, jj, kk, E_bound, P, 0, N_max, iterations, graph_ws,
                         ba_ws, ba_ws_bytes, U_max, nullptr, stream, dyn);
}

static int ba_forward_impl(float* poses, float* patches, const float* intrinsics, const float* target,
                           const float* weight, const float* lmbda, const int64_t* ii, const int64_t* jj,
                           const int64_t* kk, int64_t E, int P, int t0, int t1, int iterations,
                           const void* graph_ws, void* ba_ws, size_t ba_ws_bytes, int64_t U_max, float* dbg,
                           void* stream, const int32_t* dyn) {
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
  const BaLayout L = ba_layout(U_max, N > 0 ? N : 1, E);
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
  int ppf_hint = 0;
  {
    std::lock_guard<std::mutex> lk(g_ws_mutex);
    WsState& st = g_ws_state[ba_ws];
    fresh = !(st.valid && st.U_max == U_max && st.N == N && st.bytes == ba_ws_bytes);
    const int32_t tok0 = (fresh || st.token > 0x7ffffff0 - 4 * (iterations > 0 ? iterations : 1)) ? 0 : st.token;
    token_base = tok0;
    st = WsState{true, U_max, N, ba_ws_bytes, tok0 + (iterations > 0 ? iterations : 1)};
    auto it = g_ws_counters.find(ba_ws);
    if (it != g_ws_counters.end()) counters = it->second;
    auto ip = g_ws_ppf.find(ba_ws);
    if (ip != g_ws_ppf.end()) ppf_hint = ip->second;
  }
  // the slab paths: two launches per iteration, no float atomics (ba_win.hip up to 10 free poses, ba_mid.hip up to 32)
  const bool window = N >= 1 && N <= MID_N;
  const bool table = cdv_graph_is_table(graph_ws);
  CDV_REQUIRE(!table || window, CDV_ERR_UNSUPPORTED,
              "cdv_ba_forward: graph_ws holds a patch table (cdv_graph_build_table), which serves 1 .. 32 free poses; build "
              "the ranked index (cdv_graph_build_edges) for the global bundle adjustment");
  if (fresh) {   // (zeroing KERNELS, not hipMemsetAsync: a first call made under stream capture leaves kernel nodes only)
    const auto zero = [&](void* p, size_t bytes) {   // every area starts 256-byte aligned (ba_layout) and is a multiple of 4 bytes
      const size_t n4 = bytes / 4;
      if (n4 == 0) return;
      const int grid = (int)(cdv_div_up((int64_t)n4, 1024) < 4096 ? cdv_div_up((int64_t)n4, 1024) : 4096);
      hipLaunchKernelGGL(ba_zero_kernel, dim3(grid), dim3(256), 0, s, (uint32_t*)p, (int64_t)n4);
    };
    if (!window) zero(b + L.sy, L.zero_bytes);   // the window path keeps no accumulators
    zero(info, sizeof(int32_t) * 16 + sizeof(uint64_t) * MID_GRAN);
    if (window) zero(b + L.hand, sizeof(int32_t) * HAND_WORDS);   // token 0, no flag set
    if (big) zero(b + L.xgran, sizeof(uint64_t) * (size_t)L.npad);   // no granule carries a token
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
    wa.test = g_handoff_test.load();
    wa.dyn = dyn;
    // wide chunks cut per frame (ba_mid.hip): only when a frame's patches really are ppf consecutive table slots
    // ... and only when the per-frame cut does not need more slabs than the workspace holds (one per 16 rows of U_max): a
    // frame of fewer than 16 patches would be a workgroup -- and a slab -- of its own (ppf 4, 8, 12: tab_cap / ppf > n_ck)
    wa.ppf = 0;
    if (table && N > WIN_N && ppf_hint >= 4 && ppf_hint % 4 == 0 && wa.tab_cap % ppf_hint == 0 &&
        (int64_t)(wa.tab_cap / ppf_hint) * cdv_ba_mid_wide_per_frame(ppf_hint) <= L.n_ck)
      wa.ppf = ppf_hint;
    wa.dbg = nullptr;
    wa.token = token_base + 1;
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
  const int rb = cdv_div_up(L.U_max > N ? L.U_max : N, 64);
  const int npad = (int)L.npad, nbk = npad / CNB;
  static const bool block_steps = []() { const char* e = getenv("CDV_BA_BLOCK_STEPS"); return e && e[0] == '1'; }();
  if (big) {
    // the frame-pair index of this call's edges (both iterations use it): keys, an ordinary index build over them, the
    // (a, b) -> pair table
    int64_t* pkeys = (int64_t*)(b + L.pkeys);
    void* pws = b + L.pgraph;
    if (fresh) {
      const int rc0 = cdv_graph_workspace_init(pws, L.pgraph_bytes, L.E_max, L.pair_range, stream);
      if (rc0 != CDV_OK) return rc0;
    }
    cdv_graph_no_corr_order(pws);   // (idempotent; the flag goes when the workspace is forgotten)
    int32_t* ptab = (int32_t*)(b + L.ptab);
    hipLaunchKernelGGL(ba_pair_key_kernel, dim3(cdv_div_up(E, 256) < 2048 ? (int)cdv_div_up(E, 256) : 2048), dim3(256), 0, s, ii, jj,
                       (int32_t)E, t0, N, pkeys, ptab, (int64_t)L.pair_range);
    const int rc1 = cdv_graph_build_edges(ii, jj, pkeys, E, pws, L.pgraph_bytes, L.E_max, L.pair_range, nullptr, nullptr, stream);
    if (rc1 != CDV_OK) return rc1;
    const GraphView pv = graph_view(pws, graph_layout(L.E_max, L.pair_range));
    hipLaunchKernelGGL(ba_pair_table_kernel, dim3(256), dim3(256), 0, s, pv.meta, pv.kx, ptab);
  }
  for (int itr = 0; itr < iterations; itr++) {
    float* d = (dbg && itr == 0) ? dbg : nullptr;
    const PatchArgs pa{poses, patches, intrinsics, target, weight, ii, P, t0, N, gv.meta, gv.prec, gv.koff_u, gv.kx, Cg, ug, Edg,
                       (int)L.U_stride, (int)L.U_max, info, cmask, counters, itr == 0 ? 1 : 0};
    hipLaunchKernelGGL(ba_patch_kernel, dim3(n_chunks), dim3(64), 0, s, pa);
    if (big) {
      const GraphView pv = graph_view(b + L.pgraph, graph_layout(L.E_max, L.pair_range));
      const PairArgs qa{poses, patches, intrinsics, target, weight, ii, kk, P, t0, N, gv.meta, pv.meta, pv.prec, pv.koff_u, pv.kx,
                        sy, (float*)(b + L.pdiag), (int32_t)L.pair_cap, info};
      const int pgrid = (int)(L.pair_cap < 16384 ? L.pair_cap : 16384);
      hipLaunchKernelGGL(ba_pair_kernel, dim3(pgrid), dim3(64), 0, s, qa);
      hipLaunchKernelGGL(ba_diag_kernel, dim3(N), dim3(64), 0, s, gv.meta, pv.meta, (const int32_t*)(b + L.ptab),
                         (const float*)(b + L.pdiag), N, sy, info);
      const int npan = cdv_div_up(N, BIG_PP);
      hipLaunchKernelGGL(ba_schur_kernel, dim3(npan * (npan + 1) / 2), dim3(256), 0, s, lmbda, N, gv.meta, sy, Cg, ug, Edg,
                         (int)L.U_stride, cmask, n_chunks, info);
      hipLaunchKernelGGL(ba_big_fold_kernel, dim3(1024), dim3(256), 0, s, sy, (int)L.sy_stride, n6i, npad, Abig, gv.meta,
                         d, info, (uint64_t*)(b + L.xgran), (int32_t*)(b + L.fctl), fac_ctl_words(nbk));
      if (!block_steps) {
        // the factorisation as one launch of block work items (ba_factor.hip)
        const int rcf = cdv_ba_big_factor(Abig, npad, (int32_t*)(b + L.fctl), (float*)(b + L.ltg), gv.meta, info, g_handoff_test.load(), s);
        if (rcf != CDV_OK) return rcf;
      } else {
        // (rounds 1-3, kept for comparison: CDV_BA_BLOCK_STEPS=1) one launch per block column.  Block step 0: the panel alone;
        // block step kb >= 1: the panel together with what step kb - 1 owes the matrix
        hipLaunchKernelGGL(ba_big_panel_kernel, dim3(nbk + 1), dim3(64), 0, s, Abig, npad, 0, gv.meta, info);
        for (int kb = 1; kb < nbk; kb++) {
          const int T = nbk - kb - 1;
          hipLaunchKernelGGL(ba_big_step_kernel, dim3(nbk - kb + 1 + T * (T + 1) / 2 + T), dim3(STEP_T), 0, s, Abig, npad, kb, gv.meta,
                             info);
        }
      }
      hipLaunchKernelGGL(ba_big_backsolve_kernel, dim3(npad / CNB), dim3(64), 0, s, Abig, npad, n6i, dXg,
                         (uint64_t*)(b + L.xgran), token_base + 1 + itr, gv.meta, d, info, g_handoff_test.load());
    } else {
      // only N = 0 gets here (no free pose: depths alone are refined): q = 1 / (C + lambda)
      hipLaunchKernelGGL(ba_q_kernel, dim3(cdv_div_up(L.U_max, 256) < 1024 ? (int)cdv_div_up(L.U_max, 256) : 1024), dim3(256), 0, s,
                         lmbda, gv.meta, Cg, qg, info);
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

// Fault injection for the in-launch hand-offs (tests only; process-wide, read by the next cdv_ba_forward calls):
//   0 off; 1 the solver of the N <= 32 paths stalls BEFORE its commit (the retract workgroups abandon: nothing is applied) --
//   on the global path the back substitution withholds one block's solution (its readers time out: nothing is applied);
//   2 the solver stalls AFTER its commit (the retract workgroups lose their patience, learn that the solution is coming and
//   wait on: the update is applied as usual; the global path has no such in-launch commit and runs undisturbed);  3 (global path) a diagonal block of the factorisation launch never raises its flag
//   (everybody who needs it times out, the launch drains, nothing is applied).  The waits are shortened so that a test takes
//   milliseconds.
extern "C" int cdv_ba_test_handoff(int mode) {
  CDV_REQUIRE(mode >= 0 && mode <= 3, CDV_ERR_ARG, "cdv_ba_test_handoff: mode 0 .. 3");
  g_handoff_test.store(mode);
  return CDV_OK;
}

// PPF of cuda_ba.forward (fastba/ba.cpp:31-45 passes the patches per frame; the reference's kernels only use it in the
// block-sparse E of eff_impl): a hint that lets the 10 < N <= 32 path cut its workgroups' patch ranges per frame when the
// patch table's capacity is a multiple of it.  0 forgets the hint.  Results do not depend on it beyond summation order.
extern "C" int cdv_ba_set_patches_per_frame(void* ba_ws, int patches_per_frame) {
  CDV_REQUIRE(ba_ws != nullptr && patches_per_frame >= 0, CDV_ERR_ARG, "cdv_ba_set_patches_per_frame: arguments");
  std::lock_guard<std::mutex> lk(g_ws_mutex);
  if (patches_per_frame > 0) g_ws_ppf[ba_ws] = patches_per_frame;
  else g_ws_ppf.erase(ba_ws);
  return CDV_OK;
}

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
