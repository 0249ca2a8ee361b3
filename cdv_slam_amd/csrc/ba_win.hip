// ba_win.hip -- fastba over the optimisation window (1 <= N <= 10 free poses: every steady-state update of the default
// configuration), gfx950.  Replaces one Gauss-Newton iteration of cuda_ba.forward (cdvslam/fastba/ba_cuda.cu:462-611)
// by TWO launches, with no float atomic anywhere: every sum has one owner and a fixed order, so two runs give the same
// bits (the reference's ~16 M contended atomicAdds per iteration, ba_cuda.cu:350-402, are what this does not imitate).
//
//   1. ba_chunk_kernel   workgroup = chunk of 16 unique patches x ALL their edges (patch CSR records of the graph index:
//                        one 16-byte load gives edge id, source frame, target frame).  lane = (patch, one of four
//                        target slots), four waves, so one or two rounds cover a patch's ~25 edges.
//                          - the chunk owns its patches' E columns, C and u completely: built in LDS / registers, no
//                            accumulation across workgroups; q = 1 / (C + lambda) on the spot;
//                          - B and v: per wave the 13 x 13 Gram matrices of [Ji | Jj | r] per frame pair as f32 MFMA
//                            tiles (K = the pair's residual rows), added into the wave's PRIVATE packed-triangular
//                            copy of [S | y] in LDS;
//                          - the chunk's Schur products [E; u] diag(q) [E; u]^T (K = 16 patches) as MFMA tiles; the tile
//                            owner adds the four wave copies in fixed order, subtracts and stores ONE partial system
//                            per chunk (a "slab": packed lower triangle of S + y, 7.6 KB) with plain stores;
//                          - E columns, q, u of the chunk go to HBM for the retraction (plain stores, complete values:
//                            nothing has to be re-zeroed afterwards).
//   2. ba_finish_kernel  workgroup 0 = the solver wave; workgroups 1.. first reduce the slabs (each thread a 16-byte
//                        column over every G-th slab, butterfly over the G partials: fixed order), hand the reduced
//                        system to the solver (write-through stores, drained, one arrival count per workgroup), then
//                        preload their patches' E columns while the solver runs, wait for dX (tagged 8-byte granules:
//                        the poll is the load), and retract depths and poses (ba_cuda.cu:178-229 semantics).
//                        Solver: lane r holds row r of [S ; y^T] in registers (the right-hand side as row 60: forward
//                        substitution for free); column k is broadcast through LDS (one ds_write, then 16-byte
//                        broadcast reads) one column AHEAD of the rank-1 updates that consume it, so the LDS round
//                        trip and the rsqrt chain of column k + 1 hide under the packed FMAs of column k.
#include "cdv_ba.h"
#include "cdv_se3.h"

using namespace cdv;

namespace {

constexpr int CK = WIN_CK;
constexpr int CKW = 4;                     // waves per chunk workgroup
constexpr int SN = WIN_SN;
constexpr int TRI = WIN_TRI;
constexpr int SLAB = WIN_SLAB;
constexpr int XLD = 17;                    // floats per residual row in the Gram staging buffer (16 + 1 pad)
constexpr int XW = 128 * XLD + 64;         // per wave: [128][XLD] rows + 64 per-edge pair keys
constexpr int EDL = CK + 1;                // row stride of the chunk's [E; u] block in LDS
constexpr int LDS_CHUNK_FLOATS = CKW * XW + CKW * SLAB + 64 * EDL + CKW * 8 * CK + 2 * CK;

__device__ __forceinline__ int tri_index(int R, int Cc) { return ((R * (R + 1)) >> 1) + Cc; }

// one entry (row, col) of the 13 x 13 Gram matrix G = sum_k w_k X[k] X[k]^T, X[k] = [Ji | Jj | r], of frame pair
// (ci, cj) (free-pose indices or -1) into the wave's packed copy: B[ii] += w Ji Ji^T, B[jj] += w Jj Jj^T,
// B[ij] -= w Ji Jj^T, v[i] -= w r Ji, v[j] += w r Jj (ba_cuda.cu:364-377,393-398 semantics), lower triangle only.
// quad < 0: every entry; otherwise only the entries of Gram quadrant `quad` (row >= 6, col in [6, 12)).
__device__ __forceinline__ void tri_emit(float val, int row, int col, int ci, int cj, float* __restrict__ Sw, int quad) {
  if (row >= 12 || col >= 13) return;                 // row 12 duplicates column 12; (12, 12) = sum w r^2
  const bool ri = row < 6;
  const bool isv = col == 12;
  const bool cib = col < 6;
  if (quad >= 0 && quad != ((ri ? 0 : 2) + ((isv || cib) ? 0 : 1))) return;
  const int rb = ri ? ci : cj;
  if (rb < 0) return;
  const int R = 6 * rb + (ri ? row : row - 6);
  if (isv) {
    Sw[TRI + R] += ri ? -val : val;
    return;
  }
  const int cb = cib ? ci : cj;
  if (cb < 0) return;
  const int Cc = 6 * cb + (cib ? col : col - 6);
  if (Cc > R) return;                                  // the mirrored Gram entry lands in the lower triangle
  Sw[tri_index(R, Cc)] += (ri != cib) ? -val : val;
}

struct EdgeRec {
  int e, ix, jx;
};
struct EdgeIn {
  float pi[7], pj[7], tx, ty, wx, wy;
};

__device__ __forceinline__ EdgeRec load_rec(const BaWinArgs& A, int pos, bool has_ii) {
  const int4 r = *reinterpret_cast<const int4*>(A.prec + 4 * (size_t)pos);
  EdgeRec o;
  o.e = r.x;
  o.ix = has_ii ? r.y : (int)A.ii[r.x];
  o.jx = r.z;
  return o;
}

__device__ __forceinline__ EdgeIn load_in(const BaWinArgs& A, const EdgeRec& x) {
  EdgeIn o;
  const float* __restrict__ poses = A.poses;
#pragma unroll
  for (int a = 0; a < 7; a++) { o.pi[a] = poses[7 * (int64_t)x.ix + a]; o.pj[a] = poses[7 * (int64_t)x.jx + a]; }
  const float2 t = *reinterpret_cast<const float2*>(A.target + 2 * (int64_t)x.e);
  const float2 w = *reinterpret_cast<const float2*>(A.weight + 2 * (int64_t)x.e);
  o.tx = t.x; o.ty = t.y; o.wx = w.x; o.wy = w.y;
  return o;
}

__global__ __launch_bounds__(64 * CKW) void ba_chunk_kernel(BaWinArgs A) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Xall = smem;                         // [CKW][XW]   Gram staging rows + pair keys, per wave
  float* Sc = Xall + CKW * XW;                // [CKW][SLAB] per-wave packed copies of [S | y] (B and v parts)
  float* Ed = Sc + CKW * SLAB;                // [64][EDL]   rows 0..59 E, row 60 u, rows 61..63 zero
  float* part = Ed + 64 * EDL;                // [CKW][8][CK] per-wave partial sums: 6 rows of E_i, C, u
  float* qs = part + CKW * 8 * CK;            // [CK]
  int* ixp = reinterpret_cast<int*>(qs + CK); // [CK] free-pose index of the patch's source frame (-1: fixed / none)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int32_t* __restrict__ gmeta = A.gmeta;
  const int gerr = gmeta[GM_ERROR];
  const int U = gmeta[GM_U];
  const bool has_ii = gmeta[GM_HAS_II] != 0;
  if (blockIdx.x == 0) {
    if (tid == 0) {
      ba_begin_status(A.info, A.counters, A.first, gerr, U > A.U_max);
      *A.arrive = 0;                              // hand-off words of the finish launch that follows
    }
    if (tid < 64) A.granX[tid] = 0ull;
  }
  if (gerr || U > A.U_max) return;
  const int N = A.N, t0 = A.t0, P = A.P;
  const int n6 = 6 * N;
  const int PP = P * P;
  const int centre = (P > 1) ? (P + 1) : 0;
  const float fx = A.intr[0], fy = A.intr[1], cx = A.intr[2], cy = A.intr[3];   // row 0 only (ba_cuda.cu:253-259)
  const float lm = A.lmbda[0];
  const int p = lane & 15, sub = lane >> 4;
  const int c16 = lane & 15, g4 = lane >> 4;
  float* X = Xall + wave * XW;
  int* keys = reinterpret_cast<int*>(X + 128 * XLD);
  float* Sw = Sc + wave * SLAB;
  const int n_chunks = (U + CK - 1) / CK;

  for (int chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
    const int r0 = chunk * CK;
    const int r = r0 + p;
    const bool live = r < U;
    // ---- level 1: this lane's patch ----
    const int plo = live ? A.koff_u[r] : 0;
    const int deg = live ? A.koff_u[r + 1] - plo : 0;
    const int64_t kxr = live ? A.kx[r] : 0;
    // zero the workgroup's accumulators (the previous chunk of a grid-stride loop is done with them: barrier below)
    {
      const cdv_float4 z4 = {0.f, 0.f, 0.f, 0.f};
      cdv_float4* s4 = reinterpret_cast<cdv_float4*>(Sc);
      for (int i = tid; i < CKW * SLAB / 4; i += 64 * CKW) s4[i] = z4;
      for (int i = tid; i < 64 * EDL; i += 64 * CKW) Ed[i] = 0.f;
    }
    int maxdeg = deg;
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) maxdeg = max(maxdeg, __shfl_xor(maxdeg, o));   // over the 16 patches (all sub rows alike)
    maxdeg = __builtin_amdgcn_readfirstlane(maxdeg);
    // ---- level 2: patch centre, and the source frame every edge of this patch shares (ii = ix[kk], slam.py:331-337)
    float px = 0.f, py = 0.f, pd = 0.f;
    int ix_patch = -1;
    if (deg > 0) {
      const float* pk = A.patches + kxr * 3 * PP;
      px = pk[centre]; py = pk[PP + centre]; pd = pk[2 * PP + centre];
      ix_patch = load_rec(A, plo, has_ii).ix;
    }
    const int a0 = ix_patch - t0;
    const int ixf_patch = (deg > 0 && a0 >= 0 && a0 < N) ? a0 : -1;
    __syncthreads();   // accumulators are zero

    float Cacc = 0.f, uacc = 0.f;
    float eiacc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int step = 4 * CKW;
    int tb = 4 * wave;
    EdgeRec rec = load_rec(A, (live && tb + sub < deg) ? plo + tb + sub : 0, has_ii);
    EdgeIn in = load_in(A, rec);
    for (; tb < maxdeg; tb += step) {
      const bool active = live && (tb + sub) < deg;
      const bool more = tb + step < maxdeg;      // wave-uniform
      const EdgeRec cur = rec;
      if (more) rec = load_rec(A, (live && tb + step + sub < deg) ? plo + tb + step + sub : 0, has_ii);
      EdgeFactor J;
      fastba_factor(in.pi, in.pj, px, py, pd, in.tx, in.ty, in.wx, in.wy, fx, fy, cx, cy, J);
      if (more) in = load_in(A, rec);            // the next round's inputs travel under this round's Gram
      int ixf = -1, jxf = -1;
      if (active) {
        const int a = cur.ix - t0, b = cur.jx - t0;
        ixf = (a >= 0 && a < N) ? a : -1;
        jxf = (b >= 0 && b < N) ? b : -1;
        // E, C, u of this lane's patch (ba_cuda.cu:380-390,401-402 semantics)
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
          if (cur.ix == ix_patch) {
#pragma unroll
            for (int c = 0; c < 6; c++) eiacc[c] += ei[c];      // the patch's own frame: summed in registers, fixed order
          } else {
            // an edge list that gives one patch two source frames (never built by slam.py): still summed, through LDS
#pragma unroll
            for (int c = 0; c < 6; c++) atomicAdd(&Ed[(6 * ixf + c) * EDL + p], ei[c]);
          }
        }
        if (jxf >= 0) {
          // (patch, target frame) is unique per edge in a patch graph, so each address receives ONE add onto zero
          // (exact, order-free); duplicate edges would add in arrival order
#pragma unroll
          for (int c = 0; c < 6; c++) atomicAdd(&Ed[(6 * jxf + c) * EDL + p], ej[c]);
        }
      }
      // ---- B and v of this wave's frame pairs: Gram matrices on the matrix cores ----
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
      unsigned long long todo = __ballot(active && key != 0);
      if (todo) {
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
        while (todo) {
          const int leader = __ffsll((long long)todo) - 1;
          const int kcur = __shfl(key, leader);
          const int ci = __shfl(ixf, leader), cj = __shfl(jxf, leader);
          cdv_float4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
          const unsigned long long match = __ballot(active && key == kcur);
#pragma unroll
          for (int st = 0; st < 32; st += 2) {
            // k-steps st, st + 1 hold the rows of edges (lanes) 2 st .. 2 st + 3: skipped when none is of this pair
            if (((match >> (2 * st)) & 15ull) == 0) continue;
            const float w0 = (xk[st] == kcur) ? xw[st] : 0.f;
            const float w1 = (xk[st + 1] == kcur) ? xw[st + 1] : 0.f;
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[st], w0 * xa[st], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[st + 1], w1 * xa[st + 1], acc1, 0, 0, 0);
          }
          // D layout: col = lane & 15, row = 4 * (lane >> 4) + reg.  One owner lane per destination inside a pass;
          // a self pair (i == j) folds its four Gram quadrants onto one block: four sequential sub-steps
          if (ci == cj && ci >= 0) {
#pragma unroll
            for (int quad = 0; quad < 4; quad++) {
#pragma unroll
              for (int q = 0; q < 4; q++) tri_emit(acc0[q] + acc1[q], 4 * g4 + q, c16, ci, cj, Sw, quad);
              wave_lds_sync();
            }
          } else {
#pragma unroll
            for (int q = 0; q < 4; q++) tri_emit(acc0[q] + acc1[q], 4 * g4 + q, c16, ci, cj, Sw, -1);
          }
          wave_lds_sync();   // the next pass of this wave may touch the same entries
          todo &= ~match;
        }
      }
      wave_lds_sync();  // the next round overwrites X
    }
    // ---- the wave's partial E_i, C, u: over its four target slots in fixed order, then published ----
    {
      float v[8] = {eiacc[0], eiacc[1], eiacc[2], eiacc[3], eiacc[4], eiacc[5], Cacc, uacc};
#pragma unroll
      for (int c = 0; c < 8; c++) {
        v[c] += __shfl_xor(v[c], 16);
        v[c] += __shfl_xor(v[c], 32);
        if (sub == 0) part[(wave * 8 + c) * CK + p] = v[c];
      }
      if (wave == 0 && sub == 0) ixp[p] = ixf_patch;
    }
    __syncthreads();
    if (tid < 6 * CK) {          // E_i rows of every patch: the four wave partials in fixed order, onto the E_j entries
      const int c = tid / CK, pp = tid - c * CK;
      const float tot = (part[(0 * 8 + c) * CK + pp] + part[(1 * 8 + c) * CK + pp]) +
                        (part[(2 * 8 + c) * CK + pp] + part[(3 * 8 + c) * CK + pp]);
      const int ib = ixp[pp];
      if (ib >= 0) Ed[(6 * ib + c) * EDL + pp] += tot;
    } else if (tid < 7 * CK) {   // C, u, q of every patch
      const int pp = tid - 6 * CK;
      const float Ct = (part[(0 * 8 + 6) * CK + pp] + part[(1 * 8 + 6) * CK + pp]) +
                       (part[(2 * 8 + 6) * CK + pp] + part[(3 * 8 + 6) * CK + pp]);
      const float ut = (part[(0 * 8 + 7) * CK + pp] + part[(1 * 8 + 7) * CK + pp]) +
                       (part[(2 * 8 + 7) * CK + pp] + part[(3 * 8 + 7) * CK + pp]);
      const int rr = r0 + pp;
      const float q = (rr < U) ? 1.0f / (Ct + lm) : 0.f;      // Q = 1 / (C + lambda)   (ba_cuda.cu:548 semantics)
      qs[pp] = q;
      Ed[SN * EDL + pp] = (rr < U) ? ut : 0.f;
      A.qg[rr] = q;
      A.ug[rr] = (rr < U) ? ut : 0.f;
      if (A.dbg) {
        float* dbgp = A.dbg + (size_t)n6 * n6 + 2 * n6;
        dbgp[A.U_stride + rr] = Ct;
        dbgp[2 * (size_t)A.U_stride + rr] = ut;
      }
    }
    __syncthreads();
    // ---- the chunk's E columns for the retraction (complete values, plain stores) ----
    for (int i = tid; i < n6 * CK; i += 64 * CKW) {
      const int row = i / CK, pp = i - row * CK;
      const float v = Ed[row * EDL + pp];
      A.Edg[(size_t)row * A.U_stride + r0 + pp] = v;
      if (A.dbg) A.dbg[(size_t)n6 * n6 + 2 * n6 + 3 * (size_t)A.U_stride + (size_t)row * A.U_stride + r0 + pp] = v;
    }
    // ---- Schur products of the chunk, [E; u] diag(q) [E; u]^T on the matrix cores (K = 16 patches), and the chunk's
    // partial system: slab = (wave copies of B, v in fixed order) - products; each entry has exactly one owner ----
    float* slab = A.slabs + (size_t)chunk * SLAB;
    if (tid < SLAB - (TRI + SN)) slab[TRI + SN + tid] = 0.f;
    for (int pidx = wave; pidx < 10; pidx += CKW) {
      int ti = 0, acc_rows = 0;   // lower-triangular tile pair (ti >= tj) of the 4 x 4 tiles covering rows 0..63
      while (acc_rows + ti + 1 <= pidx) { acc_rows += ti + 1; ti++; }
      const int tj = pidx - acc_rows;
      const float* pa = Ed + (16 * ti + c16) * EDL;
      const float* pb = Ed + (16 * tj + c16) * EDL;
      cdv_float4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int st = 0; st < 4; st++) {
        const int k = 4 * st + g4;
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[k], qs[k] * pb[k], acc, 0, 0, 0);
      }
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int R = 16 * ti + 4 * g4 + q, Cc = 16 * tj + c16;
        if (R > SN || Cc >= SN || (R < SN && Cc > R)) continue;   // row 60 = y; column 60 only duplicates it
        const int idx = (R < SN) ? tri_index(R, Cc) : TRI + Cc;
        const float bsum = (Sc[idx] + Sc[SLAB + idx]) + (Sc[2 * SLAB + idx] + Sc[3 * SLAB + idx]);
        slab[idx] = bsum - acc[q];
      }
    }
    __syncthreads();   // a grid-stride successor chunk re-zeroes the accumulators
  }
}

// ---------------------------------------------------------------------------------------------------------
// finish: reduce -> solve -> retract
// ---------------------------------------------------------------------------------------------------------

typedef float cdv_float2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float ld_agent(const float* p) {   // global_load_dword sc1: past this CU's L1
  return __int_as_float((int)__hip_atomic_load(reinterpret_cast<const uint32_t*>(p), __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT));
}

// The 60 x 60 system in the registers of ONE wave.  Lane r holds row r of [S ; y^T] (lane 60 = the right-hand side).
__device__ __forceinline__ void solve_wave(const BaWinArgs& A, int RW) {
  __shared__ __attribute__((aligned(16))) float colb[64];          // the column being broadcast
  __shared__ __attribute__((aligned(16))) float Lt[(SN + 1) * 68]; // L for the back substitution (row stride 68)
  const int lane = threadIdx.x;
  const int n = 6 * A.N;
  // ---- wait for the reduce workgroups (bounded: a lost hand-off must not hang the device) ----
  bool ok = false;
  for (int spins = 0; spins < (1 << 20); spins++) {
    const int v = __hip_atomic_load(A.arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (v >= RW) { ok = true; break; }
    __builtin_amdgcn_s_sleep(1);
  }
  if (!ok) {
    if (lane == 0) ba_flag(A.info, BI_HANDOFF, 1);
    return;   // the retract workgroups time out on the dX granules and leave the state untouched
  }
  // ---- my row, write-through loads.  The packed rows follow each other in memory, so a row is read at full length:
  // its tail (the head of the next rows) sits where the upper triangle would be, which lane r computes on but nobody
  // ever reads (a pivot is lane k's own a[k][k], a broadcast value lane c's a[c][k], c > k) ----
  const int row = min(lane, SN);
  const float* rp = A.ared + ((row < SN) ? tri_index(row, 0) : TRI);
  cdv_float2 a2[SN / 2];     // the row as 30 float2 registers: rank-1 updates run two columns per v_pk_fma_f32
#pragma unroll
  for (int c = 0; c < SN; c++) {
    float v = ld_agent(rp + c);
    if (c == row) v += 1e-4f * v + 1.0f;             // S += I (1e-4 S + 1.0)  (ba_cuda.cu:589 semantics); rows >= 6 N: identity
    a2[c >> 1][c & 1] = v;
  }
  if (A.dbg && lane <= SN) {                          // damped S (both triangles) and y of iteration 0
#pragma unroll
    for (int c = 0; c < SN; c++) {
      const float v = a2[c >> 1][c & 1];
      if (lane < n && c <= lane) { A.dbg[(size_t)lane * n + c] = v; A.dbg[(size_t)c * n + lane] = v; }
      if (lane == SN && c < n) A.dbg[(size_t)n * n + c] = v;
    }
  }
  // ---- right-looking Cholesky, column k broadcast through LDS one column ahead of its rank-1 update ----
  int badk = 0;
  float Lk;
  {
    const float piv = readlane_f(a2[0][0], 0);
    if (!(piv > 0.f)) badk = 1;
    Lk = a2[0][0] * __builtin_amdgcn_rsqf(piv);
    a2[0][0] = Lk;
    colb[lane] = Lk;
  }
  cdv_float2 bcur[SN / 2], bnxt[SN / 2];   // column k / column k + 1 of L, the same in every lane (pairs of columns)
#pragma unroll
  for (int c4 = 0; c4 < SN / 4; c4++) {
    const cdv_float4 v = *reinterpret_cast<const cdv_float4*>(&colb[4 * c4]);
    bcur[2 * c4] = cdv_float2{v[0], v[1]};
    bcur[2 * c4 + 1] = cdv_float2{v[2], v[3]};
  }
#pragma unroll
  for (int k = 0; k < SN; k++) {
    float Ln = 0.f;
    if (k + 1 < SN) {
      // column k + 1 first: its one update from column k, pivot, scale, broadcast request
      float an = fmaf(-Lk, bcur[(k + 1) >> 1][(k + 1) & 1], a2[(k + 1) >> 1][(k + 1) & 1]);
      const float piv = readlane_f(an, k + 1);
      if (!(piv > 0.f) && badk == 0) badk = (k + 1) / 6 + 1;        // wave-uniform
      Ln = an * __builtin_amdgcn_rsqf(piv);
      a2[(k + 1) >> 1][(k + 1) & 1] = Ln;
      colb[lane] = Ln;      // in-order LDS: the reads of column k were issued before this write
#pragma unroll
      for (int c4 = (k + 2) / 4; c4 < SN / 4; c4++) {
        const cdv_float4 v = *reinterpret_cast<const cdv_float4*>(&colb[4 * c4]);
        bnxt[2 * c4] = cdv_float2{v[0], v[1]};
        bnxt[2 * c4 + 1] = cdv_float2{v[2], v[3]};
      }
    }
    // the rest of column k's rank-1 update (columns k + 2 ..) runs while column k + 1 travels through LDS
    if (((k + 2) & 1) && k + 2 < SN)
      a2[(k + 2) >> 1][1] = fmaf(-Lk, bcur[(k + 2) >> 1][1], a2[(k + 2) >> 1][1]);
    const cdv_float2 nLk = {-Lk, -Lk};
#pragma unroll
    for (int pp = (k + 3) >> 1; pp < SN / 2; pp++) a2[pp] = __builtin_elementwise_fma(nLk, bcur[pp], a2[pp]);
    Lk = Ln;
#pragma unroll
    for (int pp = (k + 2) >> 1; pp < SN / 2; pp++) bcur[pp] = bnxt[pp];
  }
  float a[SN];
#pragma unroll
  for (int c = 0; c < SN; c++) a[c] = a2[c >> 1][c & 1];
  // ---- L back to LDS, then lane k picks up COLUMN k: col[r] = L[r][k].  Entries above the diagonal (r < k) are
  // whatever the row held there: lane k folds them into its z only AFTER x_k has been taken from it ----
  wave_lds_sync();
  if (lane <= SN) {
#pragma unroll
    for (int c4 = 0; c4 < SN / 4; c4++)
      *reinterpret_cast<cdv_float4*>(&Lt[lane * 68 + 4 * c4]) =
          cdv_float4{a[4 * c4], a[4 * c4 + 1], a[4 * c4 + 2], a[4 * c4 + 3]};
  }
  wave_lds_sync();
  const int kc = min(lane, SN - 1);
  float col[SN];
#pragma unroll
  for (int r = 0; r < SN; r++) col[r] = Lt[r * 68 + kc];
  float z = Lt[SN * 68 + kc];                       // z = L^-1 y
  const float invd = 1.0f / Lt[kc * 68 + kc];
  // back substitution L^T x = z: x_r = z_r / L[r][r] once every x_j, j > r, has been folded into z
  float x = 0.f;
#pragma unroll
  for (int r = SN - 1; r >= 0; r--) {
    const float xr = readlane_f(z * invd, r);
    x = (lane == r) ? xr : x;
    z = fmaf(-col[r], xr, z);
  }
  if (lane < n) {
    // the data IS the flag: one 8-byte {tag = 1, value} granule per unknown, written through; the retract workgroups
    // poll the tags of the granules they read (CDNA programming guide, Guideline 16, recipe R2)
    __hip_atomic_store(&A.granX[lane], (1ull << 32) | (uint64_t)(uint32_t)__float_as_int(x), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
    A.dXg[lane] = x;
    if (A.dbg) A.dbg[(size_t)n * n + n + lane] = x;
  }
  if (lane == 0 && badk) ba_flag(A.info, BI_CHOL, badk);
}

__global__ __launch_bounds__(256) void ba_finish_kernel(BaWinArgs A) {
  const int32_t* __restrict__ gmeta = A.gmeta;
  const int U = gmeta[GM_U];
  if (gmeta[GM_ERROR] || U > A.U_max) return;
  const int RW = (int)gridDim.x - 1;
  const int tid = threadIdx.x;
  if (blockIdx.x == 0) {
    if (tid < 64) solve_wave(A, RW);
    return;
  }
  const int b = (int)blockIdx.x - 1;
  // ---- 1. reduce the chunk slabs: thread = (16-byte column, one of G interleaved slab subsets).  G depends on the
  //         number of slabs only and the butterfly over the subsets is fixed, so the sum is the same whatever the
  //         launch geometry (U_max, number of workgroups): reproducible bits ----
  {
    const int nsl = (U + CK - 1) / CK;
    const int G = nsl <= 256 ? 4 : (nsl <= 1024 ? 8 : 16);
    const int cols_per_wg = 256 / G;
    const int g = tid % G;
    for (int col = b * cols_per_wg + tid / G; col - tid / G < SLAB / 4; col += RW * cols_per_wg) {   // workgroup-uniform trip count
      cdv_float4 acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
      const bool mine = col < SLAB / 4;
      if (mine) {
        const cdv_float4* src = reinterpret_cast<const cdv_float4*>(A.slabs) + col;
        int sidx = g;
        for (; sidx + 3 * G < nsl; sidx += 4 * G) {   // four independent loads in flight per trip
#pragma unroll
          for (int u = 0; u < 4; u++) acc[u] += src[(size_t)(sidx + u * G) * (SLAB / 4)];
        }
        for (int u = 0; sidx < nsl; sidx += G, u++) acc[u] += src[(size_t)sidx * (SLAB / 4)];
      }
      cdv_float4 tot = (acc[0] + acc[1]) + (acc[2] + acc[3]);
      for (int o = 1; o < G; o <<= 1) {
#pragma unroll
        for (int j = 0; j < 4; j++) tot[j] += __shfl_xor(tot[j], o);
      }
      if (mine && g == 0) {
        // written through (8-byte agent-scope stores): the solver reads them past its L1
        uint64_t* dst = reinterpret_cast<uint64_t*>(A.ared + 4 * col);
        __hip_atomic_store(dst, ((uint64_t)(uint32_t)__float_as_int(tot[1]) << 32) | (uint32_t)__float_as_int(tot[0]),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(dst + 1, ((uint64_t)(uint32_t)__float_as_int(tot[3]) << 32) | (uint32_t)__float_as_int(tot[2]),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains before the workgroup counts itself in
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(A.arrive, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // ---- 2. before dX exists: everything of this thread's patch that does not depend on it ----
  __shared__ float sdx[64];
  const int P = A.P, PP = P * P, N = A.N;
  int r = b * 256 + tid;
  bool livep = r < U;
  float ev[SN];
#pragma unroll
  for (int i = 0; i < SN; i++) ev[i] = (livep && i < 6 * N) ? A.Edg[(size_t)i * A.U_stride + r] : 0.f;
  float uv = 0.f, qv = 0.f, d0 = 0.f;
  float* pk = nullptr;
  if (livep) {
    uv = A.ug[r]; qv = A.qg[r];
    pk = A.patches + A.kx[r] * 3 * PP + 2 * PP;
    d0 = pk[0];                      // the depth is read from pixel [0][0]   (ba_cuda.cu:218 semantics)
  }
  // ---- 3. wait for the solver (bounded) ----
  __shared__ int s_ok;
  if (tid < 64) {   // wave 0: lane t polls the granule of unknown t until its tag shows up; the poll is the load of dX
    float xv = 0.f;
    bool ok = tid >= 6 * N;
    for (int spins = 0; spins < (1 << 21); spins++) {
      if (!ok) {
        const uint64_t g = __hip_atomic_load(&A.granX[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((uint32_t)(g >> 32) == 1u) { xv = __int_as_float((int)(uint32_t)g); ok = true; }
      }
      if (__all(ok)) break;
      __builtin_amdgcn_s_sleep(2);
    }
    const bool all_ok = __all(ok);
    if (!all_ok && tid == 0) ba_flag(A.info, BI_HANDOFF, 1);
    if (tid == 0) s_ok = all_ok ? 1 : 0;
    sdx[tid] = xv;
  }
  __syncthreads();
  if (!s_ok) return;   // no update without a solution: poses and depths stay as they were
  // ---- 4. pose retraction T <- Exp(dX_i) T: the last workgroup's first N lanes ----
  if ((int)blockIdx.x == (int)gridDim.x - 1 && tid < N) {
    float* p = A.poses + 7 * (size_t)(A.t0 + tid);
    float pose[7], xi[6];
#pragma unroll
    for (int c = 0; c < 7; c++) pose[c] = p[c];
#pragma unroll
    for (int c = 0; c < 6; c++) xi[c] = sdx[6 * tid + c];
    se3_retract_raw(xi, pose);
#pragma unroll
    for (int c = 0; c < 7; c++) p[c] = pose[c];
  }
  // ---- 5. dZ = Q (u - E^T dX), inverse-depth update (ba_cuda.cu:592,209-229 semantics) ----
  for (;;) {
    if (livep) {
      float sacc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int bb = 0; bb < SN / 6; bb++)
#pragma unroll
        for (int c = 0; c < 6; c++) sacc[c] += ev[6 * bb + c] * sdx[6 * bb + c];
      const float dz = qv * (uv - (((sacc[0] + sacc[1]) + (sacc[2] + sacc[3])) + (sacc[4] + sacc[5])));
      if (A.dbg) A.dbg[(size_t)36 * N * N + 12 * N + r] = dz;
      float d = d0 + dz;
      d = (d > 20.f) ? 1.0f : d;
      d = fmaxf(d, 1e-4f);
      for (int a = 0; a < PP; a++) pk[a] = d;
    }
    // more patches than one pass of the retract workgroups covers: the next block of 256 (loaded now, dX is known)
    r += RW * 256;
    if (r - tid >= U) break;          // workgroup-uniform
    livep = r < U;
#pragma unroll
    for (int i = 0; i < SN; i++) ev[i] = (livep && i < 6 * N) ? A.Edg[(size_t)i * A.U_stride + r] : 0.f;
    if (livep) {
      uv = A.ug[r]; qv = A.qg[r];
      pk = A.patches + A.kx[r] * 3 * PP + 2 * PP;
      d0 = pk[0];
    }
  }
}

}  // namespace

int cdv::cdv_ba_window_iteration(const BaWinArgs& a, hipStream_t s) {
  static hipError_t attr_err = [] {
    return hipFuncSetAttribute((const void*)ba_chunk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)(sizeof(float) * LDS_CHUNK_FLOATS));
  }();
  CDV_HIP_CHECK(attr_err);
  const int n_ck = a.n_ck_cap < WIN_MAX_GRID ? a.n_ck_cap : WIN_MAX_GRID;
  hipLaunchKernelGGL(ba_chunk_kernel, dim3(n_ck), dim3(64 * CKW), sizeof(float) * LDS_CHUNK_FLOATS, s, a);
  // reduce / retract workgroups: one per 256 patches of capacity, at least 8 (the reduce wants the parallelism), at
  // most WIN_MAX_RW (they all poll the solver)
  int RW = cdv_div_up(a.U_max, 256);
  RW = RW < 8 ? 8 : (RW > WIN_MAX_RW ? WIN_MAX_RW : RW);
  hipLaunchKernelGGL(ba_finish_kernel, dim3(1 + RW), dim3(256), 0, s, a);
  CDV_LAUNCH_CHECK();
  return CDV_OK;
}
