// ba_mid.hip -- fastba for 10 < N <= 32 free poses (a wider OPTIMIZATION_WINDOW, the start of a global optimisation),
// gfx950.  The design of ba_win.hip -- one Gauss-Newton iteration of cuda_ba.forward (cdvslam/fastba/ba_cuda.cu:462-611)
// with one owner and a fixed order for every sum, no float atomic in HBM -- for systems whose triangle no longer fits a
// wave-private LDS copy (6N = 192: 75 KB).  Three launches per iteration:
//
//   1. ba_mid_chunk_kernel  workgroup = 16 unique patches x all their edges, as in ba_win.hip.  What a chunk adds to B
//                           and v is sparse: the block diagonal (B_jj, v_j of every target frame), and the block rows
//                           of its patches' SOURCE frames (B_ii, v_i, B_ij) -- one frame, two when the chunk straddles
//                           a frame boundary.  Each wave keeps that footprint privately ([2][27] + [N][27] + [2][N][36]
//                           floats, 8.9 KB at N = 22); the copies are summed in a fixed order and the source-frame parts
//                           folded onto the diagonal blocks.  The chunk's Schur complement E Q E^T (MFMA, K = 16 patches)
//                           goes tile by tile into a packed triangle staged in the LDS the copies no longer need, the
//                           folded footprint is scattered onto it (one addend per entry), and the slab (packed lower
//                           triangle of the 6N x 6N system + y) leaves as 16-byte stores: every word written once.
//                           Source frames are found from a 32-bit mask of the free frames the chunk's edges start from;
//                           a chunk with more than two (fewer than 8 patches per frame) takes one more pass per further
//                           pair and adds its B part onto its own slab.
//   2. ba_mid_reduce_kernel the slabs summed in a fixed order by the whole chip (9.6 MB at N = 22, U = 4,312), written as
//                           the damped dense system [6N + 1][LD].
//   3. ba_mid_finish_kernel workgroup 0 = the solver (8 waves), the others preload their patches' E columns, wait for dX
//                           (tagged granules) and retract.  Solver: [S ; y^T] dense in LDS, right-looking Cholesky over
//                           panels of 24 columns factored inside single waves (one matrix row per lane, pivots and
//                           L[m][k] by v_readlane), trailing update as 16 x 16 tiles on the matrix cores (K = 24), back
//                           substitution inside one wave.
#include <stdlib.h>

#include "cdv_ba_pairs.h"

using namespace cdv;

CDV_STAMP_TU(bam)

namespace {

constexpr int CK = WIN_CK;
// A workgroup takes SC consecutive chunks of 16 patches (a "wide chunk"): what it does AFTER the edges -- summing the wave
// copies of the footprint, folding, the Schur tiles, the scatter and the slab copy-out, half of its time at N = 22 -- is
// the same work for 16 patches as for 32, and the stress configuration's 294 chunks needed a second round of workgroups
// on the 256 CUs (one 8-wave workgroup per CU: registers): 147 wide chunks are one round, and half the slabs.
constexpr int SC = 2;
constexpr int CKS = SC * CK;               // patches per workgroup
constexpr int EDL = CKS + 1;               // row stride of the wide chunk's [E; u] block in LDS
constexpr int FT = 512;                    // threads of a finish workgroup

// ---- footprint of a chunk in B and v, per wave --------------------------------------------------------------------
//   [0, 54)                   F_ii[s]   s = 0, 1: 21 (B_ii lower triangle) + 6 (v_i) of source slot s
//   [54, 54 + 27 N)           F_jj[j]   21 + 6 of target frame j
//   [54 + 27 N, 54 + 99 N)    F_ij[s][j] 36 (B_ij, rows of the source frame, columns of the target frame)
__host__ __device__ inline int fp_floats(int N) { return (54 + 99 * N + 3) / 4 * 4; }
__host__ __device__ inline int ed_rows(int N) { return (6 * N + 1 + 15) / 16 * 16; }
// stride between the waves' footprint copies: the copies of waves 1 .. W - 1 also stage the chunk's slab, so with few waves
// the stride grows beyond the footprint itself
__host__ __device__ inline int fp_stride(int N, int waves) {
  const int slab = ((6 * N * (6 * N + 1)) / 2 + 6 * N + 7) / 8 * 8;
  const int need = ((slab + waves - 2) / (waves - 1) + 3) / 4 * 4;
  return fp_floats(N) > need ? fp_floats(N) : need;
}
inline size_t chunk_lds_bytes(int N, int waves) {
  return sizeof(float) * ((size_t)waves * fp_stride(N, waves) + (size_t)ed_rows(N) * EDL + (size_t)SC * waves * 8 * CK + 2 * CKS + 8);
}

// where value `code` of a frame pair goes inside a footprint: base + ms * (source slot) + mj * (target frame); need: bit 0
// = the source frame must be free, bit 1 = the target frame must be free
struct EmitAddr {
  int base, ms, mj, need;
};
__device__ __forceinline__ EmitAddr emit_addr(int code, int N) {
  const int kind = code >> 6, a = (code >> 3) & 7, b = code & 7;
  const int t = ((a * (a + 1)) >> 1) + b;
  EmitAddr e = {0, 0, 0, 4};   // need 4: padding, never emitted
  if (kind == 0) e = EmitAddr{t, 27, 0, 1};
  else if (kind == 1) e = EmitAddr{21 + a, 27, 0, 1};
  else if (kind == 2) e = EmitAddr{54 + t, 0, 27, 2};
  else if (kind == 3) e = EmitAddr{54 + 21 + a, 0, 27, 2};
  else if (kind == 4) e = EmitAddr{54 + 27 * N + 6 * a + b, 36 * N, 36, 3};
  return e;
}

// k-th set bit of m (k < popcount(m)), else -1
__device__ __forceinline__ int nth_bit(uint32_t m, int k) {
  for (int i = 0; i < k; i++) m &= m - 1;
  return m ? __ffs((int)m) - 1 : -1;
}

#ifdef CDV_MID_WPE
#define CDV_MID_OCC __attribute__((amdgpu_waves_per_eu(CDV_MID_WPE, CDV_MID_WPE)))
#else
#define CDV_MID_OCC
#endif
// Rows of wide chunk wc: plainly [CKS wc, + CKS); with A.ppf > 0 (a patch table whose capacity is a multiple of the patches per
// frame: a frame's patches then sit in ppf consecutive slots) the wide chunks are cut PER FRAME -- pieces of CKS rows, the
// last one shorter -- so that no workgroup holds patches of two source frames: a row of 16 lanes whose patches belong to two
// frames takes two passes of the pair products instead of one, and the workgroup that had such rows was the launch's long
// pole (stress configuration, 196 patches per frame: one wide chunk in six).
__host__ __device__ inline int wide_per_frame(int ppf) { return (ppf + CKS - 1) / CKS; }
__device__ __forceinline__ void wide_rows(const BaWinArgs& A, int wc, int& r0w, int& cnt) {
  if (A.ppf > 0) {
    const int per = wide_per_frame(A.ppf);
    const int b = wc / per, piece = wc - b * per;
    r0w = b * A.ppf + CKS * piece;
    cnt = min(CKS, A.ppf - CKS * piece);
  } else {
    r0w = wc * CKS;
    cnt = CKS;
  }
}
__device__ __forceinline__ int wide_count(const BaWinArgs& A, int U) {
  if (A.ppf > 0) return (U / A.ppf) * wide_per_frame(A.ppf);
  return ((U + CK - 1) / CK + SC - 1) / SC;
}

template <bool HAS_II, int MKW, bool TABLE>
__global__ __launch_bounds__(64 * MKW) CDV_MID_OCC void ba_mid_chunk_kernel(BaWinArgs A_in) {
  const BaWinArgs A = with_dyn(A_in);   // (a frame stream's window lives on the device: every layout below follows ITS N)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int N = A.N, t0 = A.t0, P = A.P;
  const int n6 = 6 * N;
  const int FP = fp_stride(N, MKW), ER = ed_rows(N);
  const int TRI_N = (n6 * (n6 + 1)) >> 1;
  const int slabf = (TRI_N + n6 + 7) / 8 * 8;
  float* Fw = smem;                            // [MKW][FP] per-wave footprints
  float* Ed = Fw + MKW * FP;                   // [ER][EDL]  rows 0..6N-1 E, row 6N u, the rest zero
  float* part = Ed + ER * EDL;                 // [SC][MKW][8][CK] per-wave partial sums: 6 rows of E_i, C, u
  float* qs = part + SC * MKW * 8 * CK;        // [CKS]
  int* ixp = reinterpret_cast<int*>(qs + CKS); // [CKS] free-pose index of the patch's source frame (-1: fixed / none)
  uint32_t* smask = reinterpret_cast<uint32_t*>(ixp + CKS);   // [2] free source frames of the chunk's edges (ping-pong)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: loops over a wave's share are uniform
  const int32_t* __restrict__ gmeta = A.gmeta;
  const PatchSpan sp = patch_span<TABLE>(A);
  const int gerr = graph_error_of(gmeta, TABLE);
  const int U = sp.U;
  if (blockIdx.x == 0) {
    if (tid == 0) {
      ba_begin_status(A.info, A.counters, A.first, gerr, U > A.U_max);
      *A.arrive = 0;                              // hand-off words of the finish launch that follows
      A.arrive[HO_VERDICT] = HO_UNDECIDED;
    }
    for (int i = tid; i < MID_GRAN; i += 64 * MKW) A.granX[i] = 0ull;
  }
  if (gerr || U > A.U_max) return;
  const int PP = P * P;
  const int centre = (P > 1) ? (P + 1) : 0;
  const int p = lane & 15, sub = lane >> 4;
  // (dealing the slots of a round to the waves cyclically -- slot = wave + MKW * sub -- so that the slots that carry work
  // spread over all waves was measured: 12.8 against 12.4 us; a pass of the pair products costs the same for one active
  // row as for four, so concentrating the active rows in few waves is the cheaper arrangement)
  const int so = sub;
  const int c16 = lane & 15, g4 = lane >> 4;
  float* Fme = Fw + wave * FP;
  // which of a frame pair's 90 sums this lane owns after the reductions: value 16 g + brev4(c16) of group g
  EmitAddr ea[6];
  {
    const int br = ((c16 & 1) << 3) | ((c16 & 2) << 1) | ((c16 & 4) >> 1) | ((c16 & 8) >> 3);
#pragma unroll
    for (int g = 0; g < 6; g++) ea[g] = emit_addr(pair_code_rt(g, br), N);
  }
  const int n_wide = wide_count(A, U);
  if (tid < 2) smask[tid] = 0u;
  __syncthreads();

  CDV_IF_STAMPS(const int sslot = (int)blockIdx.x * MKW + wave;)
  CDV_STAMP(bam, sslot, 0);
  CDV_STAMP_RT(bam, sslot, 14);
  int par = 0;   // which mask word this pass fills
  for (int wc = blockIdx.x; wc < n_wide; wc += gridDim.x) {
    int pass = 0, npass = 1;
    do {
      int r0w, cntw;                             // first patch row of the wide chunk, its rows
      wide_rows(A, wc, r0w, cntw);
      const int step = 4 * MKW;
      const float fx = A.intr[0], fy = A.intr[1], cx = A.intr[2], cy = A.intr[3];
      const float lm = A.lmbda[0];
      const EdgeRec safe = {A.prec[0], A.prec[1], A.prec[2]};   // stands in for slots that do not exist
      // zero the workgroup's accumulators (the previous pass is done with them: barrier at its end)
      {
        const cdv_float4 z4 = {0.f, 0.f, 0.f, 0.f};
        cdv_float4* s4 = reinterpret_cast<cdv_float4*>(Fw);
        for (int i = tid; i < MKW * FP / 4; i += 64 * MKW) s4[i] = z4;
        for (int i = tid; i < ER * EDL; i += 64 * MKW) Ed[i] = 0.f;
        if (tid == 0) smask[par ^ 1] = 0u;      // the word of the NEXT pass
      }
      // ---- the free source frames of the wide chunk's edges: every edge's (of all its SC chunks), so that an edge list
      // which gives one patch two source frames (never built by slam.py) is still summed where it belongs.  Records only:
      // this pre-pass requests the first-round record of every sub-chunk at once ----
      {
        uint32_t mbits = 0u;
#pragma unroll
        for (int sc = 0; sc < SC; sc++) {
          const int r = r0w + sc * CK + p;
          const bool live = r < U && sc * CK + p < cntw;
          const bool use_ell = TABLE || (r >> 4) < A.ell_chunks;
          const int rs = live ? r : 0;
          const int tb = 4 * wave;
          const int4* cell = use_ell ? reinterpret_cast<const int4*>(A.pell) : reinterpret_cast<const int4*>(A.prec);
          int4 raw = cell[(use_ell && tb + so < ELL_SLOTS) ? cell_index(rs, tb + so) : 0];
          const PatchRow row = patch_row<TABLE>(A, rs);
          const int plo = live ? row.plo : 0, deg = live ? row.deg : 0;
          if (!use_ell || tb + so >= ELL_SLOTS) raw = reinterpret_cast<const int4*>(A.prec)[(tb + so < deg) ? plo + tb + so : 0];
          for (int t = tb + so; t < deg; t += step) {   // the first trip uses the record in hand
            const int4 rk = (t == tb + so) ? raw
                                           : ((use_ell && t < ELL_SLOTS) ? cell[cell_index(rs, t)] : reinterpret_cast<const int4*>(A.prec)[plo + t]);
            const int a = (HAS_II ? rk.y : (int)A.ii[rk.x]) - t0;
            if (a >= 0 && a < N) mbits |= 1u << a;
          }
        }
        if (mbits) atomicOr(&smask[par], mbits);
      }
      CDV_STAMP(bam, sslot, 13);
      lds_barrier();   // accumulators are zero, the mask is complete (LDS only: loads stay in flight)
      const uint32_t mask = smask[par];
      if (pass == 0) npass = max(1, (__popc(mask) + 1) >> 1);
      const int isrc0 = nth_bit(mask, 2 * pass), isrc1 = nth_bit(mask, 2 * pass + 1);   // free-pose indices or -1
      CDV_STAMP(bam, sslot, 1);

#pragma unroll 1
      for (int sc = 0; sc < SC; sc++) {
      const int r0 = r0w + sc * CK;
      const int r = r0 + p;
      const bool live = r < U && sc * CK + p < cntw;
      // ---- level 1 (see ba_win.hip): records of this lane's first-round slot and of the patch's first edge from the
      // chunk-slot copy, the patch's CSR offsets and id; everything unconditional on clamped indices
      int tb = 4 * wave;
      const bool use_ell = TABLE || (r0 >> 4) < A.ell_chunks;
      const int rs = live ? r : 0;
      const int4* cell = use_ell ? reinterpret_cast<const int4*>(A.pell) : reinterpret_cast<const int4*>(A.prec);
      int4 raw = cell[(use_ell && tb + so < ELL_SLOTS) ? cell_index(rs, tb + so) : 0];
      int4 raw0 = cell[use_ell ? cell_index(rs, 0) : 0];
      const PatchRow row = patch_row<TABLE>(A, rs);
      const int plo = live ? row.plo : 0;
      const int deg = live ? row.deg : 0;
      const int64_t kxr = live ? row.id : 0;
      if (!use_ell || tb + so >= ELL_SLOTS) {   // beyond the chunk-slot copy: the CSR records, one round trip later
        const int4* csr = reinterpret_cast<const int4*>(A.prec);
        raw = csr[(tb + so < deg) ? plo + tb + so : 0];
        if (!use_ell) raw0 = csr[plo];
      }
      EdgeRec rec = settle_rec<HAS_II>(A, raw, tb + so < deg, safe);
      const EdgeRec rec0 = settle_rec<HAS_II>(A, raw0, deg > 0, safe);
      // ---- level 2: the patch centre, the first round's poses, target, weight
      const float* pk = A.patches + kxr * 3 * PP;
      const float px = pk[centre], py = pk[PP + centre], pd = pk[2 * PP + centre];
      const int ix_patch = deg > 0 ? rec0.ix : -1;
      EdgeIn in = load_in(A, rec);
      int maxdeg = deg;
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) maxdeg = max(maxdeg, __shfl_xor(maxdeg, o));
      maxdeg = __builtin_amdgcn_readfirstlane(maxdeg);
      const int a0 = ix_patch - t0;
      const int ixf_patch = (deg > 0 && a0 >= 0 && a0 < N) ? a0 : -1;
      const int pc = sc * CK + p;                // this lane's patch column of the wide chunk

      float Cacc = 0.f, uacc = 0.f;
      float eiacc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      for (; tb < maxdeg; tb += step) {
        const bool active = (tb + so) < deg;
        const bool more = tb + step < maxdeg;      // wave-uniform
        const EdgeRec cur = rec;
        int4 raw_nxt = {0, 0, 0, 0};
        if (more) {   // slots below ELL_SLOTS from the chunk-slot rows (a table keeps only longer lists in its CSR too)
          const int sn = tb + step + so;
          const bool from_cell = use_ell && sn < ELL_SLOTS;
          const int4* src = from_cell ? cell : reinterpret_cast<const int4*>(A.prec);
          raw_nxt = src[from_cell ? cell_index(rs, sn) : ((sn < deg) ? plo + sn : 0)];
        }
        EdgeFactor J;
        fastba_factor(in.pi, in.pj, px, py, pd, in.tx, in.ty, in.wx, in.wy, fx, fy, cx, cy, J);
        if (more) {
          rec = settle_rec<HAS_II>(A, raw_nxt, tb + step + so < deg, safe);
          in = load_in(A, rec);
        }
        int ixf = -1, jxf = -1;
        if (active) {
          const int a = cur.ix - t0, b = cur.jx - t0;
          ixf = (a >= 0 && a < N) ? a : -1;
          jxf = (b >= 0 && b < N) ? b : -1;
        }
        if (active && pass == 0) {
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
#pragma unroll
              for (int c = 0; c < 6; c++) atomicAdd(&Ed[(6 * ixf + c) * EDL + pc], ei[c]);
            }
          }
          if (jxf >= 0) {
            // (patch, target frame) is unique per edge in a patch graph: ONE add onto zero per address (exact, order-free)
#pragma unroll
            for (int c = 0; c < 6; c++) atomicAdd(&Ed[(6 * jxf + c) * EDL + pc], ej[c]);
          }
        }
        // ---- B and v: per DPP row (16 patches of one target slot) the pairs (i, j) present, one leader pass each; an
        // edge takes part in the pass of its source frame's slot pair (fixed source frames: pass 0)
        const int ord = ixf >= 0 ? __popc(mask & ((1u << ixf) - 1u)) : 0;
        const bool mine_pass = active && (ord >> 1) == pass;
        const int key = (ixf + 1) * (N + 1) + (jxf + 1);
        unsigned long long todo = __ballot(mine_pass && key != 0);
        while (todo) {
          const unsigned rowbits = (unsigned)(todo >> (16 * sub)) & 0xffffu;
          const bool row_on = rowbits != 0;
          const int leader = 16 * sub + (row_on ? __ffs((int)rowbits) - 1 : 0);
          const int kcur = __shfl(key, leader);
          const int ci = __shfl(ixf, leader), cj = __shfl(jxf, leader);
          const int cs = __shfl(ord, leader) & 1;
          const bool match = mine_pass && row_on && key == kcur;
          const PairW PW = pair_weights(J, match ? J.w[0] : 0.f, match ? J.w[1] : 0.f);
          float tot[6];
          {
            constexpr auto seq = std::make_integer_sequence<int, 16>{};
            float v16[16];
            pair_group<0>(J, PW, v16, seq); tot[0] = transpose_reduce16(v16, c16);
            pair_group<1>(J, PW, v16, seq); tot[1] = transpose_reduce16(v16, c16);
            pair_group<2>(J, PW, v16, seq); tot[2] = transpose_reduce16(v16, c16);
            pair_group<3>(J, PW, v16, seq); tot[3] = transpose_reduce16(v16, c16);
            pair_group<4>(J, PW, v16, seq); tot[4] = transpose_reduce16(v16, c16);
            pair_group<5>(J, PW, v16, seq); tot[5] = transpose_reduce16(v16, c16);
          }
          const int have = (ci >= 0 ? 1 : 0) | (cj >= 0 ? 2 : 0);
#pragma unroll
          for (int g = 0; g < 6; g++) {
            const bool ok = row_on && (ea[g].need & ~have) == 0;
            const int off = ea[g].base + ea[g].ms * cs + ea[g].mj * max(cj, 0);
            if (ok) lds_add(&Fme[off], tot[g]);
          }
          todo &= ~__ballot(match);
        }
      }
      if (sc == 0) { CDV_STAMP(bam, sslot, 12); }
      CDV_STAMP(bam, sslot, 2);
      // ---- the wave's partial E_i, C, u: over its four target slots in fixed order, then published ----
      {
        float v[8] = {eiacc[0], eiacc[1], eiacc[2], eiacc[3], eiacc[4], eiacc[5], Cacc, uacc};
#pragma unroll
        for (int c = 0; c < 8; c++) {
          v[c] += __shfl_xor(v[c], 16);
          v[c] += __shfl_xor(v[c], 32);
          if (sub == 0) part[((sc * MKW + wave) * 8 + c) * CK + p] = v[c];
        }
        if (wave == 0 && sub == 0) ixp[pc] = ixf_patch;
      }
      }   // sub-chunks
      lds_barrier();
      CDV_STAMP(bam, sslot, 9);
      // the wave copies of the footprint summed in fixed order into copy 0 (each thread its own 16-byte columns)
      {
        cdv_float4* s4 = reinterpret_cast<cdv_float4*>(Fw);
        const int F4 = FP / 4;
        for (int i = tid; i < F4; i += 64 * MKW) {
          cdv_float4 t = s4[i];
#pragma unroll
          for (int w = 1; w < MKW; w++) t += s4[w * F4 + i];
          s4[i] = t;
        }
      }
      if (pass == 0) {
        if (tid < 6 * CKS) {         // E_i rows of every patch: the wave partials in fixed order, onto the E_j entries
          const int c = tid / CKS, pp = tid - c * CKS;
          const int sc = pp / CK, pl = pp - sc * CK;
          float tot = 0.f;
#pragma unroll
          for (int w = 0; w < MKW; w++) tot += part[((sc * MKW + w) * 8 + c) * CK + pl];
          const int ib = ixp[pp];
          if (ib >= 0) Ed[(6 * ib + c) * EDL + pp] += tot;
        } else if (tid < 7 * CKS) {  // C, u, q of every patch
          const int pp = tid - 6 * CKS;
          const int sc = pp / CK, pl = pp - sc * CK;
          float Ct = 0.f, ut = 0.f;
#pragma unroll
          for (int w = 0; w < MKW; w++) { Ct += part[((sc * MKW + w) * 8 + 6) * CK + pl]; ut += part[((sc * MKW + w) * 8 + 7) * CK + pl]; }
          const int rr = r0w + pp;
          const bool mine = rr < U && pp < cntw;
          const float q = mine ? 1.0f / (Ct + lm) : 0.f;      // Q = 1 / (C + lambda)   (ba_cuda.cu:548 semantics)
          qs[pp] = q;
          Ed[n6 * EDL + pp] = mine ? ut : 0.f;
          Ed[(ER - 1) * EDL + pp] = Ct;  // (the last padding row of the block: nobody's tile reads beyond row 6N) kept for the debug dump
        }
      }
      CDV_STAMP(bam, sslot, 10);
      lds_barrier();
      CDV_STAMP(bam, sslot, 11);
      // ---- fold the source-frame parts onto the diagonal blocks of copy 0: F_jj[i_s] += F_ii[s], and a self pair
      // (i_s -> i_s) contributes B_ij + B_ij^T to its diagonal block ----
      if (tid < 54) {
        const int s = tid / 27, t = tid - 27 * s;
        const int is = s ? isrc1 : isrc0;
        if (is >= 0) {
          float add = Fw[27 * s + t];
          if (t < 21) {
            int a = 0;
            while (((a + 1) * (a + 2)) / 2 <= t) a++;
            const int b = t - (a * (a + 1)) / 2;
            const float* bij = Fw + 54 + 27 * N + 36 * (s * N + is);
            add += bij[6 * a + b] + bij[6 * b + a];
          }
          Fw[54 + 27 * is + t] += add;
        }
      } else if (tid >= 64 && tid < 100 && isrc0 >= 0 && isrc1 >= 0) {
        // the chunk's two source frames see each other: block (i1, i0) is the transpose of block (i0, i1) of the lower
        // triangle -- folded onto slot 0's copy, so that every slab entry has ONE addend below
        const int ab = tid - 64, a = ab / 6, b = ab - 6 * a;
        float* f0 = Fw + 54 + 27 * N + 36 * isrc1;
        const float* f1 = Fw + 54 + 27 * N + 36 * (N + isrc0);
        f0[ab] += f1[6 * b + a];
      }
      lds_barrier();
      CDV_STAMP(bam, sslot, 3);
      // ---- slab = B - E Q E^T, y = v - E Q u: first the Schur products of the chunk on the matrix cores (K = 16 patches)
      // tile by tile, each entry with exactly one owner lane ----
      // The entries are staged in LDS (the footprint copies 1.. are free now) and leave as 16-byte stores afterwards.
      float* Sd = Fw + FP;   // [slabf] <= (MKW - 1) FP (checked by the host)
      if (tid < slabf - (TRI_N + n6)) Sd[TRI_N + n6 + tid] = 0.f;
      const int T16 = ER / 16, ntile = T16 * (T16 + 1) / 2;
      const float* Fij0 = Fw + 54 + 27 * N;
      // three tiles of a wave in flight together (operand reads, the four dependent MFMA steps and the scattered writes of
      // one tile are ~1,500 cycles of latency when taken alone; 45 tiles at N = 22 are six per wave)
      constexpr int TU = 3;
      constexpr int KST = CKS / 4;       // k-steps of 4 patches
      float qk[KST];
#pragma unroll
      for (int st = 0; st < KST; st++) qk[st] = qs[4 * st + g4];
      for (int p0 = wave; p0 < ntile; p0 += TU * MKW) {
        int ti[TU], tj[TU];
        bool on[TU];   // wave-uniform
#pragma unroll
        for (int u = 0; u < TU; u++) {
          const int pidx = p0 + u * MKW;
          on[u] = pidx < ntile;
          const int pc = on[u] ? pidx : p0;
          int t = (int)((sqrtf(8.0f * (float)pc + 1.0f) - 1.0f) * 0.5f);   // lower-triangular tile pair (ti >= tj)
          if (((t + 1) * (t + 2)) >> 1 <= pc) t++;
          if ((t * (t + 1)) >> 1 > pc) t--;
          ti[u] = t;
          tj[u] = pc - ((t * (t + 1)) >> 1);
        }
        cdv_float4 acc[TU];
#pragma unroll
        for (int u = 0; u < TU; u++) acc[u] = cdv_float4{0.f, 0.f, 0.f, 0.f};
        if (pass == 0) {
          float av[TU][KST], bv[TU][KST];
#pragma unroll
          for (int u = 0; u < TU; u++) {
            const float* pa = Ed + (16 * ti[u] + c16) * EDL;
            const float* pb = Ed + (16 * tj[u] + c16) * EDL;
#pragma unroll
            for (int st = 0; st < KST; st++) {
              av[u][st] = pa[4 * st + g4];
              bv[u][st] = pb[4 * st + g4];
            }
          }
#pragma unroll
          for (int st = 0; st < KST; st++) {
#pragma unroll
            for (int u = 0; u < TU; u++)
              acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][st], qk[st] * bv[u][st], acc[u], 0, 0, 0);
          }
        }
#pragma unroll
        for (int u = 0; u < TU; u++) {
          if (!on[u]) continue;
          const int Cc = 16 * tj[u] + c16;
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const int R = 16 * ti[u] + 4 * g4 + q;
            if (R > n6 || Cc >= n6 || (R < n6 && Cc > R)) continue;   // row 6N = y; column 6N only duplicates it
            Sd[(R < n6) ? tri_index(R, Cc) : TRI_N + Cc] = -acc[u][q];
          }
        }
      }
      CDV_STAMP(bam, sslot, 7);
      lds_barrier();   // LDS only: the E stores above drain in the background
      CDV_STAMP(bam, sslot, 8);
      // ---- the B part: every entry of the folded footprint onto its slab entry.  Each slab entry has at most one addend
      // (the diagonal blocks and v from F_jj; block (i_s, j) from F_ij[s][j], self pairs and the transposed pair of the
      // two source frames having been folded away above): plain read-modify-write, one owner each.  A tile-side lookup
      // of the same values cost 150 vector instructions per tile. ----
      constexpr int G27 = (64 * MKW) / 27, G36 = (64 * MKW) / 36;   // thread groups of 27 / 36 the workgroup holds
      if (tid < 27 * G27) {
        const int jl = tid / 27, t = tid - 27 * jl;
        int a = 0, b = 0;
        if (t < 21) {
          while (((a + 1) * (a + 2)) / 2 <= t) a++;
          b = t - (a * (a + 1)) / 2;
        }
        for (int j = jl; j < N; j += G27) {
          const int idx = (t < 21) ? tri_index(6 * j + a, 6 * j + b) : TRI_N + 6 * j + (t - 21);
          Sd[idx] += Fw[54 + 27 * j + t];
        }
      }
      if (tid < 36 * G36) {
        const int bl = tid / 36, ab = tid - 36 * bl;
        const int a = ab / 6, b = ab - 6 * a;
        for (int blk = bl; blk < 2 * N; blk += G36) {
          const int sl = blk >= N ? 1 : 0, j = blk - N * sl;
          const int is = sl ? isrc1 : isrc0;
          // skipped: no such source frame; the self pair (folded onto the diagonal block); slot 1's view of slot 0
          if (is < 0 || j == is || (sl == 1 && j == isrc0)) continue;
          const int R = is > j ? 6 * is + a : 6 * j + b, Cc = is > j ? 6 * j + b : 6 * is + a;
          Sd[tri_index(R, Cc)] += Fij0[36 * blk + ab];
        }
      }
      CDV_STAMP(bam, sslot, 4);
      lds_barrier();
      if (pass == 0) {
        // ---- the chunk's E columns for the retraction (complete values; 16 bytes per lane: narrow stores are bound by
        // the CU's store issue, not by bandwidth).  Issued here, after the tile loop: the compiler makes whoever reuses a
        // store's data registers wait for the store (s_waitcnt vmcnt inside the tile loop: 2,000 cycles per tile) ----
        if (tid < cntw && r0w + tid < A.U_stride) {   // q, u (and C for the debug dump) of the wide chunk's patches
          const int rr = r0w + tid;
          A.qg[rr] = qs[tid];
          A.ug[rr] = Ed[n6 * EDL + tid];
          if (A.dbg) {
            float* dbgp = A.dbg + (size_t)n6 * n6 + 2 * n6;
            dbgp[A.U_stride + rr] = Ed[(ER - 1) * EDL + tid];
            dbgp[2 * (size_t)A.U_stride + rr] = Ed[n6 * EDL + tid];
          }
        }
        for (int i = tid; i < n6 * (CKS / 4); i += 64 * MKW) {
          const int row = i / (CKS / 4), p4 = (i % (CKS / 4)) * 4;
          if (r0w + p4 >= A.U_stride || p4 >= cntw) continue;   // beyond the rows / beyond this piece of the frame (a multiple of 4 rows)
          const float* e = Ed + row * EDL + p4;
          const cdv_float4 v = {e[0], e[1], e[2], e[3]};
          *reinterpret_cast<cdv_float4*>(A.Edg + (size_t)row * A.U_stride + r0w + p4) = v;
          if (A.dbg)
            *reinterpret_cast<cdv_float4*>(A.dbg + (size_t)n6 * n6 + 2 * n6 + 3 * (size_t)A.U_stride + (size_t)row * A.U_stride + r0w + p4) = v;
        }
      }
      {
        cdv_float4* dst = reinterpret_cast<cdv_float4*>(A.slabs + (size_t)wc * slabf);
        const cdv_float4* src = reinterpret_cast<const cdv_float4*>(Sd);
        if (pass == 0) {
          for (int i = tid; i < slabf / 4; i += 64 * MKW) dst[i] = src[i];
        } else {
          // a further source-frame pair of the same chunk: onto this workgroup's own slab (its stores of the previous
          // pass were drained by that pass's last barrier; read past L1)
          for (int i = tid; i < slabf / 4; i += 64 * MKW) {
            const uint64_t* d2 = reinterpret_cast<const uint64_t*>(dst + i);
            const uint64_t lo = __hip_atomic_load(d2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint64_t hi = __hip_atomic_load(d2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            cdv_float4 v = src[i];
            v[0] += __int_as_float((int)(uint32_t)lo); v[1] += __int_as_float((int)(uint32_t)(lo >> 32));
            v[2] += __int_as_float((int)(uint32_t)hi); v[3] += __int_as_float((int)(uint32_t)(hi >> 32));
            dst[i] = v;
          }
        }
      }
      CDV_STAMP(bam, sslot, 5);
      __syncthreads();   // the next pass / chunk re-zeroes the accumulators; drains this pass's stores as well
      par ^= 1;
      pass++;
    } while (pass < npass);
  }
  CDV_STAMP(bam, sslot, 6);
  CDV_STAMP_RT(bam, sslot, 15);
}

// ---------------------------------------------------------------------------------------------------------
// finish: reduce -> solve -> retract
// ---------------------------------------------------------------------------------------------------------

constexpr int NB = 24;          // columns per block step: four poses (12: twice the steps, each nearly as long)
constexpr int PROWS = 64 - NB;  // panel rows per wave (its first NB lanes hold the diagonal block's rows)

// The solver workgroup: FT threads.  A = [S ; y^T] dense in LDS, (n + 1) rows of LD floats (row n = y: the forward
// substitution happens inside the factorisation).
//
// One block step = a panel of NB columns:
//   panel    every wave that has rows below the diagonal block takes the block's rows in its lanes 0..NB-1 (redundantly)
//            and up to 52 panel rows in the others, one matrix row per lane, NB registers.  Column k: the pivot and the
//            entries L[m][k] of the block come from lanes k and m by v_readlane, so the whole panel is factored inside
//            the wave -- no LDS round trip, no barrier, ~170 instructions.
//   trailing A[R0.., R0..] -= P P^T as 16 x 16 tiles on the matrix cores (K = NB); one owner per entry and step.
// A factored diagonal block goes back in place with its diagonal inverted, once every panel wave has read it.
__device__ __forceinline__ void mid_solve(const BaWinArgs& A, float* smem) {
  const int N = A.N, n = 6 * N;
  const int LD = solve_ld(n);
  float* Am = smem;                     // [n + 1][LD]
  const int t = threadIdx.x, lane = t & 63;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  CDV_IF_STAMPS(const int sslot = 4000 + (t >> 6); unsigned long long t_pan = 0, t_tr = 0, t_x, t_pl = 0, t_pc = 0, t_ps = 0;)
  CDV_STAMP(bam, sslot, 0);
  CDV_STAMP_RT(bam, sslot, 14);
  // lower-triangular tile pairs (ti >= tj) in row-major order: the first T (T + 1) / 2 entries serve any T
  __shared__ int ttab[96];
  __shared__ int s_read, s_verdict, s_lost;
  if (t == 0) {
    s_read = 0;
    s_lost = 0;
    // the solver of this path waits for nothing outside its workgroup (the reduce was a launch of its own): it COMMITS to
    // publishing dX right away -- unless a retract workgroup has already given up on it (cdv_ba.h ho_decide)
    if (A.test == HO_TEST_STALL_BEFORE) ho_test_stall();
    s_verdict = ho_decide(&A.arrive[HO_VERDICT], HO_COMMITTED);
  }
  if (t < 96) {
    int ti = 0, acc_rows = 0;
    while (acc_rows + ti + 1 <= t) { acc_rows += ti + 1; ti++; }
    ttab[t] = (ti << 8) | (t - acc_rows);
  }
  // ---- the reduced, damped system (dense rows, written by the reduce launch) -> LDS: a straight copy, 12 16-byte loads
  // of a thread in flight together.  (Its upper triangle is whatever the buffer held: nothing below lets it reach a
  // lower entry.) ----
  {
    constexpr int LF = 12;
    const cdv_float4* src = reinterpret_cast<const cdv_float4*>(A.ared);
    cdv_float4* dst = reinterpret_cast<cdv_float4*>(Am);
    const int tot4 = (n + 1) * LD / 4;
    for (int base = 0; base < tot4; base += LF * FT) {
      cdv_float4 v[LF];
#pragma unroll
      for (int i = 0; i < LF; i++) {
        const int i4 = base + i * FT + t;
        v[i] = (i4 < tot4) ? src[i4] : cdv_float4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int i = 0; i < LF; i++) {
        const int i4 = base + i * FT + t;
        if (i4 < tot4) dst[i4] = v[i];
      }
    }
  }
  __syncthreads();
  CDV_STAMP(bam, sslot, 2);

  const int nblk = (n + NB - 1) / NB;   // n is a multiple of 6: the last block may be 6, 12 or 18 columns wide
  int readers_due = 0;
  int bad_at = 0;                       // first pose block with a non-positive pivot (1-based), wave-uniform
  for (int kb = 0; kb < nblk; kb++) {
    CDV_IF_STAMPS(t_x = cdv_now();)
    const int c0 = NB * kb;
    const int wb = min(NB, n - c0);       // 12 or 6
    const int R0 = c0 + wb;               // first matrix row below the diagonal block
    const int nrows = n + 1 - R0;         // rows R0 .. n (row n = y)
    // ---- panel ----
    readers_due += max(0, (nrows + PROWS - 1) / PROWS - 1);   // panel waves besides wave 0, counted over all steps
    if (wv * PROWS < nrows || wv == 0) {
      const int pr = wv * PROWS + lane - NB;                       // panel row of lanes >= NB
      const bool isdiag = lane < NB;
      const bool has = isdiag ? lane < wb : pr < nrows;
      const int row = isdiag ? c0 + min(lane, wb - 1) : R0 + min(pr, nrows - 1);
      float a[NB];
      {
        // three 16-byte reads whatever the block's width (beyond a 6-wide last block they fetch bytes nobody uses); a
        // 6-wide block is padded with the identity, rows that do not exist are zero
        const float* rp = &Am[row * LD + c0];
#pragma unroll
        for (int j = 0; j < NB; j += 4) {
          const cdv_float4 v = *reinterpret_cast<const cdv_float4*>(rp + j);
#pragma unroll
          for (int h = 0; h < 4; h++)
            a[j + h] = (has && (j + h) < wb) ? v[h] : ((isdiag && lane == j + h) ? 1.0f : 0.f);
        }
      }
      if (wv != 0) {   // this wave holds its copy of the diagonal block: tell wave 0 (which overwrites it at the end)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(&s_read, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      CDV_IF_STAMPS(asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long t_p0 = cdv_now(); t_pl += t_p0 - t_x;)
      // Right-looking, per column k: all broadcasts L[m][k] (v_readlane, m > k) first, then the updates.  The empty asm
      // statements pin that order: left to itself the compiler pairs broadcast and update (each pair then waits for the
      // v_readlane result) or sinks all updates of a column to just before its pivot (276 broadcast values alive in
      // SGPRs, spilled through v_writelane: 11,000 cycles per panel).  Measured alternatives, none faster than this
      // (4,600 cycles per 24-column panel): the next column's pivot chain started before the bulk of the updates; the
      // broadcasts as 16-byte LDS reads from a per-wave column buffer written one column ahead, as in ba_win.hip (6,900);
      // two halves of 12 columns with the rank-12 update between them from a 12 x 12 block in LDS (132 instead of 276
      // v_readlane: 4,800) -- the cost is the pivot chain of each column, not the broadcasts.
      float rs_k[NB];
#pragma unroll
      for (int k = 0; k < NB; k++) {
        const float piv = readlane_f(a[k], k);
        if (!(piv > 0.f) && bad_at == 0 && k < wb) bad_at = (c0 + k) / 6 + 1;
        const float rs = __builtin_amdgcn_rsqf(piv);
        rs_k[k] = rs;
        const float Lk = a[k] * rs;
        a[k] = Lk;
        float bc[NB];
#pragma unroll
        for (int m = k + 1; m < NB; m++) {
          bc[m] = readlane_f(Lk, m);
          asm volatile("" : "+s"(bc[m]));
        }
#pragma unroll
        for (int m = k + 1; m < NB; m++) {
          a[m] = fmaf(-Lk, bc[m], a[m]);
          asm volatile("" : "+v"(a[m]));
        }
      }
      CDV_IF_STAMPS(asm volatile("v_nop" :: "v"(a[NB - 1])); const unsigned long long t_p1 = cdv_now(); t_pc += t_p1 - t_p0;)
      if (!isdiag) {
        if (has) {
          float* rp = &Am[row * LD + c0];
#pragma unroll
          for (int j = 0; j < NB; j += 4)
            if (j < wb) {   // a 6-wide last block: its second 16 bytes end in two columns of padding (the row's stride covers them)
              *reinterpret_cast<cdv_float4*>(rp + j) = cdv_float4{a[j], a[j + 1], a[j + 2], a[j + 3]};
            }
        }
      } else if (wv == 0) {
        // row `lane` of the block's L goes back IN PLACE, its diagonal entry inverted (six 16-byte writes; the upper part
        // receives what the registers hold there: nobody uses it) -- once every other panel wave has its copy of the block
        // (they read it first thing after the barrier, thousands of cycles ago: the poll is there for the guarantee)
        float o[NB];
#pragma unroll
        for (int j = 0; j < NB; j++) o[j] = (lane == j) ? rs_k[j] : a[j];
        bool copied = false;
        for (int spins = 0; spins < (1 << 16); spins++) {
          if (__hip_atomic_load(&s_read, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= readers_due) { copied = true; break; }
          __builtin_amdgcn_s_sleep(1);
        }
        // a reader that never checked in: the block is NOT overwritten under it (the factor is then wrong, and said to be:
        // the hand-off word is raised like for every other lost hand-off of this path)
        if (!copied && lane == 0) { ba_flag(A.info, BI_HANDOFF, 1); s_lost = 1; }   // nothing is published below
        if (has && copied) {
          float* rp = &Am[row * LD + c0];
#pragma unroll
          for (int j = 0; j < NB; j += 4)
            if (j < wb) *reinterpret_cast<cdv_float4*>(rp + j) = cdv_float4{o[j], o[j + 1], o[j + 2], o[j + 3]};
        }
      }
    } else if (wv == FT / 64 - 1 && kb > 0) {
      // ---- the last wave (never a panel wave: 6N + 1 <= 5 PROWS) has nothing to do during a panel: it inverts the
      // PREVIOUS diagonal block, final since that step's barrier.  W = L_bb^-1, lane j = column j by forward substitution
      // on e_j with the block's rows as uniform (broadcast) LDS reads; W^T goes into the block's strictly upper triangle,
      // which nobody reads (W's diagonal is the inverted diagonal already stored).  The back substitution below then
      // solves a block by a 24 x 24 product instead of a chain of 24 dependent steps. ----
      const int cb = NB * (kb - 1);
      float w[NB];
#pragma unroll
      for (int k = 0; k < NB; k++) {
        const float* lr = &Am[(cb + k) * LD + cb];
        float s0 = (k == lane) ? 1.0f : 0.f, s1 = 0.f;
#pragma unroll
        for (int m = 0; m < k; m++) {
          if (m & 1) s1 = fmaf(-lr[m], w[m], s1);
          else s0 = fmaf(-lr[m], w[m], s0);
        }
        w[k] = (s0 + s1) * lr[k];
      }
      if (lane < NB) {
        float* rp = &Am[(cb + lane) * LD + cb];
#pragma unroll
        for (int jq = 0; jq < NB; jq += 4) {
          cdv_float4 v = *reinterpret_cast<const cdv_float4*>(rp + jq);
#pragma unroll
          for (int h = 0; h < 4; h++) v[h] = (jq + h > lane) ? w[jq + h] : v[h];
          *reinterpret_cast<cdv_float4*>(rp + jq) = v;
        }
      }
    }
    CDV_IF_STAMPS(if (wv == 0) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); t_ps += cdv_now() - t_x; })
    __syncthreads();
    CDV_IF_STAMPS({ const unsigned long long t_y = cdv_now(); t_pan += t_y - t_x; t_x = t_y; })
    // ---- trailing update: lower tiles only, dealt round-robin to the waves; the y row is simply the last panel row ----
    {
      const int c16 = lane & 15, g4 = lane >> 4;
      const int T16 = (nrows + 15) >> 4;
      const int ntile = T16 * (T16 + 1) / 2;
      // (two tiles of a wave in flight together, as in the chunk kernel's tile loop, measured here: 22.9k against 19.4k
      // cycles over the six steps -- the read-modify-writes of two tiles queue behind each other in the LDS pipe)
      constexpr int TU = 1;
      for (int p0 = wv; p0 < ntile; p0 += TU * (FT / 64)) {
        int ti[TU], tj[TU];
        bool on[TU];   // wave-uniform
        float av[TU][NB / 4], bv[TU][NB / 4];
#pragma unroll
        for (int u = 0; u < TU; u++) {
          const int pidx = p0 + u * (FT / 64);
          on[u] = pidx < ntile;
          const int tt = __builtin_amdgcn_readfirstlane(ttab[on[u] ? pidx : p0]);
          ti[u] = tt >> 8; tj[u] = tt & 255;
          const int ra = 16 * ti[u] + c16, rb = 16 * tj[u] + c16;
          const float* pa = &Am[(R0 + min(ra, nrows - 1)) * LD + c0];
          const float* pb = &Am[(R0 + min(rb, nrows - 1)) * LD + c0];
          // the k index of an MFMA step is ours to choose (the same for both operands): lane group g4 takes columns
          // 6 g4 .. 6 g4 + 5 of the panel, so each operand is three 8-byte reads; k >= wb is padding
#pragma unroll
          for (int st = 0; st < NB / 4; st += 2) {
            const int k = (NB / 4) * g4 + st;
            const float2 a2 = *reinterpret_cast<const float2*>(pa + min(k, wb - 2));
            const float2 b2 = *reinterpret_cast<const float2*>(pb + min(k, wb - 2));
            av[u][st] = (ra < nrows && k < wb) ? a2.x : 0.f; av[u][st + 1] = (ra < nrows && k + 1 < wb) ? a2.y : 0.f;
            bv[u][st] = (rb < nrows && k < wb) ? b2.x : 0.f; bv[u][st + 1] = (rb < nrows && k + 1 < wb) ? b2.y : 0.f;
          }
        }
        cdv_float4 acc0[TU], acc1[TU];
#pragma unroll
        for (int u = 0; u < TU; u++) { acc0[u] = cdv_float4{0.f, 0.f, 0.f, 0.f}; acc1[u] = cdv_float4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int st = 0; st < NB / 4; st += 2) {
#pragma unroll
          for (int u = 0; u < TU; u++) {
            acc0[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][st], bv[u][st], acc0[u], 0, 0, 0);
            acc1[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][st + 1], bv[u][st + 1], acc1[u], 0, 0, 0);
          }
        }
#pragma unroll
        for (int u = 0; u < TU; u++) {
          if (!on[u]) continue;
          const cdv_float4 acc = acc0[u] + acc1[u];
          // D: row = 16 ti + 4 g4 + q, col = 16 tj + c16
          const int cc = 16 * tj[u] + c16;
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const int rr = 16 * ti[u] + 4 * g4 + q;
            if (rr < nrows && cc < n - R0 && cc <= rr) Am[(R0 + rr) * LD + R0 + cc] -= acc[q];   // (ds_add_f32 here: 2.5 x slower)
          }
        }
      }
    }
    __syncthreads();
    CDV_IF_STAMPS(t_tr += cdv_now() - t_x;)
  }
  CDV_STAMP(bam, sslot, 3);
  CDV_STAMP_VAL(bam, sslot, 8, t_pan);
  CDV_STAMP_VAL(bam, sslot, 9, t_tr);
  CDV_STAMP_VAL(bam, sslot, 10, t_pl);
  CDV_STAMP_VAL(bam, sslot, 11, t_pc);
  CDV_STAMP_VAL(bam, sslot, 12, t_ps);
  // ---- row n of A now holds z = L^-1 y.  Back substitution L^T x = z inside ONE wave, no barrier: lane l keeps
  // components l, 64 + l, 128 + l of z in three registers.  The LAST block (6 .. 24 wide, not inverted) goes unknown by
  // unknown: x_k = z_k / L[k][k] is broadcast by v_readlane and row k of L (contiguous in LDS, requested a step ahead: it
  // does not depend on the chain) folds it into the components before it.  Every block before it, last to first, is
  // x_b = W_b^T z_b -- lane c sums its own row of the block (W^T) against the block's z as broadcast reads, four partial
  // sums -- followed by z -= L[b][:]^T x_b with x from v_readlane: ~1,200 cycles per block of 24 against 3,400 for the
  // chain.  (Block by block with the other waves' help: two barriers per block, 6,000 cycles each.) ----
  float* z = Am + (size_t)n * LD;
  if (wv == 0) {
    float zr[3], xr[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 3; q++) zr[q] = (64 * q + lane < n) ? z[64 * q + lane] : 0.f;
    const int cl = NB * (nblk - 1);   // first unknown of the last block
    // two unknowns per step (n and the block sizes are even): the chain fma -> v_readlane (~40 cycles to a dependent
    // vector instruction) -> three scalar-operand instructions is paid once per pair
    float La[3], Lb[3], Na[3], Nb[3];   // rows k, k - 1 in use; the next pair's, requested a step ahead
#pragma unroll
    for (int q = 0; q < 3; q++) {
      Na[q] = Am[(size_t)(n - 1) * LD + min(64 * q + lane, LD - 1)];
      Nb[q] = Am[(size_t)(n - 2) * LD + min(64 * q + lane, LD - 1)];
    }
#pragma unroll
    for (int r = 2; r >= 0; r--) {
      for (int k = min(n, 64 * (r + 1)) - 1; k >= max(64 * r, cl); k -= 2) {
#pragma unroll
        for (int q = 0; q < 3; q++) { La[q] = Na[q]; Lb[q] = Nb[q]; }
        const int ka = max(k - 2, 0), kb2 = max(k - 3, 0);
#pragma unroll
        for (int q = 0; q < 3; q++) {
          Na[q] = Am[(size_t)ka * LD + min(64 * q + lane, LD - 1)];
          Nb[q] = Am[(size_t)kb2 * LD + min(64 * q + lane, LD - 1)];
        }
        const int lk = k - 64 * r;
        const float dk = readlane_f(La[r], lk), dk1 = readlane_f(Lb[r], lk - 1);   // inverted diagonals
        const float lkk1 = readlane_f(La[r], lk - 1);                              // L[k][k - 1]
        const float zk = readlane_f(zr[r], lk), zk1 = readlane_f(zr[r], lk - 1);
        const float xk = zk * dk;
        const float xk1 = fmaf(-lkk1, xk, zk1) * dk1;
        xr[r] = (lane == lk) ? xk : ((lane == lk - 1) ? xk1 : xr[r]);
        // components >= k - 1 receive garbage here (the upper part of the rows): their x is kept in xr
#pragma unroll
        for (int q = 0; q <= r; q++) zr[q] = fmaf(-Lb[q], xk1, fmaf(-La[q], xk, zr[q]));
      }
    }
#pragma unroll
    for (int q = 0; q < 3; q++)
      if (64 * q + lane >= cl && 64 * q + lane < n) z[64 * q + lane] = xr[q];
    for (int kb = nblk - 2; kb >= 0; kb--) {
      const int c0 = NB * kb;
      // the block's z out of the registers (components past the block are dead: they hold garbage since their solve)
#pragma unroll
      for (int q = 0; q < 3; q++) {
        const int comp = 64 * q + lane;
        if (comp >= c0 && comp < c0 + NB) z[comp] = zr[q];
      }
      const int c = min(lane, NB - 1);
      const float* wr = &Am[(size_t)(c0 + c) * LD + c0];   // row c of the block: L left of the diagonal, W^T from it on
      const float* zb = z + c0;
      float ps[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int jq = 0; jq < NB; jq += 4) {
        const cdv_float4 w4 = *reinterpret_cast<const cdv_float4*>(wr + jq);
        const cdv_float4 z4 = *reinterpret_cast<const cdv_float4*>(zb + jq);
#pragma unroll
        for (int h = 0; h < 4; h++) ps[h] = fmaf((jq + h >= c) ? w4[h] : 0.f, z4[h], ps[h]);
      }
      const float xv = (ps[0] + ps[1]) + (ps[2] + ps[3]);
      if (lane < NB) z[c0 + lane] = xv;   // x over the block's z, which nobody needs any more
      // z of the earlier components -= L[block rows][:]^T x  (components >= c0 receive garbage: dead)
      float acc[3][2] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
      for (int r = 0; r < NB; r++) {
        const float xs = readlane_f(xv, r);
        const float* lrow = &Am[(size_t)(c0 + r) * LD];
#pragma unroll
        for (int q = 0; q < 3; q++) acc[q][r & 1] = fmaf(lrow[min(64 * q + lane, LD - 1)], xs, acc[q][r & 1]);
      }
#pragma unroll
      for (int q = 0; q < 3; q++) zr[q] -= acc[q][0] + acc[q][1];
    }
  }
  __syncthreads();
  CDV_STAMP(bam, sslot, 4);
  if (A.test == HO_TEST_STALL_AFTER && t == 0) ho_test_stall();
  if (A.test == HO_TEST_STALL_AFTER) __syncthreads();
  if (s_verdict != HO_COMMITTED || s_lost) {
    // a retract workgroup abandoned the launch before the commit (all of them have left or will), or the factor is not to
    // be trusted (a hand-off inside this workgroup failed): nothing is published, nothing is applied
    if (t == 0) {
      ba_flag(A.info, BI_HANDOFF, 1);
      if (s_verdict == HO_COMMITTED)      // retract workgroups are waiting for a solution that will not come: tell them
        __hip_atomic_store(&A.arrive[HO_VERDICT], HO_ABANDONED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return;
  }
  if (t < n) {
    const float x = z[t];
    __hip_atomic_store(&A.granX[t], (1ull << 32) | (uint64_t)(uint32_t)__float_as_int(x), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
    A.dXg[t] = x;
    if (A.dbg) A.dbg[(size_t)n * n + n + t] = x;
  }
  if (t == 0 && bad_at > 0) ba_flag(A.info, BI_CHOL, bad_at);
  CDV_STAMP(bam, sslot, 5);
  CDV_STAMP_RT(bam, sslot, 15);
}

// Sum of the chunk slabs, a launch of its own: at N = 22, U = 4,312 the slabs are 9.6 MB and a CU streams ~10 B per
// cycle, so the reduce wants the whole chip, while the solve + retract launch that follows needs the solver's CU-sized
// LDS in every workgroup.  Thread = (16-byte column, slab subset): 16 subsets in the lanes of a DPP row, W in {1, 2, 4}
// further ones in separate rows whose partials meet in LDS.  W and every summation order depend on the number of slabs
// only: the sum is the same whatever the launch geometry.
template <bool TABLE>
__global__ __launch_bounds__(256) void ba_mid_reduce_kernel(BaWinArgs A_in) {
  const BaWinArgs A = with_dyn(A_in);
  const int32_t* __restrict__ gmeta = A.gmeta;
  const PatchSpan sp = patch_span<TABLE>(A);
  const int U = sp.U;
  if (graph_error_of(gmeta, TABLE) || U > A.U_max) return;
  __shared__ cdv_float4 partial[16];
  const int n6 = 6 * A.N;
  const int TRI_N = (n6 * (n6 + 1)) >> 1;
  const int slabf = (TRI_N + n6 + 7) / 8 * 8;
  const int LD = solve_ld(n6);
  const int S4 = slabf / 4;
  const int nsl = wide_count(A, U);      // one slab per wide chunk
  const int W = nsl <= 128 ? 1 : (nsl <= 256 ? 2 : 4);
  const int cols_per_wg = 16 / W;
  const int tid = threadIdx.x, g = tid & 15, slot = tid >> 4;
  const int ws = slot % W, cs = slot / W;
  for (int col0 = (int)blockIdx.x * cols_per_wg; col0 < S4; col0 += (int)gridDim.x * cols_per_wg) {   // workgroup-uniform
    const int col = col0 + cs;
    const bool mine = col < S4;
    cdv_float4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    if (mine) {
      const cdv_float4* src = reinterpret_cast<const cdv_float4*>(A.slabs) + col;
      for (int s0 = ws * 16 + g; s0 < nsl; s0 += 8 * 16 * W) {   // 8 loads in flight: 128 W slabs per memory round trip
        cdv_float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const int sidx = s0 + u * 16 * W;
          v[u] = (sidx < nsl) ? src[(size_t)sidx * S4] : cdv_float4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < 8; u++) acc[u & 1] += v[u];
      }
    }
    cdv_float4 tot = acc[0] + acc[1];
#pragma unroll
    for (int j = 0; j < 4; j++) {   // lane g = 0 of the row collects the 16 partials with DPP row shifts (a fixed tree)
      float tt = tot[j];
      tt += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(tt), 0x101, 0xf, 0xf, true));   // row_shl:1
      tt += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(tt), 0x102, 0xf, 0xf, true));   // row_shl:2
      tt += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(tt), 0x104, 0xf, 0xf, true));   // row_shl:4
      tt += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(tt), 0x108, 0xf, 0xf, true));   // row_shl:8
      tot[j] = tt;
    }
    if (W > 1) {
      if (g == 0) partial[slot] = tot;
      __syncthreads();
      if (g == 0 && ws == 0) {
        for (int w = 1; w < W; w++) tot += partial[slot + w];
      }
      __syncthreads();
    }
    if (mine && g == 0 && ws == 0) {
      // the four entries leave as dense rows [6N + 1][LD] (row 6N = y), damped: S += I (1e-4 S + 1.0) (ba_cuda.cu:589
      // semantics) -- the solver then copies rows instead of unpacking a triangle
      const int idx0 = 4 * col;
      int R, Cc;
      if (idx0 >= TRI_N) { R = n6; Cc = idx0 - TRI_N; }
      else {
        R = (int)((sqrtf(8.0f * (float)idx0 + 1.0f) - 1.0f) * 0.5f);
        if (((R + 1) * (R + 2)) >> 1 <= idx0) R++;        // the float estimate is off by at most one
        if ((R * (R + 1)) >> 1 > idx0) R--;
        Cc = idx0 - ((R * (R + 1)) >> 1);
      }
#pragma unroll
      for (int h = 0; h < 4; h++) {
        const int idx = idx0 + h;
        if (h > 0) {
          if (R < n6 && Cc == R) { R++; Cc = 0; } else Cc++;
          if (idx == TRI_N) { R = n6; Cc = 0; }
        }
        if (idx < TRI_N + n6) {
          float sv = tot[h];
          if (R == Cc) sv += 1e-4f * sv + 1.0f;
          A.ared[(size_t)R * LD + Cc] = sv;
          if (A.dbg) {
            if (R < n6) { A.dbg[(size_t)R * n6 + Cc] = sv; A.dbg[(size_t)Cc * n6 + R] = sv; }
            else A.dbg[(size_t)n6 * n6 + Cc] = sv;
          }
        }
      }
    }
  }
}

// SNP: the E column a retract thread keeps in registers (>= 6 N)
template <int SNP, bool TABLE>
__global__ __launch_bounds__(FT) void ba_mid_finish_kernel(BaWinArgs A_in) {
  const BaWinArgs A = with_dyn(A_in);
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int32_t* __restrict__ gmeta = A.gmeta;
  const PatchSpan sp = patch_span<TABLE>(A);
  const int U = sp.U;
  if (graph_error_of(gmeta, TABLE) || U > A.U_max) return;
  const int RT = (int)gridDim.x - 1;   // retract workgroups, FT patches each per pass
  const int tid = threadIdx.x;
  if (blockIdx.x == 0) {
    mid_solve(A, smem);
    return;
  }
  const int b = (int)blockIdx.x - 1;
  const int N = A.N, n6 = 6 * N;
  CDV_IF_STAMPS(const int sslot = 4100 + b * 8 + (tid >> 6);)
  CDV_STAMP(bam, sslot, 0);
  CDV_STAMP_RT(bam, sslot, 14);
  CDV_STAMP(bam, sslot, 1);
  // ---- 2. before dX exists: everything of this thread's patch that does not depend on it ----
  float* sdx = smem;   // [SNP]
  const int P = A.P, PP = P * P;
  int r = b * FT + tid;
  bool livep = r < U;
  float ev[SNP];
#pragma unroll
  for (int i = 0; i < SNP; i++) ev[i] = (livep && i < n6) ? A.Edg[(size_t)i * A.U_stride + r] : 0.f;
  float uv = 0.f, qv = 0.f, d0 = 0.f;
  float* pk = nullptr;
  if (livep) {
    const PatchRow row = patch_row<TABLE>(A, r);
    livep = row.deg > 0;             // a table id without an edge: not part of the graph, not retracted
    uv = A.ug[r]; qv = A.qg[r];
    pk = A.patches + row.id * 3 * PP + 2 * PP;
    d0 = pk[0];                      // the depth is read from pixel [0][0]   (ba_cuda.cu:218 semantics)
  }
  CDV_STAMP(bam, sslot, 2);
  // ---- 3. wait for the solver: all-or-nothing (cdv_ba.h ho_decide; the same protocol as ba_win.hip's finish launch) ----
  __shared__ int s_bad, s_go;
  const int patience = A.test ? (1 << 10) : (1 << 21);
  for (int round = 0; round < 2; round++) {
    if (tid == 0) { s_bad = 0; s_go = 0; }
    __syncthreads();
    if (tid < MID_GRAN) {   // thread t polls the granule of unknown t until its tag shows up; the poll is the load of dX
      float xv = 0.f;
      bool ok = tid >= n6;
      for (int spins = 0; spins < (round == 0 ? patience : (1 << 24)); spins++) {
        if (!ok) {
          const uint64_t g = __hip_atomic_load(&A.granX[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if ((uint32_t)(g >> 32) == 1u) { xv = __int_as_float((int)(uint32_t)g); ok = true; }
        }
        if (__all(ok)) break;
        if ((spins & 255) == 255 &&      // the solver said it will not publish
            __hip_atomic_load(&A.arrive[HO_VERDICT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == HO_ABANDONED) break;
        __builtin_amdgcn_s_sleep(2);
      }
      if (tid < SNP) sdx[tid] = xv;
      if (!ok) s_bad = 1;
    }
    __syncthreads();
    if (!s_bad) break;
    // patience ran out: ask for the verdict.  COMMITTED: the solution is coming, wait on (once); otherwise nobody applies it
    if (tid == 0) s_go = (round == 0 && ho_decide(&A.arrive[HO_VERDICT], HO_ABANDONED) == HO_COMMITTED) ? 1 : 0;
    __syncthreads();
    if (!s_go) break;
  }
  if (s_bad) {
    if (tid == 0) ba_flag(A.info, BI_HANDOFF, 1);
    return;   // no update without a solution: poses and depths stay as they were -- in EVERY retract workgroup
  }
  CDV_STAMP(bam, sslot, 3);
  // ---- 4. pose retraction T <- Exp(dX_i) T: the first retract workgroup's first N lanes ----
  if (b == 0 && tid < N) {
    float* pp = A.poses + 7 * (size_t)(A.t0 + tid);
    float pose[7], xi[6];
#pragma unroll
    for (int c = 0; c < 7; c++) pose[c] = pp[c];
#pragma unroll
    for (int c = 0; c < 6; c++) xi[c] = sdx[6 * tid + c];
    se3_retract_raw(xi, pose);
#pragma unroll
    for (int c = 0; c < 7; c++) pp[c] = pose[c];
  }
  // ---- 5. dZ = Q (u - E^T dX), inverse-depth update (ba_cuda.cu:592,209-229 semantics) ----
  for (;;) {
    if (livep) {
      float sacc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c4 = 0; c4 < SNP / 4; c4++) {
        const cdv_float4 x4 = *reinterpret_cast<const cdv_float4*>(&sdx[4 * c4]);
#pragma unroll
        for (int j = 0; j < 4; j++) sacc[j] += ev[4 * c4 + j] * x4[j];
      }
      const float dz = qv * (uv - ((sacc[0] + sacc[1]) + (sacc[2] + sacc[3])));
      if (A.dbg) A.dbg[(size_t)36 * N * N + 12 * N + r] = dz;
      float d = d0 + dz;
      d = (d > 20.f) ? 1.0f : d;
      d = fmaxf(d, 1e-4f);
      store_depth(pk, PP, d);
    }
    r += RT * FT;
    if (r - tid >= U) { CDV_STAMP(bam, sslot, 4); CDV_STAMP_RT(bam, sslot, 15); break; }          // workgroup-uniform
    livep = r < U;
#pragma unroll
    for (int i = 0; i < SNP; i++) ev[i] = (livep && i < n6) ? A.Edg[(size_t)i * A.U_stride + r] : 0.f;
    if (livep) {
      const PatchRow row = patch_row<TABLE>(A, r);
      livep = row.deg > 0;
      uv = A.ug[r]; qv = A.qg[r];
      pk = A.patches + row.id * 3 * PP + 2 * PP;
      d0 = pk[0];
    }
  }
}

template <bool HAS_II, int MKW, bool TABLE>
hipError_t chunk_attr() {
  return hipFuncSetAttribute((const void*)ba_mid_chunk_kernel<HAS_II, MKW, TABLE>, hipFuncAttributeMaxDynamicSharedMemorySize,
                             160 * 1024 - 256);
}

template <int MKW, bool TABLE>
void launch_chunk(const BaWinArgs& a, int grid, size_t lds, hipStream_t s) {
  if (a.has_ii) hipLaunchKernelGGL((ba_mid_chunk_kernel<true, MKW, TABLE>), dim3(grid), dim3(64 * MKW), lds, s, a);
  else hipLaunchKernelGGL((ba_mid_chunk_kernel<false, MKW, TABLE>), dim3(grid), dim3(64 * MKW), lds, s, a);
}

template <int SNP, bool TABLE>
hipError_t finish_attr() {
  return hipFuncSetAttribute((const void*)ba_mid_finish_kernel<SNP, TABLE>, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024);
}

template <bool TABLE>
void launch_finish(const BaWinArgs& a, int n, int RW, size_t lds_f, hipStream_t s) {
  if (n <= 96) hipLaunchKernelGGL((ba_mid_finish_kernel<96, TABLE>), dim3(1 + RW), dim3(FT), lds_f, s, a);
  else if (n <= 144) hipLaunchKernelGGL((ba_mid_finish_kernel<144, TABLE>), dim3(1 + RW), dim3(FT), lds_f, s, a);
  else hipLaunchKernelGGL((ba_mid_finish_kernel<192, TABLE>), dim3(1 + RW), dim3(FT), lds_f, s, a);
}

}  // namespace

int cdv::cdv_ba_mid_wide_per_frame(int ppf) { return wide_per_frame(ppf); }

int cdv::cdv_ba_mid_iteration(const BaWinArgs& a, hipStream_t s) {
  static hipError_t attr_err = [] {
    hipError_t e = hipSuccess, x;
    if ((x = chunk_attr<true, 8, false>()) != hipSuccess) e = x;
    if ((x = chunk_attr<false, 8, false>()) != hipSuccess) e = x;
    if ((x = chunk_attr<true, 8, true>()) != hipSuccess) e = x;
    if ((x = chunk_attr<false, 8, true>()) != hipSuccess) e = x;
    if ((x = chunk_attr<true, 7, false>()) != hipSuccess) e = x;
    if ((x = chunk_attr<false, 7, false>()) != hipSuccess) e = x;
    if ((x = chunk_attr<true, 7, true>()) != hipSuccess) e = x;
    if ((x = chunk_attr<false, 7, true>()) != hipSuccess) e = x;
    if ((x = chunk_attr<true, 4, false>()) != hipSuccess) e = x;
    if ((x = chunk_attr<false, 4, false>()) != hipSuccess) e = x;
    if ((x = chunk_attr<true, 4, true>()) != hipSuccess) e = x;
    if ((x = chunk_attr<false, 4, true>()) != hipSuccess) e = x;
    if ((x = finish_attr<96, false>()) != hipSuccess) e = x;
    if ((x = finish_attr<144, false>()) != hipSuccess) e = x;
    if ((x = finish_attr<192, false>()) != hipSuccess) e = x;
    if ((x = finish_attr<96, true>()) != hipSuccess) e = x;
    if ((x = finish_attr<144, true>()) != hipSuccess) e = x;
    if ((x = finish_attr<192, true>()) != hipSuccess) e = x;
    return e;
  }();
  CDV_HIP_CHECK(attr_err);
  const int N = a.N;
  // 8 waves per chunk workgroup: 32 target slots per round.  (The kernel needs ~250 VGPRs, so a CU holds 8 of its waves
  // whatever the split.  Workgroups of 4 waves, two per CU -- so that the stress configuration's 294 chunks are resident at
  // once instead of taking a second round of 38 workgroups -- were measured: 49.1 against 41.7 us; the kernel is written
  // for either, CDV_MID_WAVES=4 selects them.)
  const int n_wide = a.ppf > 0 ? (a.tab_cap / a.ppf) * wide_per_frame(a.ppf) : cdv_div_up(a.n_ck_cap, SC);
  CDV_REQUIRE(n_wide <= a.n_ck_cap, CDV_ERR_WORKSPACE, "cdv_ba_forward: more wide chunks than slabs in the workspace");   // (ba.hip drops the hint before)
  const int n_ck = n_wide < WIN_MAX_GRID ? n_wide : WIN_MAX_GRID;
  const bool table = a.tab_cap > 0;
  static const int mkw_env = getenv("CDV_MID_WAVES") ? atoi(getenv("CDV_MID_WAVES")) : 8;
  const int mkw = mkw_env == 4 ? 4 : (mkw_env == 7 ? 7 : 8);
  const size_t lds = chunk_lds_bytes(N, mkw);
  CDV_REQUIRE(lds <= 160 * 1024 - 256, CDV_ERR_UNSUPPORTED, "cdv_ba_forward: chunk footprint exceeds LDS");
  if (mkw == 8) {
    if (table) launch_chunk<8, true>(a, n_ck, lds, s);
    else launch_chunk<8, false>(a, n_ck, lds, s);
  } else if (mkw == 7) {
    if (table) launch_chunk<7, true>(a, n_ck, lds, s);
    else launch_chunk<7, false>(a, n_ck, lds, s);
  } else {
    if (table) launch_chunk<4, true>(a, n_ck, lds, s);
    else launch_chunk<4, false>(a, n_ck, lds, s);
  }
  const int n = 6 * N;
  {
    const int S4 = mid_slab(N) / 4;
    const int grid = cdv_div_up(S4, 4);    // one trip at W = 4; fewer columns per workgroup only shorten the loop
    if (table) hipLaunchKernelGGL(ba_mid_reduce_kernel<true>, dim3(grid < 1024 ? grid : 1024), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(ba_mid_reduce_kernel<false>, dim3(grid < 1024 ? grid : 1024), dim3(256), 0, s, a);
  }
  // solver + retract workgroups (FT patches each per pass; they all poll the solver)
  int RW = cdv_div_up(a.U_max, FT);
  RW = RW < 1 ? 1 : (RW > MID_MAX_RW ? MID_MAX_RW : RW);
  const size_t lds_f = sizeof(float) * ((size_t)(n + 1) * solve_ld(n) + 8);
  if (table) launch_finish<true>(a, n, RW, lds_f, s);
  else launch_finish<false>(a, n, RW, lds_f, s);
  CDV_LAUNCH_CHECK();
  return CDV_OK;
}
