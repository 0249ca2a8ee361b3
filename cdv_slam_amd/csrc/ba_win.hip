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
//                          - B and v: every lane forms the 90 distinct products of its edge (the lower triangles of
//                            Ji^T w Ji and Jj^T w Jj, Ji^T w Jj, the two right-hand sides); the 16 patches of a target
//                            slot share their frame pair, so the pair's sums are wavefront shuffle reductions (DPP
//                            row rotations, as a transpose-reduce: 45 instructions per 16 values, lane l ends up
//                            owning one sum) added into the wave's PRIVATE packed-triangular copy of [S | y] in LDS;
//                          - the chunk's Schur products [E; u] diag(q) [E; u]^T (K = 16 patches) as MFMA tiles; the tile
//                            owner adds the four wave copies in fixed order, subtracts and stores ONE partial system
//                            per chunk (a "slab": packed lower triangle of S + y, 7.6 KB) with plain stores;
//                          - E columns, q, u of the chunk go to HBM for the retraction (plain stores, complete values:
//                            nothing has to be re-zeroed afterwards).
//   2. ba_finish_kernel  workgroup 0 = the solver wave; workgroups 1.. (119: a CU streams ~10 B per cycle, the 1.5 MB of
//                        slabs want many) first reduce the slabs -- thread = (16-byte column, one of 64 slab subsets),
//                        DPP row shifts over 16 of them, the four waves' partials through LDS: a fixed order -- hand the
//                        reduced system to the solver (write-through stores, drained, then the launch's token into the
//                        workgroup's own flag word: no shared counter), and the first U / 256 of them preload their
//                        patches' E columns while the solver runs, wait for dX (tagged 8-byte granules: the poll is the
//                        load), and retract depths and poses (ba_cuda.cu:178-229 semantics).
//                        Solver: lane r holds row r of [S ; y^T] in registers (the right-hand side as row 60: forward
//                        substitution for free); column k is broadcast through LDS (one ds_write, then 16-byte
//                        broadcast reads) one column AHEAD of the rank-1 updates that consume it, so the LDS round
//                        trip and the rsqrt chain of column k + 1 hide under the packed FMAs of column k.
#include "cdv_ba_pairs.h"

using namespace cdv;

CDV_STAMP_TU(baw)

namespace {

constexpr int CK = WIN_CK;
constexpr int CKW = 8;                     // waves per chunk workgroup: two per SIMD (a lone wave issues a vector
                                           // instruction every 4 cycles, two alternate at 2), 32 target slots per round
constexpr int SN = WIN_SN;
constexpr int TRI = WIN_TRI;
constexpr int SLAB = WIN_SLAB;
constexpr int EDL = CK + 1;                // row stride of the chunk's [E; u] block in LDS
constexpr int LDS_CHUNK_FLOATS = CKW * SLAB + 64 * EDL + CKW * 8 * CK + 2 * CK;

// one reduced value of frame pair (ci, cj) (free-pose indices or -1) into the wave's packed copy of [S | y]
__device__ __forceinline__ void pair_emit(float total, int code, int ci, int cj, bool row_on, float* __restrict__ Sw) {
  const int kind = code >> 6, a = (code >> 3) & 7, b = code & 7;
  const bool isv = (kind == 1) || (kind == 3);
  const int rp = (kind == 2 || kind == 3) ? cj : ci;     // pose of the row index
  const int cp = (kind == 0) ? ci : cj;                  // pose of the column index (unused for v)
  const bool ok = row_on && kind != 7 && rp >= 0 && (isv || cp >= 0);
  const int R = 6 * rp + a, Cc = 6 * cp + b;
  const int hi = max(R, Cc), lo = min(R, Cc);            // B_ij lands in the block of the lower triangle
  const int idx = isv ? TRI + R : tri_index(hi, lo);
  // a self pair (i == j) folds B_ij and its transpose onto one block: the diagonal receives both
  const float v = (kind == 4 && ci == cj && a == b) ? 2.0f * total : total;
  if (ok) lds_add(&Sw[idx], v);
}

// ---------------------------------------------------------------------------------------------------------
// finish: reduce -> solve -> retract
// ---------------------------------------------------------------------------------------------------------

typedef float cdv_float2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float ld_agent(const float* p) {   // global_load_dword sc1: past this CU's L1
  return __int_as_float((int)__hip_atomic_load(reinterpret_cast<const uint32_t*>(p), __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT));
}

// Reduce share of workgroup b of RW.
__device__ __forceinline__ void reduce_slabs(const BaWinArgs& A, int b, int RW, int U, int tid, bool flag_it) {
  const bool act = tid < 256;
// ---- reduce the chunk slabs.  A CU streams ~10 B per cycle, so the 1.5 MB of slabs want many CUs: a workgroup takes
  //         FOUR 16-byte columns, thread = (column, one of 64 interleaved slab subsets: 16 in the lanes of a DPP row x the 4
  //         waves).  Per-thread sums in slab order, the row's 16 partials by DPP row shifts, the four waves' partials in
  //         LDS: a fixed tree that depends on nothing but the number of slabs -- reproducible bits whatever the launch
  //         geometry. ----
  {
    const int nsl = (U + CK - 1) / CK;
    const int g = tid & 15, cs = (tid >> 4) & 3, w = (tid >> 6) & 3;
    __shared__ cdv_float4 s_part[4][4];
    const int token = A.token;
    for (int col0 = b * 4; col0 < SLAB / 4; col0 += RW * 4) {   // workgroup-uniform trip count
      const int col = col0 + cs;
      const bool mine = act && col < SLAB / 4;
      cdv_float4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
      if (mine) {
        const cdv_float4* src = reinterpret_cast<const cdv_float4*>(A.slabs) + col;
        for (int s0 = g + 16 * w; s0 < nsl; s0 += 8 * 64) {   // 8 loads in flight: 512 slabs per memory round trip
          cdv_float4 v[8];
#pragma unroll
          for (int u = 0; u < 8; u++) {
            const int sidx = s0 + 64 * u;
            v[u] = (sidx < nsl) ? src[(size_t)sidx * (SLAB / 4)] : cdv_float4{0.f, 0.f, 0.f, 0.f};
          }
#pragma unroll
          for (int u = 0; u < 8; u++) acc[u & 1] += v[u];
        }
      }
      cdv_float4 tot = acc[0] + acc[1];
#pragma unroll
      for (int j = 0; j < 4; j++) {   // lane g = 0 of the row collects the 16 partials with DPP row shifts
        float t = tot[j];
        t += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), 0x101, 0xf, 0xf, true));   // row_shl:1
        t += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), 0x102, 0xf, 0xf, true));   // row_shl:2
        t += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), 0x104, 0xf, 0xf, true));   // row_shl:4
        t += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), 0x108, 0xf, 0xf, true));   // row_shl:8
        tot[j] = t;
      }
      if (act && g == 0) s_part[w][cs] = tot;
      __syncthreads();
      if (w == 0 && g == 0 && mine) {
        const cdv_float4 t4 = ((s_part[0][cs] + s_part[1][cs]) + s_part[2][cs]) + s_part[3][cs];
        // written through (8-byte agent-scope stores): the solver reads them past its L1
        uint64_t* dst = reinterpret_cast<uint64_t*>(A.ared + 4 * col);
        __hip_atomic_store(dst, ((uint64_t)(uint32_t)__float_as_int(t4[1]) << 32) | (uint32_t)__float_as_int(t4[0]),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(dst + 1, ((uint64_t)(uint32_t)__float_as_int(t4[3]) << 32) | (uint32_t)__float_as_int(t4[2]),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (col0 + RW * 4 < SLAB / 4) __syncthreads();   // another trip reuses s_part (workgroup-uniform)
    }
    if (tid < 64 && flag_it) {   // wave 0 holds every store of this workgroup: it drains, then raises the flag (no barrier needed)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (tid == 0) __hip_atomic_store(&A.arrive[16 + b], token, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// (Measured and rejected: a second wave on another SIMD applying the published columns to columns [34, 60) until the
// chain wave gets there -- half the rank-1 work per wave -- left the factorisation at 14.4k cycles: a lone wave issues a
// v_pk_fma_f32 every 8 cycles and a ds_read_b128 every ~8, and the ~30 instructions of a column step, chain included,
// are paid per column whoever does the bulk of the update.)
// The 60 x 60 system in the registers of ONE wave.  Lane r holds row r of [S ; y^T] (lane 60 = the right-hand side).
__device__ __forceinline__ void solve_wave(const BaWinArgs& A, int RW) {
  __shared__ __attribute__((aligned(16))) float colb[64];          // the column being broadcast
  __shared__ __attribute__((aligned(16))) float Lt[(SN + 1) * 68]; // L for the back substitution (row stride 68)
  const int lane = threadIdx.x;
  const int n = 6 * A.N;
  CDV_IF_STAMPS(const int sslot = 4000;)
  CDV_STAMP(baw, sslot, 0);
  CDV_STAMP_RT(baw, sslot, 14);
  // ---- wait for the reduce workgroups: each stores the launch's token into its own flag word once its part of the
  // reduced system is written through -- no shared counter (same-address atomics serialise, ~90 ns each).  Bounded: a
  // lost hand-off must not hang the device. ----
  const int token = A.token;
  bool ok = false;
  for (int spins = 0; spins < (1 << 20); spins++) {
    const int f0 = lane < RW ? __hip_atomic_load(&A.arrive[16 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : token;
    const int f1 = lane + 64 < RW ? __hip_atomic_load(&A.arrive[16 + 64 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : token;
    if (__all(f0 == token && f1 == token)) { ok = true; break; }
    __builtin_amdgcn_s_sleep(1);
  }
  if (A.test == HO_TEST_STALL_BEFORE) ho_test_stall();
  // ---- ONE decision per launch (cdv_ba.h ho_decide): from here on the solver has nothing left to wait for, so it COMMITS to
  // publishing dX -- unless a retract workgroup has given up first, in which case nothing of this iteration may be
  // applied by anybody.  The compare-and-swap is issued now and looked at before the granules go out: its round trip
  // lies under the loads and the factorisation. ----
  int verdict = HO_ABANDONED;
  if (lane == 0) verdict = ho_decide(&A.arrive[HO_VERDICT], ok ? HO_COMMITTED : HO_ABANDONED);
  if (!ok) {
    if (lane == 0) ba_flag(A.info, BI_HANDOFF, 1);
    return;   // the retract workgroups read the verdict and leave the state untouched
  }
  CDV_STAMP(baw, sslot, 1);
  // ---- the reduced system: coalesced 8-byte write-through loads (every lane 15 of them, one memory round trip) into
  // LDS, then my row from there.  The packed rows follow each other, so a row is read at full length: its tail (the
  // head of the next rows) sits where the upper triangle would be, which lane r computes on but nobody ever reads (a
  // pivot is lane k's own a[k][k], a broadcast value lane c's a[c][k], c > k) ----
  {
    uint64_t v[SLAB / 128 + 1];
    const uint64_t* src = reinterpret_cast<const uint64_t*>(A.ared);
#pragma unroll
    for (int i = 0; i < SLAB / 128 + 1; i++)
      v[i] = (64 * i + lane < SLAB / 2) ? __hip_atomic_load(src + 64 * i + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
    uint64_t* dst = reinterpret_cast<uint64_t*>(Lt);   // (SN + 1) * 68 floats >= SLAB
#pragma unroll
    for (int i = 0; i < SLAB / 128 + 1; i++)
      if (64 * i + lane < SLAB / 2) dst[64 * i + lane] = v[i];
  }
  wave_lds_sync();
  // S += I (1e-4 S + 1.0)  (ba_cuda.cu:589 semantics; rows >= 6 N: identity) on the packed triangle in LDS, lane r its own
  // diagonal entry -- as a select per column while the row is read (`if (c == row)`) it was 60 compares on hoisted lane masks
  if (lane < SN) {
    float* dp = Lt + tri_index(lane, lane);
    float v = *dp;
    v += 1e-4f * v + 1.0f;
    *dp = v;
  }
  wave_lds_sync();
  const int row = min(lane, SN);
  const float* rp = Lt + ((row < SN) ? tri_index(row, 0) : TRI);
  cdv_float2 a2[SN / 2];     // the row as 30 float2 registers: rank-1 updates run two columns per v_pk_fma_f32
#pragma unroll
  for (int c = 0; c < SN; c++) a2[c >> 1][c & 1] = rp[c];
  if (A.dbg && (lane < n || lane == SN)) {            // damped S (both triangles) and y of iteration 0: a rolled loop over the
    const int cmax = lane < n ? lane : n - 1;         // row in LDS (unrolled over the registers it was 120 hoisted lane masks)
#pragma unroll 1
    for (int c = 0; c <= cmax; c++) {
      const float v = rp[c];
      if (lane < n) { A.dbg[(size_t)lane * n + c] = v; A.dbg[(size_t)c * n + lane] = v; }
      else A.dbg[(size_t)n * n + c] = v;
    }
  }
  wave_lds_sync();   // Lt is reused for L below
  CDV_IF_STAMPS(asm volatile("v_nop" :: "v"(a2[29][1] + a2[0][0]));)
  CDV_STAMP(baw, sslot, 2);
  // ---- right-looking Cholesky, column k broadcast through LDS one column ahead of its rank-1 update ----
  // (no pivot test inside the chain: a pivot that is not positive leaves a NaN on L's diagonal -- piv * rsq(piv) -- and is
  // found there afterwards.  Tested per column, the compiler kept all 60 pivots in spilled scalar registers and evaluated the
  // tests after the loop: 60 x (v_readlane of a spill lane, compare, or) at the END of the critical path.)
  float Lk;
  {
    const float piv = readlane_f(a2[0][0], 0);
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
      // L[k+1][k] straight from lane k + 1 (the LDS copy of column k only feeds the rest of the update: the chain
      // pivot -> scale -> next pivot never waits for an LDS round trip)
      float an = fmaf(-Lk, readlane_f(Lk, k + 1), a2[(k + 1) >> 1][(k + 1) & 1]);
      const float piv = readlane_f(an, k + 1);
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
  CDV_STAMP(baw, sslot, 3);
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
  const float dgk = Lt[kc * 68 + kc];
  const float invd = 1.0f / dgk;
  // the first pivot that was not positive (ba_cuda.cu:576,590 ignore cholesky_ex's info; here it is reported): block index + 1
  const unsigned long long badm = __ballot(lane < SN && !(dgk > 0.f));
  const int badk = badm ? (__ffsll((long long)badm) - 1) / 6 + 1 : 0;
  // back substitution L^T x = z: x_r = z_r / L[r][r] once every x_j, j > r, has been folded into z.  x_r is wave-uniform when
  // it is known (a scalar register) and goes into lane r of x with ONE v_writelane -- `x = (lane == r) ? xr : x` cost a
  // hoisted 64-bit lane mask per step, 60 of them spilled and read back with v_readlane at the end of the kernel's critical path
  // The sweep runs on the SCALED unknown zs_k = z_k / L[k][k] with columns scaled alike, so that a step's chain is
  // v_readlane -> v_fma (the multiply by 1 / L[k][k] sat between them: 50 cycles a step, 3.0k of the solver's 24k)
  float x = 0.f;
#pragma unroll
  for (int r = 0; r < SN; r++) col[r] *= invd;
  float zs = z * invd;
#pragma unroll
  for (int r = SN - 1; r >= 0; r--) {
    const float xr = readlane_f(zs, r);
    asm("v_writelane_b32 %0, %1, %2" : "+v"(x) : "s"(xr), "n"(r));   // (x_r is a data operand here, not the lane select: no hazard wait)
    zs = fmaf(-col[r], xr, zs);
  }
  CDV_STAMP(baw, sslot, 4);
  if (A.test == HO_TEST_STALL_AFTER) ho_test_stall();
  if (__builtin_amdgcn_readfirstlane(verdict) != HO_COMMITTED) {   // a retract workgroup gave up before the commit: all of them did
    if (lane == 0) ba_flag(A.info, BI_HANDOFF, 1);
    return;
  }
  if (lane < n) {
    // the data IS the flag: one 8-byte {tag = this launch's token, value} granule per unknown, written through; the retract workgroups
    // poll the tags of the granules they read (CDNA programming guide, Guideline 16, recipe R2)
    __hip_atomic_store(&A.granX[lane], ((uint64_t)(uint32_t)token << 32) | (uint64_t)(uint32_t)__float_as_int(x),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    A.dXg[lane] = x;
    if (A.dbg) A.dbg[(size_t)n * n + n + lane] = x;
  }
  if (lane == 0 && badk) ba_flag(A.info, BI_CHOL, badk);
  CDV_STAMP(baw, sslot, 5);
  CDV_STAMP_RT(baw, sslot, 15);
}

template <bool HAS_II, bool TABLE>
__global__ __launch_bounds__(64 * CKW) void ba_chunk_kernel(BaWinArgs A_in) {
  const BaWinArgs A = with_dyn(A_in);
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Sc = smem;                           // [CKW][SLAB] per-wave packed copies of [S | y] (B and v parts)
  float* Ed = Sc + CKW * SLAB;                // [64][EDL]   rows 0..59 E, row 60 u, rows 61..63 zero
  float* part = Ed + 64 * EDL;                // [CKW][8][CK] per-wave partial sums: 6 rows of E_i, C, u
  float* qs = part + CKW * 8 * CK;            // [CK]
  int* ixp = reinterpret_cast<int*>(qs + CK); // [CK] free-pose index of the patch's source frame (-1: fixed / none)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int32_t* __restrict__ gmeta = A.gmeta;
  const PatchSpan sp = patch_span<TABLE>(A);
  const int gerr = graph_error_of(gmeta, TABLE);
  const int U = sp.U;
  if (blockIdx.x == 0 && tid == 0) ba_begin_status(A.info, A.counters, A.first, gerr, U > A.U_max);
  // the hand-off words of the finish launch that follows (arrival flags, verdict, dX granules) lose their tags here: the
  // tags are per-launch tokens taken from host state when the call is ENQUEUED, so a captured hipGraph replays the same
  // tokens -- with the words reset by the launch in front, a replay never meets a tag of its own from the time before
  if (blockIdx.x == 0) {
    for (int i = tid; i < HAND_WORDS - 16; i += (int)blockDim.x) A.arrive[16 + i] = 0;
    for (int i = tid; i < 64; i += (int)blockDim.x) A.granX[i] = 0ull;
    if (tid == 0) A.arrive[HO_VERDICT] = HO_UNDECIDED;
  }
  if (gerr || U > A.U_max) return;
  const int N = A.N, t0 = A.t0, P = A.P;
  const int n6 = 6 * N;
  const int PP = P * P;
  const int centre = (P > 1) ? (P + 1) : 0;
  const int p = lane & 15, sub = lane >> 4;
  // (dealing the slots of a round to the waves cyclically -- slot = wave + CKW * sub -- so that the slots that carry work
  // spread over all waves was measured: 12.8 against 12.4 us; a pass of the pair products costs the same for one active
  // row as for four, so concentrating the active rows in few waves is the cheaper arrangement)
  const int so = sub;
  const int c16 = lane & 15, g4 = lane >> 4;
  float* Sw = Sc + wave * SLAB;
  // which of a frame pair's 90 sums this lane owns after the reductions: value 16 g + brev4(c16) of group g
  int codes[6];
  {
    const int br = ((c16 & 1) << 3) | ((c16 & 2) << 1) | ((c16 & 4) >> 1) | ((c16 & 8) >> 3);
#pragma unroll
    for (int g = 0; g < 6; g++) codes[g] = pair_code_rt(g, br);
  }
  const int n_chunks = (U + CK - 1) / CK;

  CDV_IF_STAMPS(const int sslot = (int)blockIdx.x * CKW + wave; unsigned long long t_fac = 0, t_ej = 0, t_xw = 0, t_rd = 0, t_mf = 0, t_em = 0, t_x;)
  CDV_STAMP(baw, sslot, 0);
  CDV_STAMP_RT(baw, sslot, 14);
  for (int chunk = (int)blockIdx.x; chunk < n_chunks; chunk += (int)gridDim.x) {
    const int r0 = chunk * CK;
    const int r = r0 + p;
    const bool live = r < U;
    // ---- level 1, every load unconditional on a clamped index and all of them issued together: the records of this
    // lane's first-round slot and of the patch's first edge from the chunk-slot copy (addresses that need no CSR
    // offset), the patch's CSR offsets and id
    const int step = 4 * CKW;
    int tb = 4 * wave;
    const bool use_ell = TABLE || chunk < A.ell_chunks;   // workgroup-uniform; false only beyond 65,536 unique patches
    // (no branch around these loads: a chunk without a chunk-slot copy reads CSR record 0 here and the real ones below)
    const int rs = live ? r : 0;
    const int4* cell = use_ell ? reinterpret_cast<const int4*>(A.pell) : reinterpret_cast<const int4*>(A.prec);
    int4 raw = cell[use_ell ? cell_index(rs, tb + so) : 0];
    int4 raw0 = cell[use_ell ? cell_index(rs, 0) : 0];
    const PatchRow row = patch_row<TABLE>(A, rs);
    // the wave-uniform inputs travel with level 1 as well (read here, not before the loop: nothing waits for them
    // before the loads above are out): intrinsics of row 0 (ba_cuda.cu:253-259), lambda, CSR record 0
    const float fx = A.intr[0], fy = A.intr[1], cx = A.intr[2], cy = A.intr[3];
    const float lm = A.lmbda[0];
    const EdgeRec safe = {A.prec[0], A.prec[1], A.prec[2]};   // stands in for slots that do not exist
    // zero the workgroup's accumulators (the previous chunk of a grid-stride loop is done with them: barrier below)
    {
      const cdv_float4 z4 = {0.f, 0.f, 0.f, 0.f};
      cdv_float4* s4 = reinterpret_cast<cdv_float4*>(Sc);
      for (int i = tid; i < CKW * SLAB / 4; i += 64 * CKW) s4[i] = z4;
      for (int i = tid; i < 64 * EDL; i += 64 * CKW) Ed[i] = 0.f;
    }
    const int plo = live ? row.plo : 0;
    const int deg = live ? row.deg : 0;
    const int64_t kxr = live ? row.id : 0;
    if (!use_ell) {   // beyond the chunk-slot copy: the CSR records, one round trip later
      const int4* csr = reinterpret_cast<const int4*>(A.prec);
      raw = csr[(tb + so < deg) ? plo + tb + so : 0];
      raw0 = csr[plo];
    }
    // a slot beyond the patch's degree holds anything: replaced by CSR record 0 (always valid) before use (selects)
    EdgeRec rec = settle_rec<HAS_II>(A, raw, tb + so < deg, safe);
    const EdgeRec rec0 = settle_rec<HAS_II>(A, raw0, deg > 0, safe);
    // ---- level 2: the patch centre (patch 0 for a lane without a patch), the first round's poses, target, weight
    const float* pk = A.patches + kxr * 3 * PP;
    const float px = pk[centre], py = pk[PP + centre], pd = pk[2 * PP + centre];
    const int ix_patch = deg > 0 ? rec0.ix : -1;
    EdgeIn in = load_in(A, rec);
    int maxdeg = deg;
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) maxdeg = max(maxdeg, __shfl_xor(maxdeg, o));   // over the 16 patches (all sub rows alike)
    maxdeg = __builtin_amdgcn_readfirstlane(maxdeg);
    const int a0 = ix_patch - t0;
    const int ixf_patch = (deg > 0 && a0 >= 0 && a0 < N) ? a0 : -1;
    lds_barrier();   // accumulators are zero (LDS only: the loads above stay in flight)
    const float pdu = pd;
    CDV_STAMP(baw, sslot, 1);

    float Cacc = 0.f, uacc = 0.f;
    float eiacc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (; tb < maxdeg; tb += step) {
      const bool active = (tb + so) < deg;
      const bool more = tb + step < maxdeg;      // wave-uniform; only patches with more than 4 CKW = 32 edges
      const EdgeRec cur = rec;
      int4 raw_nxt = {0, 0, 0, 0};
      if (more) raw_nxt = reinterpret_cast<const int4*>(A.prec)[(tb + step + so < deg) ? plo + tb + step + so : 0];
      CDV_IF_STAMPS(t_x = cdv_now();)
      EdgeFactor J;
      fastba_factor(in.pi, in.pj, px, py, pdu, in.tx, in.ty, in.wx, in.wy, fx, fy, cx, cy, J);
      CDV_IF_STAMPS(asm volatile("v_nop" :: "v"(J.Ji[11] + J.Jz[1])); { const unsigned long long t_y = cdv_now(); t_fac += t_y - t_x; t_x = t_y; })
      if (more) {   // rare: this wave's next round (its record was requested above)
        rec = settle_rec<HAS_II>(A, raw_nxt, tb + step + so < deg, safe);
        in = load_in(A, rec);
      }
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
      CDV_IF_STAMPS({ const unsigned long long t_y = cdv_now(); t_ej += t_y - t_x; t_x = t_y; })
      // ---- B and v: the 16 patches of a target slot (one DPP row) normally share their frame pair (i, j); every pass
      // takes, per row, the pair of its lowest unprocessed lane and sums the lanes of that pair (an irregular edge list
      // costs more passes, nothing else) ----
      const int key = (ixf + 1) * (N + 1) + (jxf + 1);
      unsigned long long todo = __ballot(active && key != 0);
      while (todo) {
        const unsigned rowbits = (unsigned)(todo >> (16 * sub)) & 0xffffu;
        const bool row_on = rowbits != 0;
        const int leader = 16 * sub + (row_on ? __ffs((int)rowbits) - 1 : 0);
        const int kcur = __shfl(key, leader);
        const int ci = __shfl(ixf, leader), cj = __shfl(jxf, leader);
        const bool match = active && row_on && key == kcur;
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
        CDV_IF_STAMPS({ asm volatile("v_nop" :: "v"(tot[0] + tot[5])); const unsigned long long t_y = cdv_now(); t_mf += t_y - t_x; t_x = t_y; })
#pragma unroll
        for (int g = 0; g < 6; g++) pair_emit(tot[g], codes[g], ci, cj, row_on, Sw);
        CDV_IF_STAMPS({ const unsigned long long t_y = cdv_now(); t_em += t_y - t_x; t_x = t_y; })
        todo &= ~__ballot(match);
      }
    }
    CDV_STAMP(baw, sslot, 2);
    CDV_STAMP_VAL(baw, sslot, 8, t_fac); CDV_STAMP_VAL(baw, sslot, 9, t_ej); CDV_STAMP_VAL(baw, sslot, 10, t_xw);
    CDV_STAMP_VAL(baw, sslot, 11, t_rd); CDV_STAMP_VAL(baw, sslot, 12, t_mf); CDV_STAMP_VAL(baw, sslot, 13, t_em);
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
    lds_barrier();
    // the wave copies of B, v summed in fixed order into copy 0 (each thread its own 16-byte columns)
    {
      cdv_float4* s4 = reinterpret_cast<cdv_float4*>(Sc);
      for (int i = tid; i < SLAB / 4; i += 64 * CKW) {
        cdv_float4 t = s4[i];
#pragma unroll
        for (int w = 1; w < CKW; w++) t += s4[w * (SLAB / 4) + i];
        s4[i] = t;
      }
    }
    if (tid < 6 * CK) {          // E_i rows of every patch: the four wave partials in fixed order, onto the E_j entries
      const int c = tid / CK, pp = tid - c * CK;
      float tot = 0.f;
#pragma unroll
      for (int w = 0; w < CKW; w++) tot += part[(w * 8 + c) * CK + pp];
      const int ib = ixp[pp];
      if (ib >= 0) Ed[(6 * ib + c) * EDL + pp] += tot;
    } else if (tid < 7 * CK) {   // C, u, q of every patch
      const int pp = tid - 6 * CK;
      float Ct = 0.f, ut = 0.f;
#pragma unroll
      for (int w = 0; w < CKW; w++) { Ct += part[(w * 8 + 6) * CK + pp]; ut += part[(w * 8 + 7) * CK + pp]; }
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
    CDV_STAMP(baw, sslot, 3);
    // ---- the chunk's E columns for the retraction (complete values, plain stores) ----
    for (int i = tid; i < n6 * CK; i += 64 * CKW) {
      const int row = i / CK, pp = i - row * CK;
      const float v = Ed[row * EDL + pp];
      A.Edg[(size_t)row * A.U_stride + r0 + pp] = v;
      if (A.dbg) A.dbg[(size_t)n6 * n6 + 2 * n6 + 3 * (size_t)A.U_stride + (size_t)row * A.U_stride + r0 + pp] = v;
    }
    // ---- Schur products of the chunk, [E; u] diag(q) [E; u]^T on the matrix cores (K = 16 patches), subtracted from
    // the combined copy of B, v: the chunk's partial system; each entry has exactly one owner lane, which stores it
    // (a pass of 16-byte stores behind one more barrier was measured: slower -- every wave then waits for the two that
    // own two tile pairs) ----
    CDV_STAMP(baw, sslot, 4);
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
        slab[idx] = Sc[idx] - acc[q];
      }
    }
    CDV_STAMP(baw, sslot, 5);
    __syncthreads();   // a grid-stride successor chunk re-zeroes the accumulators
  }
  CDV_STAMP(baw, sslot, 6);
  CDV_STAMP_RT(baw, sslot, 15);
}


// (waves_per_eu 1..2: one workgroup per CU matters here, not occupancy -- left to itself the scheduler sinks the solver's
// look-ahead broadcast reads to reach a smaller register bucket and the factorisation runs 50 % longer: 18.8k against 12.6k
// cycles, measured when the kernel dropped from 180 to 140 registers)
template <bool TABLE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void ba_finish_kernel(BaWinArgs A_in) {
  const BaWinArgs A = with_dyn(A_in);
  const int32_t* __restrict__ gmeta = A.gmeta;
  const PatchSpan sp = patch_span<TABLE>(A);
  const int U = sp.U;
  if (graph_error_of(gmeta, TABLE) || U > A.U_max) return;
  const int RW = (int)gridDim.x - 1;
  const int tid = threadIdx.x;
  if (blockIdx.x == 0) {
    if (tid < 64) solve_wave(A, RW);
    return;
  }
  const int b = (int)blockIdx.x - 1;
  CDV_IF_STAMPS(const int sslot = 4100 + b * 4 + (tid >> 6);)
  CDV_STAMP(baw, sslot, 0);
  CDV_STAMP_RT(baw, sslot, 14);
  reduce_slabs(A, b, RW, U, tid, true);
  // only the first RT workgroups go on to the retraction (256 patches each per pass); the others were here for the reduce
  const int RT = min(RW, max(1, (U + 255) / 256));
  if (b >= RT) return;
  CDV_STAMP(baw, sslot, 1);
  // ---- 2. before dX exists: everything of this thread's patch that does not depend on it ----
  __shared__ __attribute__((aligned(16))) float sdx[64];
  const int P = A.P, PP = P * P, N = A.N;
  int r = b * 256 + tid;
  bool livep = r < U;
  float ev[SN];
#pragma unroll
  for (int i = 0; i < SN; i++) ev[i] = (livep && i < 6 * N) ? A.Edg[(size_t)i * A.U_stride + r] : 0.f;
  float uv = 0.f, qv = 0.f, d0 = 0.f;
  float* pk = nullptr;
  if (livep) {
    const PatchRow row = patch_row<TABLE>(A, r);
    livep = row.deg > 0;             // a table id without an edge: not part of the graph, not retracted
    uv = A.ug[r]; qv = A.qg[r];
    pk = A.patches + row.id * 3 * PP + 2 * PP;
    d0 = pk[0];                      // the depth is read from pixel [0][0]   (ba_cuda.cu:218 semantics)
  }
  CDV_IF_STAMPS(asm volatile("v_nop" :: "v"(ev[0] + ev[59] + d0));)
  CDV_STAMP(baw, sslot, 2);
  // ---- 3. wait for the solver.  All-or-nothing (cdv_ba.h ho_decide): a workgroup whose patience runs out does not simply
  // leave -- others may already hold the solution -- it asks for the launch's verdict: if the solver has committed, dX WILL
  // come and the wait goes on; if nobody has decided yet, this workgroup decides ABANDONED, which the solver honours by
  // publishing nothing, so that every retract workgroup ends here ----
  __shared__ int s_ok;
  if (tid < 64) {   // wave 0: lane t polls the granule of unknown t until its tag shows up; the poll is the load of dX
    float xv = 0.f;
    bool ok = tid >= 6 * N;
    const int patience = A.test ? (1 << 10) : (1 << 21);
    bool all_ok = false;
    for (int round = 0; round < 2 && !all_ok; round++) {
      for (int spins = 0; spins < (round == 0 ? patience : (1 << 24)); spins++) {
        if (!ok) {
          const uint64_t g = __hip_atomic_load(&A.granX[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if ((uint32_t)(g >> 32) == (uint32_t)A.token) { xv = __int_as_float((int)(uint32_t)g); ok = true; }
        }
        if (__all(ok)) { all_ok = true; break; }
        if ((spins & 255) == 255 &&      // the solver may have said it will not publish (its own wait failed)
            __hip_atomic_load(&A.arrive[HO_VERDICT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == HO_ABANDONED) break;
        __builtin_amdgcn_s_sleep(2);
      }
      if (all_ok) break;
      int verdict = 0;
      if (tid == 0) verdict = ho_decide(&A.arrive[HO_VERDICT], HO_ABANDONED);
      if (__builtin_amdgcn_readfirstlane(verdict) != HO_COMMITTED) break;   // abandoned (by this workgroup or before): nobody applies it
    }
    if (!all_ok && tid == 0) ba_flag(A.info, BI_HANDOFF, 1);
    if (tid == 0) s_ok = all_ok ? 1 : 0;
    sdx[tid] = xv;
  }
  __syncthreads();
  CDV_STAMP(baw, sslot, 3);
  if (!s_ok) return;   // no update without a solution: poses and depths stay as they were
  // ---- 4. pose retraction T <- Exp(dX_i) T: the first retract workgroup's first N lanes ----
  if (b == 0 && tid < N) {
    float* p = A.poses + 7 * (size_t)(A.t0 + tid);
    const float* ps = p;
    float pose[7], xi[6];
#pragma unroll
    for (int c = 0; c < 7; c++) pose[c] = ps[c];
#pragma unroll
    for (int c = 0; c < 6; c++) xi[c] = sdx[6 * tid + c];
    se3_retract_raw(xi, pose);
#pragma unroll
    for (int c = 0; c < 7; c++) p[c] = pose[c];
  }
  // ---- 5. dZ = Q (u - E^T dX), inverse-depth update (ba_cuda.cu:592,209-229 semantics) ----
  for (;;) {
    if (livep) {
      float sacc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c4 = 0; c4 < SN / 4; c4++) {
        const cdv_float4 x4 = *reinterpret_cast<const cdv_float4*>(&sdx[4 * c4]);
#pragma unroll
        for (int j = 0; j < 4; j++) sacc[j] = __builtin_fmaf(ev[4 * c4 + j], x4[j], sacc[j]);
      }
      const float dz = __fmul_rn(qv, uv - ((sacc[0] + sacc[1]) + (sacc[2] + sacc[3])));
      if (A.dbg) A.dbg[(size_t)36 * N * N + 12 * N + r] = dz;
      float d = __fadd_rn(d0, dz);
      d = (d > 20.f) ? 1.0f : d;
      d = fmaxf(d, 1e-4f);
      store_depth(pk, PP, d);
    }
    // more patches than one pass of the retract workgroups covers: the next block of 256 (loaded now, dX is known)
    r += RT * 256;
    if (r - tid >= U) { CDV_STAMP(baw, sslot, 4); CDV_STAMP_RT(baw, sslot, 15); break; }          // workgroup-uniform
    livep = r < U;
#pragma unroll
    for (int i = 0; i < SN; i++) ev[i] = (livep && i < 6 * N) ? A.Edg[(size_t)i * A.U_stride + r] : 0.f;
    if (livep) {
      const PatchRow row = patch_row<TABLE>(A, r);
      livep = row.deg > 0;
      uv = A.ug[r]; qv = A.qg[r];
      pk = A.patches + row.id * 3 * PP + 2 * PP;
      d0 = pk[0];
    }
  }
}

}  // namespace

namespace {

hipError_t win_attrs() {
  static hipError_t attr_err = [] {
    hipError_t e = hipSuccess, x;
    const int lds = (int)(sizeof(float) * LDS_CHUNK_FLOATS);
#define CDV_ATTR(...) if ((x = hipFuncSetAttribute((const void*)ba_chunk_kernel<__VA_ARGS__>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)) != hipSuccess) e = x;
    CDV_ATTR(true, false) CDV_ATTR(false, false) CDV_ATTR(true, true) CDV_ATTR(false, true)
#undef CDV_ATTR
    return e;
  }();
  return attr_err;
}

void launch_chunk_win(const BaWinArgs& a, int grid, hipStream_t s) {
  const bool table = a.tab_cap > 0;
  const size_t lds = sizeof(float) * LDS_CHUNK_FLOATS;
  if (a.has_ii && table) hipLaunchKernelGGL((ba_chunk_kernel<true, true>), dim3(grid), dim3(64 * CKW), lds, s, a);
  else if (a.has_ii) hipLaunchKernelGGL((ba_chunk_kernel<true, false>), dim3(grid), dim3(64 * CKW), lds, s, a);
  else if (table) hipLaunchKernelGGL((ba_chunk_kernel<false, true>), dim3(grid), dim3(64 * CKW), lds, s, a);
  else hipLaunchKernelGGL((ba_chunk_kernel<false, false>), dim3(grid), dim3(64 * CKW), lds, s, a);
}

void launch_finish_win(const BaWinArgs& a, hipStream_t s) {
  // reduce workgroups: one per four 16-byte columns of a slab (119), at least one per 256 patches of capacity for the
  // retraction the first of them go on to, at most WIN_MAX_RW (their arrival flags)
  int RW = cdv_div_up(WIN_SLAB / 4, 4);
  const int rw_p = cdv_div_up(a.U_max, 256);
  RW = RW < rw_p ? rw_p : RW;
  RW = RW > WIN_MAX_RW ? WIN_MAX_RW : RW;
  if (a.tab_cap > 0) hipLaunchKernelGGL(ba_finish_kernel<true>, dim3(1 + RW), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(ba_finish_kernel<false>, dim3(1 + RW), dim3(256), 0, s, a);
}

}  // namespace

// one Gauss-Newton iteration as two launches: chunk systems, then reduce + solve + retract
int cdv::cdv_ba_window_iteration(const BaWinArgs& a, hipStream_t s) {
  CDV_HIP_CHECK(win_attrs());
  const int n_ck = a.n_ck_cap < WIN_MAX_GRID ? a.n_ck_cap : WIN_MAX_GRID;
  launch_chunk_win(a, n_ck, s);
  launch_finish_win(a, s);
  CDV_LAUNCH_CHECK();
  return CDV_OK;
}

