// cdv_ba.h -- internal: workspace layout and launch interface shared by ba.hip (dispatch, larger systems) and
// ba_win.hip (the optimisation-window path, N <= 10 free poses).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "cdv_common.h"
#include "cdv_graph.h"

namespace cdv {

constexpr int BA_CHUNK = 64;      // unique patches per workgroup of the N > 10 path (= lanes of a wave)
constexpr int BA_NMAX = 32;       // free poses supported by the single-workgroup solver
constexpr int BA_NBIG = 1024;     // free poses supported by the global-BA path (dense E in HBM, blocked Cholesky)
constexpr int CNB = 64;           // block size of the multi-workgroup Cholesky
constexpr int BIG_PP = 8;         // poses per panel of the global path's Schur products (48 rows; <= 128 panels: four mask words)
constexpr int BIG_MW = 4;         // mask words per chunk of 64 patches: which panels have a non-zero E block in it
constexpr int PDIAG = 48;         // floats per (frame pair, side) partial of a diagonal block: 36 (6 x 6) + 6 (v), padded

// ---- window path (N <= 10): no float atomics anywhere, bitwise reproducible ---------------------------------
constexpr int WIN_N = 10;                        // free poses
constexpr int WIN_SN = 6 * WIN_N;                // unknowns
constexpr int WIN_TRI = WIN_SN * (WIN_SN + 1) / 2;   // packed lower triangle of S
constexpr int WIN_SLAB = 1896;                   // floats per partial system: TRI + 60 (y), padded to a multiple of 8
constexpr int WIN_CK = 16;                       // unique patches per chunk workgroup
constexpr int WIN_MAX_GRID = 1024;               // chunk workgroups per launch (grid-stride over the chunks beyond)
constexpr int WIN_MAX_RW = 128;                  // reduce workgroups of the finish launch (the first few also retract)
constexpr int HAND_WORDS = 16 + WIN_MAX_RW;      // hand-off words: [HO_VERDICT] the launch's verdict, [16 + b] arrival flag of reduce workgroup b

// ---- the solve -> retract hand-off inside a finish launch is ALL-OR-NOTHING.  dX travels as tagged granules (the data is the
// flag); whether it travels at all is ONE word per launch, decided by whoever gets its compare-and-swap in first:
//   the solver, once it has nothing left to wait for: UNDECIDED -> COMMITTED  (it then publishes, always);
//   a retract workgroup whose patience has run out:    UNDECIDED -> ABANDONED  (the solver then publishes nothing).
// A retract workgroup that loses the swap to COMMITTED keeps waiting -- the solution is on its way; one that finds ABANDONED
// leaves.  So either every retract workgroup applies the update or none does, however the workgroups of the launch are
// delayed against each other (another stream, another process on the device).  The word is reset by the chunk launch in
// front (like the flags and the granules, so that a replayed hipGraph starts from UNDECIDED).
constexpr int HO_VERDICT = 2;                    // index into the hand-off words
enum { HO_UNDECIDED = 0, HO_COMMITTED = 1, HO_ABANDONED = 2 };
// test switch (cdv_ba_test_handoff): the solver stalls before / after its commit, the retract workgroups' patience is short
enum { HO_TEST_OFF = 0, HO_TEST_STALL_BEFORE = 1, HO_TEST_STALL_AFTER = 2,
       HO_TEST_FACTOR = 3 };   // global path: a block of the factorisation launch never raises its flag

// ---- the same design for 10 < N <= BA_NMAX free poses (ba_mid.hip): the slab is the packed triangle of a 6N x 6N system
constexpr int MID_N = BA_NMAX;
constexpr int MID_GRAN = 256;                    // dX granules of the solve -> retract hand-off (>= 6 MID_N)
constexpr int MID_MAX_RW = 64;                   // retract workgroups of the mid finish launch
inline int mid_tri(int N) { return 6 * N * (6 * N + 1) / 2; }
inline int mid_slab(int N) { return (mid_tri(N) + 6 * N + 7) / 8 * 8; }   // floats per partial system
// row stride of the dense system the mid solver works on: >= n, = 4 (mod 32) floats -- the 16-byte row reads and writes of
// 8 consecutive rows then cover the 32 LDS banks once
__host__ __device__ inline int solve_ld(int n) { return (n + 27) / 32 * 32 + 4; }

// status words of a workspace (int32 info[16] on the device; cdv_ba_status reads the first four)
enum { BI_CHOL = 0, BI_OVERFLOW = 1, BI_HANDOFF = 2, BI_GRAPH = 3 };

// control words of the factorisation launch (ba_factor.hip): 16 + one flag per block of the lower triangle and of the
// right-hand-side row + one per block row (its P item), rounded to 16 bytes
inline int fac_ctl_words(int nb) { return (16 + (nb + 1) * nb + nb + 3) / 4 * 4; }

struct BaLayout {
  size_t sy, C, u, Ed, cmask, zero_bytes, q, dX, info, Abig, xgran, fctl, ltg, slabs, ared, hand, pnext, ptab, pdiag, pkeys, pgraph, pgraph_bytes, total;
  int64_t E_max, pair_cap, pair_range;  // global path: edges the pair index is sized for, frame pairs it can hold, key range
  int64_t npad;                         // global-BA path: 6 N rounded up to the Cholesky block (0: not used)
  int64_t U_max, U_stride, sy_stride;   // sy_stride: floats between two copies of [S | y]
  int64_t n_ck;                         // window path: chunk slabs
  int N_max;
};

inline BaLayout ba_layout(int64_t U_max, int N_max, int64_t E_max = 1) {
  BaLayout L;
  L.U_max = U_max; L.N_max = N_max; L.E_max = E_max < 1 ? 1 : E_max;
  L.U_stride = (U_max + BA_CHUNK - 1) / BA_CHUNK * BA_CHUNK;
  const size_t n6 = 6 * (size_t)N_max;
  size_t o = 0;
  // accumulators of the N > 10 path: zeroed once, then kept zero by their consumers
  L.sy_stride = (int64_t)((n6 * n6 + n6 + 1023) / 1024 * 1024);
  L.sy = o;   o = align256(o + sizeof(float) * (size_t)L.sy_stride);   // [S | y] of the global path: one owner per entry
  L.C = o;    o = align256(o + sizeof(float) * (size_t)L.U_stride);
  L.u = o;    o = align256(o + sizeof(float) * (size_t)L.U_stride);
  L.Ed = o;   o = align256(o + sizeof(float) * n6 * (size_t)L.U_stride);
  L.cmask = o; o = align256(o + sizeof(uint32_t) * BIG_MW * (size_t)(L.U_stride / BA_CHUNK));   // active pose panels per chunk
  L.zero_bytes = o;
  L.q = o;    o = align256(o + sizeof(float) * (size_t)L.U_stride);
  L.dX = o;   o = align256(o + sizeof(float) * (n6 + 8));
  L.info = o; o = align256(o + sizeof(int32_t) * 16 + sizeof(uint64_t) * MID_GRAN);   // 16 words + the dX granules of the solve -> retract hand-off
  L.npad = 0; L.Abig = o; L.xgran = o; L.fctl = o; L.ltg = o;
  if (N_max > BA_NMAX) {   // working copy of [S ; y^T] for the blocked Cholesky, padded with identity to 64-blocks
    L.npad = (int64_t)((n6 + CNB - 1) / CNB * CNB);
    o = align256(o + sizeof(float) * (size_t)(L.npad + 1) * (size_t)L.npad);
    // the solution as {launch token, value} granules: the hand-off between the workgroups of the back-substitution launch
    L.xgran = o; o = align256(o + sizeof(uint64_t) * (size_t)L.npad);
    // the factorisation launch's ticket counter, abort word and one flag per 64 x 64 block (ba_factor.hip)
    L.fctl = o; o = align256(o + sizeof(int32_t) * (size_t)fac_ctl_words((int)(L.npad / CNB)));
    L.ltg = o;  o = align256(o + sizeof(float) * (size_t)L.npad * CNB);
  }
  // window path: one partial system per chunk of 16 patches, the reduced system, the arrival counter
  L.n_ck = 0; L.slabs = o; L.ared = o; L.hand = o; L.pnext = o;
  if (N_max >= 1 && N_max <= MID_N) {
    const size_t slab = N_max <= WIN_N ? (size_t)WIN_SLAB : (size_t)mid_slab(N_max);
    L.n_ck = (U_max + WIN_CK - 1) / WIN_CK;
    L.slabs = o; o = align256(o + sizeof(float) * slab * (size_t)L.n_ck);
    // the reduced system: packed (window path) or as dense rows [6N + 1][solve_ld] (mid path)
    const size_t ared = N_max <= WIN_N ? slab : (size_t)(6 * N_max + 1) * (size_t)solve_ld(6 * N_max);
    L.ared = o;  o = align256(o + sizeof(float) * ared);
    L.hand = o;  o = align256(o + sizeof(int32_t) * HAND_WORDS);
    L.pnext = o; o = align256(o + sizeof(float) * 7 * WIN_N);
  }
  // global path (N > 32): the frame-pair index of the edges -- an ordinary patch-graph index (cdv_graph.h) built over the
  // key (a, b) = the two poses of an edge as free-pose numbers + 1 (0: fixed), a <= b --, the pair table (a, b) -> pair
  // number + 1, and the per-pair partials of the diagonal blocks.  LAST in the layout: everything before it keeps its
  // offset whatever the number of edges.
  L.ptab = o; L.pdiag = o; L.pkeys = o; L.pgraph = o; L.pgraph_bytes = 0; L.pair_cap = 0; L.pair_range = 0;
  if (N_max > BA_NMAX) {
    const int64_t np1 = (int64_t)N_max + 1;
    L.pair_range = np1 * np1;
    const int64_t allp = np1 * (np1 + 1) / 2;
    L.pair_cap = L.E_max < allp ? L.E_max : allp;
    L.ptab = o;   o = align256(o + sizeof(int32_t) * (size_t)L.pair_range);
    // (the index first: its address -- the key of the index registry -- then depends on (U_max, N) only, and a call with
    // another number of edges is seen as a change of ITS layout, which re-initialises it)
    L.pgraph_bytes = graph_layout(L.E_max, L.pair_range).total;
    L.pgraph = o; o = align256(o + L.pgraph_bytes);
    L.pkeys = o;  o = align256(o + sizeof(int64_t) * (size_t)L.E_max);
    L.pdiag = o;  o = align256(o + sizeof(float) * 2 * PDIAG * (size_t)L.pair_cap);
  }
  L.total = o;
  return L;
}

struct BaWinArgs {
  float* poses;
  float* patches;
  const float *intr, *target, *weight, *lmbda;
  const int64_t* ii;                 // only read when the graph's CSR records carry no source frames
  int P, t0, N;
  const int32_t* gmeta;
  const int32_t *prec, *koff_u;
  const int32_t* pell;               // chunk-slot copy of the records (cdv_graph.h: ELL_SLOTS, ELL_CHUNKS)
  int ell_chunks;
  // patch TABLE instead of the ranked index (cdv_ba_pairs.h patch_span): pell = the table, prec = its overflow CSR, and
  // per slot its degree, overflow offset and patch id
  const int32_t *tdeg, *tplo, *tkid;
  int tab_cap;                       // > 0: the index is a patch table of this many slots (cdv_ba_pairs.h patch_span)
  const int64_t* kx;
  float *slabs, *ared;
  int32_t* arrive;                   // hand-off words (HAND_WORDS): token and arrival flags of the reduce workgroups
  uint64_t* granX;                   // dX granules {tag, value}
  float *Cg, *ug, *qg, *Edg, *dXg;
  int U_stride, U_max, n_ck_cap;
  int32_t* info;
  int32_t* counters;                 // optional host-visible event counters (cdv_ba_bind_status_counters), may be NULL
  float* dbg;                        // iteration-0 dump (see cdv_ba_forward), may be NULL
  int token;                         // tag of this launch's in-launch hand-offs (arrival flags, dX granules): never 0, new per launch
  int test;                          // HO_TEST_*: hand-off fault injection (tests only), 0 in production
  int ppf;                           // > 0: patches per frame, with a patch table whose capacity is a multiple of it and ppf % 4 == 0:
                                     // the 10 < N <= 32 path cuts its wide chunks per frame (ba_mid.hip wide_rows); 0: plain
  const int32_t* dyn;                // != NULL: t0 and N are dyn[CDV_DYN_T0], dyn[CDV_DYN_NFREE] (sizes on the device; N <= the N above)
  int first;                         // first iteration of a call: clears the sticky status words
  int has_ii;                        // the graph's records carry the source frames (it was built with ii)
};

// sizes on the device: the window [t0, t0 + N) of this update comes from the dynamic block (two scalar loads that travel
// with the first load level); the N of the arguments bounds it (it dimensioned the launches and the workspace)
__device__ __forceinline__ BaWinArgs with_dyn(const BaWinArgs& a) {
  BaWinArgs A = a;
  if (a.dyn) {
    A.t0 = a.dyn[CDV_DYN_T0];
    A.N = min(a.dyn[CDV_DYN_NFREE], a.N);
  }
  return A;
}

// the blocked Cholesky factorisation of the global path as one launch (ba_factor.hip): A [(npad + 1)][npad] in place; ctl
// (fac_ctl_words(npad / CNB) words) must be zero when the launch starts; ltg: npad * CNB floats of scratch (the diagonal blocks'
// factors in the layout the item workgroups solve against)
int cdv_ba_big_factor(float* A, int npad, int32_t* ctl, float* ltg, const int32_t* gmeta, int32_t* info, int test, hipStream_t s);
int cdv_ba_big_factor_items(int nb);

// one Gauss-Newton iteration of the window path: two launches on `s`
int cdv_ba_window_iteration(const BaWinArgs& a, hipStream_t s);
// the same for 10 < N <= MID_N (ba_mid.hip)
int cdv_ba_mid_iteration(const BaWinArgs& a, hipStream_t s);
// wide chunks (workgroups, slabs) one frame of ppf patches is cut into on that path
int cdv_ba_mid_wide_per_frame(int ppf);

// LDS hand-off between the lanes of ONE wave: the LDS unit executes a wave's DS instructions in order, so only the
// compiler has to be kept from moving accesses across this point.  (A workgroup-scope release fence would also drain
// vmcnt, i.e. wait for the wave's outstanding global memory operations: ~3000 cycles each time.)
__device__ __forceinline__ void wave_lds_sync() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

// workgroup barrier over LDS only: waits for this wave's LDS operations, NOT for its outstanding global loads (which
// __syncthreads() would drain: the next level of a dependent load chain then starts a memory round trip late)
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// try to decide the launch's verdict; returns the verdict that holds afterwards (one lane calls it)
__device__ __forceinline__ int ho_decide(int32_t* word, int want) {
  int expected = HO_UNDECIDED;
  __hip_atomic_compare_exchange_strong(word, &expected, want, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return expected == HO_UNDECIDED ? want : expected;
}

// fault injection: ~30 ms of sleep (tests only)
__device__ __forceinline__ void ho_test_stall() {
  for (int i = 0; i < (1 << 13); i++) __builtin_amdgcn_s_sleep(127);
}

__device__ __forceinline__ float readlane_f(float v, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

// The first launch of every iteration publishes the (optional) host-visible event counters of the workspace in the
// info block, so that every later kernel can count a failure without another argument.
__device__ __forceinline__ void ba_publish_counters(int32_t* info, int32_t* counters) {
  *reinterpret_cast<int32_t**>(info + 8) = counters;
}

// a failure event: the sticky word of this call and, when bound, the host-visible counter
__device__ __forceinline__ void ba_flag(int32_t* info, int which, int value) {
  info[which] = value;
  int32_t* counters = *reinterpret_cast<int32_t* const*>(info + 8);
  if (counters) {   // host-visible memory: a system-scope load and store (no PCIe atomic needed; events are rare and a lost
                    // increment between two simultaneous reporters only under-counts)
    const int32_t c = __hip_atomic_load(&counters[which], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&counters[which], c + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// the P x P inverse-depth pixels of a patch, all set to d (ba_cuda.cu:223-227): 36 contiguous bytes on a 4-byte boundary when
// P = 3 -- two 16-byte stores and one 4-byte store instead of nine 4-byte ones (global memory takes unaligned 16-byte accesses)
__device__ __forceinline__ void store_depth(float* __restrict__ pk, int PP, float d) {
  if (PP == 9) {
    typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
    const f4u v = {d, d, d, d};
    *reinterpret_cast<f4u*>(pk) = v;
    *reinterpret_cast<f4u*>(pk + 4) = v;
    pk[8] = d;
  } else {
    for (int a = 0; a < PP; a++) pk[a] = d;
  }
}

// status handling of the first launch of an iteration (one thread): sticky words cleared by the first iteration of a
// call; overflow / graph error rewritten by every iteration, so a workspace recovers on the next well-sized call
__device__ __forceinline__ void ba_begin_status(int32_t* info, int32_t* counters, int first, int gerr, bool overflow) {
  ba_publish_counters(info, counters);
  if (first) { info[BI_CHOL] = 0; info[BI_HANDOFF] = 0; }
  info[BI_GRAPH] = 0;
  info[BI_OVERFLOW] = 0;
  if (gerr) ba_flag(info, BI_GRAPH, 1);
  else if (overflow) ba_flag(info, BI_OVERFLOW, 1);
}

}  // namespace cdv
