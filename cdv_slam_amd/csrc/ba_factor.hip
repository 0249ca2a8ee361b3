// ba_factor.hip -- the Cholesky factorisation of the global bundle adjustment's reduced system (32 < N <= 1024 free poses,
// the dense `torch.linalg.cholesky` of 6N unknowns behind ba_cuda.cu:567-594 / slam.py:460-478) as ONE launch.
//
// The matrix A [(npad + 1)][npad] (ba.hip ba_big_fold_kernel: S with the damping of ba_cuda.cu:589, identity on the padded
// diagonal, row npad = y^T) is cut into 64 x 64 blocks; the result -- L in the lower blocks, z = L^-1 y in the last row --
// is what ba_big_backsolve_kernel reads.  Rounds 1-3 ran one launch per block column (28 x 26 us at N = 299: a one-wave
// panel chain behind a launch boundary each).  Here a block is computed LEFT-LOOKING,
//
//     block (r, c) = ( A(r, c) - sum_{k < c} L(r, k) L(c, k)^T ) L(c, c)^-T         (the right-hand side: r = nb, one row)
//
// with the sum kept in the matrix cores' accumulators, the inputs taken from whoever produced them as soon as their flags
// are up (the hand-off of the programming guide: write-through stores, one flag per block, every load of a handed-off byte
// past the L1), and nothing written but finished blocks.
//
// Two kinds of workgroup share the launch:
//  * ONE chain workgroup (whoever draws ticket 0) walks down the diagonal and never hands the critical path to anybody:
//    stage c = wave 0 factors the diagonal block (c, c) (the register scheme of ba_win.hip's solver) and leaves every finished
//    column in a column store in LDS; waves 1, 2 solve the lower neighbour (c + 1, c) against it column by column BEHIND the
//    factorisation (solve_cols2) and end a few columns after it; the helper waves 4 .. 7 take the neighbour's product off the
//    next diagonal block (matrix cores) -- L(c, c) and L(c + 1, c) never leave the LDS on their way to the next stage.  Underneath
//    the factorisation the helpers prepare the next stage's two blocks (fetch what an item workgroup pre-accumulated, apply the
//    one product that was still missing); helper 0 writes L(c, c) out -- as the column store holds it, flag right behind --,
//    helper 1 L(c + 1, c).  The waves of this workgroup meet through counters in LDS, not through s_barrier: a wave that waits
//    for its stores to drain holds nobody up.
//  * item workgroups draw the other tickets: O items -- a block (r, c), r >= c + 2, or the right-hand side's block of column
//    c: all its products, then the solve against L(c, c) (solve_cols4, against the published column store); P items -- the two
//    blocks of block row c that the chain will take over (its neighbour block (c, c - 1) and the diagonal block), with the
//    products of the columns k <= c - 3.
//    Tickets are dealt in an order in which an item only needs items with smaller tickets and chain stages that, in turn,
//    only need such items: the lowest unfinished ticket is always held by a running workgroup whose inputs will arrive, so the
//    launch finishes whatever the number of resident workgroups and the order they start in.
#include <hip/hip_runtime.h>

#include "cdv_ba.h"

CDV_STAMP_TU(baf)
// (only the waves with a slot stamp: hundreds of waves storing to one dummy slot queue up behind each other for ~100 us)
#define FSTAMP(id) do { CDV_IF_STAMPS(if (sslot >= 0) { CDV_STAMP(baf, sslot, id); }) } while (0)
#define FSTAMP_RT(id) do { CDV_IF_STAMPS(if (sslot >= 0) { CDV_STAMP_RT(baf, sslot, id); }) } while (0)

namespace cdv {
namespace {

constexpr int FT = 512;                 // threads of a workgroup: 8 waves, two 16 x 16 tiles of a block each
constexpr int FLD = CNB + 4;            // LDS row stride of a block (16-byte aligned rows, 8 rows cover the banks once)
constexpr int FBUF = CNB * FLD;         // floats of a block in LDS
constexpr int AUX_SC1 = 16;             // buffer load / store past the L1, write-through

typedef float f2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

struct FacArgs {
  float* A;
  int npad, nb, total;     // total: tickets (1 + items)
  int32_t* ctl;            // [0] ticket counter, [1] abort word, [16 + r * nb + c] flag of block (r, c) (r = nb: right-hand side),
                           // [16 + (nb + 1) * nb + c] flag of the P item of block row c
  float* Ltg;              // [nb][64][64]: L(c, c) as the chain workgroup's column store held it (column k at [64 k + row]): what the
                           // item workgroups solve against
  const int32_t* gmeta;
  int32_t* info;
  int test;
};

// ticket (>= 1) -> item.  Order: for g = 0 .. nb - 1: O(g) = the blocks (r, g), r = g + 2 .. nb - 1, and the right-hand side's
// block of column g; then P(g + 3).  What an item needs:
//   (r, c) of O(c): L(r, k), k < c (items of O(k)); L(c, k), k <= c - 2 (O(k)); L(c, c - 1) and L(c, c) (chain stages c - 1, c);
//   P(c):           L(c, k), L(c - 1, k), k <= c - 3 (items of O(k), k <= c - 3);
//   chain stage c (factor (c, c), solve (c + 1, c)): P(c + 1), the block (c + 1, c - 1) of O(c - 1), chain stage c - 1:
// smaller tickets, or chain stages that need nothing but smaller tickets.
__host__ __device__ __forceinline__ bool fac_decode(int t, int nb, int& kind, int& c, int& r) {
  t -= 1;
  for (int g = 0; g < nb; g++) {
    const int nrows = nb - g - 2 > 0 ? nb - g - 2 : 0;
    if (t < nrows + 1) { kind = 1; c = g; r = t < nrows ? g + 2 + t : nb; return true; }
    t -= nrows + 1;
    if (g + 3 < nb) {
      if (t == 0) { kind = 2; c = g + 3; r = g + 3; return true; }
      t -= 1;
    }
  }
  return false;
}

__device__ __forceinline__ int ld_flag(const int32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// one wave: wait until both flags are up (f1 may equal f0).  false: gave up (the abort word is set: by us on a timeout, or by
// another workgroup) -- uniform over the wave.
__device__ __forceinline__ bool fac_wait(const int32_t* f0, const int32_t* f1, int32_t* ctl, int32_t* info, int lane, int test) {
  const int limit = test == HO_TEST_FACTOR ? (1 << 11) : (1 << 21);
  const int32_t* p = lane == 0 ? f0 : lane == 1 ? f1 : ctl + 1;
  for (int spins = 0; spins < limit; spins++) {
    const int v = lane < 3 ? ld_flag(p) : 1;
    const bool up = lane == 2 ? true : v != 0;
    const bool ab = lane == 2 && v != 0;
    if (__any(ab)) return false;
    if (__all(up)) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // no instruction: keeps the compiler from moving the loads above the poll
      return true;
    }
    __builtin_amdgcn_s_sleep(2);
  }
  if (lane == 0) {
    __hip_atomic_store(ctl + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ba_flag(info, BI_HANDOFF, 1);
  }
  return false;
}

// ---- the waves of the chain workgroup meet through words in LDS.  The LDS unit executes a wave's instructions in order and
// holds the only copy of the data: a wave that has read a word's new value reads everything its writer wrote before it.
// Only the compiler has to be kept from moving accesses across.
enum { SY_L = 0, SY_X = 1, SY_S = 2, SY_U = 3, SY_PUB1 = 4, SY_PUB2 = 5, SY_ABORT = 6, SY_F = 7 };
__device__ __forceinline__ int lds_get(int* sy, int word) {
  return __hip_atomic_load(&sy[word], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_set(int* sy, int word, int value) {
  __hip_atomic_store(&sy[word], value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ bool lds_wait(int* sy, int word, int target) {
  for (int spins = 0; spins < (1 << 24); spins++) {
    if (lds_get(sy, word) >= target) { asm volatile("" ::: "memory"); return true; }
    if (lds_get(sy, SY_ABORT)) return false;
    __builtin_amdgcn_s_sleep(1);
  }
  lds_set(sy, SY_ABORT, 1);
  return false;
}
__device__ __forceinline__ void lds_post(int* sy, int word, int value) {   // one writer per word
  asm volatile("" ::: "memory");
  lds_set(sy, word, value);
}
__device__ __forceinline__ void lds_arrive(int* sy, int word, int lane) {           // several writers: a counter
  asm volatile("" ::: "memory");
  if (lane == 0) __hip_atomic_fetch_add(&sy[word], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

__device__ __forceinline__ cdv_float4 ld4(__amdgpu_buffer_rsrc_t rs, size_t elem) {
  return __builtin_bit_cast(cdv_float4, (i32x4)__builtin_amdgcn_raw_buffer_load_b128(rs, (int)(unsigned)(elem * 4u), 0, AUX_SC1));
}
__device__ __forceinline__ float ld1(__amdgpu_buffer_rsrc_t rs, size_t elem) {
  return __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)(unsigned)(elem * 4u), 0, AUX_SC1));
}
__device__ __forceinline__ void st4(__amdgpu_buffer_rsrc_t rs, size_t elem, cdv_float4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, v), rs, (int)(unsigned)(elem * 4u), 0, AUX_SC1);
}

// the tiles (ti, 0 .. NT - 1) of X Y^T for two blocks in LDS, K = 64, added to acc: lane (c16, g4) holds rows 16 ti + 4 g4 + q,
// column 16 tj + c16
template <int NT>
__device__ __forceinline__ void tiles_row(const float* X, const float* Y, int ti, int c16, int g4, cdv_float4 (&acc)[NT]) {
  const float* pa = X + (size_t)(16 * ti + c16) * FLD + 16 * g4;
  const float* pb = Y + (size_t)c16 * FLD + 16 * g4;
#pragma unroll
  for (int s4 = 0; s4 < 4; s4++) {
    const cdv_float4 av = *reinterpret_cast<const cdv_float4*>(pa + 4 * s4);
#pragma unroll
    for (int tj = 0; tj < NT; tj++) {
      const cdv_float4 bv = *reinterpret_cast<const cdv_float4*>(pb + (size_t)16 * tj * FLD + 4 * s4);
#pragma unroll
      for (int j = 0; j < 4; j++) acc[tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], bv[j], acc[tj], 0, 0, 0);
    }
  }
}

// the two tiles (ti, tj0), (ti, tj0 + 1) of X Y^T for two blocks in LDS, K = 64, added to acc[0], acc[1]: lane (c16, g4) holds
// rows 16 ti + 4 g4 + q, column 16 tj + c16 (lane group g4 takes the K slots 16 g4 .. 16 g4 + 15 of both operands)
__device__ __forceinline__ void tiles_xyt(const float* X, const float* Y, int ti, int tj0, int c16, int g4, cdv_float4 (&acc)[2]) {
  const float* pa = X + (size_t)(16 * ti + c16) * FLD + 16 * g4;
  const float* pb0 = Y + (size_t)(16 * tj0 + c16) * FLD + 16 * g4;
  const float* pb1 = pb0 + 16 * FLD;
#pragma unroll
  for (int s4 = 0; s4 < 4; s4++) {
    const cdv_float4 av = *reinterpret_cast<const cdv_float4*>(pa + 4 * s4);
    const cdv_float4 b0 = *reinterpret_cast<const cdv_float4*>(pb0 + 4 * s4);
    const cdv_float4 b1 = *reinterpret_cast<const cdv_float4*>(pb1 + 4 * s4);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], b0[j], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], b1[j], acc[1], 0, 0, 0);
    }
  }
}

// one wave, lane = row: the 64 x 64 block in Db (row-major, stride FLD; only its lower triangle matters) -> its Cholesky
// factor, row `lane` returned in a2 (pairs of columns; entries right of the diagonal are not part of it).  Right-looking in
// the wave's registers, column k + 1 broadcast through LDS while column k's rank-1 update runs (ba_win.hip's solver).  A pivot
// that is not positive leaves a NaN or an infinity on the diagonal below it: the caller looks there.
// (Measured and dropped: the same by four waves, 16 columns each, the waves behind the panel wave taking every finished column
// off their own columns as it appears in LDS: 23,800 cycles against 16,500 -- a column costs ~260 cycles here and ~280 there, both
// times the chain pivot -> v_readlane -> scale -> v_readlane -> FMA, not the rank-1 FMAs that the other waves take over; updating
// two or four columns ahead through v_readlane and applying the LDS read-backs three steps late changed nothing either.)
__device__ __forceinline__ void factor_rows(const float* Db, float* colb, float* Lt, int* sy, int base, int lane, f2 (&a2)[CNB / 2]) {
#pragma unroll
  for (int c4 = 0; c4 < CNB / 4; c4++) {
    const cdv_float4 q = *reinterpret_cast<const cdv_float4*>(&Db[lane * FLD + 4 * c4]);
    a2[2 * c4] = f2{q[0], q[1]};
    a2[2 * c4 + 1] = f2{q[2], q[3]};
  }
  float Lk;
  {
    const float piv = readlane_f(a2[0][0], 0);
    Lk = a2[0][0] * __builtin_amdgcn_rsqf(piv);
    a2[0][0] = Lk;
    colb[lane] = Lk;
    Lt[lane] = Lk;             // every column also into the column store, the progress word behind it (solve_cols2 follows)
    lds_set(sy, SY_F, base + 1);
  }
  f2 bcur[CNB / 2], bnxt[CNB / 2];
#pragma unroll
  for (int c4 = 0; c4 < CNB / 4; c4++) {
    const cdv_float4 v = *reinterpret_cast<const cdv_float4*>(&colb[4 * c4]);
    bcur[2 * c4] = f2{v[0], v[1]};
    bcur[2 * c4 + 1] = f2{v[2], v[3]};
  }
#pragma unroll
  for (int k = 0; k < CNB; k++) {
    float Ln = 0.f;
    if (k + 1 < CNB) {
      const float an = fmaf(-Lk, readlane_f(Lk, k + 1), a2[(k + 1) >> 1][(k + 1) & 1]);
      const float piv = readlane_f(an, k + 1);
      Ln = an * __builtin_amdgcn_rsqf(piv);
      a2[(k + 1) >> 1][(k + 1) & 1] = Ln;
      colb[lane] = Ln;      // in-order LDS: the reads of column k were issued before this write
      Lt[(k + 1) * CNB + lane] = Ln;
      lds_set(sy, SY_F, base + k + 2);
#pragma unroll
      for (int c4 = (k + 2) / 4; c4 < CNB / 4; c4++) {
        const cdv_float4 v = *reinterpret_cast<const cdv_float4*>(&colb[4 * c4]);
        bnxt[2 * c4] = f2{v[0], v[1]};
        bnxt[2 * c4 + 1] = f2{v[2], v[3]};
      }
    }
    if (((k + 2) & 1) && k + 2 < CNB)
      a2[(k + 2) >> 1][1] = fmaf(-Lk, bcur[(k + 2) >> 1][1], a2[(k + 2) >> 1][1]);
    const f2 nLk = {-Lk, -Lk};
#pragma unroll
    for (int pp = (k + 3) >> 1; pp < CNB / 2; pp++) a2[pp] = __builtin_elementwise_fma(nLk, bcur[pp], a2[pp]);
    Lk = Ln;
#pragma unroll
    for (int pp = (k + 2) >> 1; pp < CNB / 2; pp++) bcur[pp] = bnxt[pp];
  }
}


// X L^T = S for 32 rows, COLUMN BY COLUMN behind the factorisation: step k needs column k of L -- what factor_rows has just put
// into the column store -- so the solve of the neighbour block runs underneath the factorisation of the diagonal block and ends a
// step after it.  Two lanes per row (lane = 2 r + g): lane g holds the row's columns 8 m + 4 g .. + 3, m = 0 .. 7; step k: the
// owner's x_k (times 1 / L[k][k]) goes to its partner by a quad permute, both take x_k L[c][k] off their columns c > k.  What the
// column store holds above the diagonal is never multiplied (compile-time masks per step, one lane test).  false: gave up.
__device__ __forceinline__ bool solve_cols2(const float* Sb, const float* Lt, int* sy, int base, float* Xb, int R0, int lane) {
  const int r = lane >> 1, g = lane & 1;
  cdv_float4 xs[8];
#pragma unroll
  for (int m = 0; m < 8; m++) xs[m] = *reinterpret_cast<const cdv_float4*>(&Sb[(R0 + r) * FLD + 8 * m + 4 * g]);
  const bool g0 = g == 0;
  bool ok = true;
  cdv_float4 cur[8], nxt[8];
  // (the lane's offset into a column behind an empty asm: otherwise the addresses below are loop-invariant for the caller's stage
  // loop, get hoisted out of it as ~100 registers, and the solve spills)
  int go = 4 * g;
  asm volatile("" : "+v"(go));
  const float* Lg = Lt + go;
  // (the progress word is looked at once per eight columns: a test per column leaves the unrolled code -- xs is indexed by k, so it
  // has to be unrolled -- with 64 spin loops in its control flow and the register allocator with 470 spills.  A column's pieces are
  // requested a step ahead; the scheduling barrier keeps the compiler from requesting all eight steps' at once, which spills too.)
#pragma clang loop unroll(full)
  for (int k = 0; k < CNB; k++) {
    const int mk = k >> 3, gk = (k >> 2) & 1, ek = k & 3;
    if ((k & 7) == 0) {
      // (every x pinned to a register here: the compiler otherwise sinks the updates of the columns further right across the spin
      // loops down to where those columns are next read, and keeps eight steps' pieces of L alive for each of them -- 430 spills)
#pragma unroll
      for (int m = 0; m < 8; m++) asm volatile("" : "+v"(xs[m][0]), "+v"(xs[m][1]), "+v"(xs[m][2]), "+v"(xs[m][3]));
      if (ok) {
        int spins = 0, have;
        do {
          have = lds_get(sy, SY_F) - base;
        } while (have < k + 8 && !lds_get(sy, SY_ABORT) && ++spins < (1 << 22));
        ok = have >= k + 8;
      }
      asm volatile("" ::: "memory");
#pragma unroll
      for (int m = mk; m < 8; m++) cur[m] = *reinterpret_cast<const cdv_float4*>(&Lg[k * CNB + 8 * m]);
    }
    if (((k + 1) & 7) != 0) {
#pragma unroll
      for (int m = (k + 1) >> 3; m < 8; m++) nxt[m] = *reinterpret_cast<const cdv_float4*>(&Lg[(k + 1) * CNB + 8 * m]);
    }
    // x_k and the diagonal L[k][k] sit in the owner's pieces: both to its partner by the same quad permute
    const float xraw = xs[mk][ek], draw = cur[mk][ek];
    float xb, db;
    if (gk) {
      xb = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(xraw), 0xF5, 0xf, 0xf, false));   // quad_perm:[1,1,3,3]
      db = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(draw), 0xF5, 0xf, 0xf, false));
    } else {
      xb = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(xraw), 0xA0, 0xf, 0xf, false));   // quad_perm:[0,0,2,2]
      db = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(draw), 0xA0, 0xf, 0xf, false));
    }
    const float xk = __builtin_amdgcn_rcpf(db) * xb;
#pragma unroll
    for (int m = mk; m < 8; m++) {
      const cdv_float4 l = cur[m];
      if (m > mk) {
        const f2 nx = {-xk, -xk};
        const f2 lo = __builtin_elementwise_fma(nx, f2{l[0], l[1]}, f2{xs[m][0], xs[m][1]});
        const f2 hi = __builtin_elementwise_fma(nx, f2{l[2], l[3]}, f2{xs[m][2], xs[m][3]});
        xs[m] = cdv_float4{lo[0], lo[1], hi[0], hi[1]};
      } else {
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const bool c0 = 8 * mk + j > k, c1 = 8 * mk + 4 + j > k;   // lane g = 0 / g = 1: is this column right of k?
          const float u = fmaf(-xk, l[j], xs[m][j]);
          if (c0 && c1) xs[m][j] = u;
          else if (c0) xs[m][j] = g0 ? u : xs[m][j];
          else if (c1) xs[m][j] = g0 ? xs[m][j] : u;
        }
      }
    }
    xs[mk][ek] = ((gk == 0) == g0) ? xk : xs[mk][ek];
#pragma unroll
    for (int m = 0; m < 8; m++) cur[m] = nxt[m];
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int m = 0; m < 8; m++) *reinterpret_cast<cdv_float4*>(&Xb[(R0 + r) * FLD + 8 * m + 4 * g]) = xs[m];
  if (!ok) lds_set(sy, SY_ABORT, 1);
  return ok;
}

// The same column by column for an item workgroup (all of L(c, c) is there: no progress word): 16 rows per wave, FOUR lanes per row
// (lane = 4 r + g): lane g holds the row's columns 16 m + 4 g .. + 3, m = 0 .. 3.  Lt: the column store as the chain workgroup
// published it (column k at Lt[64 k + row]; what lies above the diagonal is never multiplied).
__device__ __forceinline__ void solve_cols4(const float* Sb, const float* Lt, float* Xb, int R0, int lane) {
  const int r = lane >> 2, g = lane & 3;
  cdv_float4 xs[4];
#pragma unroll
  for (int m = 0; m < 4; m++) xs[m] = *reinterpret_cast<const cdv_float4*>(&Sb[(R0 + r) * FLD + 16 * m + 4 * g]);
  int go = 4 * g;
  asm volatile("" : "+v"(go));   // (see solve_cols2)
  const float* Lg = Lt + go;
  cdv_float4 cur[4], nxt[4];
#pragma unroll
  for (int m = 0; m < 4; m++) cur[m] = *reinterpret_cast<const cdv_float4*>(&Lg[16 * m]);
#pragma clang loop unroll(full)
  for (int k = 0; k < CNB; k++) {
    const int mk = k >> 4, gk = (k >> 2) & 3, ek = k & 3;
    if (k + 1 < CNB) {
#pragma unroll
      for (int m = (k + 1) >> 4; m < 4; m++) nxt[m] = *reinterpret_cast<const cdv_float4*>(&Lg[(k + 1) * CNB + 16 * m]);
    }
    const float xraw = xs[mk][ek], draw = cur[mk][ek];
    float xb, db;   // x_k and L[k][k] from their owner to the four lanes of the row
    if (gk == 0) {
      xb = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(xraw), 0x00, 0xf, 0xf, false));
      db = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(draw), 0x00, 0xf, 0xf, false));
    } else if (gk == 1) {
      xb = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(xraw), 0x55, 0xf, 0xf, false));
      db = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(draw), 0x55, 0xf, 0xf, false));
    } else if (gk == 2) {
      xb = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(xraw), 0xAA, 0xf, 0xf, false));
      db = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(draw), 0xAA, 0xf, 0xf, false));
    } else {
      xb = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(xraw), 0xFF, 0xf, 0xf, false));
      db = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(draw), 0xFF, 0xf, 0xf, false));
    }
    const float xk = __builtin_amdgcn_rcpf(db) * xb;
#pragma unroll
    for (int m = mk; m < 4; m++) {
      const cdv_float4 l = cur[m];
      if (m > mk) {
        const f2 nx = {-xk, -xk};
        const f2 lo = __builtin_elementwise_fma(nx, f2{l[0], l[1]}, f2{xs[m][0], xs[m][1]});
        const f2 hi = __builtin_elementwise_fma(nx, f2{l[2], l[3]}, f2{xs[m][2], xs[m][3]});
        xs[m] = cdv_float4{lo[0], lo[1], hi[0], hi[1]};
      } else {
        // columns 16 mk + 4 g + j: right of k for the lanes g > gk, and for g == gk where j > ek
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const float u = fmaf(-xk, l[j], xs[m][j]);
          const bool take = j > ek ? g >= gk : g > gk;
          xs[m][j] = take ? u : xs[m][j];
        }
      }
    }
    xs[mk][ek] = (g == gk) ? xk : xs[mk][ek];
#pragma unroll
    for (int m = 0; m < 4; m++) cur[m] = nxt[m];
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int m = 0; m < 4; m++) *reinterpret_cast<cdv_float4*>(&Xb[(R0 + r) * FLD + 16 * m + 4 * g]) = xs[m];
}

constexpr int NSOLVE = 4;              // item workgroups: waves 0 .. 3 solve (16 rows each, solve_cols4)
constexpr int NTW = 2;                 // chain workgroup: waves 1, 2 solve the neighbour behind the factorisation (32 rows each, solve_cols2)
constexpr int NHELP = 4;               // its helper waves 4 .. 7: helper h prepares the tile row h of the next stage's blocks

__global__ __launch_bounds__(FT) void ba_big_factor_kernel(FacArgs P) {
  if (P.gmeta[GM_ERROR] || P.info[BI_OVERFLOW]) return;
  // six block buffers and the column store: 118 KB of the CU's 160 -- ONE workgroup per CU (the form of the hand-off used here was measured that way)
  __shared__ __attribute__((aligned(16))) float BB[6 * FBUF];
  __shared__ __attribute__((aligned(16))) float Lt[CNB * CNB];   // chain workgroup: the column store (column k of L(c, c) at Lt[64 k + row])
  __shared__ __attribute__((aligned(16))) float colb[CNB];
  float* const B0 = BB;
  float* const B1 = BB + FBUF;
  float* const B2 = BB + 2 * FBUF;
  float* const B3 = BB + 3 * FBUF;
  __shared__ int sh[8];    // [0] kind, [1] c, [2] r, [3] go on, [4], [5] wait verdicts of even / odd k, [6] of the diagonal block
  __shared__ int sy[8];    // chain workgroup: SY_* words
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, c16 = lane & 15, g4 = lane >> 4;
  const int nb = P.nb, npad = P.npad;
  const size_t lda = (size_t)npad;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      P.A, (short)0, (int)(unsigned)((size_t)(npad + 1) * lda * sizeof(float)), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc(
      P.Ltg, (short)0, (int)(unsigned)((size_t)nb * CNB * CNB * sizeof(float)), 0x00020000);
  int32_t* const flags = P.ctl + 16;
  int32_t* const pflags = flags + (nb + 1) * nb;
  const int ti = wave >> 1, tj0 = 2 * (wave & 1);   // item workgroups: this wave's two tiles of a block: (ti, tj0), (ti, tj0 + 1)
  // first element of row `row` of block row blk / of the right-hand side (one row: the others read as zero)
  auto row_base = [&](int blk, int row) -> size_t {
    return blk == nb ? (size_t)npad * lda : (size_t)(CNB * blk + row) * lda;
  };

  for (;;) {
    if (t == 0) {
      const int ticket = atomicAdd(P.ctl, 1);
      int kind = 0, c = 0, r = 0;
      const bool have = ticket < P.total && (ticket == 0 || fac_decode(ticket, nb, kind, c, r));
      sh[0] = kind; sh[1] = c; sh[2] = r; sh[3] = have ? 1 : 0;
      for (int i = 0; i < 8; i++) sy[i] = 0;
    }
    __syncthreads();
    const int kind = sh[0], c = sh[1], r = sh[2];
    if (!sh[3]) return;

    if (kind == 0) {
      // =====================================================================================================
      // the chain workgroup.  Wave 0 factors, waves 1 and 2 solve the neighbour behind it (32 rows each), wave 3 has nothing to do;
      // waves 4 .. 7: helpers (helper 0 also writes L(c, c) out, helper 1 L(c + 1, c))
      // =====================================================================================================
      float* const Db = BB;                // the diagonal block to factor
      float* const Lb = BB + FBUF;         // L(c, c), row-major as the factoring wave's registers held it (the publisher cuts off the upper part)
      float* const Sb = BB + 2 * FBUF;     // the neighbour (c + 1, c), all products of the columns left of it applied
      float* const Xs = BB + 3 * FBUF;     // L(c + 1, c)
      float* const Y01 = BB + 4 * FBUF;    // L(c + 1, c - 1) as fetched by the helpers, two buffers: stages alternate
      int* const vy = sy;
      auto give_up = [&]() {   // an LDS wait ran out (cannot happen unless a wave of this workgroup died): everybody leaves
        if (lane == 0) {
          __hip_atomic_store(P.ctl + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (P.info[BI_HANDOFF] == 0) ba_flag(P.info, BI_HANDOFF, 1);
        }
      };
      if (wave == 0) {
        // ---- the factoring wave ----
        {   // the first diagonal block straight from the matrix
          cdv_float4 pl[CNB / 4];
#pragma unroll
          for (int u = 0; u < CNB / 4; u++) pl[u] = ld4(rs, (size_t)(4 * u + g4) * lda + 4 * c16);
#pragma unroll
          for (int u = 0; u < CNB / 4; u++) *reinterpret_cast<cdv_float4*>(&Db[(4 * u + g4) * FLD + 4 * c16]) = pl[u];
          wave_lds_sync();
        }
        for (int cs = 0; cs < nb; cs++) {
          CDV_IF_STAMPS(const int sslot = cs;)
          if (cs > 0 && !lds_wait(vy, SY_U, NHELP * cs)) break;            // the diagonal block is complete
          FSTAMP(0);
          FSTAMP_RT(14);
          f2 a2[CNB / 2];
          factor_rows(Db, colb, Lt, sy, CNB * cs, lane, a2);
          FSTAMP(1);
          if (cs > 0 && !lds_wait(vy, SY_PUB1, cs)) break;                 // the previous L(c, c) has been taken
          // L(c, c) row-major into LDS for the publisher (which cuts off what lies right of the diagonal): off the critical path,
          // the solve took the columns as they came
#pragma unroll
          for (int c4 = 0; c4 < CNB / 4; c4++)
            *reinterpret_cast<cdv_float4*>(&Lb[lane * FLD + 4 * c4]) = cdv_float4{a2[2 * c4][0], a2[2 * c4][1], a2[2 * c4 + 1][0], a2[2 * c4 + 1][1]};
          lds_post(vy, SY_L, cs + 1);
          wave_lds_sync();
          const float dg = Lb[lane * FLD + lane];   // a pivot that was not positive has left a NaN (or an infinity) on the diagonal below it
          const bool bad = !(dg > 0.f && dg < 3.0e38f);
          if (__any(bad) && lane == 0 && P.info[BI_CHOL] == 0) ba_flag(P.info, BI_CHOL, cs + 1);
          FSTAMP(2);
        }
        if (lds_get(vy, SY_ABORT)) give_up();
        return;
      }
      if (wave <= NTW) {
        // ---- the solving waves: the neighbour (c + 1, c), 32 rows each, behind the factorisation of (c, c) ----
        for (int cs = 0; cs + 1 < nb; cs++) {
          CDV_IF_STAMPS(const int sslot = wave == 1 ? 32 + cs : -1;)
          FSTAMP(0);
          if (!lds_wait(vy, SY_S, NHELP * (cs + 1))) break;                // the neighbour is complete (and the helpers are done with Xs)
          if (cs > 0 && !lds_wait(vy, SY_PUB2, cs)) break;                 // the previous L(c + 1, c) has been taken
          FSTAMP(3);
          if (!solve_cols2(Sb, Lt, sy, CNB * cs, Xs, 32 * (wave - 1), lane)) break;
          lds_arrive(sy, SY_X, lane);
          FSTAMP(4);
        }
        return;
      }
      if (wave < 8 - NHELP) return;
      // ---- the helpers: helper h owns the tile row h (rows 16 h .. 16 h + 15) of the two blocks of the next stage ----
      const int h = wave - (8 - NHELP);
      // publishing: a block in LDS (row-major) -> the matrix, whole lines per store instruction; the stores are left in flight,
      // the flag goes up later (publish_flag), after the wave has had other things to do
      cdv_float4 pubv[CNB / 4];
      auto publish_issue = [&](const float* src, bool lower) {   // lower: zeros right of the diagonal
        // (the lane's coordinates behind an empty asm: what is compared with them is then not hoisted out of the stage loop as ~64
        // lane masks that live in spilled scalar registers)
        int g4o = g4, c16o = c16;
        asm volatile("" : "+v"(g4o), "+v"(c16o));
#pragma unroll
        for (int u = 0; u < CNB / 4; u++) {
          pubv[u] = *reinterpret_cast<const cdv_float4*>(&src[(4 * u + g4) * FLD + 4 * c16]);
          if (lower) {
#pragma unroll
            for (int j = 0; j < 4; j++) pubv[u][j] = (4 * c16o + j <= 4 * u + g4o) ? pubv[u][j] : 0.f;
          }
        }
        wave_lds_sync();
      };
      auto publish_store = [&](int br, int bcn) {
#pragma unroll
        for (int u = 0; u < CNB / 4; u++) st4(rs, (size_t)(CNB * br + 4 * u + g4) * lda + CNB * bcn + 4 * c16, pubv[u]);
      };
      auto publish_flag = [&](int br, int bcn, bool withhold) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0 && !withhold) __hip_atomic_store(&flags[br * nb + bcn], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      };
      for (int cs = 0; cs + 1 < nb; cs++) {
        CDV_IF_STAMPS(const int sslot = h == NHELP - 1 ? 64 + cs : -1;)
        FSTAMP(0);
        const int rn = cs + 1;
        if (rn >= 3 && !fac_wait(&pflags[rn], &pflags[rn], P.ctl, P.info, lane, P.test)) { lds_set(vy, SY_ABORT, 1); return; }
        float sT[4][4], dT[4][4];
#pragma unroll
        for (int tj = 0; tj < 4; tj++)
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const size_t rowp = (size_t)(CNB * rn + 16 * h + 4 * g4 + q) * lda;
            sT[tj][q] = ld1(rs, rowp + CNB * cs + 16 * tj + c16);
            dT[tj][q] = ld1(rs, rowp + CNB * rn + 16 * tj + c16);
          }
        FSTAMP(1);
        float* const Yb = Y01 + (cs & 1) * FBUF;
        if (cs >= 1) {
          // the one product the neighbour still lacks: L(c + 1, c - 1) L(c, c - 1)^T -- this helper's 16 rows of the first, fetched
          // by itself; the second is what the solve left in Xs a stage ago
          if (!fac_wait(&flags[rn * nb + cs - 1], &flags[rn * nb + cs - 1], P.ctl, P.info, lane, P.test)) { lds_set(vy, SY_ABORT, 1); return; }
          FSTAMP(2);
          cdv_float4 py[4];
#pragma unroll
          for (int u = 0; u < 4; u++) py[u] = ld4(rs, (size_t)(CNB * rn + 16 * h + 4 * u + g4) * lda + CNB * (cs - 1) + 4 * c16);
#pragma unroll
          for (int u = 0; u < 4; u++) *reinterpret_cast<cdv_float4*>(&Yb[(16 * h + 4 * u + g4) * FLD + 4 * c16]) = py[u];
          wave_lds_sync();
          FSTAMP(3);
          if (!lds_wait(vy, SY_X, NTW * cs)) return;
          cdv_float4 acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
          tiles_row<4>(Yb, Xs, h, c16, g4, acc);
#pragma unroll
          for (int tj = 0; tj < 4; tj++)
#pragma unroll
            for (int q = 0; q < 4; q++) sT[tj][q] -= acc[tj][q];
        }
#pragma unroll
        for (int tj = 0; tj < 4; tj++)
#pragma unroll
          for (int q = 0; q < 4; q++) Sb[(16 * h + 4 * g4 + q) * FLD + 16 * tj + c16] = sT[tj][q];
        lds_arrive(sy, SY_S, lane);
        FSTAMP(4);
        if (cs >= 1) {
          if (!lds_wait(vy, SY_S, NHELP * (cs + 1))) return;   // every helper's rows of L(c + 1, c - 1) are in
          cdv_float4 acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
          tiles_row<4>(Yb, Yb, h, c16, g4, acc);
#pragma unroll
          for (int tj = 0; tj < 4; tj++)
#pragma unroll
            for (int q = 0; q < 4; q++) dT[tj][q] -= acc[tj][q];
        }
        FSTAMP(5);
        if (h == 0) {
          // L(c, c) out as soon as its last column is in the column store: the column store itself first (what the item workgroups
          // solve against -- they are on the path to the next stage's neighbour block), the flag behind it; then row-major into the
          // matrix (what the back substitution launch reads; nobody in this launch waits for it)
          if (!lds_wait(vy, SY_F, CNB * (cs + 1))) return;
#pragma unroll
          for (int u = 0; u < CNB / 4; u++) pubv[u] = *reinterpret_cast<const cdv_float4*>(&Lt[(4 * u + g4) * CNB + 4 * c16]);
          wave_lds_sync();
#pragma unroll
          for (int u = 0; u < CNB / 4; u++) st4(rs2, (size_t)cs * CNB * CNB + (4 * u + g4) * CNB + 4 * c16, pubv[u]);
          // fault injection (tests): the second diagonal block never becomes visible
          publish_flag(cs, cs, P.test == HO_TEST_FACTOR && cs == 1);
          if (!lds_wait(vy, SY_L, cs + 1)) return;
          publish_issue(Lb, true);
          lds_post(vy, SY_PUB1, cs + 1);
          publish_store(cs, cs);
        }
        if (!lds_wait(vy, SY_X, NTW * (cs + 1))) return;        // L(c + 1, c) is in Xs
        FSTAMP(6);
        if (h == 1) {
          publish_issue(Xs, false);
          lds_post(vy, SY_PUB2, cs + 1);
          publish_store(cs + 1, cs);
        }
        {
          cdv_float4 acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
          tiles_row<4>(Xs, Xs, h, c16, g4, acc);
#pragma unroll
          for (int tj = 0; tj < 4; tj++)
#pragma unroll
            for (int q = 0; q < 4; q++) Db[(16 * h + 4 * g4 + q) * FLD + 16 * tj + c16] = dT[tj][q] - acc[tj][q];
        }
        lds_arrive(sy, SY_U, lane);
        FSTAMP(7);
        if (h == 1) publish_flag(cs + 1, cs, false);
      }
      if (h == 0) {   // the last diagonal block (the right-hand side's last block is solved against it)
        if (!lds_wait(vy, SY_F, CNB * nb)) return;
#pragma unroll
        for (int u = 0; u < CNB / 4; u++) pubv[u] = *reinterpret_cast<const cdv_float4*>(&Lt[(4 * u + g4) * CNB + 4 * c16]);
        wave_lds_sync();
#pragma unroll
        for (int u = 0; u < CNB / 4; u++) st4(rs2, (size_t)(nb - 1) * CNB * CNB + (4 * u + g4) * CNB + 4 * c16, pubv[u]);
        publish_flag(nb - 1, nb - 1, false);
        if (!lds_wait(vy, SY_L, nb)) return;
        publish_issue(Lb, true);
        publish_store(nb - 1, nb - 1);
      }
      return;
    }

    // =======================================================================================================
    // item workgroups
    // =======================================================================================================
    const bool pitem = kind == 2;
    const bool rhs = r == nb;
    CDV_IF_STAMPS(const int sslot = wave != 0 ? -1 : (!pitem && r == c + 2 ? 192 + c : -1);)
    FSTAMP(0);
    // the block that is accumulated as a product of two different block rows: (r, bc); a P item: the neighbour (c, c - 1)
    const int bc = pitem ? c - 1 : c;
    const int nk = pitem ? c - 2 : c;   // products to apply: the columns k < nk
    // ---- the blocks as the fold launch left them: this wave's tiles, requested now ----
    float s0[2][4], d0[2][4];
#pragma unroll
    for (int u = 0; u < 2; u++)
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int row = 16 * ti + 4 * g4 + q, col = 16 * (tj0 + u) + c16;
        s0[u][q] = (!rhs || row == 0) ? ld1(rs, row_base(r, row) + CNB * bc + col) : 0.f;
        d0[u][q] = pitem ? ld1(rs, row_base(c, row) + CNB * c + col) : 0.f;
      }
    cdv_float4 accS[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    cdv_float4 accD[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    // ---- sum over k < nk of L(r, k) L(bc, k)^T (a P item: and of L(c, k) L(c, k)^T), the next k's blocks in flight under the products ----
    // thread -> two 16-byte pieces of each operand block: piece i = t + 512 u: row i >> 4, columns 4 (i & 15) ..
    cdv_float4 px[2], py[2];
    auto request = [&](int k) {
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const int i = t + FT * u, row = i >> 4, c4 = i & 15;
        px[u] = (!rhs || row == 0) ? ld4(rs, row_base(r, row) + CNB * k + 4 * c4) : cdv_float4{0.f, 0.f, 0.f, 0.f};
        py[u] = ld4(rs, row_base(bc, row) + CNB * k + 4 * c4);
      }
    };
    bool ok = true;
    if (nk > 0) {
      if (wave == 0) {
        const bool w = fac_wait(&flags[r * nb + 0], &flags[bc * nb + 0], P.ctl, P.info, lane, P.test);
        if (lane == 0) sh[4] = w ? 1 : 0;
      }
      __syncthreads();
      ok = sh[4] != 0;
      if (ok) request(0);
    }
    for (int k = 0; ok && k < nk; k++) {
      float* const X = BB + 2 * (k & 1) * FBUF;
      float* const Y = X + FBUF;
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const int i = t + FT * u, row = i >> 4, c4 = i & 15;
        *reinterpret_cast<cdv_float4*>(&X[row * FLD + 4 * c4]) = px[u];
        *reinterpret_cast<cdv_float4*>(&Y[row * FLD + 4 * c4]) = py[u];
      }
      if (wave == 0 && k + 1 < nk) {
        const bool w = fac_wait(&flags[r * nb + k + 1], &flags[bc * nb + k + 1], P.ctl, P.info, lane, P.test);
        if (lane == 0) sh[4 + ((k + 1) & 1)] = w ? 1 : 0;   // (two slots: a slow wave may still be reading the verdict on k)
      }
      __syncthreads();   // operands of k in LDS; the verdict on k + 1
      if (k + 1 < nk) {
        ok = sh[4 + ((k + 1) & 1)] != 0;
        if (ok) request(k + 1);
      }
      tiles_xyt(X, Y, ti, tj0, c16, g4, accS);
      if (pitem) tiles_xyt(X, X, ti, tj0, c16, g4, accD);
    }
    if (!ok) return;   // uniform: the abort word is up, everybody leaves as they find out
    __syncthreads();   // the last products have read their operands: the buffers are free
    FSTAMP(1);
    float* const Sb = B0;
    float* const Db = B1;
    float* const Lb = B2;
    float* const Xs = B3;
#pragma unroll
    for (int u = 0; u < 2; u++)
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int row = 16 * ti + 4 * g4 + q, col = 16 * (tj0 + u) + c16;
        Sb[row * FLD + col] = s0[u][q] - accS[u][q];
        if (pitem) Db[row * FLD + col] = d0[u][q] - accD[u][q];
      }
    if (pitem) {
      // the two blocks go back where they came from, for the chain workgroup to take over (whole lines per store instruction)
      __syncthreads();
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const int i = t + FT * u, row = i >> 4, c4 = i & 15;
        st4(rs, (size_t)(CNB * c + row) * lda + CNB * bc + 4 * c4, *reinterpret_cast<const cdv_float4*>(&Sb[row * FLD + 4 * c4]));
        st4(rs, (size_t)(CNB * c + row) * lda + CNB * c + 4 * c4, *reinterpret_cast<const cdv_float4*>(&Db[row * FLD + 4 * c4]));
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (t == 0) __hip_atomic_store(&pflags[c], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      if (wave == 0) {
        FSTAMP(2);
        const bool w = fac_wait(&flags[bc * nb + bc], &flags[bc * nb + bc], P.ctl, P.info, lane, P.test);
        if (lane == 0) sh[6] = w ? 1 : 0;
        FSTAMP(3);
      }
      __syncthreads();   // the block's rows are in Sb; the verdict on L(c, c)
      if (!sh[6]) return;
      {
        // L(c, c) as the chain workgroup's column store held it: two 16-byte pieces per thread
        cdv_float4 pl[2];
#pragma unroll
        for (int u = 0; u < 2; u++) pl[u] = ld4(rs2, (size_t)bc * CNB * CNB + 4 * (t + FT * u));
#pragma unroll
        for (int u = 0; u < 2; u++) *reinterpret_cast<cdv_float4*>(&Lb[4 * (t + FT * u)]) = pl[u];
      }
      FSTAMP(4);
      __syncthreads();
      FSTAMP(5);
      if (wave < NSOLVE) solve_cols4(Sb, Lb, Xs, 16 * wave, lane);
      FSTAMP(6);
      // the finished block goes out through LDS: whole lines per store instruction, every wave stores (the right-hand side: one row)
      __syncthreads();
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const int i = t + FT * u, row = i >> 4, c4 = i & 15;
        if (!rhs || row == 0) st4(rs, row_base(r, row) + CNB * bc + 4 * c4, *reinterpret_cast<const cdv_float4*>(&Xs[row * FLD + 4 * c4]));
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (t == 0) __hip_atomic_store(&flags[r * nb + bc], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      FSTAMP(10);
    }
    __syncthreads();   // sh and the buffers are rewritten by the next item
  }
}

int g_fac_cus = 0;

}  // namespace

// tickets of a launch: the chain workgroup's + one per item (fac_decode)
int cdv_ba_big_factor_items(int nb) {
  int total = 1;
  for (int g = 0; g < nb; g++) total += (nb - g - 2 > 0 ? nb - g - 2 : 0) + 1 + (g + 3 < nb ? 1 : 0);
  return total;
}

}  // namespace cdv

// The work item behind a ticket of the factorisation launch for a matrix of nb block columns (host copy of the kernel's own
// decoding, for tests of the ticket order): out[0] = 0 chain workgroup / 1 block (r, c), solved against L(c, c) / 2 the two
// pre-accumulated blocks of block row c; out[1] = c, out[2] = r (nb: the right-hand side).  Returns the number of tickets.
extern "C" int cdv_ba_factor_ticket(int ticket, int nb, int32_t* out) {
  const int total = cdv::cdv_ba_big_factor_items(nb);
  if (out && ticket >= 0 && ticket < total) {
    int kind = 0, c = 0, r = 0;
    if (ticket > 0) cdv::fac_decode(ticket, nb, kind, c, r);
    out[0] = kind; out[1] = c; out[2] = r;
  }
  return total;
}

namespace cdv {

int cdv_ba_big_factor(float* A, int npad, int32_t* ctl, float* ltg, const int32_t* gmeta, int32_t* info, int test, hipStream_t s) {
  if (g_fac_cus == 0) {
    int dev = 0, cus = 0;
    CDV_HIP_CHECK(hipGetDevice(&dev));
    CDV_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    g_fac_cus = cus > 0 ? cus : 256;
  }
  FacArgs P;
  P.A = A; P.npad = npad; P.nb = npad / CNB; P.total = cdv_ba_big_factor_items(P.nb);
  P.ctl = ctl; P.Ltg = ltg; P.gmeta = gmeta; P.info = info; P.test = test;
  const int grid = P.total < g_fac_cus ? P.total : g_fac_cus;
  hipLaunchKernelGGL(ba_big_factor_kernel, dim3(grid), dim3(FT), 0, s, P);
  return CDV_OK;
}

}  // namespace cdv
