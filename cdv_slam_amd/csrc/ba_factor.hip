// ba_factor.hip -- the Cholesky factorisation of the global bundle adjustment's reduced system (32 < N <= 1024 free poses,
// the dense `torch.linalg.cholesky` of 6N unknowns behind ba_cuda.cu:567-594 / slam.py:460-478) as ONE launch.
//
// The matrix A [(npad + 1)][npad] (ba.hip ba_big_fold_kernel: S with the damping of ba_cuda.cu:589, identity on the padded
// diagonal, row npad = y^T) is cut into 64 x 64 blocks; the result -- L in the lower blocks, z = L^-1 y in the last row --
// is what ba_big_backsolve_kernel reads.  Rounds 1-3 ran one launch per block column (28 x 26 us at N = 299: a one-wave
// panel chain behind a launch boundary each).  Here a block is a WORK ITEM that is computed left-looking by one workgroup,
//
//     block (r, c) = ( A(r, c) - sum_{k < c} L(r, k) L(c, k)^T ) L(c, c)^-T         (the right-hand side: r = nb, one row)
//
// with the sum kept in the matrix cores' accumulators for the whole life of the item, the inputs taken from the workgroups
// that produced them as soon as their flags are up (the hand-off of the programming guide: write-through stores, one flag
// per block, every load of a handed-off byte past the L1), and nothing written but the finished block.  Items are dealt by
// a ticket counter in an order in which every item only needs items with smaller tickets: the lowest unfinished ticket is
// always held by a running workgroup whose inputs are complete, so the launch finishes whatever the number of resident
// workgroups and whatever order they start in.
//
// What is on the critical path is the chain of diagonal blocks.  The work item D(c) owns the diagonal block (c, c) AND its
// left neighbour (c, c - 1): when L(c - 1, c - 1) arrives it solves the neighbour (one wave, forward substitution along the
// rows), takes the neighbour's product off the diagonal block on the matrix cores and factors it (one wave, the register
// scheme of ba_win.hip's solver) -- one hand-off per block column instead of two, and no launch boundary.
#include <hip/hip_runtime.h>

#include "cdv_ba.h"

namespace cdv {
namespace {

constexpr int FT = 512;                 // threads of a workgroup: 8 waves, two 16 x 16 tiles of a block each
constexpr int FLD = CNB + 4;            // LDS row stride of a block (16-byte aligned rows, 8 rows cover the banks once)
constexpr int FBUF = CNB * FLD;         // floats of a block in LDS
constexpr int FAC_PAD = 16 * 1024;      // bytes of dynamic LDS asked for on top of the 70 KB used: more than half a CU's together, so
                                        // that ONE workgroup runs per CU (the form of the hand-off used here was measured that way)
constexpr int AUX_SC1 = 16;             // buffer load / store past the L1, write-through

typedef float f2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

struct FacArgs {
  float* A;
  int npad, nb, total;
  int32_t* ctl;            // [0] ticket counter, [1] abort word, [FAC_CTL + r * nb + c] flag of block (r, c), r = nb: right-hand side
  const int32_t* gmeta;
  int32_t* info;
  int test;
};

// ticket -> work item.  Order: D(0); then for g = 1 .. nb: D(g) (g < nb) followed by O(g - 1) = the blocks (r, g - 1),
// r = g + 1 .. nb - 1, and the right-hand side's block of column g - 1.
//   D(c) needs L(c, k), L(c - 1, k), k <= c - 2 (items of O(k), or D(c - 1) for (c - 1, c - 2)) and L(c - 1, c - 1) (D(c - 1));
//   (r, c) of O(c) needs L(r, k) (O(k), k < c), L(c, k) (O(k) for k < c - 1, D(c) for k = c - 1) and L(c, c) (D(c)):
// all with smaller tickets.
__device__ __forceinline__ bool fac_decode(int t, int nb, int& kind, int& c, int& r) {
  if (t == 0) { kind = 0; c = 0; r = 0; return true; }
  t -= 1;
  for (int g = 1; g <= nb; g++) {
    const int nd = g < nb ? 1 : 0;
    const int co = g - 1;
    const int nrows = nb - co - 2 > 0 ? nb - co - 2 : 0;
    const int sz = nd + nrows + 1;
    if (t < sz) {
      if (t < nd) { kind = 0; c = g; r = g; return true; }
      t -= nd;
      kind = 1; c = co; r = t < nrows ? co + 2 + t : nb;
      return true;
    }
    t -= sz;
  }
  return false;
}

__device__ __forceinline__ int ld_flag(const int32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// wave 0: wait until both flags are up (f1 may equal f0).  false: gave up (the abort word is set: by us on a timeout, or by
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

__device__ __forceinline__ cdv_float4 ld4(__amdgpu_buffer_rsrc_t rs, size_t elem) {
  return __builtin_bit_cast(cdv_float4, (i32x4)__builtin_amdgcn_raw_buffer_load_b128(rs, (int)(unsigned)(elem * 4u), 0, AUX_SC1));
}
__device__ __forceinline__ float ld1(__amdgpu_buffer_rsrc_t rs, size_t elem) {
  return __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)(unsigned)(elem * 4u), 0, AUX_SC1));
}
__device__ __forceinline__ void st4(__amdgpu_buffer_rsrc_t rs, size_t elem, cdv_float4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, v), rs, (int)(unsigned)(elem * 4u), 0, AUX_SC1);
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

// one wave, lane = row: X L^T = S by forward substitution along the row; S rows in Sb, L (lower triangle + diagonal) in Lb,
// both row-major with stride FLD; rinvb: scratch of 64 floats.  Returns the row in x.
__device__ __forceinline__ void solve_rows(const float* Sb, const float* Lb, float* rinvb, int lane, float (&x)[CNB]) {
#pragma unroll
  for (int c4 = 0; c4 < CNB / 4; c4++) {
    const cdv_float4 q = *reinterpret_cast<const cdv_float4*>(&Sb[lane * FLD + 4 * c4]);
    x[4 * c4] = q[0]; x[4 * c4 + 1] = q[1]; x[4 * c4 + 2] = q[2]; x[4 * c4 + 3] = q[3];
  }
  rinvb[lane] = 1.0f / Lb[lane * FLD + lane];
  wave_lds_sync();
#pragma unroll
  for (int c = 0; c < CNB; c++) {
    // sum_{j < c} x_j L[c][j]: two chains over the whole quads (packed FMAs), the quad that holds the diagonal by itself
    f2 sa = {0.f, 0.f}, sb = {0.f, 0.f};
#pragma unroll
    for (int j4 = 0; j4 < c / 4; j4++) {
      const cdv_float4 l = *reinterpret_cast<const cdv_float4*>(&Lb[c * FLD + 4 * j4]);
      sa = __builtin_elementwise_fma(f2{x[4 * j4], x[4 * j4 + 1]}, f2{l[0], l[1]}, sa);
      sb = __builtin_elementwise_fma(f2{x[4 * j4 + 2], x[4 * j4 + 3]}, f2{l[2], l[3]}, sb);
    }
    float s = (sa[0] + sa[1]) + (sb[0] + sb[1]);
    if (c & 3) {
      const cdv_float4 l = *reinterpret_cast<const cdv_float4*>(&Lb[c * FLD + (c & ~3)]);
#pragma unroll
      for (int j = 0; j < (c & 3); j++) s = fmaf(x[(c & ~3) + j], l[j], s);
    }
    x[c] = (x[c] - s) * rinvb[c];
  }
}

// one wave, lane = row: the 64 x 64 block in Db (row-major, stride FLD; only its lower triangle matters) -> its Cholesky
// factor, row `lane` returned in a2 (pairs of columns; entries right of the diagonal are not part of it).  Right-looking in
// the wave's registers, column k + 1 broadcast through LDS while column k's rank-1 update runs (ba_win.hip's solver).
__device__ __forceinline__ bool factor_rows(const float* Db, float* colb, int lane, f2 (&a2)[CNB / 2]) {
#pragma unroll
  for (int c4 = 0; c4 < CNB / 4; c4++) {
    const cdv_float4 q = *reinterpret_cast<const cdv_float4*>(&Db[lane * FLD + 4 * c4]);
    a2[2 * c4] = f2{q[0], q[1]};
    a2[2 * c4 + 1] = f2{q[2], q[3]};
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
      bad = bad || !(piv > 0.f);                          // wave-uniform
      Ln = an * __builtin_amdgcn_rsqf(piv);
      a2[(k + 1) >> 1][(k + 1) & 1] = Ln;
      colb[lane] = Ln;      // in-order LDS: the reads of column k were issued before this write
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
  return bad;
}

__global__ __launch_bounds__(FT) void ba_big_factor_kernel(FacArgs P) {
  if (P.gmeta[GM_ERROR] || P.info[BI_OVERFLOW]) return;
  // (static: the addresses are instruction offsets; the launch asks for FAC_PAD more bytes of dynamic LDS on top)
  __shared__ __attribute__((aligned(16))) float B0[FBUF];
  __shared__ __attribute__((aligned(16))) float B1[FBUF];
  __shared__ __attribute__((aligned(16))) float B2[FBUF];
  __shared__ __attribute__((aligned(16))) float B3[FBUF];
  __shared__ __attribute__((aligned(16))) float colb[CNB];
  __shared__ __attribute__((aligned(16))) float rinvb[CNB];
  __shared__ int sh[8];   // [0] kind, [1] c, [2] r, [3] go on, [4], [5] wait verdicts of even / odd k, [6] of the diagonal block
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, c16 = lane & 15, g4 = lane >> 4;
  const int nb = P.nb, npad = P.npad;
  const size_t lda = (size_t)npad;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      P.A, (short)0, (int)(unsigned)((size_t)(npad + 1) * lda * sizeof(float)), 0x00020000);
  int32_t* const flags = P.ctl + 16;
  const int ti = wave >> 1, tj0 = 2 * (wave & 1);   // this wave's two tiles of a block: (ti, tj0), (ti, tj0 + 1)

  for (;;) {
    if (t == 0) {
      const int ticket = atomicAdd(P.ctl, 1);
      int kind = 0, c = 0, r = 0;
      const bool have = ticket < P.total && fac_decode(ticket, nb, kind, c, r);
      sh[0] = kind; sh[1] = c; sh[2] = r; sh[3] = have ? 1 : 0;
    }
    __syncthreads();
    const int kind = sh[0], c = sh[1], r = sh[2];
    if (!sh[3]) return;
    const bool diag = kind == 0;
    const bool rhs = r == nb;
    // the block that is accumulated as a product of two different rows: (rowblk, bc); D(c): its left neighbour (c, c - 1)
    const int bc = diag ? c - 1 : c;
    const bool has_s = bc >= 0;
    const int nk = has_s ? bc : 0;
    // first element of row `row` of block row r / the right-hand side (one row: the others read as zero)
    auto row_base = [&](int blk, int row) -> size_t {
      return blk == nb ? (size_t)npad * lda : (size_t)(CNB * blk + row) * lda;
    };
    // ---- the blocks as the fold launch left them: this wave's tiles, requested now ----
    float s0[2][4], d0[2][4];
#pragma unroll
    for (int u = 0; u < 2; u++)
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int row = 16 * ti + 4 * g4 + q, col = 16 * (tj0 + u) + c16;
        s0[u][q] = (has_s && (!rhs || row == 0)) ? ld1(rs, row_base(r, row) + CNB * bc + col) : 0.f;
        d0[u][q] = diag ? ld1(rs, row_base(c, row) + CNB * c + col) : 0.f;
      }
    cdv_float4 accS[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    cdv_float4 accD[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    // ---- sum over k < bc of L(r, k) L(bc, k)^T (and, D(c): of L(c, k) L(c, k)^T), the next k's blocks in flight under the products ----
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
      float* const X = (k & 1) ? B2 : B0;
      float* const Y = (k & 1) ? B3 : B1;
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
      if (diag) tiles_xyt(X, X, ti, tj0, c16, g4, accD);
    }
    if (!ok) return;   // uniform: the abort word is up, everybody leaves as they find out
    __syncthreads();   // the last products have read their operands: B0 .. B3 are free
    float* const Sb = B0;
    float* const Db = B1;
    float* const Lb = B2;
    float* const Xs = B3;
    if (has_s) {
#pragma unroll
      for (int u = 0; u < 2; u++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int row = 16 * ti + 4 * g4 + q, col = 16 * (tj0 + u) + c16;
          Sb[row * FLD + col] = s0[u][q] - accS[u][q];
        }
      if (wave == 0) {
        // L(bc, bc): this wave waits for it and fetches it by itself (16 pieces per lane), no barrier between the flag and the solve
        const bool w = fac_wait(&flags[bc * nb + bc], &flags[bc * nb + bc], P.ctl, P.info, lane, P.test);
        if (lane == 0) sh[6] = w ? 1 : 0;
        if (w) {
          cdv_float4 pl[CNB / 4];
#pragma unroll
          for (int u = 0; u < CNB / 4; u++) {
            const int row = 4 * u + g4;
            pl[u] = ld4(rs, (size_t)(CNB * bc + row) * lda + CNB * bc + 4 * c16);
          }
#pragma unroll
          for (int u = 0; u < CNB / 4; u++) {
            const int row = 4 * u + g4;
            *reinterpret_cast<cdv_float4*>(&Lb[row * FLD + 4 * c16]) = pl[u];
          }
        }
      }
      __syncthreads();   // the neighbour's rows are in Sb; the verdict
      if (!sh[6]) return;
      if (wave == 0) {
        float x[CNB];
        solve_rows(Sb, Lb, rinvb, lane, x);
        if (!diag) {
          // an item of O(c): the finished block goes out from this wave's registers (the right-hand side: one row)
          if (!rhs || lane == 0) {
#pragma unroll
            for (int c4 = 0; c4 < CNB / 4; c4++)
              st4(rs, row_base(r, lane) + CNB * bc + 4 * c4, cdv_float4{x[4 * c4], x[4 * c4 + 1], x[4 * c4 + 2], x[4 * c4 + 3]});
          }
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          if (lane == 0) __hip_atomic_store(&flags[r * nb + bc], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
#pragma unroll
          for (int c4 = 0; c4 < CNB / 4; c4++)
            *reinterpret_cast<cdv_float4*>(&Xs[lane * FLD + 4 * c4]) = cdv_float4{x[4 * c4], x[4 * c4 + 1], x[4 * c4 + 2], x[4 * c4 + 3]};
        }
      }
    }
    if (diag) {
      if (has_s) {
        __syncthreads();   // L(c, c - 1) is in Xs
        tiles_xyt(Xs, Xs, ti, tj0, c16, g4, accD);
      }
#pragma unroll
      for (int u = 0; u < 2; u++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int row = 16 * ti + 4 * g4 + q, col = 16 * (tj0 + u) + c16;
          Db[row * FLD + col] = d0[u][q] - accD[u][q];
        }
      __syncthreads();
      if (has_s && wave == FT / 64 - 1) {
        // L(c, c - 1) goes out from the last wave while wave 0 factors (the blocks right of it need it a whole factorisation later)
        cdv_float4 xr[CNB / 4];
#pragma unroll
        for (int c4 = 0; c4 < CNB / 4; c4++) xr[c4] = *reinterpret_cast<const cdv_float4*>(&Xs[lane * FLD + 4 * c4]);
#pragma unroll
        for (int c4 = 0; c4 < CNB / 4; c4++) st4(rs, (size_t)(CNB * c + lane) * lda + CNB * bc + 4 * c4, xr[c4]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(&flags[c * nb + bc], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (wave == 0) {
        f2 a2[CNB / 2];
        const bool bad = factor_rows(Db, colb, lane, a2);
#pragma unroll
        for (int c4 = 0; c4 < CNB / 4; c4++) {
          cdv_float4 q;
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const int cc = 4 * c4 + j;
            q[j] = (cc <= lane) ? a2[cc >> 1][cc & 1] : 0.f;
          }
          st4(rs, (size_t)(CNB * c + lane) * lda + CNB * c + 4 * c4, q);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // fault injection (tests): the second diagonal block never becomes visible
        if (lane == 0 && !(P.test == HO_TEST_FACTOR && c == 1))
          __hip_atomic_store(&flags[c * nb + c], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (lane == 0 && bad && P.info[BI_CHOL] == 0) ba_flag(P.info, BI_CHOL, c + 1);
      }
    }
    __syncthreads();   // sh and the buffers are rewritten by the next item
  }
}

int g_fac_cus = 0;

}  // namespace

int cdv_ba_big_factor_items(int nb) {
  int total = 1;
  for (int g = 1; g <= nb; g++) total += (g < nb ? 1 : 0) + (nb - g - 1 > 0 ? nb - g - 1 : 0) + 1;
  return total;
}

int cdv_ba_big_factor(float* A, int npad, int32_t* ctl, const int32_t* gmeta, int32_t* info, int test, hipStream_t s) {
  if (g_fac_cus == 0) {
    int dev = 0, cus = 0;
    CDV_HIP_CHECK(hipGetDevice(&dev));
    CDV_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    CDV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(ba_big_factor_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      FAC_PAD));
    g_fac_cus = cus > 0 ? cus : 256;
  }
  FacArgs P;
  P.A = A; P.npad = npad; P.nb = npad / CNB; P.total = cdv_ba_big_factor_items(P.nb);
  P.ctl = ctl; P.gmeta = gmeta; P.info = info; P.test = test;
  const int grid = P.total < g_fac_cus ? P.total : g_fac_cus;
  hipLaunchKernelGGL(ba_big_factor_kernel, dim3(grid), dim3(FT), FAC_PAD, s, P);
  return CDV_OK;
}

}  // namespace cdv
