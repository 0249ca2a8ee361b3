// cdv_ba_pairs.h -- internal: device helpers shared by the chunk kernels of ba_win.hip (N <= 10) and ba_mid.hip
// (10 < N <= 32): the per-edge products of a frame pair, their 16-lane transpose-reduce, edge records and inputs.
#pragma once
#include <utility>

#include "cdv_ba.h"
#include "cdv_se3.h"

namespace cdv {


constexpr int NV = 90;                     // distinct sums of one frame pair: 21 + 6 (B_ii, v_i) + 21 + 6 (B_jj, v_j) + 36 (B_ij)

__device__ __forceinline__ int tri_index(int R, int Cc) { return ((R * (R + 1)) >> 1) + Cc; }

// ds_add_f32 without return: issued and forgotten (a read-modify-write through registers would expose an LDS round trip
// per entry).  Used on data only this wave touches, one lane per address inside an instruction: no contention, and the
// LDS unit executes a wave's instructions in order, so successive adds to one address apply in program order.
__device__ __forceinline__ void lds_add(float* p, float v) {
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// ---- the sums of one frame pair (i, j) over the edges of a target slot ---------------------------------------------------
// Flat order of the NV = 90 values (signs included, ba_cuda.cu:364-377,393-398 semantics):
//   [0, 21)   B_ii += w Ji Ji^T   lower triangle, (a, b) with a >= b in row-major order
//   [21, 27)  v_i  -= w r Ji
//   [27, 48)  B_jj += w Jj Jj^T   lower triangle
//   [48, 54)  v_j  += w r Jj
//   [54, 90)  B_ij -= w Ji Jj^T   all 36, (a, b) row-major
// code of a value: kind << 6 | a << 3 | b, kind 7 = padding
constexpr int pair_code(int vi) {
  int a = 0, b = 0, kind = 7, t = 0;
  if (vi < 21) { kind = 0; t = vi; }
  else if (vi < 27) { kind = 1; a = vi - 21; }
  else if (vi < 48) { kind = 2; t = vi - 27; }
  else if (vi < 54) { kind = 3; a = vi - 48; }
  else if (vi < NV) { kind = 4; a = (vi - 54) / 6; b = (vi - 54) - 6 * a; }
  if (kind == 0 || kind == 2) {   // (a, b) of the t-th entry of a lower triangle in row-major order
    a = 0;
    while (((a + 1) * (a + 2)) / 2 <= t) a++;
    b = t - (a * (a + 1)) / 2;
  }
  return (kind << 6) | (a << 3) | b;
}

// the weighted Jacobian rows every product starts from
struct PairW {
  float wi0[6], wi1[6], wj0[6], wj1[6];
};
__device__ __forceinline__ PairW pair_weights(const EdgeFactor& J, float w0, float w1) {
  PairW P;
#pragma unroll
  for (int a = 0; a < 6; a++) {
    P.wi0[a] = w0 * J.Ji[a]; P.wi1[a] = w1 * J.Ji[6 + a];
    P.wj0[a] = w0 * J.Jj[a]; P.wj1[a] = w1 * J.Jj[6 + a];
  }
  return P;
}

// value VI of the flat order above.  Generated 16 at a time, right before their reduction, so that at most 16 of the 90
// are alive.
template <int VI>
__device__ __forceinline__ float pair_value(const EdgeFactor& J, const PairW& P) {
  constexpr int code = pair_code(VI);
  constexpr int kind = code >> 6, a = (code >> 3) & 7, b = code & 7;
  if constexpr (kind == 0) return fmaf(P.wi1[a], J.Ji[6 + b], P.wi0[a] * J.Ji[b]);
  else if constexpr (kind == 1) return -fmaf(P.wi1[a], J.r[1], P.wi0[a] * J.r[0]);
  else if constexpr (kind == 2) return fmaf(P.wj1[a], J.Jj[6 + b], P.wj0[a] * J.Jj[b]);
  else if constexpr (kind == 3) return fmaf(P.wj1[a], J.r[1], P.wj0[a] * J.r[0]);
  else if constexpr (kind == 4) return -fmaf(P.wi1[a], J.Jj[6 + b], P.wi0[a] * J.Jj[b]);
  else return 0.f;
}

template <int G, int... I>
__device__ __forceinline__ void pair_group(const EdgeFactor& J, const PairW& P, float (&v)[16], std::integer_sequence<int, I...>) {
  ((v[I] = pair_value<16 * G + I>(J, P)), ...);
}

// the code of value 16 g + br (br known at run time only: the lane's bit-reversed index)
__device__ __forceinline__ int pair_code_rt(int g, int br) {
  int code = pair_code(95);
#pragma unroll
  for (int vi = 0; vi < 96; vi++)
    if (vi == 16 * g + br) code = pair_code(vi);
  return code;
}

// value of lane (l + n) or (l - n) mod 16 of the same DPP row (row_ror:n); the reduction below works with either
template <int N>
__device__ __forceinline__ float row_ror(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + N, 0xf, 0xf, false));
}

// Transpose-reduce: 16 values per lane, summed over the 16 lanes of each DPP row; lane l of the row ends up with the
// total of value 8 b0(l) + 4 b1(l) + 2 b2(l) + b3(l).  Step d in {1, 2, 4, 8}: a lane keeps the half of its values that
// its bit log2(d) selects and adds the same half of the lane d away (whose bit log2(d) is the opposite one and whose
// lower bits -- hence its value set -- are the same): 8 + 4 + 2 + 1 adds, a fixed summation tree.
__device__ __forceinline__ float transpose_reduce16(const float* v, int c16) {
  const bool b0 = (c16 & 1) != 0, b1 = (c16 & 2) != 0, b2 = (c16 & 4) != 0, b3 = (c16 & 8) != 0;
  float u8[8], u4[4], u2[2];
#pragma unroll
  for (int i = 0; i < 8; i++) u8[i] = (b0 ? v[i + 8] : v[i]) + row_ror<1>(b0 ? v[i] : v[i + 8]);
#pragma unroll
  for (int i = 0; i < 4; i++) u4[i] = (b1 ? u8[i + 4] : u8[i]) + row_ror<2>(b1 ? u8[i] : u8[i + 4]);
#pragma unroll
  for (int i = 0; i < 2; i++) u2[i] = (b2 ? u4[i + 2] : u4[i]) + row_ror<4>(b2 ? u4[i] : u4[i + 2]);
  return (b3 ? u2[1] : u2[0]) + row_ror<8>(b3 ? u2[0] : u2[1]);
}

// The "ranks" the slab kernels work through, for either form of the index (cdv_graph.h):
//   ranked CSR index  U unique patches (read from the index); rank r: id kx[r], CSR segment koff_u[r] .. koff_u[r + 1],
//                     chunk-slot row r
//   patch table       U = the table's capacity (a kernel argument: nothing of the span is read from memory, so the first
//                     load level does not wait for another one); rank r IS slot r = id mod capacity: id tkid[r], degree
//                     tdeg[r] (0: no patch in this slot -- it contributes nothing and is not retracted; a chunk without
//                     any patch leaves a zero slab), overflow segment tplo[r], table row r
// Slot order is not id order where the ids wrap around the capacity; every sum still has one owner and a fixed order.
// (the form of the index is a TEMPLATE parameter of the kernels: as a run-time choice the compiler puts branches around
// the first-level loads, which serialises the load levels -- measured: +1.1 us on the chunk kernel)
struct PatchSpan {
  int U;
};
template <bool TABLE>
__device__ __forceinline__ PatchSpan patch_span(const BaWinArgs& A) {
  PatchSpan s;
  s.U = TABLE ? A.tab_cap : A.gmeta[GM_U];
  return s;
}
// int4 index of record t of row u in the chunk-slot layout (row = unique rank, or slot of a table)
__device__ __forceinline__ size_t cell_index(int u, int t) { return (size_t)((u >> 4) * ELL_SLOTS + t) * 16 + (u & 15); }

// CSR offset / degree / id of rank rs (< U), all loads unconditional (select-computed addresses: see the load levels)
struct PatchRow {
  int plo, deg;
  int64_t id;
};
template <bool TABLE>
__device__ __forceinline__ PatchRow patch_row(const BaWinArgs& A, int rs) {
  PatchRow r;
  if (TABLE) {
    const int plo = A.tplo[rs], deg = A.tdeg[rs], kid = A.tkid[rs];
    r.plo = plo; r.deg = deg; r.id = (int64_t)(kid < 0 ? 0 : kid);
  } else {
    const int lo = A.koff_u[rs], hi = A.koff_u[rs + 1];
    r.plo = lo; r.deg = hi - lo; r.id = A.kx[rs];
  }
  return r;
}

struct EdgeRec {
  int e, ix, jx;
};
struct EdgeIn {
  float pi[7], pj[7], tx, ty, wx, wy;
};

// A record becomes usable once the patch's degree is known: a slot that does not exist (its memory may hold anything)
// is replaced by `safe` (CSR record 0, always valid) BEFORE any of its fields is used as an index -- by selects.  A
// graph built without source frames (HAS_II false) gets them from ii here: one more dependent load.
template <bool HAS_II>
__device__ __forceinline__ EdgeRec settle_rec(const BaWinArgs& A, int4 raw, bool exists, const EdgeRec& safe) {
  EdgeRec r;
  r.e = exists ? raw.x : safe.e;
  r.jx = exists ? raw.z : safe.jx;
  if (HAS_II) r.ix = exists ? raw.y : safe.ix;
  else r.ix = (int)A.ii[r.e];
  return r;
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

}  // namespace cdv
