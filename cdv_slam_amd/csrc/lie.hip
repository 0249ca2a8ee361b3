// lie.hip -- batched SO3 / SE3 forward ops (lietorch_backends replacement), gfx950.
//
// One lane per batch element, fixed-size math in registers (cdv_se3.h restates
// cdvslam/lietorch/include/so3.h, se3.h).  Reference kernels: lietorch/src/lietorch_gpu.cu:25-299.
#include "cdv_common.h"
#include "cdv_se3.h"

namespace {

enum { OP_EXP = 0, OP_LOG, OP_INV, OP_MUL, OP_ADJ, OP_ADJT, OP_ACT, OP_ACT4, OP_MATRIX };

template <typename T, bool SE3, int OP>
__global__ __launch_bounds__(256) void lie_kernel(int64_t n, const T* __restrict__ x, const T* __restrict__ y,
                                                  T* __restrict__ z) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  constexpr int N = SE3 ? 7 : 4, K = SE3 ? 6 : 3;
  if constexpr (OP == OP_EXP) {
    T a[K], X[N];
#pragma unroll
    for (int c = 0; c < K; c++) a[c] = x[K * i + c];
    if constexpr (SE3) cdv::lt_se3_exp(a, X); else cdv::lt_so3_exp(a, X);
#pragma unroll
    for (int c = 0; c < N; c++) z[N * i + c] = X[c];
  } else if constexpr (OP == OP_LOG) {
    T X[N], a[K];
#pragma unroll
    for (int c = 0; c < N; c++) X[c] = x[N * i + c];
    if constexpr (SE3) cdv::lt_se3_log(X, a); else cdv::lt_so3_log(X, a);
#pragma unroll
    for (int c = 0; c < K; c++) z[K * i + c] = a[c];
  } else if constexpr (OP == OP_INV) {
    T X[N], Y[N];
#pragma unroll
    for (int c = 0; c < N; c++) X[c] = x[N * i + c];
    if constexpr (SE3) {
      cdv::lt_se3_inv(X, Y);
    } else {
      T q[4], qc[4];
      cdv::lt_quat_load(X, q);
      qc[0] = -q[0]; qc[1] = -q[1]; qc[2] = -q[2]; qc[3] = q[3];
      cdv::lt_quat_load(qc, Y);
    }
#pragma unroll
    for (int c = 0; c < N; c++) z[N * i + c] = Y[c];
  } else if constexpr (OP == OP_MUL) {
    T X[N], Y[N], Z[N];
#pragma unroll
    for (int c = 0; c < N; c++) { X[c] = x[N * i + c]; Y[c] = y[N * i + c]; }
    if constexpr (SE3) {
      cdv::lt_se3_mul(X, Y, Z);
    } else {
      T a[4], b[4], r[4];
      cdv::lt_quat_load(X, a);
      cdv::lt_quat_load(Y, b);
      cdv::lt_quat_mul(a, b, r);
      cdv::lt_quat_load(r, Z);
    }
#pragma unroll
    for (int c = 0; c < N; c++) z[N * i + c] = Z[c];
  } else if constexpr (OP == OP_ADJ || OP == OP_ADJT) {
    T X[N], a[K], b[K];
#pragma unroll
    for (int c = 0; c < N; c++) X[c] = x[N * i + c];
#pragma unroll
    for (int c = 0; c < K; c++) a[c] = y[K * i + c];
    if constexpr (SE3) {
      if constexpr (OP == OP_ADJ) cdv::lt_se3_adj(X, a, b); else cdv::lt_se3_adjT(X, a, b);
    } else {
      T q[4], R[9];
      cdv::lt_quat_load(X, q);
      cdv::lt_quat_to_R(q, R);
      if constexpr (OP == OP_ADJ) cdv::mat3_vec(R, a, b); else cdv::mat3T_vec(R, a, b);
    }
#pragma unroll
    for (int c = 0; c < K; c++) z[K * i + c] = b[c];
  } else if constexpr (OP == OP_ACT || OP == OP_ACT4) {
    constexpr int D = (OP == OP_ACT) ? 3 : 4;
    T X[N], p[4], o[4], t[3] = {0, 0, 0}, q[4];
#pragma unroll
    for (int c = 0; c < N; c++) X[c] = x[N * i + c];
#pragma unroll
    for (int c = 0; c < D; c++) p[c] = y[D * i + c];
    if constexpr (D == 3) p[3] = T(1);
    if constexpr (SE3) cdv::lt_se3_load(X, t, q); else cdv::lt_quat_load(X, q);
    cdv::lt_act4_loaded(t, q, p, o);
#pragma unroll
    for (int c = 0; c < D; c++) z[D * i + c] = o[c];
  } else if constexpr (OP == OP_MATRIX) {
    T X[N], t[3] = {0, 0, 0}, q[4], R[9];
#pragma unroll
    for (int c = 0; c < N; c++) X[c] = x[N * i + c];
    if constexpr (SE3) cdv::lt_se3_load(X, t, q); else cdv::lt_quat_load(X, q);
    cdv::lt_quat_to_R(q, R);
    T* M = z + 16 * i;
#pragma unroll
    for (int a = 0; a < 3; a++) {
#pragma unroll
      for (int b = 0; b < 3; b++) M[4 * a + b] = R[3 * a + b];
      M[4 * a + 3] = t[a];
    }
    M[12] = 0; M[13] = 0; M[14] = 0; M[15] = 1;
  }
}

template <typename T, bool SE3>
int launch_op(int op, int64_t n, const T* x, const T* y, T* z, hipStream_t s) {
  const int threads = 256;
  const int blocks = cdv_div_up(n, threads);
#define CDV_LIE_CASE(O)                                                                                  \
  case O:                                                                                                \
    hipLaunchKernelGGL((lie_kernel<T, SE3, O>), dim3(blocks), dim3(threads), 0, s, n, x, y, z);          \
    break;
  switch (op) {
    CDV_LIE_CASE(OP_EXP)
    CDV_LIE_CASE(OP_LOG)
    CDV_LIE_CASE(OP_INV)
    CDV_LIE_CASE(OP_MUL)
    CDV_LIE_CASE(OP_ADJ)
    CDV_LIE_CASE(OP_ADJT)
    CDV_LIE_CASE(OP_ACT)
    CDV_LIE_CASE(OP_ACT4)
    CDV_LIE_CASE(OP_MATRIX)
    default:
      cdv_set_error(CDV_ERR_ARG, "cdv_lie_op: unknown op");
      return CDV_ERR_ARG;
  }
#undef CDV_LIE_CASE
  CDV_LAUNCH_CHECK();
  return CDV_OK;
}

}  // namespace

extern "C" int cdv_lie_op(int group, int op, int dtype, int64_t n, const void* x, const void* y, void* z,
                          void* stream) {
  CDV_REQUIRE(group == 1 || group == 3, CDV_ERR_UNSUPPORTED,
              "cdv_lie_op: only SO3 (1) and SE3 (3) are on the update path; RxSO3/Sim3 are out of scope");
  CDV_REQUIRE(dtype == CDV_F32 || dtype == CDV_F64, CDV_ERR_UNSUPPORTED, "cdv_lie_op: dtype must be f32 or f64");
  const bool binary = (op == OP_MUL || op == OP_ADJ || op == OP_ADJT || op == OP_ACT || op == OP_ACT4);
  CDV_REQUIRE(!binary || y != nullptr, CDV_ERR_ARG, "cdv_lie_op: binary op needs y");
  if (n == 0) return CDV_OK;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == CDV_F32) {
    return group == 3 ? launch_op<float, true>(op, n, (const float*)x, (const float*)y, (float*)z, s)
                      : launch_op<float, false>(op, n, (const float*)x, (const float*)y, (float*)z, s);
  }
  return group == 3 ? launch_op<double, true>(op, n, (const double*)x, (const double*)y, (double*)z, s)
                    : launch_op<double, false>(op, n, (const double*)x, (const double*)y, (double*)z, s);
}
