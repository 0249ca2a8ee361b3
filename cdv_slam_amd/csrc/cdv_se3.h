// cdv_se3.h -- device-side SO3 / SE3 arithmetic (gfx950).
//
// Two families, because the reference has two:
//  * lt_*  : lietorch semantics (cdvslam/lietorch/include/so3.h, se3.h): every load of a group
//            element re-normalises the quaternion (so3.h:30-37); used by pops.transform and the
//            lietorch_backends ops.
//  * fastba semantics (se3_between_raw, se3_row_times_adj_raw, se3_retract_raw, fastba_factor): group elements
//            are used as stored (no normalisation), fastba's own series thresholds (ba_cuda.cu:97,140); written on
//            the same primitives as the lt_* family.
// Data layout: SE3 = (tx,ty,tz, qx,qy,qz,qw); tangent = (tau[3], phi[3]).
#pragma once
#include <hip/hip_runtime.h>

#define CDV_LIE_EPS 1e-6

namespace cdv {

template <typename T> __device__ __forceinline__ T t_sqrt(T x);
template <> __device__ __forceinline__ float t_sqrt<float>(float x) { return sqrtf(x); }
template <> __device__ __forceinline__ double t_sqrt<double>(double x) { return sqrt(x); }
template <typename T> __device__ __forceinline__ T t_sin(T x);
template <> __device__ __forceinline__ float t_sin<float>(float x) { return sinf(x); }
template <> __device__ __forceinline__ double t_sin<double>(double x) { return sin(x); }
template <typename T> __device__ __forceinline__ T t_cos(T x);
template <> __device__ __forceinline__ float t_cos<float>(float x) { return cosf(x); }
template <> __device__ __forceinline__ double t_cos<double>(double x) { return cos(x); }
template <typename T> __device__ __forceinline__ T t_atan(T x);
template <> __device__ __forceinline__ float t_atan<float>(float x) { return atanf(x); }
template <> __device__ __forceinline__ double t_atan<double>(double x) { return atan(x); }

#ifdef CDV_EXP_CONTRACT
#define CDV_NOCONTRACT
#else
#define CDV_NOCONTRACT _Pragma("clang fp contract(off)")
#endif

template <typename T>
__device__ __forceinline__ void cross3(const T* a, const T* b, T* o) {
  CDV_NOCONTRACT
  T x = a[1] * b[2] - a[2] * b[1];
  T y = a[2] * b[0] - a[0] * b[2];
  T z = a[0] * b[1] - a[1] * b[0];
  o[0] = x; o[1] = y; o[2] = z;
}

// ---- lietorch family ---------------------------------------------------------------------------
// (the quaternion / rotation primitives run WITHOUT multiply-add contraction: `#pragma clang fp contract(off)` in their
// bodies.  Contraction is decided per call site after inlining, so two kernels running the same formula could differ in
// the last bit -- the reprojection of cdv_transform and of the table prologue must not, cdv_parts.h tf_pixel.)

template <typename T>
__device__ __forceinline__ void lt_quat_load(const T* d, T* q) {  // so3.h:30-37
  CDV_NOCONTRACT
  T n = t_sqrt<T>(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]);
  q[0] = d[0] / n; q[1] = d[1] / n; q[2] = d[2] / n; q[3] = d[3] / n;
}

template <typename T>
__device__ __forceinline__ void lt_quat_mul(const T* a, const T* b, T* o) {
  CDV_NOCONTRACT
  T w = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
  T x = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  T y = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
  T z = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
  o[0] = x; o[1] = y; o[2] = z; o[3] = w;
}

template <typename T>
__device__ __forceinline__ void lt_rot(const T* q, const T* p, T* o) {  // so3.h:54-59
  CDV_NOCONTRACT
  T uv[3], c[3];
  cross3(q, p, uv);
  uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
  cross3(q, uv, c);
  T x = p[0] + q[3] * uv[0] + c[0];
  T y = p[1] + q[3] * uv[1] + c[1];
  T z = p[2] + q[3] * uv[2] + c[2];
  o[0] = x; o[1] = y; o[2] = z;
}

template <typename T>
__device__ __forceinline__ void lt_quat_to_R(const T* q, T* R) {
  T tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
  T twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
  T txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
  T tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
  R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

template <typename T>
__device__ __forceinline__ void lt_se3_load(const T* d, T* t, T* q) {  // se3.h:36
  CDV_NOCONTRACT
  t[0] = d[0]; t[1] = d[1]; t[2] = d[2];
  lt_quat_load(d + 3, q);
}

template <typename T>
__device__ __forceinline__ void lt_se3_inv(const T* X, T* Y) {  // se3.h:38-40
  CDV_NOCONTRACT
  T t[3], q[4], qc[4], qi[4], r[3];
  lt_se3_load(X, t, q);
  qc[0] = -q[0]; qc[1] = -q[1]; qc[2] = -q[2]; qc[3] = q[3];
  lt_quat_load(qc, qi);
  lt_rot(qi, t, r);
  Y[0] = -r[0]; Y[1] = -r[1]; Y[2] = -r[2];
  Y[3] = qi[0]; Y[4] = qi[1]; Y[5] = qi[2]; Y[6] = qi[3];
}

template <typename T>
__device__ __forceinline__ void lt_se3_mul(const T* X, const T* Y, T* Z) {  // se3.h:47-49
  CDV_NOCONTRACT
  T t1[3], q1[4], t2[3], q2[4], qr[4], q[4], r[3];
  lt_se3_load(X, t1, q1);
  lt_se3_load(Y, t2, q2);
  lt_quat_mul(q1, q2, qr);
  lt_quat_load(qr, q);
  lt_rot(q1, t2, r);
  Z[0] = t1[0] + r[0]; Z[1] = t1[1] + r[1]; Z[2] = t1[2] + r[2];
  Z[3] = q[0]; Z[4] = q[1]; Z[5] = q[2]; Z[6] = q[3];
}

// act4 with an already loaded (normalised) element (se3.h:55-58)
template <typename T>
__device__ __forceinline__ void lt_act4_loaded(const T* t, const T* q, const T* p, T* o) {
  CDV_NOCONTRACT
  T r[3];
  lt_rot(q, p, r);
  T w = p[3];
  o[0] = r[0] + t[0] * w; o[1] = r[1] + t[1] * w; o[2] = r[2] + t[2] * w; o[3] = w;
}

template <typename T>
__device__ __forceinline__ void hat3(const T* p, T* M) {
  M[0] = 0;     M[1] = -p[2]; M[2] = p[1];
  M[3] = p[2];  M[4] = 0;     M[5] = -p[0];
  M[6] = -p[1]; M[7] = p[0];  M[8] = 0;
}

template <typename T>
__device__ __forceinline__ void mat3_mul(const T* A, const T* B, T* C) {
  T R[9];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) R[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
#pragma unroll
  for (int i = 0; i < 9; i++) C[i] = R[i];
}

template <typename T>
__device__ __forceinline__ void mat3_vec(const T* A, const T* v, T* o) {
  T x = A[0] * v[0] + A[1] * v[1] + A[2] * v[2];
  T y = A[3] * v[0] + A[4] * v[1] + A[5] * v[2];
  T z = A[6] * v[0] + A[7] * v[1] + A[8] * v[2];
  o[0] = x; o[1] = y; o[2] = z;
}

template <typename T>
__device__ __forceinline__ void mat3T_vec(const T* A, const T* v, T* o) {
  T x = A[0] * v[0] + A[3] * v[1] + A[6] * v[2];
  T y = A[1] * v[0] + A[4] * v[1] + A[7] * v[2];
  T z = A[2] * v[0] + A[5] * v[1] + A[8] * v[2];
  o[0] = x; o[1] = y; o[2] = z;
}

// Adj(X) a = (R tau + t x (R phi), R phi)      se3.h:60-69,84-86
template <typename T>
__device__ __forceinline__ void lt_se3_adj(const T* X, const T* a, T* b) {
  T t[3], q[4], R[9], Rt[3], Rp[3], c[3];
  lt_se3_load(X, t, q);
  lt_quat_to_R(q, R);
  mat3_vec(R, a, Rt);
  mat3_vec(R, a + 3, Rp);
  cross3(t, Rp, c);
  b[0] = Rt[0] + c[0]; b[1] = Rt[1] + c[1]; b[2] = Rt[2] + c[2];
  b[3] = Rp[0]; b[4] = Rp[1]; b[5] = Rp[2];
}

// Adj(X)^T a = (R^T a1, R^T (a1 x t) + R^T a2)   se3.h:88-90  ((tx R)^T a1 = R^T tx^T a1 = -R^T (t x a1))
template <typename T>
__device__ __forceinline__ void lt_se3_adjT_loaded(const T* t, const T* R, const T* a, T* b) {
  T c[3], s[3];
  cross3(a, t, c);  // a1 x t = -(t x a1)
  s[0] = c[0] + a[3]; s[1] = c[1] + a[4]; s[2] = c[2] + a[5];
  mat3T_vec(R, a, b);
  mat3T_vec(R, s, b + 3);
}

template <typename T>
__device__ __forceinline__ void lt_se3_adjT(const T* X, const T* a, T* b) {
  T t[3], q[4], R[9];
  lt_se3_load(X, t, q);
  lt_quat_to_R(q, R);
  lt_se3_adjT_loaded(t, R, a, b);
}

template <typename T>
__device__ __forceinline__ void lt_so3_exp(const T* phi, T* q) {  // so3.h:153-170
  T theta2 = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2];
  T theta = t_sqrt<T>(theta2);
  T imag, real;
  if (theta < CDV_LIE_EPS) {
    T theta4 = theta2 * theta2;
    imag = T(0.5) - T(1.0 / 48.0) * theta2 + T(1.0 / 3840.0) * theta4;
    real = T(1) - T(1.0 / 8.0) * theta2 + T(1.0 / 384.0) * theta4;
  } else {
    imag = (T)(sin(.5 * (double)theta) / (double)theta);
    real = (T)cos(.5 * (double)theta);
  }
  T raw[4] = {imag * phi[0], imag * phi[1], imag * phi[2], real};
  lt_quat_load(raw, q);
}

template <typename T>
__device__ __forceinline__ void lt_so3_log(const T* qd, T* phi) {  // so3.h:115-151
  T q[4];
  lt_quat_load(qd, q);
  T sq = q[0] * q[0] + q[1] * q[1] + q[2] * q[2];
  T w = q[3];
  T f;
  if (sq < CDV_LIE_EPS * CDV_LIE_EPS) {
    T w2 = w * w;
    f = T(2) / w - T(2.0 / 3.0) * sq / (w * w2);
  } else {
    T n = t_sqrt<T>(sq);
    T aw = w < 0 ? -w : w;
    if (aw < CDV_LIE_EPS) {
      f = (w > 0) ? T(3.14159265358979323846) / n : -T(3.14159265358979323846) / n;
    } else {
      f = T(2) * t_atan<T>(n / w) / n;
    }
  }
  phi[0] = f * q[0]; phi[1] = f * q[1]; phi[2] = f * q[2];
}

template <typename T>
__device__ __forceinline__ void lt_so3_left_jacobian(const T* phi, T* J) {  // so3.h:172-191
  T Phi[9], Phi2[9];
  hat3(phi, Phi);
  mat3_mul(Phi, Phi, Phi2);
  T theta2 = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2];
  T theta = t_sqrt<T>(theta2);
  T c1, c2;
  if (theta < CDV_LIE_EPS) {
    c1 = T(1.0 / 2.0) - T(1.0 / 24.0) * theta2;
    c2 = T(1.0 / 6.0) - T(1.0 / 120.0) * theta2;
  } else {
    c1 = (T)((1.0 - t_cos<T>(theta)) / theta2);
    c2 = (T)((theta - t_sin<T>(theta)) / (theta2 * theta));
  }
#pragma unroll
  for (int i = 0; i < 9; i++) J[i] = ((i % 4 == 0) ? T(1) : T(0)) + c1 * Phi[i] + c2 * Phi2[i];
}

template <typename T>
__device__ __forceinline__ void lt_so3_left_jacobian_inverse(const T* phi, T* J) {  // so3.h:193-210
  T Phi[9], Phi2[9];
  hat3(phi, Phi);
  mat3_mul(Phi, Phi, Phi2);
  T theta2 = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2];
  T theta = t_sqrt<T>(theta2);
  T half = T(0.5) * theta;
  T c2 = (theta < CDV_LIE_EPS)
             ? T(1.0 / 12.0)
             : (T(1) - theta * t_cos<T>(half) / (T(2) * t_sin<T>(half))) / (theta * theta);
#pragma unroll
  for (int i = 0; i < 9; i++) J[i] = ((i % 4 == 0) ? T(1) : T(0)) + T(-0.5) * Phi[i] + c2 * Phi2[i];
}

template <typename T>
__device__ __forceinline__ void lt_se3_exp(const T* xi, T* X) {  // se3.h:134-142
  T q[4], J[9], t[3];
  lt_so3_exp(xi + 3, q);
  lt_so3_left_jacobian(xi + 3, J);
  mat3_vec(J, xi, t);
  X[0] = t[0]; X[1] = t[1]; X[2] = t[2];
  X[3] = q[0]; X[4] = q[1]; X[5] = q[2]; X[6] = q[3];
}

template <typename T>
__device__ __forceinline__ void lt_se3_log(const T* X, T* xi) {  // se3.h:124-132
  T phi[3], Vinv[9], tau[3];
  lt_so3_log(X + 3, phi);
  lt_so3_left_jacobian_inverse(phi, Vinv);
  mat3_vec(Vinv, X, tau);
  xi[0] = tau[0]; xi[1] = tau[1]; xi[2] = tau[2];
  xi[3] = phi[0]; xi[4] = phi[1]; xi[5] = phi[2];
}

// ---- fastba semantics, on the primitives above (float only) -----------------------------------------------
// fastba differs from lietorch in conventions, not in geometry: group elements are used as
// stored -- NO quaternion re-normalisation anywhere (ba_cuda.cu:74-85 vs so3.h:30-37) -- the exponential switches to
// its series at theta^2 < 1e-8 and adds the translation's rotation coupling only for theta > 1e-4 (ba_cuda.cu:97,140).
// Everything below is the lt_* arithmetic (cross3 / lt_rot / lt_quat_mul / lt_act4_loaded) with those conventions.

// G_ij = G_j G_i^-1 of two stored poses (t, q_xyzw), raw
__device__ __forceinline__ void se3_between_raw(const float* Pi, const float* Pj, float* t, float* q) {
  const float qi_conj[4] = {-Pi[3], -Pi[4], -Pi[5], Pi[6]};
  lt_quat_mul(Pj + 3, qi_conj, q);
  float moved[3];
  lt_rot(q, Pi, moved);               // where G_ij's rotation takes camera i's translation
#pragma unroll
  for (int a = 0; a < 3; a++) t[a] = Pj[a] - moved[a];
}

// row vector (lin, ang) times Adj(t, q):  (R^T lin, R^T (ang + lin x t)); R^T v = rotation by the conjugate
__device__ __forceinline__ void se3_row_times_adj_raw(const float* t, const float* q, const float* row, float* out) {
  const float qc[4] = {-q[0], -q[1], -q[2], q[3]};
  float coupled[3];
  cross3(row, t, coupled);
#pragma unroll
  for (int a = 0; a < 3; a++) coupled[a] += row[3 + a];
  lt_rot(qc, row, out);
  lt_rot(qc, coupled, out + 3);
}

// pose retraction T <- Exp(xi) T, xi = (tau, phi), in place on a stored pose (ba_cuda.cu:178-206 semantics)
__device__ __forceinline__ void se3_retract_raw(const float* xi, float* P) {
  CDV_NOCONTRACT   // the same bits wherever it is inlined: the fused BA launch retracts per lane what the finish launch stores
  const float* tau = xi;
  const float* phi = xi + 3;
  const float th2 = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2];
  const float th = sqrtf(th2);
  // unit quaternion of the rotation vector phi: (sin(th/2)/th * phi, cos(th/2)), series below theta^2 = 1e-8
  const bool tiny = th2 < 1e-8f;
  const float th4 = th2 * th2;
  const float sv = tiny ? 0.5f - th2 * (1.0f / 48.0f) + th4 * (1.0f / 3840.0f) : sinf(0.5f * th) / th;
  const float cw = tiny ? 1.0f - th2 * (1.0f / 8.0f) + th4 * (1.0f / 384.0f) : cosf(0.5f * th);
  const float dq[4] = {sv * phi[0], sv * phi[1], sv * phi[2], cw};
  // translation of Exp: V tau = tau + a (phi x tau) + b (phi x (phi x tau)); the coupling terms only for theta > 1e-4
  float dt[3] = {tau[0], tau[1], tau[2]};
  if (th > 1e-4f) {
    float w1[3], w2[3];
    cross3(phi, tau, w1);
    cross3(phi, w1, w2);
    const float a = (1.0f - cosf(th)) / th2, b = (th - sinf(th)) / (th * th2);
#pragma unroll
    for (int c = 0; c < 3; c++) dt[c] += a * w1[c] + b * w2[c];
  }
  float qn[4], tn[3];
  lt_quat_mul(dq, P + 3, qn);
  lt_rot(dq, P, tn);
#pragma unroll
  for (int c = 0; c < 3; c++) P[c] = tn[c] + dt[c];
#pragma unroll
  for (int c = 0; c < 4; c++) P[3 + c] = qn[c];
}

// One reprojection factor of fastba (ba_cuda.cu:261-342 semantics): the patch centre (px, py, inverse depth pd) of a
// patch in frame i seen in frame j through the pinhole (fx, fy, cx, cy).
//   residual r = target - pi(X_j);  valid: |r| < 128, Z > 0.2, projection inside the image grown by 64 px
//   g_r  = d pi_r / d X_j = (f d, 0, -f X d^2) resp. (0, f d, -f Y d^2),  d = 1/Z (0 below Z = 0.2)
//   X_j moves by  w tau + phi x X_j  under a left perturbation (tau, phi) of G_j  (w = inverse depth)
//     => J_j = [ w g | X_j x g ],   J_i = -(J_j Adj(G_ij))  (the sign is applied where J_i is used),   J_z = g . t_ij
struct EdgeFactor {
  float r[2], w[2], Jz[2], Ji[12], Jj[12];
};

__device__ __forceinline__ void fastba_factor(const float* Pi, const float* Pj, float px, float py, float pd, float tx,
                                              float ty, float wx, float wy, float fx, float fy, float cx, float cy,
                                              EdgeFactor& o) {
  float t[3], q[4];
  se3_between_raw(Pi, Pj, t, q);
  const float ray[4] = {(px - cx) / fx, (py - cy) / fy, 1.0f, pd};
  float Xj[4];
  lt_act4_loaded(t, q, ray, Xj);
  const float Z = Xj[2];
  const float d = (Z >= 0.2f) ? 1.0f / Z : 0.0f;
  const float u = fx * (Xj[0] / Z) + cx, v = fy * (Xj[1] / Z) + cy;
  o.r[0] = tx - u;
  o.r[1] = ty - v;
  const bool ok = (sqrtf(o.r[0] * o.r[0] + o.r[1] * o.r[1]) < 128.f) && (Z > 0.2f) && (u > -64.f) && (v > -64.f) &&
                  (u < 2 * cx + 64.f) && (v < 2 * cy + 64.f);
  o.w[0] = ok ? wx : 0.0f;
  o.w[1] = ok ? wy : 0.0f;
  const float g[2][3] = {{fx * d, 0.0f, -fx * Xj[0] * d * d}, {0.0f, fy * d, -fy * Xj[1] * d * d}};
#pragma unroll
  for (int row = 0; row < 2; row++) {
    float* Jj = o.Jj + 6 * row;
#pragma unroll
    for (int a = 0; a < 3; a++) Jj[a] = Xj[3] * g[row][a];
    cross3(Xj, g[row], Jj + 3);
    se3_row_times_adj_raw(t, q, Jj, o.Ji + 6 * row);
    o.Jz[row] = g[row][0] * t[0] + g[row][1] * t[1] + g[row][2] * t[2];
  }
}

}  // namespace cdv
