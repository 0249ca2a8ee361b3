// cdv_se3.h -- device-side SO3 / SE3 arithmetic (gfx950).
//
// Two families, because the reference has two:
//  * lt_*  : lietorch semantics (cdvslam/lietorch/include/so3.h, se3.h): every load of a group
//            element re-normalises the quaternion (so3.h:30-37); used by pops.transform and the
//            lietorch_backends ops.
//  * fb_*  : fastba's own float helpers (cdvslam/fastba/ba_cuda.cu:36-174): no normalisation,
//            its own Taylor thresholds.
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

template <typename T>
__device__ __forceinline__ void cross3(const T* a, const T* b, T* o) {
  T x = a[1] * b[2] - a[2] * b[1];
  T y = a[2] * b[0] - a[0] * b[2];
  T z = a[0] * b[1] - a[1] * b[0];
  o[0] = x; o[1] = y; o[2] = z;
}

// ---- lietorch family ---------------------------------------------------------------------------

template <typename T>
__device__ __forceinline__ void lt_quat_load(const T* d, T* q) {  // so3.h:30-37
  T n = t_sqrt<T>(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]);
  q[0] = d[0] / n; q[1] = d[1] / n; q[2] = d[2] / n; q[3] = d[3] / n;
}

template <typename T>
__device__ __forceinline__ void lt_quat_mul(const T* a, const T* b, T* o) {
  T w = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
  T x = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  T y = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
  T z = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
  o[0] = x; o[1] = y; o[2] = z; o[3] = w;
}

template <typename T>
__device__ __forceinline__ void lt_rot(const T* q, const T* p, T* o) {  // so3.h:54-59
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
  t[0] = d[0]; t[1] = d[1]; t[2] = d[2];
  lt_quat_load(d + 3, q);
}

template <typename T>
__device__ __forceinline__ void lt_se3_inv(const T* X, T* Y) {  // se3.h:38-40
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

// ---- fastba family (float only) ------------------------------------------------------------------

__device__ __forceinline__ void fb_actSO3(const float* q, const float* X, float* Y) {  // ba_cuda.cu:36-46
  float uv[3];
  uv[0] = 2.0f * (q[1] * X[2] - q[2] * X[1]);
  uv[1] = 2.0f * (q[2] * X[0] - q[0] * X[2]);
  uv[2] = 2.0f * (q[0] * X[1] - q[1] * X[0]);
  float y0 = X[0] + q[3] * uv[0] + (q[1] * uv[2] - q[2] * uv[1]);
  float y1 = X[1] + q[3] * uv[1] + (q[2] * uv[0] - q[0] * uv[2]);
  float y2 = X[2] + q[3] * uv[2] + (q[0] * uv[1] - q[1] * uv[0]);
  Y[0] = y0; Y[1] = y1; Y[2] = y2;
}

__device__ __forceinline__ void fb_actSE3(const float* t, const float* q, const float* X, float* Y) {  // :48-55
  fb_actSO3(q, X, Y);
  Y[3] = X[3];
  Y[0] += X[3] * t[0];
  Y[1] += X[3] * t[1];
  Y[2] += X[3] * t[2];
}

__device__ __forceinline__ void fb_adjSE3(const float* t, const float* q, const float* X, float* Y) {  // :57-72
  float qinv[4] = {-q[0], -q[1], -q[2], q[3]};
  fb_actSO3(qinv, &X[0], &Y[0]);
  fb_actSO3(qinv, &X[3], &Y[3]);
  float u[3], v[3];
  u[0] = t[2] * X[1] - t[1] * X[2];
  u[1] = t[0] * X[2] - t[2] * X[0];
  u[2] = t[1] * X[0] - t[0] * X[1];
  fb_actSO3(qinv, u, v);
  Y[3] += v[0];
  Y[4] += v[1];
  Y[5] += v[2];
}

__device__ __forceinline__ void fb_relSE3(const float* ti, const float* qi, const float* tj, const float* qj,
                                          float* tij, float* qij) {  // :74-85
  qij[0] = -qj[3] * qi[0] + qj[0] * qi[3] - qj[1] * qi[2] + qj[2] * qi[1];
  qij[1] = -qj[3] * qi[1] + qj[1] * qi[3] - qj[2] * qi[0] + qj[0] * qi[2];
  qij[2] = -qj[3] * qi[2] + qj[2] * qi[3] - qj[0] * qi[1] + qj[1] * qi[0];
  qij[3] = qj[3] * qi[3] + qj[0] * qi[0] + qj[1] * qi[1] + qj[2] * qi[2];
  fb_actSO3(qij, ti, tij);
  tij[0] = tj[0] - tij[0];
  tij[1] = tj[1] - tij[1];
  tij[2] = tj[2] - tij[2];
}

__device__ __forceinline__ void fb_expSO3(const float* phi, float* q) {  // :89-112
  float theta_sq = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2];
  float theta_p4 = theta_sq * theta_sq;
  float theta = sqrtf(theta_sq);
  float imag, real;
  if (theta_sq < 1e-8f) {
    imag = 0.5f - (1.0f / 48.0f) * theta_sq + (1.0f / 3840.0f) * theta_p4;
    real = 1.0f - (1.0f / 8.0f) * theta_sq + (1.0f / 384.0f) * theta_p4;
  } else {
    imag = sinf(0.5f * theta) / theta;
    real = cosf(0.5f * theta);
  }
  q[0] = imag * phi[0]; q[1] = imag * phi[1]; q[2] = imag * phi[2]; q[3] = real;
}

__device__ __forceinline__ void fb_crossInplace(const float* a, float* b) {  // :114-125
  float x[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
  b[0] = x[0]; b[1] = x[1]; b[2] = x[2];
}

__device__ __forceinline__ void fb_expSE3(const float* xi, float* t, float* q) {  // :127-154
  fb_expSO3(xi + 3, q);
  float tau[3] = {xi[0], xi[1], xi[2]};
  float phi[3] = {xi[3], xi[4], xi[5]};
  float theta_sq = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2];
  float theta = sqrtf(theta_sq);
  t[0] = tau[0]; t[1] = tau[1]; t[2] = tau[2];
  if (theta > 1e-4f) {
    float a = (1 - cosf(theta)) / theta_sq;
    fb_crossInplace(phi, tau);
    t[0] += a * tau[0]; t[1] += a * tau[1]; t[2] += a * tau[2];
    float b = (theta - sinf(theta)) / (theta * theta_sq);
    fb_crossInplace(phi, tau);
    t[0] += b * tau[0]; t[1] += b * tau[1]; t[2] += b * tau[2];
  }
}

__device__ __forceinline__ void fb_retrSE3(const float* xi, const float* t, const float* q, float* t1,
                                           float* q1) {  // :157-174
  float dt[3] = {0, 0, 0};
  float dq[4] = {0, 0, 0, 1};
  fb_expSE3(xi, dt, dq);
  q1[0] = dq[3] * q[0] + dq[0] * q[3] + dq[1] * q[2] - dq[2] * q[1];
  q1[1] = dq[3] * q[1] + dq[1] * q[3] + dq[2] * q[0] - dq[0] * q[2];
  q1[2] = dq[3] * q[2] + dq[2] * q[3] + dq[0] * q[1] - dq[1] * q[0];
  q1[3] = dq[3] * q[3] - dq[0] * q[0] - dq[1] * q[1] - dq[2] * q[2];
  fb_actSO3(dq, t, t1);
  t1[0] += dt[0]; t1[1] += dt[1]; t1[2] += dt[2];
}

}  // namespace cdv
