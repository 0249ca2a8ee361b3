/*
 * oracle/lie_impl.h -- TEST INFRASTRUCTURE ONLY (CPU oracle, never shipped, never timed as product).
 *
 * SO3 / SE3 group arithmetic restated from the reference's Eigen-templated headers:
 *   cdvslam/lietorch/include/so3.h:30-221, se3.h:30-217, common.h:7 (EPS = 1e-6).
 * Eigen 3.4.0 is not vendored in the reference tree, so the Eigen primitives used there
 * (Quaternion::normalize, operator*, toRotationMatrix) are restated from their published
 * definitions.  This file is included twice (REAL = float, REAL = double).
 *
 * Parity status: pinned only by the reference's own property tests
 * (cdvslam/lietorch/run_tests.py:16-52: Log(Exp(x))==x, X*X^-1==I, adjoint identity,
 * act vs 4x4 matrix), replicated in tests/test_oracle_lie.py.  No known-answer vectors exist
 * in the reference.
 *
 * Layouts: SO3 data = quaternion (x,y,z,w); SE3 data = (tx,ty,tz, qx,qy,qz,qw);
 * SE3 tangent = (tau[3], phi[3]).
 */

#ifndef REAL
#error "define REAL and SUF(name) before including lie_impl.h"
#endif

#define LIE_EPS 1e-6
#define LIE_PI 3.14159265358979323846

/* ---- small helpers -------------------------------------------------------------------- */

static inline void SUF(cross3)(const REAL *a, const REAL *b, REAL *o) {
  REAL x = a[1] * b[2] - a[2] * b[1];
  REAL y = a[2] * b[0] - a[0] * b[2];
  REAL z = a[0] * b[1] - a[1] * b[0];
  o[0] = x; o[1] = y; o[2] = z;
}

/* Eigen::Quaternion::normalize(): coeffs /= sqrt(squaredNorm)   (so3.h:30-37 ctor calls it) */
static inline void SUF(quat_load)(const REAL *d, REAL *q) {
  REAL n2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3];
  REAL n = (REAL)sqrt((double)n2);
  if (sizeof(REAL) == 4) n = (REAL)sqrtf((float)n2);
  q[0] = d[0] / n; q[1] = d[1] / n; q[2] = d[2] / n; q[3] = d[3] / n;
}

/* Eigen quaternion product a*b, coefficients stored (x,y,z,w) */
static inline void SUF(quat_mul_raw)(const REAL *a, const REAL *b, REAL *o) {
  REAL w = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
  REAL x = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  REAL y = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
  REAL z = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
  o[0] = x; o[1] = y; o[2] = z; o[3] = w;
}

/* so3.h:54-59  rotation of a point: uv = q.vec x p; uv += uv; p + w*uv + q.vec x uv */
static inline void SUF(so3_rot)(const REAL *q, const REAL *p, REAL *o) {
  REAL uv[3], c[3];
  SUF(cross3)(q, p, uv);
  uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
  SUF(cross3)(q, uv, c);
  REAL x = p[0] + q[3] * uv[0] + c[0];
  REAL y = p[1] + q[3] * uv[1] + c[1];
  REAL z = p[2] + q[3] * uv[2] + c[2];
  o[0] = x; o[1] = y; o[2] = z;
}

/* Eigen toRotationMatrix, row-major 3x3 */
static inline void SUF(quat_to_R)(const REAL *q, REAL *R) {
  REAL tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
  REAL twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
  REAL txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
  REAL tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
  R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

/* ---- SO3 --------------------------------------------------------------------------------- */

/* so3.h:153-170 */
static inline void SUF(so3_exp)(const REAL *phi, REAL *q) {
  REAL theta2 = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2];
  REAL theta = (sizeof(REAL) == 4) ? (REAL)sqrtf((float)theta2) : (REAL)sqrt((double)theta2);
  REAL imag, real;
  if (theta < LIE_EPS) {
    REAL theta4 = theta2 * theta2;
    imag = (REAL)0.5 - (REAL)(1.0 / 48.0) * theta2 + (REAL)(1.0 / 3840.0) * theta4;
    real = (REAL)1 - (REAL)(1.0 / 8.0) * theta2 + (REAL)(1.0 / 384.0) * theta4;
  } else {
    /* so3.h:164-165: `.5 * theta` is a double expression for either Scalar */
    imag = (REAL)(sin(.5 * (double)theta) / (double)theta);
    real = (REAL)cos(.5 * (double)theta);
  }
  REAL raw[4] = {imag * phi[0], imag * phi[1], imag * phi[2], real};
  SUF(quat_load)(raw, q); /* SO3(q) ctor normalises */
}

/* so3.h:115-151 atan-based log */
static inline void SUF(so3_log)(const REAL *qd, REAL *phi) {
  REAL q[4];
  SUF(quat_load)(qd, q);
  REAL sq = q[0] * q[0] + q[1] * q[1] + q[2] * q[2];
  REAL w = q[3];
  REAL f;
  if (sq < LIE_EPS * LIE_EPS) {
    REAL w2 = w * w;
    f = (REAL)2 / w - (REAL)(2.0 / 3.0) * sq / (w * w2);
  } else {
    REAL n = (sizeof(REAL) == 4) ? (REAL)sqrtf((float)sq) : (REAL)sqrt((double)sq);
    REAL aw = w < 0 ? -w : w;
    if (aw < LIE_EPS) {
      f = (w > 0) ? (REAL)LIE_PI / n : -(REAL)LIE_PI / n;
    } else {
      REAL at = (sizeof(REAL) == 4) ? (REAL)atanf((float)(n / w)) : (REAL)atan((double)(n / w));
      f = (REAL)2 * at / n;
    }
  }
  phi[0] = f * q[0]; phi[1] = f * q[1]; phi[2] = f * q[2];
}

static inline void SUF(hat3)(const REAL *p, REAL *M) {
  M[0] = 0;     M[1] = -p[2]; M[2] = p[1];
  M[3] = p[2];  M[4] = 0;     M[5] = -p[0];
  M[6] = -p[1]; M[7] = p[0];  M[8] = 0;
}

static inline void SUF(mat3_mul)(const REAL *A, const REAL *B, REAL *C) {
  REAL T[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      REAL s = 0;
      for (int k = 0; k < 3; k++) s += A[3 * i + k] * B[3 * k + j];
      T[3 * i + j] = s;
    }
  for (int i = 0; i < 9; i++) C[i] = T[i];
}

static inline void SUF(mat3_vec)(const REAL *A, const REAL *v, REAL *o) {
  REAL x = A[0] * v[0] + A[1] * v[1] + A[2] * v[2];
  REAL y = A[3] * v[0] + A[4] * v[1] + A[5] * v[2];
  REAL z = A[6] * v[0] + A[7] * v[1] + A[8] * v[2];
  o[0] = x; o[1] = y; o[2] = z;
}

/* so3.h:172-191 */
static inline void SUF(so3_left_jacobian)(const REAL *phi, REAL *J) {
  REAL Phi[9], Phi2[9];
  SUF(hat3)(phi, Phi);
  SUF(mat3_mul)(Phi, Phi, Phi2);
  REAL theta2 = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2];
  REAL theta = (sizeof(REAL) == 4) ? (REAL)sqrtf((float)theta2) : (REAL)sqrt((double)theta2);
  REAL c1, c2;
  if (theta < LIE_EPS) {
    c1 = (REAL)(1.0 / 2.0) - (REAL)(1.0 / 24.0) * theta2;
    c2 = (REAL)(1.0 / 6.0) - (REAL)(1.0 / 120.0) * theta2;
  } else if (sizeof(REAL) == 4) {
    c1 = (REAL)((1.0 - cosf((float)theta)) / theta2);
    c2 = (REAL)((theta - sinf((float)theta)) / (theta2 * theta));
  } else {
    c1 = (REAL)((1.0 - cos((double)theta)) / theta2);
    c2 = (REAL)((theta - sin((double)theta)) / (theta2 * theta));
  }
  for (int i = 0; i < 9; i++) J[i] = ((i % 4 == 0) ? (REAL)1 : (REAL)0) + c1 * Phi[i] + c2 * Phi2[i];
}

/* so3.h:193-210 */
static inline void SUF(so3_left_jacobian_inverse)(const REAL *phi, REAL *J) {
  REAL Phi[9], Phi2[9];
  SUF(hat3)(phi, Phi);
  SUF(mat3_mul)(Phi, Phi, Phi2);
  REAL theta2 = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2];
  REAL theta = (sizeof(REAL) == 4) ? (REAL)sqrtf((float)theta2) : (REAL)sqrt((double)theta2);
  REAL half = (REAL)0.5 * theta;
  REAL c2;
  if (theta < LIE_EPS) {
    c2 = (REAL)(1.0 / 12.0);
  } else if (sizeof(REAL) == 4) {
    c2 = ((REAL)1 - theta * (REAL)cosf((float)half) / ((REAL)2 * (REAL)sinf((float)half))) / (theta * theta);
  } else {
    c2 = ((REAL)1 - theta * (REAL)cos((double)half) / ((REAL)2 * (REAL)sin((double)half))) / (theta * theta);
  }
  for (int i = 0; i < 9; i++) J[i] = ((i % 4 == 0) ? (REAL)1 : (REAL)0) + (REAL)(-0.5) * Phi[i] + c2 * Phi2[i];
}

/* ---- SE3 (se3.h) --------------------------------------------------------------------------- */

/* se3.h:36 ctor: translation(data), so3(data+3) [normalised] */
static inline void SUF(se3_load)(const REAL *d, REAL *t, REAL *q) {
  t[0] = d[0]; t[1] = d[1]; t[2] = d[2];
  SUF(quat_load)(d + 3, q);
}

/* se3.h:38-40: SE3(so3.inv(), -(so3.inv()*translation)); so3.inv() = SO3(conjugate) [normalised] */
static inline void SUF(se3_inv)(const REAL *X, REAL *Y) {
  REAL t[3], q[4], qc[4], qi[4], r[3];
  SUF(se3_load)(X, t, q);
  qc[0] = -q[0]; qc[1] = -q[1]; qc[2] = -q[2]; qc[3] = q[3];
  SUF(quat_load)(qc, qi);
  SUF(so3_rot)(qi, t, r);
  Y[0] = -r[0]; Y[1] = -r[1]; Y[2] = -r[2];
  Y[3] = qi[0]; Y[4] = qi[1]; Y[5] = qi[2]; Y[6] = qi[3];
}

/* se3.h:47-49: SE3(so3*other.so3 [normalised], translation + so3*other.translation) */
static inline void SUF(se3_mul)(const REAL *X, const REAL *Y, REAL *Z) {
  REAL t1[3], q1[4], t2[3], q2[4], qr[4], q[4], r[3];
  SUF(se3_load)(X, t1, q1);
  SUF(se3_load)(Y, t2, q2);
  SUF(quat_mul_raw)(q1, q2, qr);
  SUF(quat_load)(qr, q);
  SUF(so3_rot)(q1, t2, r);
  Z[0] = t1[0] + r[0]; Z[1] = t1[1] + r[1]; Z[2] = t1[2] + r[2];
  Z[3] = q[0]; Z[4] = q[1]; Z[5] = q[2]; Z[6] = q[3];
}

/* se3.h:51-53 */
static inline void SUF(se3_act)(const REAL *X, const REAL *p, REAL *o) {
  REAL t[3], q[4], r[3];
  SUF(se3_load)(X, t, q);
  SUF(so3_rot)(q, p, r);
  o[0] = r[0] + t[0]; o[1] = r[1] + t[1]; o[2] = r[2] + t[2];
}

/* se3.h:55-58 */
static inline void SUF(se3_act4)(const REAL *X, const REAL *p, REAL *o) {
  REAL t[3], q[4], r[3];
  SUF(se3_load)(X, t, q);
  SUF(so3_rot)(q, p, r);
  REAL w = p[3];
  o[0] = r[0] + t[0] * w; o[1] = r[1] + t[1] * w; o[2] = r[2] + t[2] * w; o[3] = w;
}

/* se3.h:60-69: Ad = [R, tx*R; 0, R]  (row-major 6x6) */
static inline void SUF(se3_Adj_matrix)(const REAL *X, REAL *Ad) {
  REAL t[3], q[4], R[9], T[9], TR[9];
  SUF(se3_load)(X, t, q);
  SUF(quat_to_R)(q, R);
  SUF(hat3)(t, T);
  SUF(mat3_mul)(T, R, TR);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      Ad[6 * i + j] = R[3 * i + j];
      Ad[6 * i + 3 + j] = TR[3 * i + j];
      Ad[6 * (i + 3) + j] = 0;
      Ad[6 * (i + 3) + 3 + j] = R[3 * i + j];
    }
}

/* se3.h:84-86 / 88-90 */
static inline void SUF(se3_adj)(const REAL *X, const REAL *a, REAL *b) {
  REAL Ad[36], o[6];
  SUF(se3_Adj_matrix)(X, Ad);
  for (int i = 0; i < 6; i++) {
    REAL s = 0;
    for (int j = 0; j < 6; j++) s += Ad[6 * i + j] * a[j];
    o[i] = s;
  }
  for (int i = 0; i < 6; i++) b[i] = o[i];
}

static inline void SUF(se3_adjT)(const REAL *X, const REAL *a, REAL *b) {
  REAL Ad[36], o[6];
  SUF(se3_Adj_matrix)(X, Ad);
  for (int i = 0; i < 6; i++) {
    REAL s = 0;
    for (int j = 0; j < 6; j++) s += Ad[6 * j + i] * a[j];
    o[i] = s;
  }
  for (int i = 0; i < 6; i++) b[i] = o[i];
}

/* se3.h:71-76 row-major 4x4 */
static inline void SUF(se3_matrix)(const REAL *X, REAL *M) {
  REAL t[3], q[4], R[9];
  SUF(se3_load)(X, t, q);
  SUF(quat_to_R)(q, R);
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) M[4 * i + j] = R[3 * i + j];
    M[4 * i + 3] = t[i];
  }
  M[12] = 0; M[13] = 0; M[14] = 0; M[15] = 1;
}

/* se3.h:134-142 */
static inline void SUF(se3_exp)(const REAL *xi, REAL *X) {
  REAL q[4], J[9], t[3];
  SUF(so3_exp)(xi + 3, q);
  SUF(so3_left_jacobian)(xi + 3, J);
  SUF(mat3_vec)(J, xi, t);
  X[0] = t[0]; X[1] = t[1]; X[2] = t[2];
  X[3] = q[0]; X[4] = q[1]; X[5] = q[2]; X[6] = q[3];
}

/* se3.h:124-132 */
static inline void SUF(se3_log)(const REAL *X, REAL *xi) {
  REAL phi[3], Vinv[9], tau[3];
  SUF(so3_log)(X + 3, phi);
  SUF(so3_left_jacobian_inverse)(phi, Vinv);
  SUF(mat3_vec)(Vinv, X, tau);
  xi[0] = tau[0]; xi[1] = tau[1]; xi[2] = tau[2];
  xi[3] = phi[0]; xi[4] = phi[1]; xi[5] = phi[2];
}

/* SO3 group-level wrappers on raw data (each load normalises, as the SO3 ctor does) */
static inline void SUF(so3_inv)(const REAL *X, REAL *Y) {
  REAL q[4], qc[4];
  SUF(quat_load)(X, q);
  qc[0] = -q[0]; qc[1] = -q[1]; qc[2] = -q[2]; qc[3] = q[3];
  SUF(quat_load)(qc, Y);
}
static inline void SUF(so3_mul)(const REAL *X, const REAL *Y, REAL *Z) {
  REAL a[4], b[4], r[4];
  SUF(quat_load)(X, a);
  SUF(quat_load)(Y, b);
  SUF(quat_mul_raw)(a, b, r);
  SUF(quat_load)(r, Z);
}
static inline void SUF(so3_act)(const REAL *X, const REAL *p, REAL *o) {
  REAL q[4];
  SUF(quat_load)(X, q);
  SUF(so3_rot)(q, p, o);
}
static inline void SUF(so3_act4)(const REAL *X, const REAL *p, REAL *o) {
  REAL q[4];
  SUF(quat_load)(X, q);
  SUF(so3_rot)(q, p, o);
  o[3] = p[3];
}
static inline void SUF(so3_adj)(const REAL *X, const REAL *a, REAL *b) {
  REAL q[4], R[9];
  SUF(quat_load)(X, q);
  SUF(quat_to_R)(q, R);
  SUF(mat3_vec)(R, a, b);
}
static inline void SUF(so3_adjT)(const REAL *X, const REAL *a, REAL *b) {
  REAL q[4], R[9], Rt[9];
  SUF(quat_load)(X, q);
  SUF(quat_to_R)(q, R);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) Rt[3 * i + j] = R[3 * j + i];
  SUF(mat3_vec)(Rt, a, b);
}
