/*
 * oracle/fastba_impl.h -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * Sequential CPU restatement of the reference's fastba bundle adjustment:
 *   cdvslam/fastba/ba_cuda.cu:36-85   actSO3 / actSE3 / adjSE3 / relSE3
 *   cdvslam/fastba/ba_cuda.cu:89-174  expSO3 / expSE3 / retrSE3
 *   cdvslam/fastba/ba_cuda.cu:178-229 pose_retr_kernel / patch_retr_kernel
 *   cdvslam/fastba/ba_cuda.cu:232-405 reprojection_residuals_and_hessian
 *   cdvslam/fastba/ba_cuda.cu:462-611 cuda_ba host loop (dense-E path, eff_impl == false)
 * and of pops.transform (cdvslam/projective_ops.py:19-113).
 *
 * The reference accumulates with unordered float atomics; this oracle accumulates in edge
 * order.  The dense Schur solve replaces ATen matmul / linalg_cholesky_ex / cholesky_solve
 * (pytorch 2.3.1, not in the reference tree) by a textbook lower Cholesky.
 * Included twice (REAL=float mirrors the kernel's precision, REAL=double is the "truth").
 *
 * Parity status: UNPINNED by the reference (it holds no golden vectors / tests for fastba).
 * Pinned instead by cross-agreement with the reference's own cdvslam/ba.py executed under a
 * shimmed import (tests/golden/make_golden.py) where the two algorithms' gates coincide.
 */

#ifndef REAL
#error "define REAL and SUF"
#endif

/* ba_cuda.cu:36-46.  The `2.0 *` literal makes the product a double expression. */
static inline void SUF(fb_actSO3)(const REAL *q, const REAL *X, REAL *Y) {
  REAL uv[3];
  uv[0] = (REAL)(2.0 * (q[1] * X[2] - q[2] * X[1]));
  uv[1] = (REAL)(2.0 * (q[2] * X[0] - q[0] * X[2]));
  uv[2] = (REAL)(2.0 * (q[0] * X[1] - q[1] * X[0]));
  REAL y0 = X[0] + q[3] * uv[0] + (q[1] * uv[2] - q[2] * uv[1]);
  REAL y1 = X[1] + q[3] * uv[1] + (q[2] * uv[0] - q[0] * uv[2]);
  REAL y2 = X[2] + q[3] * uv[2] + (q[0] * uv[1] - q[1] * uv[0]);
  Y[0] = y0; Y[1] = y1; Y[2] = y2;
}

/* ba_cuda.cu:48-55 */
static inline void SUF(fb_actSE3)(const REAL *t, const REAL *q, const REAL *X, REAL *Y) {
  SUF(fb_actSO3)(q, X, Y);
  Y[3] = X[3];
  Y[0] += X[3] * t[0];
  Y[1] += X[3] * t[1];
  Y[2] += X[3] * t[2];
}

/* ba_cuda.cu:57-72 */
static inline void SUF(fb_adjSE3)(const REAL *t, const REAL *q, const REAL *X, REAL *Y) {
  REAL qinv[4] = {-q[0], -q[1], -q[2], q[3]};
  SUF(fb_actSO3)(qinv, &X[0], &Y[0]);
  SUF(fb_actSO3)(qinv, &X[3], &Y[3]);
  REAL u[3], v[3];
  u[0] = t[2] * X[1] - t[1] * X[2];
  u[1] = t[0] * X[2] - t[2] * X[0];
  u[2] = t[1] * X[0] - t[0] * X[1];
  SUF(fb_actSO3)(qinv, u, v);
  Y[3] += v[0];
  Y[4] += v[1];
  Y[5] += v[2];
}

/* ba_cuda.cu:74-85 (no quaternion normalisation) */
static inline void SUF(fb_relSE3)(const REAL *ti, const REAL *qi, const REAL *tj, const REAL *qj, REAL *tij,
                                  REAL *qij) {
  qij[0] = -qj[3] * qi[0] + qj[0] * qi[3] - qj[1] * qi[2] + qj[2] * qi[1];
  qij[1] = -qj[3] * qi[1] + qj[1] * qi[3] - qj[2] * qi[0] + qj[0] * qi[2];
  qij[2] = -qj[3] * qi[2] + qj[2] * qi[3] - qj[0] * qi[1] + qj[1] * qi[0];
  qij[3] = qj[3] * qi[3] + qj[0] * qi[0] + qj[1] * qi[1] + qj[2] * qi[2];
  SUF(fb_actSO3)(qij, ti, tij);
  tij[0] = tj[0] - tij[0];
  tij[1] = tj[1] - tij[1];
  tij[2] = tj[2] - tij[2];
}

/* ba_cuda.cu:89-112 */
static inline void SUF(fb_expSO3)(const REAL *phi, REAL *q) {
  REAL theta_sq = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2];
  REAL theta_p4 = theta_sq * theta_sq;
  REAL theta = (sizeof(REAL) == 4) ? (REAL)sqrtf((float)theta_sq) : (REAL)sqrt((double)theta_sq);
  REAL imag, real;
  if (theta_sq < 1e-8) {
    imag = (REAL)(0.5 - (1.0 / 48.0) * theta_sq + (1.0 / 3840.0) * theta_p4);
    real = (REAL)(1.0 - (1.0 / 8.0) * theta_sq + (1.0 / 384.0) * theta_p4);
  } else if (sizeof(REAL) == 4) {
    imag = (REAL)(sinf((float)(0.5 * theta)) / theta);
    real = (REAL)cosf((float)(0.5 * theta));
  } else {
    imag = (REAL)(sin(0.5 * (double)theta) / theta);
    real = (REAL)cos(0.5 * (double)theta);
  }
  q[0] = imag * phi[0];
  q[1] = imag * phi[1];
  q[2] = imag * phi[2];
  q[3] = real;
}

static inline void SUF(fb_crossInplace)(const REAL *a, REAL *b) {
  REAL x[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
  b[0] = x[0]; b[1] = x[1]; b[2] = x[2];
}

/* ba_cuda.cu:127-154 */
static inline void SUF(fb_expSE3)(const REAL *xi, REAL *t, REAL *q) {
  SUF(fb_expSO3)(xi + 3, q);
  REAL tau[3] = {xi[0], xi[1], xi[2]};
  REAL phi[3] = {xi[3], xi[4], xi[5]};
  REAL theta_sq = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2];
  REAL theta = (sizeof(REAL) == 4) ? (REAL)sqrtf((float)theta_sq) : (REAL)sqrt((double)theta_sq);
  t[0] = tau[0]; t[1] = tau[1]; t[2] = tau[2];
  if (theta > 1e-4) {
    REAL a, b;
    if (sizeof(REAL) == 4) {
      a = (1 - (REAL)cosf((float)theta)) / theta_sq;
    } else {
      a = (1 - (REAL)cos((double)theta)) / theta_sq;
    }
    SUF(fb_crossInplace)(phi, tau);
    t[0] += a * tau[0]; t[1] += a * tau[1]; t[2] += a * tau[2];
    if (sizeof(REAL) == 4) {
      b = (theta - (REAL)sinf((float)theta)) / (theta * theta_sq);
    } else {
      b = (theta - (REAL)sin((double)theta)) / (theta * theta_sq);
    }
    SUF(fb_crossInplace)(phi, tau);
    t[0] += b * tau[0]; t[1] += b * tau[1]; t[2] += b * tau[2];
  }
}

/* ba_cuda.cu:157-174 */
static inline void SUF(fb_retrSE3)(const REAL *xi, const REAL *t, const REAL *q, REAL *t1, REAL *q1) {
  REAL dt[3] = {0, 0, 0};
  REAL dq[4] = {0, 0, 0, 1};
  SUF(fb_expSE3)(xi, dt, dq);
  q1[0] = dq[3] * q[0] + dq[0] * q[3] + dq[1] * q[2] - dq[2] * q[1];
  q1[1] = dq[3] * q[1] + dq[1] * q[3] + dq[2] * q[0] - dq[0] * q[2];
  q1[2] = dq[3] * q[2] + dq[2] * q[3] + dq[0] * q[1] - dq[1] * q[0];
  q1[3] = dq[3] * q[3] - dq[0] * q[0] - dq[1] * q[1] - dq[2] * q[2];
  SUF(fb_actSO3)(dq, t, t1);
  t1[0] += dt[0]; t1[1] += dt[1]; t1[2] += dt[2];
}

/* Per-edge residual / Jacobian rows, ba_cuda.cu:261-342.
 * Out: r[2], w[2] (mask applied), Jz[2], Ji[2][6], Jj[2][6].  Returns the in_bounds mask. */
static inline int SUF(fb_edge)(const REAL *poses, const REAL *patches, const REAL *intr, const REAL *target,
                               const REAL *weight, long ix, long jx, long kx, int P, long n, REAL *r, REAL *w,
                               REAL *Jz, REAL *Ji, REAL *Jj) {
  const REAL fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];
  const REAL *pi = poses + 7 * ix, *pj = poses + 7 * jx;
  REAL ti[3] = {pi[0], pi[1], pi[2]}, tj[3] = {pj[0], pj[1], pj[2]};
  REAL qi[4] = {pi[3], pi[4], pi[5], pi[6]}, qj[4] = {pj[3], pj[4], pj[5], pj[6]};
  const REAL *pk = patches + kx * 3 * P * P;
  const int c = (P > 1) ? (1 * P + 1) : 0; /* [.][1][1]  (ba_cuda.cu:282-285) */
  REAL Xi[4], Xj[4];
  Xi[0] = (pk[0 * P * P + c] - cx) / fx;
  Xi[1] = (pk[1 * P * P + c] - cy) / fy;
  Xi[2] = 1.0;
  Xi[3] = pk[2 * P * P + c];
  REAL tij[3], qij[4];
  SUF(fb_relSE3)(ti, qi, tj, qj, tij, qij);
  SUF(fb_actSE3)(tij, qij, Xi, Xj);
  const REAL X = Xj[0], Y = Xj[1], Z = Xj[2], W = Xj[3];
  const REAL d = (Z >= 0.2) ? (REAL)(1.0 / Z) : (REAL)0.0;
  const REAL d2 = d * d;
  const REAL x1 = fx * (X / Z) + cx;
  const REAL y1 = fy * (Y / Z) + cy;
  const REAL rx = target[2 * n + 0] - x1;
  const REAL ry = target[2 * n + 1] - y1;
  REAL nr = (sizeof(REAL) == 4) ? (REAL)sqrtf((float)(rx * rx + ry * ry)) : (REAL)sqrt((double)(rx * rx + ry * ry));
  const int in_bounds =
      (nr < 128) && (Z > 0.2) && (x1 > -64) && (y1 > -64) && (x1 < 2 * cx + 64) && (y1 < 2 * cy + 64);
  const REAL mask = in_bounds ? (REAL)1.0 : (REAL)0.0;

  r[0] = rx;
  w[0] = mask * weight[2 * n + 0];
  Jz[0] = fx * (tij[0] * d - tij[2] * X * d2);
  REAL *J0 = Jj;
  J0[0] = fx * W * d;
  J0[1] = 0;
  J0[2] = -fx * X * W * d2;
  J0[3] = -fx * X * Y * d2;
  J0[4] = fx * (1 + X * X * d2);
  J0[5] = -fx * Y * d;
  r[1] = ry;
  w[1] = mask * weight[2 * n + 1];
  Jz[1] = fy * (tij[1] * d - tij[2] * Y * d2);
  REAL *J1 = Jj + 6;
  J1[0] = 0;
  J1[1] = fy * W * d;
  J1[2] = -fy * Y * W * d2;
  J1[3] = -fy * (1 + Y * Y * d2);
  J1[4] = fy * X * Y * d2;
  J1[5] = fy * X * d;
  SUF(fb_adjSE3)(tij, qij, J0, Ji);
  SUF(fb_adjSE3)(tij, qij, J1, Ji + 6);
  return in_bounds;
}

/* lower Cholesky A = L L^T in place (n x n row-major); returns 0 on success, k+1 if pivot k <= 0 */
static int SUF(chol_lower)(REAL *A, int n) {
  int info = 0;
  for (int j = 0; j < n; j++) {
    REAL s = A[j * n + j];
    for (int k = 0; k < j; k++) s -= A[j * n + k] * A[j * n + k];
    if (!(s > 0) && !info) info = j + 1;
    REAL d = (sizeof(REAL) == 4) ? (REAL)sqrtf((float)s) : (REAL)sqrt((double)s);
    A[j * n + j] = d;
    for (int i = j + 1; i < n; i++) {
      REAL t = A[i * n + j];
      for (int k = 0; k < j; k++) t -= A[i * n + k] * A[j * n + k];
      A[i * n + j] = t / d;
    }
  }
  return info;
}

static void SUF(chol_solve)(const REAL *L, int n, REAL *b) {
  for (int i = 0; i < n; i++) {
    REAL s = b[i];
    for (int k = 0; k < i; k++) s -= L[i * n + k] * b[k];
    b[i] = s / L[i * n + i];
  }
  for (int i = n - 1; i >= 0; i--) {
    REAL s = b[i];
    for (int k = i + 1; k < n; k++) s -= L[k * n + i] * b[k];
    b[i] = s / L[i * n + i];
  }
}

/*
 * Assemble B [6N x 6N], E [6N x U], C [U], v [6N], u [U]   (ba_cuda.cu:344-403).
 * ku = inverse index of kk into the sorted unique patch ids.
 */
static void SUF(fb_assemble)(const REAL *poses, const REAL *patches, const REAL *intr, const REAL *target,
                             const REAL *weight, const long *ii, const long *jj, const long *kk, const long *ku,
                             long E_, int P, int t0, int N, long U, REAL *B, REAL *Em, REAL *C, REAL *v, REAL *u,
                             double *r_total) {
  const long n6 = 6L * N;
  for (long a = 0; a < n6 * n6; a++) B[a] = 0;
  for (long a = 0; a < n6 * U; a++) Em[a] = 0;
  for (long a = 0; a < U; a++) { C[a] = 0; u[a] = 0; }
  for (long a = 0; a < n6; a++) v[a] = 0;
  double rt = 0;
  for (long n = 0; n < E_; n++) {
    REAL r[2], w[2], Jz[2], Ji[12], Jj[12];
    SUF(fb_edge)(poses, patches, intr, target, weight, ii[n], jj[n], kk[n], P, n, r, w, Jz, Ji, Jj);
    const long k = ku[n];
    long ix = ii[n] - t0, jx = jj[n] - t0;
    /* the reference only tests >= 0 (ba_cuda.cu:366-376); frames >= t1 would index out of bounds there,
       the oracle (and the HIP path) treat them as fixed too */
    const int fi = (ix >= 0 && ix < N), fj = (jx >= 0 && jx < N);
    for (int row = 0; row < 2; row++) {
      const REAL *Jir = Ji + 6 * row, *Jjr = Jj + 6 * row;
      const REAL wr = w[row] * r[row];
      const REAL wz = w[row] * Jz[row];
      rt += (double)(wr * r[row]);
      for (int i = 0; i < 6; i++) {
        const REAL wJi = w[row] * Jir[i], wJj = w[row] * Jjr[i];
        for (int j = 0; j < 6; j++) {
          if (fi) B[(6 * ix + i) * n6 + 6 * ix + j] += wJi * Jir[j];
          if (fj) B[(6 * jx + i) * n6 + 6 * jx + j] += wJj * Jjr[j];
          if (fi && fj) {
            const REAL ct = -wJi * Jjr[j];
            B[(6 * ix + i) * n6 + 6 * jx + j] += ct;
            B[(6 * jx + j) * n6 + 6 * ix + i] += ct;
          }
        }
      }
      for (int i = 0; i < 6; i++) {
        if (fi) Em[(6 * ix + i) * U + k] += -wz * Jir[i];
        if (fj) Em[(6 * jx + i) * U + k] += wz * Jjr[i];
      }
      for (int i = 0; i < 6; i++) {
        if (fi) v[6 * ix + i] += -wr * Jir[i];
        if (fj) v[6 * jx + i] += wr * Jjr[i];
      }
      C[k] += wz * Jz[row];
      u[k] += wr * Jz[row];
    }
  }
  if (r_total) *r_total = rt;
}

/*
 * Full BA, in place on poses [*,7] and patches [*,3,P,P]   (ba_cuda.cu:462-611, dense path).
 * dbg (optional, may be NULL): receives iteration-0 B,E,C,v,u,S,y,dX,dZ concatenated:
 *   [B 36N^2 | E 6N*U | C U | v 6N | u U | S 36N^2 | y 6N | dX 6N | dZ U]
 * Returns the Cholesky info of the last iteration (0 = ok).
 */
static int SUF(fb_ba)(REAL *poses, REAL *patches, const REAL *intr, const REAL *target, const REAL *weight,
                      REAL lmbda, const long *ii, const long *jj, const long *kk, const long *kx,
                      const long *ku, long E_, long U, int P, int t0, int t1, int iterations, REAL *dbg) {
  const int N = t1 - t0;
  const long n6 = 6L * N;
  REAL *B = (REAL *)calloc((size_t)(n6 * n6 + 1), sizeof(REAL));
  REAL *Em = (REAL *)calloc((size_t)(n6 * U + 1), sizeof(REAL));
  REAL *C = (REAL *)calloc((size_t)U + 1, sizeof(REAL));
  REAL *v = (REAL *)calloc((size_t)n6 + 1, sizeof(REAL));
  REAL *u = (REAL *)calloc((size_t)U + 1, sizeof(REAL));
  REAL *Q = (REAL *)calloc((size_t)U + 1, sizeof(REAL));
  REAL *S = (REAL *)calloc((size_t)(n6 * n6 + 1), sizeof(REAL));
  REAL *y = (REAL *)calloc((size_t)n6 + 1, sizeof(REAL));
  REAL *dZ = (REAL *)calloc((size_t)U + 1, sizeof(REAL));
  int info = 0;
  for (int itr = 0; itr < iterations; itr++) {
    SUF(fb_assemble)(poses, patches, intr, target, weight, ii, jj, kk, ku, E_, P, t0, N, U, B, Em, C, v, u, 0);
    for (long k = 0; k < U; k++) Q[k] = (REAL)(1.0 / (C[k] + lmbda)); /* ba_cuda.cu:548 */
    if (N == 0) {
      for (long k = 0; k < U; k++) dZ[k] = Q[k] * u[k]; /* ba_cuda.cu:550-560 */
    } else {
      /* S = B - (E*Q) E^T ; y = v - (E*Q) u   (ba_cuda.cu:583-587).  The sums run over k in ascending order as a dense
       * matmul row would; columns in which row a of E is exactly zero contribute exactly zero and are skipped (a pose
       * sees a few hundred of the tens of thousands of patches of a global BA), rows run in parallel under OpenMP. */
#ifdef _OPENMP
#pragma omp parallel
#endif
      {
        long *nz = (long *)malloc(sizeof(long) * (size_t)(U + 1));
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 4)
#endif
        for (long a = 0; a < n6; a++) {
          long nnz = 0;
          for (long k = 0; k < U; k++)
            if (Em[a * U + k] != 0) nz[nnz++] = k;
          for (long b = 0; b < n6; b++) {
            REAL s = 0;
            for (long t = 0; t < nnz; t++) { const long k = nz[t]; s += (Em[a * U + k] * Q[k]) * Em[b * U + k]; }
            S[a * n6 + b] = B[a * n6 + b] - s;
          }
          REAL s = 0;
          for (long t = 0; t < nnz; t++) { const long k = nz[t]; s += (Em[a * U + k] * Q[k]) * u[k]; }
          y[a] = v[a] - s;
        }
        free(nz);
      }
      /* S += I * (1e-4 * S + 1.0)   (ba_cuda.cu:589) */
      for (long a = 0; a < n6; a++) S[a * n6 + a] += (REAL)1e-4 * S[a * n6 + a] + (REAL)1.0;
      if (dbg && itr == 0) {
        REAL *p = dbg;
        memcpy(p, B, sizeof(REAL) * n6 * n6); p += n6 * n6;
        memcpy(p, Em, sizeof(REAL) * n6 * U); p += n6 * U;
        memcpy(p, C, sizeof(REAL) * U); p += U;
        memcpy(p, v, sizeof(REAL) * n6); p += n6;
        memcpy(p, u, sizeof(REAL) * U); p += U;
        memcpy(p, S, sizeof(REAL) * n6 * n6); p += n6 * n6;
        memcpy(p, y, sizeof(REAL) * n6);
      }
      info = SUF(chol_lower)(S, (int)n6);
      SUF(chol_solve)(S, (int)n6, y); /* y <- dX */
      /* dZ = Q * (u - E^T dX)   (ba_cuda.cu:592) */
      for (long k = 0; k < U; k++) {
        REAL s = 0;
        for (long a = 0; a < n6; a++) s += Em[a * U + k] * y[a];
        dZ[k] = Q[k] * (u[k] - s);
      }
      if (dbg && itr == 0) {
        REAL *p = dbg + 2 * n6 * n6 + n6 * U + 2 * U + 2 * n6;
        memcpy(p, y, sizeof(REAL) * n6); p += n6;
        memcpy(p, dZ, sizeof(REAL) * U);
      }
      /* pose_retr_kernel, ba_cuda.cu:178-206 */
      for (int i = 0; i < N; i++) {
        REAL *p = poses + 7 * (long)(t0 + i);
        REAL tt[3] = {p[0], p[1], p[2]}, qq[4] = {p[3], p[4], p[5], p[6]}, tn[3], qn[4];
        SUF(fb_retrSE3)(y + 6 * i, tt, qq, tn, qn);
        p[0] = tn[0]; p[1] = tn[1]; p[2] = tn[2];
        p[3] = qn[0]; p[4] = qn[1]; p[5] = qn[2]; p[6] = qn[3];
      }
    }
    /* patch_retr_kernel, ba_cuda.cu:209-229: reads pixel [0][0], writes all P*P */
    for (long n = 0; n < U; n++) {
      REAL *pk = patches + kx[n] * 3 * P * P + 2 * P * P;
      REAL d = pk[0];
      d = d + dZ[n];
      d = (d > 20) ? (REAL)1.0 : d;
      d = (d > (REAL)1e-4) ? d : (REAL)1e-4;
      for (int a = 0; a < P * P; a++) pk[a] = d;
    }
  }
  free(B); free(Em); free(C); free(v); free(u); free(Q); free(S); free(y); free(dZ);
  return info;
}

/*
 * pops.transform (projective_ops.py:53-113) through the lietorch op sequence
 * (Inv, Mul, Act4 -- each reloads and re-normalises the quaternion, so3.h:30-37).
 *  coords  [E][P][P][2]
 *  validpx [E][P][P]      (X1[...,2] > 0.2, projective_ops.py:110-111) or NULL
 *  jac: valid [E] (centre Z > 0.2), Ji [E][2][6], Jj [E][2][6], Jz [E][2] or all NULL
 *  tonly: replace the rotation of Gij by identity (projective_ops.py:62-63)
 */
static void SUF(pops_transform)(const REAL *poses, const REAL *patches, const REAL *intr, const long *ii,
                                const long *jj, const long *kk, long E_, int P, int tonly, REAL *coords,
                                REAL *validpx, REAL *valid, REAL *Ji, REAL *Jj, REAL *Jz) {
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
  for (long n = 0; n < E_; n++) {
    const REAL *Ki = intr + 4 * ii[n], *Kj = intr + 4 * jj[n];
    const REAL *pk = patches + kk[n] * 3 * P * P;
    REAL Pinv[7], G[7];
    SUF(se3_inv)(poses + 7 * ii[n], Pinv);
    SUF(se3_mul)(poses + 7 * jj[n], Pinv, G);
    if (tonly) { G[3] = 0; G[4] = 0; G[5] = 0; G[6] = 1; }
    REAL Xc[4] = {0, 0, 0, 0};
    for (int a = 0; a < P * P; a++) {
      REAL X0[4], X1[4];
      X0[0] = (pk[0 * P * P + a] - Ki[2]) / Ki[0];
      X0[1] = (pk[1 * P * P + a] - Ki[3]) / Ki[1];
      X0[2] = 1;
      X0[3] = pk[2 * P * P + a];
      SUF(se3_act4)(G, X0, X1);
      REAL zc = X1[2] < (REAL)0.1 ? (REAL)0.1 : X1[2]; /* Z.clamp(min=0.1) */
      REAL d = (REAL)1.0 / zc;
      coords[(n * P * P + a) * 2 + 0] = Kj[0] * (d * X1[0]) + Kj[2];
      coords[(n * P * P + a) * 2 + 1] = Kj[1] * (d * X1[1]) + Kj[3];
      if (validpx) validpx[n * P * P + a] = (X1[2] > (REAL)0.2) ? 1 : 0;
      if (a == (P / 2) * P + P / 2) { Xc[0] = X1[0]; Xc[1] = X1[1]; Xc[2] = X1[2]; Xc[3] = X1[3]; }
    }
    if (Ji) {
      const REAL X = Xc[0], Y = Xc[1], Z = Xc[2], H = Xc[3];
      const REAL fx = Kj[0], fy = Kj[1];
      REAL az = Z < 0 ? -Z : Z;
      REAL d = (az > (REAL)0.2) ? (REAL)1.0 / Z : (REAL)0;
      /* Ja 4x6, Jp 2x4 (projective_ops.py:84-101) */
      REAL Ja[24] = {H, 0, 0, 0, Z, -Y, 0, H, 0, -Z, 0, X, 0, 0, H, Y, -X, 0, 0, 0, 0, 0, 0, 0};
      REAL Jp[8] = {fx * d, 0, -fx * X * d * d, 0, 0, fy * d, -fy * Y * d * d, 0};
      REAL M[16];
      SUF(se3_matrix)(G, M);
      for (int r = 0; r < 2; r++) {
        REAL row[6];
        for (int c = 0; c < 6; c++) {
          REAL s = 0;
          for (int k = 0; k < 4; k++) s += Jp[4 * r + k] * Ja[6 * k + c];
          row[c] = s;
          Jj[(n * 2 + r) * 6 + c] = s;
        }
        REAL o[6];
        SUF(se3_adjT)(G, row, o);
        for (int c = 0; c < 6; c++) Ji[(n * 2 + r) * 6 + c] = -o[c];
        REAL s = 0;
        for (int k = 0; k < 4; k++) s += Jp[4 * r + k] * M[4 * k + 3];
        Jz[n * 2 + r] = s;
      }
      valid[n] = (Z > (REAL)0.2) ? 1 : 0;
    }
  }
}
