/*
 * oracle/cdv_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of the reference's per-frame update hot path, used by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg as the CHECKER.  Nothing under
 * cdv_slam_amd/ may import, link or call this file; the product path is the HIP library.
 *
 * What is restated, and from where (paths relative to the reference root):
 *   altcorr corr forward + bilinear blend   cdvslam/altcorr/correlation_kernel.cu:82-136,193-233
 *   altcorr patchify forward                cdvslam/altcorr/correlation_kernel.cu:16-47,288-307
 *   lietorch SO3/SE3 forward ops            cdvslam/lietorch/include/so3.h, se3.h   (lie_impl.h)
 *   pops.transform (+jacobian)              cdvslam/projective_ops.py:19-113        (fastba_impl.h)
 *   fastba BA (dense-E path)                cdvslam/fastba/ba_cuda.cu:36-611        (fastba_impl.h)
 *   fastba reproject                        cdvslam/fastba/ba_cuda.cu:408-458
 *   fastba neighbors                        cdvslam/fastba/ba.cpp:59-97
 *   torch::_unique(sorted, inverse)         ATen (pytorch 2.3.1, not in the reference tree):
 *                                           sorted unique values + inverse index
 *   (the block-sparse E storage of cdvslam/fastba/block_e.cu is NOT restated: the oracle keeps the
 *    dense E, which yields the same S, y, dX, dZ)
 *
 * PARITY STATUS: the reference ships no golden vectors or tests for altcorr / fastba /
 * projective_ops, and its CUDA/Eigen sources cannot be built in this image (no nvcc, Eigen
 * 3.4.0 not vendored).  => "parity unpinned" for corr, fastba, neighbors; the Lie ops are pinned
 * by the reference's own property tests (lietorch/run_tests.py:16-52) and pops.transform / ba.py
 * by executing the reference's Python files under a shimmed import (tests/golden/make_golden.py).
 * Since round 3 the fastba restatement is additionally anchored on a RUN of the reference's ba.py at
 * benchmark size (tests/golden/ba_py_pr1.npz: one fastba iteration == one ba.py call at ep = 1.0).
 */

#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------- */
/* IEEE binary16 <-> binary32, round-to-nearest-even (c10::Half semantics)                      */
/* ------------------------------------------------------------------------------------------- */

static inline float h2f(uint16_t h) {
  uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  uint32_t exp = (h >> 10) & 0x1fu;
  uint32_t man = h & 0x3ffu;
  uint32_t bits;
  if (exp == 0) {
    if (man == 0) {
      bits = sign;
    } else { /* subnormal */
      int e = -1;
      do { e++; man <<= 1; } while ((man & 0x400u) == 0);
      man &= 0x3ffu;
      bits = sign | ((uint32_t)(127 - 15 - e) << 23) | (man << 13);
    }
  } else if (exp == 31) {
    bits = sign | 0x7f800000u | (man << 13);
  } else {
    bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
  }
  float f;
  memcpy(&f, &bits, 4);
  return f;
}

static inline uint16_t f2h(float f) {
  uint32_t x;
  memcpy(&x, &f, 4);
  uint32_t sign = (x >> 16) & 0x8000u;
  uint32_t ax = x & 0x7fffffffu;
  if (ax >= 0x7f800000u) { /* inf / nan */
    return (uint16_t)(sign | 0x7c00u | ((ax > 0x7f800000u) ? 0x200u : 0));
  }
  if (ax >= 0x477ff000u) { /* rounds to >= 65520 -> inf */
    return (uint16_t)(sign | 0x7c00u);
  }
  if (ax < 0x38800000u) { /* subnormal half or zero: |f| < 2^-14 */
    if (ax < 0x33000000u) return (uint16_t)sign; /* < 2^-25 -> 0 */
    int e = (int)(ax >> 23);                      /* biased exponent */
    uint32_t man = (ax & 0x7fffffu) | 0x800000u;  /* 24-bit significand */
    const int shift = 126 - e;                    /* half subnormal unit is 2^-24 */
    uint32_t half = man >> shift;
    uint32_t rem = man & ((1u << shift) - 1);
    uint32_t mid = 1u << (shift - 1);
    if (rem > mid || (rem == mid && (half & 1))) half++;
    return (uint16_t)(sign | half);
  }
  uint32_t e = (ax >> 23) - 127 + 15;
  uint32_t man = ax & 0x7fffffu;
  uint32_t half = (e << 10) | (man >> 13);
  uint32_t rem = man & 0x1fffu;
  if (rem > 0x1000u || (rem == 0x1000u && (half & 1))) half++;
  return (uint16_t)(sign | half);
}

/* round a float to the nearest half and back (one c10::Half arithmetic result) */
static inline float rh(float f) { return h2f(f2h(f)); }

/* threads the edge-parallel loops (corr, transform) use; 1 without OpenMP */
int orc_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

void orc_f2h(long n, const float *x, uint16_t *y) { for (long i = 0; i < n; i++) y[i] = f2h(x[i]); }
void orc_h2f(long n, const uint16_t *x, float *y) { for (long i = 0; i < n; i++) y[i] = h2f(x[i]); }

/* ------------------------------------------------------------------------------------------- */
/* Lie groups + fastba + pops, instantiated for float and double                                */
/* ------------------------------------------------------------------------------------------- */

#define REAL float
#define SUF(n) n##_f32
#include "lie_impl.h"
#include "fastba_impl.h"
#undef REAL
#undef SUF

#define REAL double
#define SUF(n) n##_f64
#include "lie_impl.h"
#include "fastba_impl.h"
#undef REAL
#undef SUF

/* op codes shared with oracle/oracle.py */
enum { OP_EXP = 0, OP_LOG = 1, OP_INV = 2, OP_MUL = 3, OP_ADJ = 4, OP_ADJT = 5, OP_ACT = 6, OP_ACT4 = 7, OP_MATRIX = 8 };

#define LIE_DISPATCH(T, S)                                                                               \
  int orc_lie_##S(int group, int op, long n, const T *x, const T *y, T *z) {                             \
    const int se3 = (group == 3);                                                                        \
    if (group != 1 && group != 3) return -1;                                                             \
    const int N = se3 ? 7 : 4, K = se3 ? 6 : 3;                                                          \
    for (long i = 0; i < n; i++) {                                                                       \
      switch (op) {                                                                                      \
        case OP_EXP: if (se3) se3_exp_##S(x + K * i, z + N * i); else so3_exp_##S(x + K * i, z + N * i); break; \
        case OP_LOG: if (se3) se3_log_##S(x + N * i, z + K * i); else so3_log_##S(x + N * i, z + K * i); break; \
        case OP_INV: if (se3) se3_inv_##S(x + N * i, z + N * i); else so3_inv_##S(x + N * i, z + N * i); break; \
        case OP_MUL: if (se3) se3_mul_##S(x + N * i, y + N * i, z + N * i); else so3_mul_##S(x + N * i, y + N * i, z + N * i); break; \
        case OP_ADJ: if (se3) se3_adj_##S(x + N * i, y + K * i, z + K * i); else so3_adj_##S(x + N * i, y + K * i, z + K * i); break; \
        case OP_ADJT: if (se3) se3_adjT_##S(x + N * i, y + K * i, z + K * i); else so3_adjT_##S(x + N * i, y + K * i, z + K * i); break; \
        case OP_ACT: if (se3) se3_act_##S(x + N * i, y + 3 * i, z + 3 * i); else so3_act_##S(x + N * i, y + 3 * i, z + 3 * i); break; \
        case OP_ACT4: if (se3) se3_act4_##S(x + N * i, y + 4 * i, z + 4 * i); else so3_act4_##S(x + N * i, y + 4 * i, z + 4 * i); break; \
        case OP_MATRIX:                                                                                  \
          if (se3) se3_matrix_##S(x + N * i, z + 16 * i);                                                \
          else {                                                                                         \
            T q[4], R[9];                                                                                \
            quat_load_##S(x + N * i, q);                                                                 \
            quat_to_R_##S(q, R);                                                                         \
            T *M = z + 16 * i;                                                                           \
            for (int a = 0; a < 16; a++) M[a] = 0;                                                       \
            for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) M[4 * a + b] = R[3 * a + b];         \
            M[15] = 1;                                                                                   \
          }                                                                                              \
          break;                                                                                         \
        default: return -2;                                                                              \
      }                                                                                                  \
    }                                                                                                    \
    return 0;                                                                                            \
  }
LIE_DISPATCH(float, f32)
LIE_DISPATCH(double, f64)

#define POPS_EXPORT(T, S)                                                                                \
  void orc_transform_##S(const T *poses, const T *patches, const T *intr, const long *ii, const long *jj, \
                         const long *kk, long E, int P, int tonly, T *coords, T *validpx, T *valid, T *Ji, \
                         T *Jj, T *Jz) {                                                                 \
    pops_transform_##S(poses, patches, intr, ii, jj, kk, E, P, tonly, coords, validpx, valid, Ji, Jj, Jz); \
  }                                                                                                      \
  int orc_fastba_##S(T *poses, T *patches, const T *intr, const T *target, const T *weight, T lmbda,     \
                     const long *ii, const long *jj, const long *kk, const long *kx, const long *ku,     \
                     long E, long U, int P, int t0, int t1, int iterations, T *dbg) {                    \
    return fb_ba_##S(poses, patches, intr, target, weight, lmbda, ii, jj, kk, kx, ku, E, U, P, t0, t1,   \
                     iterations, dbg);                                                                   \
  }                                                                                                      \
  /* the per-edge factor of reprojection_residuals_and_hessian (ba_cuda.cu:262-342) on its own: residual rows r [E][2],  \
     masked weights w [E][2], Jz [E][2], Ji / Jj [E][2][6] -- what the block-sparse E of block_e.cu is filled from     \
     (oracle/block_e_py.py) */                                                                             \
  void orc_fastba_edges_##S(const T *poses, const T *patches, const T *intr, const T *target, const T *weight, \
                            const long *ii, const long *jj, const long *kk, long E, int P, T *r, T *w, T *Jz, \
                            T *Ji, T *Jj) {                                                              \
    for (long n = 0; n < E; n++)                                                                         \
      fb_edge_##S(poses, patches, intr, target, weight, ii[n], jj[n], kk[n], P, n, r + 2 * n, w + 2 * n, \
                  Jz + 2 * n, Ji + 12 * n, Jj + 12 * n);                                                 \
  }                                                                                                      \
  /* fastba `reproject` kernel, ba_cuda.cu:408-458: intrinsics row 0, no depth clamp -> [E][2][P][P] */  \
  void orc_fastba_reproject_##S(const T *poses, const T *patches, const T *intr, const long *ii,         \
                                const long *jj, const long *kk, long E, int P, T *coords) {              \
    const T fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];                                      \
    for (long n = 0; n < E; n++) {                                                                       \
      const T *pi = poses + 7 * ii[n], *pj = poses + 7 * jj[n];                                          \
      T tij[3], qij[4];                                                                                  \
      fb_relSE3_##S(pi, pi + 3, pj, pj + 3, tij, qij);                                                   \
      const T *pk = patches + kk[n] * 3 * P * P;                                                         \
      for (int a = 0; a < P * P; a++) {                                                                  \
        T Xi[4], Xj[4];                                                                                  \
        Xi[0] = (pk[a] - cx) / fx;                                                                       \
        Xi[1] = (pk[P * P + a] - cy) / fy;                                                               \
        Xi[2] = 1.0;                                                                                     \
        Xi[3] = pk[2 * P * P + a];                                                                       \
        fb_actSE3_##S(tij, qij, Xi, Xj);                                                                 \
        coords[(n * 2 + 0) * P * P + a] = fx * (Xj[0] / Xj[2]) + cx;                                     \
        coords[(n * 2 + 1) * P * P + a] = fy * (Xj[1] / Xj[2]) + cy;                                     \
      }                                                                                                  \
    }                                                                                                    \
  }
POPS_EXPORT(float, f32)
POPS_EXPORT(double, f64)

/* ------------------------------------------------------------------------------------------- */
/* torch::_unique(sorted=true, return_inverse=true) and fastba.neighbors                        */
/* ------------------------------------------------------------------------------------------- */

static int cmp_long(const void *a, const void *b) {
  long x = *(const long *)a, y = *(const long *)b;
  return (x > y) - (x < y);
}

/* returns U; uniq must hold n entries, inverse n entries */
long orc_unique(long n, const long *x, long *uniq, long *inverse) {
  if (n == 0) return 0;
  long *s = (long *)malloc(sizeof(long) * (size_t)n);
  memcpy(s, x, sizeof(long) * (size_t)n);
  qsort(s, (size_t)n, sizeof(long), cmp_long);
  long U = 0;
  for (long i = 0; i < n; i++)
    if (i == 0 || s[i] != s[i - 1]) uniq[U++] = s[i];
  for (long i = 0; i < n; i++) { /* binary search */
    long lo = 0, hi = U - 1;
    while (lo < hi) {
      long mid = (lo + hi) / 2;
      if (uniq[mid] < x[i]) lo = mid + 1; else hi = mid;
    }
    inverse[i] = lo;
  }
  free(s);
  return U;
}

typedef struct { long j; long e; } je_t;
static int cmp_je(const void *a, const void *b) {
  const je_t *x = (const je_t *)a, *y = (const je_t *)b;
  if (x->j != y->j) return (x->j > y->j) - (x->j < y->j);
  return (x->e > y->e) - (x->e < y->e); /* std::stable_sort on an ascending index list */
}

/* ba.cpp:59-97: neighbors(ii=patch ids, jj=frame ids) -> ix (previous edge in time), jx (next) */
void orc_neighbors(long n, const long *ii, const long *jj, long *ix, long *jx) {
  if (n == 0) return;
  long *uniq = (long *)malloc(sizeof(long) * (size_t)n), *inv = (long *)malloc(sizeof(long) * (size_t)n);
  long U = orc_unique(n, ii, uniq, inv);
  long *cnt = (long *)calloc((size_t)U + 1, sizeof(long));
  for (long i = 0; i < n; i++) cnt[inv[i] + 1]++;
  for (long k = 0; k < U; k++) cnt[k + 1] += cnt[k];
  je_t *buf = (je_t *)malloc(sizeof(je_t) * (size_t)n);
  long *cur = (long *)malloc(sizeof(long) * (size_t)U);
  for (long k = 0; k < U; k++) cur[k] = cnt[k];
  for (long i = 0; i < n; i++) { buf[cur[inv[i]]].j = jj[i]; buf[cur[inv[i]]].e = i; cur[inv[i]]++; }
  for (long k = 0; k < U; k++) {
    long a = cnt[k], b = cnt[k + 1];
    qsort(buf + a, (size_t)(b - a), sizeof(je_t), cmp_je);
    for (long t = a; t < b; t++) {
      ix[buf[t].e] = (t > a) ? buf[t - 1].e : -1;
      jx[buf[t].e] = (t < b - 1) ? buf[t + 1].e : -1;
    }
  }
  free(uniq); free(inv); free(cnt); free(buf); free(cur);
}

/* ------------------------------------------------------------------------------------------- */
/* altcorr                                                                                      */
/* ------------------------------------------------------------------------------------------- */

/*
 * corr forward (correlation_kernel.cu:82-136) + bilinear blend and permute (:213-232).
 *   fmap1 [N1][C][H][W]   patch feature tiles (H = W = P)
 *   fmap2 [N2][C][H2][W2] frame feature maps
 *   coords [M][2][H][W] f32, us[M] -> fmap1 index, vs[M] -> fmap2 index
 *   out   [M][D-1 (x)][D-1 (y)][H][W]     (the permuted order the reference returns), D = 2R+2
 * mode 0: inputs/outputs are binary16 bit patterns; every multiply, add and blend product is
 *         rounded to binary16 exactly where c10::Half / ATen half kernels round (reference-faithful).
 * mode 1: float in, float accumulate, float out (MIXED_PRECISION False path).
 * mode 2: inputs binary16 bit patterns, double accumulate + double blend, double out ("truth").
 * mode 3: float in, double accumulate, double out.
 */
/* The half path of one patch pixel (raw 8x8 volume in c10::Half arithmetic + the four-slice half blend), generated twice:
 * with the software conversions above, and with the F16C instructions (same IEEE round-to-nearest-even results, ~6x
 * faster: the closed-loop stream tests run hundreds of updates through this path).  Picked at run time by what the host
 * CPU has; tests/test_corr_independent.py holds either against torch's half bit for bit. */
#define ORC_HALF_PIXEL(NAME, ATTR, H2F, RH, F2H)                                                                      \
  ATTR static void NAME(const uint16_t *f1, const uint16_t *f2, long f1_base, long f1_cs, long f2_base, long f2_cs,   \
                        int H2, int W2, int C, int R, float x, float y, uint16_t *out, long o_base, long o_sx,       \
                        long o_sy) {                                                                                  \
    const int D = 2 * R + 2, D1 = D - 1;                                                                              \
    float raw[16 * 16];                                                                                               \
    const int fy = (int)floorf(y), fx = (int)floorf(x);                                                               \
    for (int a = 0; a < D; a++)                                                                                       \
      for (int b = 0; b < D; b++) {                                                                                   \
        const int i1 = fy + (a - R), j1 = fx + (b - R);                                                               \
        float sh = 0;                                                                                                 \
        if (i1 >= 0 && i1 < H2 && j1 >= 0 && j1 < W2) {                                                               \
          const long q = f2_base + (long)i1 * W2 + j1;                                                                \
          for (int c = 0; c < C; c++) {                                                                               \
            float p = RH(H2F(f1[f1_base + c * f1_cs]) * H2F(f2[q + c * f2_cs]));                                      \
            sh = RH(sh + p);                                                                                          \
          }                                                                                                           \
        }                                                                                                             \
        raw[a * D + b] = sh;                                                                                          \
      }                                                                                                               \
    const float dx = RH(x - floorf(x)), dy = RH(y - floorf(y));                                                       \
    const float omx = RH(1.0f - dx), omy = RH(1.0f - dy);                                                             \
    const float w00 = RH(omx * omy), w01 = RH(dx * omy), w10 = RH(omx * dy), w11 = RH(dx * dy);                       \
    for (int a = 0; a < D1; a++)                                                                                      \
      for (int b = 0; b < D1; b++) {                                                                                  \
        float acc = RH(w00 * raw[a * D + b]);                                                                         \
        acc = RH(acc + RH(w01 * raw[a * D + b + 1]));                                                                 \
        acc = RH(acc + RH(w10 * raw[(a + 1) * D + b]));                                                               \
        acc = RH(acc + RH(w11 * raw[(a + 1) * D + b + 1]));                                                           \
        out[o_base + b * o_sx + a * o_sy] = F2H(acc);                                                                 \
      }                                                                                                               \
  }

ORC_HALF_PIXEL(half_pixel_sw, , h2f, rh, f2h)
#if defined(__x86_64__) && defined(__GNUC__)
#include <immintrin.h>
#define ORC_H2F_HW(h) _cvtsh_ss(h)
#define ORC_F2H_HW(f) _cvtss_sh((f), 0)
#define ORC_RH_HW(f) _cvtsh_ss(_cvtss_sh((f), 0))
ORC_HALF_PIXEL(half_pixel_hw, __attribute__((target("f16c"))), ORC_H2F_HW, ORC_RH_HW, ORC_F2H_HW)
static int orc_have_f16c(void) { return __builtin_cpu_supports("f16c"); }
#else
#define half_pixel_hw half_pixel_sw
static int orc_have_f16c(void) { return 0; }
#endif
static int orc_force_sw = 0;
void orc_set_half_software(int on) { orc_force_sw = on; } /* tests: compare the two conversions */
int orc_half_is_hardware(void) { return !orc_force_sw && orc_have_f16c(); }

void orc_corr(int mode, const void *fmap1_, const void *fmap2_, const float *coords, const long *us,
              const long *vs, long M, int C, int H, int W, int H2, int W2, int R, void *out_) {
  const int D = 2 * R + 2, D1 = D - 1;
#ifdef _OPENMP
#pragma omp parallel
#endif
  {
  double *raw = (double *)malloc(sizeof(double) * (size_t)(D * D));
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 16)
#endif
  for (long m = 0; m < M; m++) {
    const long ix = us[m], jx = vs[m];
    for (int i0 = 0; i0 < H; i0++)
      for (int j0 = 0; j0 < W; j0++) {
        const float x = coords[((m * 2 + 0) * H + i0) * W + j0];
        const float y = coords[((m * 2 + 1) * H + i0) * W + j0];
        if (mode == 0 && D <= 16) { /* the half path as one function (software or F16C conversions, same results) */
          (orc_half_is_hardware() ? half_pixel_hw : half_pixel_sw)(
              (const uint16_t *)fmap1_, (const uint16_t *)fmap2_, ((ix * C) * H + i0) * W + j0, (long)H * W,
              (jx * C) * (long)H2 * W2, (long)H2 * W2, H2, W2, C, R, x, y, (uint16_t *)out_,
              ((m * D1) * D1 * H + i0) * W + j0, (long)D1 * H * W, (long)H * W);
          continue;
        }
        const int fy = (int)floorf(y), fx = (int)floorf(x);
        for (int a = 0; a < D; a++)
          for (int b = 0; b < D; b++) {
            const int i1 = fy + (a - R), j1 = fx + (b - R);
            double s = 0;
            if (i1 >= 0 && i1 < H2 && j1 >= 0 && j1 < W2) {
              if (mode == 0) {
                const uint16_t *f1 = (const uint16_t *)fmap1_, *f2 = (const uint16_t *)fmap2_;
                float sh = 0;
                for (int c = 0; c < C; c++) {
                  float p = rh(h2f(f1[((ix * C + c) * H + i0) * W + j0]) * h2f(f2[((jx * C + c) * H2 + i1) * W2 + j1]));
                  sh = rh(sh + p);
                }
                s = sh;
              } else if (mode == 1) {
                const float *f1 = (const float *)fmap1_, *f2 = (const float *)fmap2_;
                float sf = 0;
                for (int c = 0; c < C; c++)
                  sf += f1[((ix * C + c) * H + i0) * W + j0] * f2[((jx * C + c) * H2 + i1) * W2 + j1];
                s = sf;
              } else if (mode == 2) {
                const uint16_t *f1 = (const uint16_t *)fmap1_, *f2 = (const uint16_t *)fmap2_;
                for (int c = 0; c < C; c++)
                  s += (double)h2f(f1[((ix * C + c) * H + i0) * W + j0]) * (double)h2f(f2[((jx * C + c) * H2 + i1) * W2 + j1]);
              } else {
                const float *f1 = (const float *)fmap1_, *f2 = (const float *)fmap2_;
                for (int c = 0; c < C; c++)
                  s += (double)f1[((ix * C + c) * H + i0) * W + j0] * (double)f2[((jx * C + c) * H2 + i1) * W2 + j1];
              }
            }
            raw[a * D + b] = s;
          }
        /* blend, correlation_kernel.cu:221-230; dims of raw are [ii=y][jj=x] */
        const float dxf = x - floorf(x), dyf = y - floorf(y);
        for (int a = 0; a < D1; a++)
          for (int b = 0; b < D1; b++) {
            const long o = (((m * D1 + b) * D1 + a) * H + i0) * W + j0; /* permute -> [x][y] */
            if (mode == 0) {
              const float dx = rh(dxf), dy = rh(dyf);
              const float c00 = (float)raw[a * D + b], c01 = (float)raw[a * D + b + 1];
              const float c10 = (float)raw[(a + 1) * D + b], c11 = (float)raw[(a + 1) * D + b + 1];
              const float omx = rh(1.0f - dx), omy = rh(1.0f - dy);
              float acc = rh(rh(omx * omy) * c00);
              acc = rh(acc + rh(rh(dx * omy) * c01));
              acc = rh(acc + rh(rh(omx * dy) * c10));
              acc = rh(acc + rh(rh(dx * dy) * c11));
              ((uint16_t *)out_)[o] = f2h(acc);
            } else if (mode == 1) {
              const float dx = dxf, dy = dyf;
              const float c00 = (float)raw[a * D + b], c01 = (float)raw[a * D + b + 1];
              const float c10 = (float)raw[(a + 1) * D + b], c11 = (float)raw[(a + 1) * D + b + 1];
              float acc = (1 - dx) * (1 - dy) * c00;
              acc += (dx) * (1 - dy) * c01;
              acc += (1 - dx) * (dy)*c10;
              acc += (dx) * (dy)*c11;
              ((float *)out_)[o] = acc;
            } else {
              /* truth: the half path blends with dx,dy rounded to half (they are cast before the blend) */
              const double dx = (mode == 2) ? (double)rh(dxf) : (double)dxf;
              const double dy = (mode == 2) ? (double)rh(dyf) : (double)dyf;
              ((double *)out_)[o] = (1 - dx) * (1 - dy) * raw[a * D + b] + dx * (1 - dy) * raw[a * D + b + 1] +
                                    (1 - dx) * dy * raw[(a + 1) * D + b] + dx * dy * raw[(a + 1) * D + b + 1];
            }
          }
      }
  }
  free(raw);
  }
}

/*
 * patchify forward (correlation_kernel.cu:16-47,288-307): net [C][H][W], coords [M][2] ->
 * patches [M][C][D][D], zero when the integer sample is out of bounds.  esize = 2 or 4 bytes.
 */
void orc_patchify(int esize, const void *net, const float *coords, long M, int C, int H, int W, int R,
                  void *patches) {
  const int D = 2 * R + 2;
  memset(patches, 0, (size_t)(M * C * D * D) * (size_t)esize);
  for (long m = 0; m < M; m++) {
    const float x = coords[2 * m + 0], y = coords[2 * m + 1];
    const int fy = (int)floorf(y), fx = (int)floorf(x);
    for (int a = 0; a < D; a++)
      for (int b = 0; b < D; b++) {
        const int i = fy + (a - R), j = fx + (b - R);
        if (i < 0 || i >= H || j < 0 || j >= W) continue;
        for (int c = 0; c < C; c++)
          memcpy((char *)patches + (size_t)(((m * C + c) * D + a) * D + b) * esize,
                 (const char *)net + (size_t)((c * H + i) * W + j) * esize, (size_t)esize);
      }
  }
}
