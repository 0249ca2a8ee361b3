// reproject.hip -- fused patch reprojection (pops.transform) and fastba's reproject, gfx950.
//
// Replaces the ~15 launches of cdvslam/projective_ops.py:53-113 (iproj gather, lietorch Inv / Mul /
// Act4 on `repeat`-expanded inputs, proj, stack, permute) by one kernel: one lane per edge, the
// relative pose G_ij is built once and applied to all P*P patch pixels in registers.
#include "cdv_common.h"
#include "cdv_se3.h"

namespace {

template <int P>
__global__ __launch_bounds__(256) void transform_kernel(const float* __restrict__ poses,
                                                        const float* __restrict__ patches,
                                                        const float* __restrict__ intr,
                                                        const int64_t* __restrict__ ii,
                                                        const int64_t* __restrict__ jj,
                                                        const int64_t* __restrict__ kk, int64_t E, int flags,
                                                        float* __restrict__ coords, float* __restrict__ validpx,
                                                        float* __restrict__ valid, float* __restrict__ Ji,
                                                        float* __restrict__ Jj, float* __restrict__ Jz) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= E) return;
  const int64_t ix = ii[n], jx = jj[n], kx = kk[n];
  constexpr int PP = P * P;

  float Pi[7], Pj[7], Pinv[7], G[7];
#pragma unroll
  for (int a = 0; a < 7; a++) { Pi[a] = poses[7 * ix + a]; Pj[a] = poses[7 * jx + a]; }
  cdv::lt_se3_inv(Pi, Pinv);          // poses[:, ii].inv()          projective_ops.py:60
  cdv::lt_se3_mul(Pj, Pinv, G);       // poses[:, jj] * ...
  if (flags & CDV_TF_TONLY) { G[3] = 0.f; G[4] = 0.f; G[5] = 0.f; G[6] = 1.f; }
  float t[3], q[4];
  cdv::lt_se3_load(G, t, q);          // Act4 reloads (re-normalises) Gij  so3.h:30-37

  const float fxi = intr[4 * ix + 0], fyi = intr[4 * ix + 1], cxi = intr[4 * ix + 2], cyi = intr[4 * ix + 3];
  const float fxj = intr[4 * jx + 0], fyj = intr[4 * jx + 1], cxj = intr[4 * jx + 2], cyj = intr[4 * jx + 3];
  const float* pk = patches + kx * 3 * PP;
  const bool e2pp = (flags & CDV_TF_LAYOUT_E2PP) != 0;

  float Xc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int a = 0; a < PP; a++) {
    float X0[4], X1[4];
    X0[0] = (pk[a] - cxi) / fxi;              // iproj, projective_ops.py:19-29
    X0[1] = (pk[PP + a] - cyi) / fyi;
    X0[2] = 1.f;
    X0[3] = pk[2 * PP + a];
    cdv::lt_act4_loaded(t, q, X0, X1);
    const float d = 1.0f / fmaxf(X1[2], 0.1f); // proj, projective_ops.py:43
    const float x = fxj * (d * X1[0]) + cxj;
    const float y = fyj * (d * X1[1]) + cyj;
    if (e2pp) {
      coords[(n * 2 + 0) * PP + a] = x;
      coords[(n * 2 + 1) * PP + a] = y;
    } else {
      coords[(n * PP + a) * 2 + 0] = x;
      coords[(n * PP + a) * 2 + 1] = y;
    }
    if (validpx) validpx[n * PP + a] = (X1[2] > 0.2f) ? 1.f : 0.f;
    if (a == (P / 2) * P + P / 2) { Xc[0] = X1[0]; Xc[1] = X1[1]; Xc[2] = X1[2]; Xc[3] = X1[3]; }
  }

  if (Ji) {  // projective_ops.py:71-108
    const float X = Xc[0], Y = Xc[1], Z = Xc[2], H = Xc[3];
    const float d = (fabsf(Z) > 0.2f) ? 1.0f / Z : 0.f;
    float R[9];
    cdv::lt_quat_to_R(q, R);
    // rows of Jp*Ja: Jp = [fx d, 0, -fx X d^2, 0 ; 0, fy d, -fy Y d^2, 0], Ja = [H I | -[X]x ; 0]
    float row[2][6];
    const float a0 = fxj * d, a2 = -fxj * X * d * d;
    row[0][0] = a0 * H; row[0][1] = 0.f;    row[0][2] = a2 * H;
    row[0][3] = a2 * Y; row[0][4] = a0 * Z - a2 * X; row[0][5] = -a0 * Y;
    const float b1 = fyj * d, b2 = -fyj * Y * d * d;
    row[1][0] = 0.f;    row[1][1] = b1 * H; row[1][2] = b2 * H;
    row[1][3] = -b1 * Z + b2 * Y; row[1][4] = -b2 * X; row[1][5] = b1 * X;
#pragma unroll
    for (int r = 0; r < 2; r++) {
      float o[6];
      cdv::lt_se3_adjT_loaded(t, R, row[r], o);   // Ji = -Gij.adjT(Jj)
#pragma unroll
      for (int c = 0; c < 6; c++) {
        Jj[(n * 2 + r) * 6 + c] = row[r][c];
        Ji[(n * 2 + r) * 6 + c] = -o[c];
      }
    }
    // Jz = Jp * Gij.matrix()[:, 3]  (column (t, 1); Jp's 4th column is zero)
    Jz[n * 2 + 0] = a0 * t[0] + a2 * t[2];
    Jz[n * 2 + 1] = b1 * t[1] + b2 * t[2];
    valid[n] = (Z > 0.2f) ? 1.f : 0.f;
  }
}

template <int P>
__global__ __launch_bounds__(256) void fastba_reproject_kernel(const float* __restrict__ poses,
                                                               const float* __restrict__ patches,
                                                               const float* __restrict__ intr,
                                                               const int64_t* __restrict__ ii,
                                                               const int64_t* __restrict__ jj,
                                                               const int64_t* __restrict__ kk, int64_t E,
                                                               float* __restrict__ coords) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= E) return;
  constexpr int PP = P * P;
  const float fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];  // ba_cuda.cu:417-423
  const int64_t ix = ii[n], jx = jj[n], kx = kk[n];
  float ti[3], tj[3], qi[4], qj[4], tij[3], qij[4];
#pragma unroll
  for (int a = 0; a < 3; a++) { ti[a] = poses[7 * ix + a]; tj[a] = poses[7 * jx + a]; }
#pragma unroll
  for (int a = 0; a < 4; a++) { qi[a] = poses[7 * ix + 3 + a]; qj[a] = poses[7 * jx + 3 + a]; }
  cdv::fb_relSE3(ti, qi, tj, qj, tij, qij);
  const float* pk = patches + kx * 3 * PP;
#pragma unroll
  for (int a = 0; a < PP; a++) {
    float Xi[4], Xj[4];
    Xi[0] = (pk[a] - cx) / fx;
    Xi[1] = (pk[PP + a] - cy) / fy;
    Xi[2] = 1.0f;
    Xi[3] = pk[2 * PP + a];
    cdv::fb_actSE3(tij, qij, Xi, Xj);
    coords[(n * 2 + 0) * PP + a] = fx * (Xj[0] / Xj[2]) + cx;
    coords[(n * 2 + 1) * PP + a] = fy * (Xj[1] / Xj[2]) + cy;
  }
}

}  // namespace

extern "C" int cdv_transform(const float* poses, const float* patches, const float* intrinsics, const int64_t* ii,
                             const int64_t* jj, const int64_t* kk, int64_t E, int P, int flags, float* coords,
                             float* validpx, float* valid, float* Ji, float* Jj, float* Jz, void* stream) {
  CDV_REQUIRE(P == 3 || P == 1, CDV_ERR_UNSUPPORTED, "cdv_transform: patch size P must be 3 or 1");
  CDV_REQUIRE((Ji == nullptr) == (Jj == nullptr) && (Ji == nullptr) == (Jz == nullptr) &&
                  (Ji == nullptr) == (valid == nullptr),
              CDV_ERR_ARG, "cdv_transform: valid/Ji/Jj/Jz must be all set or all NULL");
  if (E == 0) return CDV_OK;
  hipStream_t s = (hipStream_t)stream;
  const int threads = 64;  // E ~ 5e4: small blocks spread the edges over all 256 CUs
  const int blocks = cdv_div_up(E, threads);
  if (P == 3)
    hipLaunchKernelGGL(transform_kernel<3>, dim3(blocks), dim3(threads), 0, s, poses, patches, intrinsics, ii, jj,
                       kk, E, flags, coords, validpx, valid, Ji, Jj, Jz);
  else
    hipLaunchKernelGGL(transform_kernel<1>, dim3(blocks), dim3(threads), 0, s, poses, patches, intrinsics, ii, jj,
                       kk, E, flags, coords, validpx, valid, Ji, Jj, Jz);
  CDV_LAUNCH_CHECK();
  return CDV_OK;
}

extern "C" int cdv_fastba_reproject(const float* poses, const float* patches, const float* intrinsics,
                                    const int64_t* ii, const int64_t* jj, const int64_t* kk, int64_t E, int P,
                                    float* coords, void* stream) {
  CDV_REQUIRE(P == 3 || P == 1, CDV_ERR_UNSUPPORTED, "cdv_fastba_reproject: patch size P must be 3 or 1");
  if (E == 0) return CDV_OK;
  hipStream_t s = (hipStream_t)stream;
  const int threads = 64;
  const int blocks = cdv_div_up(E, threads);
  if (P == 3)
    hipLaunchKernelGGL(fastba_reproject_kernel<3>, dim3(blocks), dim3(threads), 0, s, poses, patches, intrinsics,
                       ii, jj, kk, E, coords);
  else
    hipLaunchKernelGGL(fastba_reproject_kernel<1>, dim3(blocks), dim3(threads), 0, s, poses, patches, intrinsics,
                       ii, jj, kk, E, coords);
  CDV_LAUNCH_CHECK();
  return CDV_OK;
}
