// reproject.hip -- fused patch reprojection (pops.transform) and fastba's reproject, gfx950.
//
// Replaces the ~15 launches of cdvslam/projective_ops.py:53-113 (iproj gather, lietorch Inv / Mul /
// Act4 on `repeat`-expanded inputs, proj, stack, permute) by one kernel: one lane per edge, the
// relative pose G_ij is built once and applied to all P*P patch pixels in registers.
#include "cdv_common.h"
#include "cdv_parts.h"
#include "cdv_se3.h"

namespace {

template <int P>
__global__ __launch_bounds__(256) void transform_kernel(cdv::TfArgs A) {
  cdv::transform_body<P>(A, (int64_t)blockIdx.x * blockDim.x + threadIdx.x);
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
  const cdv::TfArgs A{poses, patches, intrinsics, ii, jj, kk, E, flags, coords, validpx, valid, Ji, Jj, Jz};
  if (P == 3)
    hipLaunchKernelGGL(transform_kernel<3>, dim3(blocks), dim3(threads), 0, s, A);
  else
    hipLaunchKernelGGL(transform_kernel<1>, dim3(blocks), dim3(threads), 0, s, A);
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
