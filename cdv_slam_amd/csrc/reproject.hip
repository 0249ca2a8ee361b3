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

// The common call -- 3 x 3 patches, coordinates only (SLAM.reproject, slam.py:325-329) -- with the memory side done wave-wide:
// the 27 floats of a patch arrive as seven wide loads instead of 27 scalar ones, and the 64 edges of a workgroup leave as
// ONE contiguous block of 64 x 72 bytes (staged in LDS, 16 bytes per lane) instead of 18 stores per lane 72 bytes apart.
// Same arithmetic as transform_body (tf_relative / tf_pixel): bit-identical coordinates.
__global__ __launch_bounds__(64) void transform_coords3_kernel(cdv::TfArgs A) {
  __shared__ __attribute__((aligned(16))) float s_xy[64 * 18];
  const int tid = threadIdx.x;
  const int64_t e0 = (int64_t)blockIdx.x * 64, n = e0 + tid;
  const bool e2pp = (A.flags & CDV_TF_LAYOUT_E2PP) != 0;
  if (n < A.E) {
    const int64_t ix = A.ii[n], jx = A.jj[n], kx = A.kk[n];
    float G[7], t[3], q[4];
    cdv::tf_relative(A.poses, ix, jx, (A.flags & CDV_TF_TONLY) != 0, G, t, q);
    const cdv_float4 Ki = *reinterpret_cast<const cdv_float4*>(A.intr + 4 * ix);
    const cdv_float4 Kj = *reinterpret_cast<const cdv_float4*>(A.intr + 4 * jx);
    typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
    const float* pk = A.patches + kx * 27;
    float pv[28];
#pragma unroll
    for (int v = 0; v < 6; v++) {
      const f4u x4 = *reinterpret_cast<const f4u*>(pk + 4 * v);
      pv[4 * v] = x4[0]; pv[4 * v + 1] = x4[1]; pv[4 * v + 2] = x4[2]; pv[4 * v + 3] = x4[3];
    }
    pv[24] = pk[24]; pv[25] = pk[25]; pv[26] = pk[26];
#pragma unroll
    for (int a = 0; a < 9; a++) {
      float X1[4], x, y;
      cdv::tf_pixel(t, q, pv[a], pv[9 + a], pv[18 + a], Ki[0], Ki[1], Ki[2], Ki[3], Kj[0], Kj[1], Kj[2], Kj[3], x, y, X1);
      if (e2pp) { s_xy[tid * 18 + a] = x; s_xy[tid * 18 + 9 + a] = y; }
      else { s_xy[tid * 18 + 2 * a] = x; s_xy[tid * 18 + 2 * a + 1] = y; }
    }
  }
  __syncthreads();
  const int n_e = (int)((A.E - e0) < 64 ? (A.E - e0) : 64);
  float* dst = A.coords + e0 * 18;                              // 64 x 72 bytes per workgroup: 16-byte aligned
  const int n4 = (n_e * 18) >> 2;
  for (int v = tid; v < n4; v += 64)
    reinterpret_cast<cdv_float4*>(dst)[v] = reinterpret_cast<const cdv_float4*>(s_xy)[v];
  if (tid == 0 && ((n_e * 18) & 3)) { dst[4 * n4] = s_xy[4 * n4]; dst[4 * n4 + 1] = s_xy[4 * n4 + 1]; }
}

// cuda_ba.reproject: the projection fastba's residuals use -- stored (not re-normalised) poses, the intrinsics of row 0,
// no depth clamp (ba_cuda.cu:408-458 semantics) -- for every pixel of the patch; one lane per edge
template <int P>
__global__ __launch_bounds__(256) void fastba_reproject_kernel(const float* __restrict__ poses,
                                                               const float* __restrict__ patches,
                                                               const float* __restrict__ intr,
                                                               const int64_t* __restrict__ ii,
                                                               const int64_t* __restrict__ jj,
                                                               const int64_t* __restrict__ kk, int64_t E,
                                                               float* __restrict__ coords) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  constexpr int PP = P * P;
  float Pi[7], Pj[7], t[3], q[4];
#pragma unroll
  for (int a = 0; a < 7; a++) { Pi[a] = poses[7 * ii[e] + a]; Pj[a] = poses[7 * jj[e] + a]; }
  cdv::se3_between_raw(Pi, Pj, t, q);
  const float fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];
  const float* xs = patches + kk[e] * 3 * PP;
  const float *ys = xs + PP, *ds = ys + PP;
  float* ou = coords + e * 2 * PP;
#pragma unroll
  for (int a = 0; a < PP; a++) {
    const float ray[4] = {(xs[a] - cx) / fx, (ys[a] - cy) / fy, 1.0f, ds[a]};
    float X[4];
    cdv::lt_act4_loaded(t, q, ray, X);
    ou[a] = fx * (X[0] / X[2]) + cx;
    ou[PP + a] = fy * (X[1] / X[2]) + cy;
  }
}

// pops.flow_mag (projective_ops.py:120-130): three reprojections per edge -- (i -> i), (i -> j), (i -> j, translation
// only) -- in one pass; flow = beta |x_ij - x_ii| + (1 - beta) |x_ij^t - x_ii| per patch pixel, valid = X_ij.z > 0.2.
// The (i -> i) pose goes through the same inv / mul / re-normalised load as the reference's Gij = Pi * Pi^-1.
template <int P>
__global__ __launch_bounds__(256) void flow_mag_kernel(const float* __restrict__ poses, const float* __restrict__ patches,
                                                       const float* __restrict__ intr, const int64_t* __restrict__ ii,
                                                       const int64_t* __restrict__ jj, const int64_t* __restrict__ kk,
                                                       int64_t E, float beta, float* __restrict__ flow,
                                                       uint8_t* __restrict__ valid) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= E) return;
  constexpr int PP = P * P;
  const int64_t ix = ii[n], jx = jj[n], kx = kk[n];
  float Pi[7], Pj[7], Pinv[7], G[3][7];
#pragma unroll
  for (int a = 0; a < 7; a++) { Pi[a] = poses[7 * ix + a]; Pj[a] = poses[7 * jx + a]; }
  cdv::lt_se3_inv(Pi, Pinv);
  cdv::lt_se3_mul(Pi, Pinv, G[0]);
  cdv::lt_se3_mul(Pj, Pinv, G[1]);
#pragma unroll
  for (int a = 0; a < 3; a++) G[2][a] = G[1][a];
  G[2][3] = 0.f; G[2][4] = 0.f; G[2][5] = 0.f; G[2][6] = 1.f;   // tonly (projective_ops.py:62)
  float t[3][3], q[3][4];
#pragma unroll
  for (int v = 0; v < 3; v++) cdv::lt_se3_load(G[v], t[v], q[v]);
  const float fxi = intr[4 * ix + 0], fyi = intr[4 * ix + 1], cxi = intr[4 * ix + 2], cyi = intr[4 * ix + 3];
  const float fxj = intr[4 * jx + 0], fyj = intr[4 * jx + 1], cxj = intr[4 * jx + 2], cyj = intr[4 * jx + 3];
  const float* pk = patches + kx * 3 * PP;
#pragma unroll
  for (int a = 0; a < PP; a++) {
    float X0[4], X1[4], xy[3][2];
    X0[0] = (pk[a] - cxi) / fxi;
    X0[1] = (pk[PP + a] - cyi) / fyi;
    X0[2] = 1.f;
    X0[3] = pk[2 * PP + a];
    bool ok = false;
#pragma unroll
    for (int v = 0; v < 3; v++) {
      cdv::lt_act4_loaded(t[v], q[v], X0, X1);
      const float d = 1.0f / fmaxf(X1[2], 0.1f);
      const float fx = v == 0 ? fxi : fxj, fy = v == 0 ? fyi : fyj, cx = v == 0 ? cxi : cxj, cy = v == 0 ? cyi : cyj;
      xy[v][0] = fx * (d * X1[0]) + cx;
      xy[v][1] = fy * (d * X1[1]) + cy;
      if (v == 1) ok = X1[2] > 0.2f;
    }
    const float ax = xy[1][0] - xy[0][0], ay = xy[1][1] - xy[0][1];
    const float bx = xy[2][0] - xy[0][0], by = xy[2][1] - xy[0][1];
    flow[n * PP + a] = beta * sqrtf(ax * ax + ay * ay) + (1.0f - beta) * sqrtf(bx * bx + by * by);
    valid[n * PP + a] = ok ? 1 : 0;
  }
}

// patchgraph.edges_loop (patchgraph.py:71-97), the device part: one wave per candidate pair (target frame j, source
// frame f).  The reference builds nj * nf * M candidate edges with flatmeshgrid, runs pops.flow_mag on the patch CENTRES
// (three reprojections, a dozen launches over ~10^6 rows) and reduces groups of M with einops; here the wave reprojects
// the M centres of frame f into frame j (the poses of the pair are wave-uniform), sums flow * valid and valid over its
// lanes and writes the pair's mean flow, or inf when not more than 0.75 M centres are valid (patchgraph.py:88-90).
__global__ __launch_bounds__(64) void loop_flow_kernel(const float* __restrict__ poses, const float* __restrict__ patches,
                                                       const float* __restrict__ intr, const int64_t* __restrict__ ix,
                                                       int M, int PP, int centre, int j0, int f0, int nf, float beta,
                                                       float* __restrict__ out) {
  const int pair = (int)blockIdx.x;
  const int jl = pair / nf, fl = pair - jl * nf;
  const int64_t jx = j0 + jl;
  const int64_t k0 = (int64_t)(f0 + fl) * M;
  const int64_t iframe = ix[k0];   // ii[::M] of the reference: the source frame of the group
  const int lane = (int)threadIdx.x;
  float Pi[7], Pj[7], Pinv[7], G[3][7];
#pragma unroll
  for (int a = 0; a < 7; a++) { Pi[a] = poses[7 * iframe + a]; Pj[a] = poses[7 * jx + a]; }
  cdv::lt_se3_inv(Pi, Pinv);
  cdv::lt_se3_mul(Pi, Pinv, G[0]);
  cdv::lt_se3_mul(Pj, Pinv, G[1]);
#pragma unroll
  for (int a = 0; a < 3; a++) G[2][a] = G[1][a];
  G[2][3] = 0.f; G[2][4] = 0.f; G[2][5] = 0.f; G[2][6] = 1.f;   // tonly (projective_ops.py:62)
  float t[3][3], q[3][4];
#pragma unroll
  for (int v = 0; v < 3; v++) cdv::lt_se3_load(G[v], t[v], q[v]);
  const float fxi = intr[4 * iframe + 0], fyi = intr[4 * iframe + 1], cxi = intr[4 * iframe + 2], cyi = intr[4 * iframe + 3];
  const float fxj = intr[4 * jx + 0], fyj = intr[4 * jx + 1], cxj = intr[4 * jx + 2], cyj = intr[4 * jx + 3];
  float fsum = 0.f, nval = 0.f;
  for (int m = lane; m < M; m += 64) {
    const float* pk = patches + (k0 + m) * 3 * PP;
    float X0[4], X1[4], xy[3][2];
    X0[0] = (pk[centre] - cxi) / fxi;
    X0[1] = (pk[PP + centre] - cyi) / fyi;
    X0[2] = 1.f;
    X0[3] = pk[2 * PP + centre];
    bool ok = false;
#pragma unroll
    for (int v = 0; v < 3; v++) {
      cdv::lt_act4_loaded(t[v], q[v], X0, X1);
      const float d = 1.0f / fmaxf(X1[2], 0.1f);
      const float fx = v == 0 ? fxi : fxj, fy = v == 0 ? fyi : fyj, cx = v == 0 ? cxi : cxj, cy = v == 0 ? cyi : cyj;
      xy[v][0] = fx * (d * X1[0]) + cx;
      xy[v][1] = fy * (d * X1[1]) + cy;
      if (v == 1) ok = X1[2] > 0.2f;
    }
    const float ax = xy[1][0] - xy[0][0], ay = xy[1][1] - xy[0][1];
    const float bx = xy[2][0] - xy[0][0], by = xy[2][1] - xy[0][1];
    const float fl1 = beta * sqrtf(ax * ax + ay * ay) + (1.0f - beta) * sqrtf(bx * bx + by * by);
    if (ok) { fsum += fl1; nval += 1.0f; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { fsum += __shfl_xor(fsum, o); nval += __shfl_xor(nval, o); }
  if (lane == 0) out[pair] = (nval > 0.75f * (float)M) ? fsum / fmaxf(nval, 1.0f) : __builtin_inff();
}

// pops.point_cloud (projective_ops.py:115-117): X = P_ix^-1 * iproj(patch) for every patch pixel, one thread per patch
template <int P>
__global__ __launch_bounds__(256) void point_cloud_kernel(const float* __restrict__ poses, const float* __restrict__ patches,
                                                          const float* __restrict__ intr, const int64_t* __restrict__ ix,
                                                          int64_t M, float* __restrict__ points) {
  const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  constexpr int PP = P * P;
  const int64_t f = ix[m];
  float Pi[7], Pinv[7], t[3], q[4];
#pragma unroll
  for (int a = 0; a < 7; a++) Pi[a] = poses[7 * f + a];
  cdv::lt_se3_inv(Pi, Pinv);
  cdv::lt_se3_load(Pinv, t, q);
  const float fx = intr[4 * f + 0], fy = intr[4 * f + 1], cx = intr[4 * f + 2], cy = intr[4 * f + 3];
  const float* pk = patches + m * 3 * PP;
#pragma unroll
  for (int a = 0; a < PP; a++) {
    float X0[4], X1[4];
    X0[0] = (pk[a] - cx) / fx;
    X0[1] = (pk[PP + a] - cy) / fy;
    X0[2] = 1.f;
    X0[3] = pk[2 * PP + a];
    cdv::lt_act4_loaded(t, q, X0, X1);
#pragma unroll
    for (int c = 0; c < 4; c++) points[(m * PP + a) * 4 + c] = X1[c];
  }
}

}  // namespace

extern "C" int cdv_flow_mag(const float* poses, const float* patches, const float* intrinsics, const int64_t* ii,
                            const int64_t* jj, const int64_t* kk, int64_t E, int P, float beta, float* flow,
                            uint8_t* valid, void* stream) {
  CDV_REQUIRE(P == 3 || P == 1, CDV_ERR_UNSUPPORTED, "cdv_flow_mag: patch size P must be 3 or 1");
  if (E == 0) return CDV_OK;
  const int blocks = cdv_div_up(E, 64);
  if (P == 3)
    hipLaunchKernelGGL(flow_mag_kernel<3>, dim3(blocks), dim3(64), 0, (hipStream_t)stream, poses, patches, intrinsics, ii,
                       jj, kk, E, beta, flow, valid);
  else
    hipLaunchKernelGGL(flow_mag_kernel<1>, dim3(blocks), dim3(64), 0, (hipStream_t)stream, poses, patches, intrinsics, ii,
                       jj, kk, E, beta, flow, valid);
  CDV_LAUNCH_CHECK();
  return CDV_OK;
}

extern "C" int cdv_loop_flow(const float* poses, const float* patches, const float* intrinsics, const int64_t* ix, int M,
                             int P, int j0, int nj, int f0, int nf, float beta, float* flow_out, void* stream) {
  CDV_REQUIRE(P == 3 || P == 1, CDV_ERR_UNSUPPORTED, "cdv_loop_flow: patch size P must be 3 or 1");
  CDV_REQUIRE(M > 0 && j0 >= 0 && f0 >= 0 && nj >= 0 && nf >= 0, CDV_ERR_ARG, "cdv_loop_flow: bad range");
  if (nj == 0 || nf == 0) return CDV_OK;
  CDV_REQUIRE((int64_t)nj * nf < ((int64_t)1 << 31), CDV_ERR_ARG, "cdv_loop_flow: too many candidate pairs");
  hipLaunchKernelGGL(loop_flow_kernel, dim3(nj * nf), dim3(64), 0, (hipStream_t)stream, poses, patches, intrinsics, ix, M,
                     P * P, P > 1 ? P + 1 : 0, j0, f0, nf, beta, flow_out);
  CDV_LAUNCH_CHECK();
  return CDV_OK;
}

extern "C" int cdv_point_cloud(const float* poses, const float* patches, const float* intrinsics, const int64_t* ix,
                               int64_t M, int P, float* points, void* stream) {
  CDV_REQUIRE(P == 3 || P == 1, CDV_ERR_UNSUPPORTED, "cdv_point_cloud: patch size P must be 3 or 1");
  if (M == 0) return CDV_OK;
  const int blocks = cdv_div_up(M, 64);
  if (P == 3)
    hipLaunchKernelGGL(point_cloud_kernel<3>, dim3(blocks), dim3(64), 0, (hipStream_t)stream, poses, patches, intrinsics,
                       ix, M, points);
  else
    hipLaunchKernelGGL(point_cloud_kernel<1>, dim3(blocks), dim3(64), 0, (hipStream_t)stream, poses, patches, intrinsics,
                       ix, M, points);
  CDV_LAUNCH_CHECK();
  return CDV_OK;
}

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
  if (P == 3 && !validpx && !Ji && ((uintptr_t)coords & 15) == 0)
    hipLaunchKernelGGL(transform_coords3_kernel, dim3(blocks), dim3(64), 0, s, A);
  else if (P == 3)
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
