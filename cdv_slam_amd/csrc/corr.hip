// corr.hip -- altcorr: patch-to-frame local correlation + bilinear blend, gfx950.
//
// Reference: cdvslam/altcorr/correlation_kernel.cu:82-136 (27.5 M threads per level, each doing 24
// strided 2-byte loads from channel-planar maps), :213-232 (raw 8x8 volume written to HBM and re-read
// by four ATen slice-multiply-add passes), then torch.stack of the two levels (slam.py:323).
//
// Here (cdv_corr_fused): ONE launch for both pyramid levels.  One wave per edge:
//   * the 3x3x24 patch tile is the MFMA A operand (9 of 16 rows used), gathered once per edge;
//   * the union window of the 9 patch pixels (<= 16x16 feature pixels, typically 10x10) is the B
//     operand, loaded straight from a CHANNELS-LAST feature ring (one pixel = C contiguous halves, so
//     a lane's 8 k-values are one 16-byte load -- no LDS staging of the inputs at all);
//   * v_mfma_f32_16x16x32_f16 produces, per window row, the 9 x 16 correlations in f32;
//   * the raw volume lives only in LDS (9.2 KB per wave); the 8x8 -> 7x7 bilinear blend of every
//     patch pixel reads it back with its own sub-pixel offset, separably (16 LDS reads per lane);
//   * the [882]-half row of the edge is staged in LDS and leaves as 256-byte coalesced stores.
// Algorithmic HBM bytes per edge: 1764 out + 72 coords + 16 idx (+ the feature maps once): DESIGN.md.
#include "cdv_common.h"

namespace {

constexpr int RAW_ROWS = 16;                    // window rows held per patch pixel
constexpr int RAW_MS = RAW_ROWS * 16 + 4;       // floats per patch pixel (+4: spreads the 4 row-groups over banks)
constexpr int RAW_FLOATS = 9 * RAW_MS;          // 2340
constexpr int OUT_HALFS = 896;                  // 882 rounded up to a multiple of 64 bytes
constexpr int WAVE_LDS_BYTES = RAW_FLOATS * 4 + OUT_HALFS * 2;  // 11,152 B per wave

__device__ __forceinline__ void wave_lds_sync() {
  // LDS operations of one wave execute in order; this only has to stop the compiler from moving the
  // reads above the writes and to wait for outstanding DS ops (s_waitcnt lgkmcnt(0)).
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
  return v;
}

struct LevelParams {
  const _Float16* fmap;  // [slots][H][W][C]
  int H, W;
  float scale;
};

template <int KS>
__global__ __launch_bounds__(256) void corr_fused_kernel(const _Float16* __restrict__ gmap, LevelParams L0,
                                                         LevelParams L1, const float* __restrict__ coords,
                                                         const int64_t* __restrict__ kk,
                                                         const int64_t* __restrict__ jj,
                                                         const int32_t* __restrict__ order,
                                                         _Float16* __restrict__ out, int E, int64_t Ng, int64_t slots,
                                                         int C, int nlev, int64_t kmod, int64_t jmod) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int p = blockIdx.x * 4 + wave;
  if (p >= E) return;  // no block-wide barriers below: waves are independent
  const int e = order ? order[p] : p;
  float* raw = reinterpret_cast<float*>(smem_raw + (size_t)wave * WAVE_LDS_BYTES);
  _Float16* outT = reinterpret_cast<_Float16*>(raw + RAW_FLOATS);

  int64_t kpatch = kk[e], jslot = jj[e];
  if (kmod > 0) kpatch %= kmod;
  if (jmod > 0) jslot %= jmod;
  const bool idx_ok = kpatch >= 0 && kpatch < Ng && jslot >= 0 && jslot < slots;

  const int n = lane & 15, g = lane >> 4;

  // ---- A operand: patch tile, rows = patch pixels (i0*3+j0), k = channels -----------------------
  cdv_half8 afrag[KS];
#pragma unroll
  for (int s = 0; s < KS; s++) {
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int ch = 32 * s + 8 * g + j;
      _Float16 v = (_Float16)0.f;
      if (idx_ok && n < 9 && ch < C) v = gmap[(kpatch * C + ch) * 9 + n];
      afrag[s][j] = v;
    }
  }

  // ---- per patch pixel coordinates (lanes 0..8 own pixel m = lane; others mirror pixel 0) -------
  const int mm = lane < 9 ? lane : 0;
  const float cxm = coords[(int64_t)e * 18 + mm];
  const float cym = coords[(int64_t)e * 18 + 9 + mm];

  for (int lev = 0; lev < nlev; lev++) {
    const LevelParams LP = lev == 0 ? L0 : L1;
    const float x = cxm / LP.scale, y = cym / LP.scale;  // slam.py:321-322 (coords / 1, coords / 4)
    const float fxf = floorf(x), fyf = floorf(y);
    // dx, dy are cast to the feature dtype before the blend (correlation_kernel.cu:223-224)
    const float dx = (float)(_Float16)(x - fxf), dy = (float)(_Float16)(y - fyf);
    const int ixm = (int)fminf(fmaxf(fxf, -1.0e6f), 1.0e6f);
    const int iym = (int)fminf(fmaxf(fyf, -1.0e6f), 1.0e6f);
    // wave-uniform window box, forced into SGPRs so that the row loop is a scalar loop
    const int x0 = __builtin_amdgcn_readfirstlane(wave_min_i(ixm) - 3);
    const int y0 = __builtin_amdgcn_readfirstlane(wave_min_i(iym) - 3);
    const int Wb = __builtin_amdgcn_readfirstlane(wave_max_i(ixm) + 4 - x0 + 1);
    const int Hb = __builtin_amdgcn_readfirstlane(wave_max_i(iym) + 4 - y0 + 1);
    const bool fast = (Wb <= 16) && (Hb <= RAW_ROWS);
    const _Float16* fbase = LP.fmap + (size_t)jslot * LP.H * LP.W * C;

    if (fast) {
      // one MFMA per window row: 16 columns x 9 patch pixels x C channels
      constexpr int RB = (KS == 1) ? 8 : 2;  // rows in flight (register budget)
      for (int tb = 0; tb < Hb; tb += RB) {
        cdv_half8 bfrag[RB][KS];
#pragma unroll
        for (int r = 0; r < RB; r++) {
          const int py = y0 + tb + r, px = x0 + n;
          const bool ok = idx_ok && (tb + r) < Hb && n < Wb && py >= 0 && py < LP.H && px >= 0 && px < LP.W;
#pragma unroll
          for (int s = 0; s < KS; s++) {
            cdv_half8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (ok && (32 * s + 8 * g) < C)
              v = *reinterpret_cast<const cdv_half8*>(fbase + ((size_t)py * LP.W + px) * C + 32 * s + 8 * g);
            bfrag[r][s] = v;
          }
        }
#pragma unroll
        for (int r = 0; r < RB; r++) {
          if (tb + r < Hb) {
            cdv_float4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < KS; s++) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(afrag[s], bfrag[r][s], acc, 0, 0, 0);
            // D layout: col = lane & 15 (window column), row = 4 * (lane >> 4) + reg (patch pixel)
#pragma unroll
            for (int q = 0; q < 4; q++) {
              const int m = 4 * g + q;
              if (m < 9) raw[m * RAW_MS + (tb + r) * 16 + n] = acc[q];
            }
          }
        }
      }
    } else {
      // wide reprojection footprint (strong zoom / rotation): every patch pixel gets its own 8x8
      // window; two window rows share one MFMA (columns 0-7 | 8-15), only row m of D is kept.
      for (int m = 0; m < 9; m++) {
        const int xm = __shfl(ixm, m) - 3, ym = __shfl(iym, m) - 3;
#pragma unroll
        for (int t2 = 0; t2 < 4; t2++) {
          const int py = ym + 2 * t2 + (n >> 3), px = xm + (n & 7);
          const bool ok = idx_ok && py >= 0 && py < LP.H && px >= 0 && px < LP.W;
          cdv_float4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int s = 0; s < KS; s++) {
            cdv_half8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (ok && (32 * s + 8 * g) < C)
              v = *reinterpret_cast<const cdv_half8*>(fbase + ((size_t)py * LP.W + px) * C + 32 * s + 8 * g);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(afrag[s], v, acc, 0, 0, 0);
          }
          if (g == (m >> 2)) {
            const int q = m & 3;
            const float val = q == 0 ? acc[0] : q == 1 ? acc[1] : q == 2 ? acc[2] : acc[3];
            raw[m * RAW_MS + (2 * t2 + (n >> 3)) * 16 + (n & 7)] = val;
          }
        }
      }
    }
    wave_lds_sync();

    // ---- bilinear blend 8x8 -> 7x7, lane = (patch pixel m, x offset xo) ---------------------------
    {
      const int m = lane / 7, xo = lane - 7 * m;
      const int msrc = m < 9 ? m : 0;
      const int bx = fast ? (__shfl(ixm, msrc) - 3 - x0) : 0;
      const int by = fast ? (__shfl(iym, msrc) - 3 - y0) : 0;
      const float dxm = __shfl(dx, msrc), dym = __shfl(dy, msrc);
      if (lane < 63) {
        const float* rp = raw + m * RAW_MS + by * 16 + bx + xo;
        float h[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
          const float c0 = rp[r * 16], c1 = rp[r * 16 + 1];
          h[r] = (1.f - dxm) * c0 + dxm * c1;
        }
#pragma unroll
        for (int yo = 0; yo < 7; yo++) {
          const float v = (1.f - dym) * h[yo] + dym * h[yo + 1];
          outT[((xo * 7 + yo) * 9 + m) * nlev + lev] = (_Float16)v;
        }
      }
    }
    wave_lds_sync();
  }

  // ---- coalesced store of the edge's row ---------------------------------------------------------
  if (nlev == 2) {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(outT);
    uint32_t* dst = reinterpret_cast<uint32_t*>(out) + (size_t)e * 441;
#pragma unroll
    for (int i = 0; i < 7; i++) {
      const int t = i * 64 + lane;
      if (t < 441) dst[t] = src[t];
    }
  } else {
    _Float16* dst = out + (size_t)e * 441;
#pragma unroll
    for (int i = 0; i < 7; i++) {
      const int t = i * 64 + lane;
      if (t < 441) dst[t] = outT[t];
    }
  }
}

// ---- generic per-level kernel: planar layouts, any C / P / radius, f16 or f32 ----------------------
template <typename T>
__global__ __launch_bounds__(256) void corr_generic_kernel(const T* __restrict__ fmap1, const T* __restrict__ fmap2,
                                                           const float* __restrict__ coords,
                                                           const int64_t* __restrict__ us,
                                                           const int64_t* __restrict__ vs, T* __restrict__ out,
                                                           int64_t M, int64_t N1, int64_t N2, int C, int P, int H2,
                                                           int W2, int R) {
  const int D1 = 2 * R + 1;
  const int64_t total = M * D1 * D1 * P * P;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    int64_t t = idx;
    const int j0 = (int)(t % P); t /= P;
    const int i0 = (int)(t % P); t /= P;
    const int yo = (int)(t % D1); t /= D1;
    const int xo = (int)(t % D1); t /= D1;
    const int64_t m = t;
    const int64_t ix = us[m], jx = vs[m];
    const float x = coords[((m * 2 + 0) * P + i0) * P + j0];
    const float y = coords[((m * 2 + 1) * P + i0) * P + j0];
    const float fxf = floorf(x), fyf = floorf(y);
    const float dx = (float)(T)(x - fxf), dy = (float)(T)(y - fyf);
    const int fx = (int)fminf(fmaxf(fxf, -1.0e6f), 1.0e6f), fy = (int)fminf(fmaxf(fyf, -1.0e6f), 1.0e6f);
    float c[2][2];
    const bool idx_ok = ix >= 0 && ix < N1 && jx >= 0 && jx < N2;
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
      for (int b = 0; b < 2; b++) {
        const int i1 = fy + yo + a - R, j1 = fx + xo + b - R;
        float s = 0.f;
        if (idx_ok && i1 >= 0 && i1 < H2 && j1 >= 0 && j1 < W2) {
          const T* p1 = fmap1 + ((ix * C) * P + i0) * P + j0;
          const T* p2 = fmap2 + ((jx * C) * (int64_t)H2 + i1) * W2 + j1;
          for (int ch = 0; ch < C; ch++) s += (float)p1[(int64_t)ch * P * P] * (float)p2[(int64_t)ch * H2 * W2];
        }
        c[a][b] = s;
      }
    const float v = (1.f - dx) * (1.f - dy) * c[0][0] + dx * (1.f - dy) * c[0][1] + (1.f - dx) * dy * c[1][0] +
                    dx * dy * c[1][1];
    out[idx] = (T)v;
  }
}

// ---- layout kernels -----------------------------------------------------------------------------------
// planar [N][C][H][W] -> channels-last [N][H][W][C]; one thread per (pixel, 8-channel group): 8 strided
// 2-byte reads (coalesced across the wave along W), one 16-byte write.
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const _Float16* __restrict__ src,
                                                           _Float16* __restrict__ dst, int64_t first, int64_t count,
                                                           int C, int H, int W) {
  const int G = C / 8;
  const int64_t total = count * H * W * G;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    // x fastest so that the planar reads of a wave are contiguous
    int64_t t = idx;
    const int xw = (int)(t % W); t /= W;
    const int gq = (int)(t % G); t /= G;
    const int yh = (int)(t % H); t /= H;
    const int64_t nslot = first + t;
    cdv_half8 v;
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = src[((nslot * C + 8 * gq + j) * H + yh) * W + xw];
    *reinterpret_cast<cdv_half8*>(dst + ((nslot * H + yh) * W + xw) * C + 8 * gq) = v;
  }
}

// one frame [C][H][W] -> ring slot of the level-0 NHWC ring and its 4x4 average pool into level 1
// (F.avg_pool2d(fmap, 4, 4), slam.py:682: f16 in, f32 sum of 16, * 1/16, rounded to f16)
__global__ __launch_bounds__(256) void fmap_ingest_kernel(const _Float16* __restrict__ src,
                                                          _Float16* __restrict__ f1_nhwc,
                                                          _Float16* __restrict__ f2_nhwc,
                                                          _Float16* __restrict__ f1_nchw,
                                                          _Float16* __restrict__ f2_nchw, int slot, int C, int H,
                                                          int W) {
  const int G = C / 8, H4 = H / 4, W4 = W / 4;
  const int64_t total = (int64_t)H4 * W4 * G;  // one thread per pooled pixel x channel group: handles a 4x4 block
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    int64_t t = idx;
    const int xq = (int)(t % W4); t /= W4;
    const int gq = (int)(t % G); t /= G;
    const int yq = (int)t;
    float sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
      for (int b = 0; b < 4; b++) {
        const int yh = 4 * yq + a, xw = 4 * xq + b;
        cdv_half8 v;
#pragma unroll
        for (int j = 0; j < 8; j++) {
          const _Float16 s = src[((int64_t)(8 * gq + j) * H + yh) * W + xw];
          v[j] = s;
          sum[j] += (float)s;
          if (f1_nchw) f1_nchw[(((int64_t)slot * C + 8 * gq + j) * H + yh) * W + xw] = s;
        }
        *reinterpret_cast<cdv_half8*>(f1_nhwc + (((int64_t)slot * H + yh) * W + xw) * C + 8 * gq) = v;
      }
    cdv_half8 pv;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      pv[j] = (_Float16)(sum[j] * (1.0f / 16.0f));
      if (f2_nchw) f2_nchw[(((int64_t)slot * C + 8 * gq + j) * H4 + yq) * W4 + xq] = pv[j];
    }
    *reinterpret_cast<cdv_half8*>(f2_nhwc + (((int64_t)slot * H4 + yq) * W4 + xq) * C + 8 * gq) = pv;
  }
}

// patchify forward (correlation_kernel.cu:16-47): gather (2R+2)^2 tiles, zero when OOB
template <typename T>
__global__ __launch_bounds__(256) void patchify_kernel(const T* __restrict__ net, const float* __restrict__ coords,
                                                       T* __restrict__ patches, int B, int64_t M, int C, int H, int W,
                                                       int R) {
  const int D = 2 * R + 2;
  const int64_t total = (int64_t)B * M * C * D * D;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    int64_t t = idx;
    const int b2 = (int)(t % D); t /= D;
    const int a2 = (int)(t % D); t /= D;
    const int ch = (int)(t % C); t /= C;
    const int64_t m = t % M; t /= M;
    const int bb = (int)t;
    const float x = coords[(bb * M + m) * 2 + 0], y = coords[(bb * M + m) * 2 + 1];
    const int i = (int)fminf(fmaxf(floorf(y), -1.0e6f), 1.0e6f) + (a2 - R);
    const int j = (int)fminf(fmaxf(floorf(x), -1.0e6f), 1.0e6f) + (b2 - R);
    T v = (T)0.f;
    if (i >= 0 && i < H && j >= 0 && j < W) v = net[(((int64_t)bb * C + ch) * H + i) * W + j];
    patches[idx] = v;
  }
}

}  // namespace

extern "C" int cdv_corr_fwd(const void* fmap1, const void* fmap2, const float* coords, const int64_t* us,
                            const int64_t* vs, void* out, int64_t M, int64_t N1, int64_t N2, int C, int P, int H2,
                            int W2, int radius, int dtype, void* stream) {
  CDV_REQUIRE(dtype == CDV_F16 || dtype == CDV_F32, CDV_ERR_UNSUPPORTED, "cdv_corr_fwd: dtype must be f16 or f32");
  CDV_REQUIRE(C > 0 && P > 0 && radius >= 0 && H2 > 0 && W2 > 0, CDV_ERR_ARG, "cdv_corr_fwd: bad shape");
  if (M == 0) return CDV_OK;
  const int D1 = 2 * radius + 1;
  const int64_t total = M * D1 * D1 * P * P;
  const int blocks = cdv_div_up(total, 256) < 65536 ? cdv_div_up(total, 256) : 65536;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == CDV_F16)
    hipLaunchKernelGGL(corr_generic_kernel<_Float16>, dim3(blocks), dim3(256), 0, s, (const _Float16*)fmap1,
                       (const _Float16*)fmap2, coords, us, vs, (_Float16*)out, M, N1, N2, C, P, H2, W2, radius);
  else
    hipLaunchKernelGGL(corr_generic_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)fmap1,
                       (const float*)fmap2, coords, us, vs, (float*)out, M, N1, N2, C, P, H2, W2, radius);
  CDV_LAUNCH_CHECK();
  return CDV_OK;
}

extern "C" int cdv_fmap_to_nhwc(const void* src_nchw, void* dst_nhwc, int64_t N, int C, int H, int W, int64_t first,
                                int64_t count, void* stream) {
  CDV_REQUIRE(C % 8 == 0 && C > 0, CDV_ERR_ARG, "cdv_fmap_to_nhwc: C must be a multiple of 8");
  CDV_REQUIRE(first >= 0 && count >= 0 && first + count <= N, CDV_ERR_ARG, "cdv_fmap_to_nhwc: slot range");
  if (count == 0) return CDV_OK;
  const int64_t total = count * H * W * (C / 8);
  const int blocks = cdv_div_up(total, 256) < 16384 ? cdv_div_up(total, 256) : 16384;
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const _Float16*)src_nchw,
                     (_Float16*)dst_nhwc, first, count, C, H, W);
  CDV_LAUNCH_CHECK();
  return CDV_OK;
}

extern "C" int cdv_fmap_ingest(const void* fmap_chw, void* fmap1_nhwc, void* fmap2_nhwc, void* fmap1_nchw,
                               void* fmap2_nchw, int slot, int C, int H, int W, void* stream) {
  CDV_REQUIRE(C % 8 == 0 && C > 0, CDV_ERR_ARG, "cdv_fmap_ingest: C must be a multiple of 8");
  CDV_REQUIRE(H % 4 == 0 && W % 4 == 0, CDV_ERR_ARG, "cdv_fmap_ingest: H and W must be multiples of 4");
  CDV_REQUIRE(slot >= 0, CDV_ERR_ARG, "cdv_fmap_ingest: slot");
  const int64_t total = (int64_t)(H / 4) * (W / 4) * (C / 8);
  const int blocks = cdv_div_up(total, 256);
  hipLaunchKernelGGL(fmap_ingest_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const _Float16*)fmap_chw,
                     (_Float16*)fmap1_nhwc, (_Float16*)fmap2_nhwc, (_Float16*)fmap1_nchw, (_Float16*)fmap2_nchw, slot,
                     C, H, W);
  CDV_LAUNCH_CHECK();
  return CDV_OK;
}

extern "C" int cdv_corr_fused(const void* gmap, const void* fmap0_nhwc, const void* fmap1_nhwc, const float* coords,
                              const int64_t* kk, const int64_t* jj, const int32_t* order, void* out, int64_t E,
                              int64_t Ng, int64_t slots, int C, int H0, int W0, int H1, int W1, float scale0,
                              float scale1, int nlev, int64_t kmod, int64_t jmod, void* stream) {
  CDV_REQUIRE(nlev == 1 || nlev == 2, CDV_ERR_ARG, "cdv_corr_fused: nlev must be 1 or 2");
  CDV_REQUIRE(C % 8 == 0 && C > 0 && C <= 128, CDV_ERR_UNSUPPORTED, "cdv_corr_fused: C must be a multiple of 8, <= 128");
  CDV_REQUIRE(E >= 0 && E < ((int64_t)1 << 31), CDV_ERR_ARG, "cdv_corr_fused: E out of range");
  CDV_REQUIRE(scale0 > 0.f && (nlev == 1 || scale1 > 0.f), CDV_ERR_ARG, "cdv_corr_fused: scales must be positive");
  CDV_REQUIRE(fmap0_nhwc != nullptr && (nlev == 1 || fmap1_nhwc != nullptr), CDV_ERR_ARG, "cdv_corr_fused: NULL map");
  if (E == 0) return CDV_OK;
  LevelParams L0{(const _Float16*)fmap0_nhwc, H0, W0, scale0};
  LevelParams L1{(const _Float16*)fmap1_nhwc, H1, W1, scale1};
  const int blocks = cdv_div_up(E, 4);
  const size_t smem = 4 * (size_t)WAVE_LDS_BYTES;
  hipStream_t s = (hipStream_t)stream;
  if (C <= 32)
    hipLaunchKernelGGL(corr_fused_kernel<1>, dim3(blocks), dim3(256), smem, s, (const _Float16*)gmap, L0, L1, coords,
                       kk, jj, order, (_Float16*)out, (int)E, Ng, slots, C, nlev, kmod, jmod);
  else
    hipLaunchKernelGGL(corr_fused_kernel<4>, dim3(blocks), dim3(256), smem, s, (const _Float16*)gmap, L0, L1, coords,
                       kk, jj, order, (_Float16*)out, (int)E, Ng, slots, C, nlev, kmod, jmod);
  CDV_LAUNCH_CHECK();
  return CDV_OK;
}

extern "C" int cdv_patchify_fwd(const void* net, const float* coords, void* patches, int B, int64_t M, int C, int H,
                                int W, int radius, int dtype, void* stream) {
  CDV_REQUIRE(dtype == CDV_F16 || dtype == CDV_F32, CDV_ERR_UNSUPPORTED, "cdv_patchify_fwd: dtype must be f16 or f32");
  const int D = 2 * radius + 2;
  const int64_t total = (int64_t)B * M * C * D * D;
  if (total == 0) return CDV_OK;
  const int blocks = cdv_div_up(total, 256) < 16384 ? cdv_div_up(total, 256) : 16384;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == CDV_F16)
    hipLaunchKernelGGL(patchify_kernel<_Float16>, dim3(blocks), dim3(256), 0, s, (const _Float16*)net, coords,
                       (_Float16*)patches, B, M, C, H, W, radius);
  else
    hipLaunchKernelGGL(patchify_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)net, coords,
                       (float*)patches, B, M, C, H, W, radius);
  CDV_LAUNCH_CHECK();
  return CDV_OK;
}
