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
//   * v_mfma_f32_16x16x32_f16 produces, per window row, the 16 x 9 correlations (f32 accumulate);
//   * all window rows of BOTH levels are requested before the first MFMA (two dependent memory round
//     trips per edge in total: indices+coordinates, then patch tile + windows);
//   * the raw volume lives only in LDS (f16 like the reference's, 3.6 KB per wave); the 8x8 -> 7x7
//     bilinear blend of every patch pixel reads it back with its own sub-pixel offset, separably;
//   * the [882]-half row of the edge is staged in LDS and leaves as 256-byte coalesced stores.
// Algorithmic HBM bytes per edge: 1764 out + 72 coords + 16 idx (+ the feature maps once): DESIGN.md.
#include <stdlib.h>

#include "cdv_common.h"

CDV_STAMP_TU(corr)

namespace {

// Per-wave LDS: the raw correlation volume of ONE level in f16 (the reference's raw volume is f16 too,
// correlation_kernel.cu:207) + the staged output row of the edge.
constexpr int RAW_ROWS = 12;                    // window rows the fast path holds (typical: 10-11 / 8-9)
constexpr int RAW_MSH = RAW_ROWS * 16 + 8;      // halfs per patch pixel (+8: 16-byte skew between pixels)
constexpr int RAW_HALFS = 9 * RAW_MSH;          // 1800
constexpr int OUT_HALFS = 896;                  // 882 rounded up to a multiple of 64 bytes
constexpr int WAVE_LDS_BYTES = RAW_HALFS * 2 + OUT_HALFS * 2;  // 5,392 B per wave

typedef _Float16 cdv_half4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void wave_lds_sync() {
  // LDS operations of one wave execute in order; this only has to stop the compiler from moving the
  // reads above the writes and to wait for outstanding DS ops (s_waitcnt lgkmcnt(0)).
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// min / max over lanes 0..15 (lanes >= 9 mirror lane 0), result valid in every lane of the group
__device__ __forceinline__ int min16(int v) {
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ int max16(int v) {
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
  return v;
}

struct LevelParams {
  const _Float16* fmap;  // [slots][H][W][C]
  int H, W;
  float scale;
};

// per-level window geometry of one edge (wave-uniform parts live in SGPRs)
struct Box {
  int x0, y0, Wb, Hb;   // union window of the 9 patch pixels: origin and extent
  int ixm, iym;         // per lane (patch pixel m = lane, lanes >= 9 mirror pixel 0): floor of the coordinate
  float dx, dy;         // per lane: sub-pixel offset, rounded to f16 as the reference does before blending
  bool fast;
};

__device__ __forceinline__ Box make_box(float cxm, float cym, float scale) {
  Box b;
  const float x = cxm / scale, y = cym / scale;  // slam.py:321-322 (coords / 1, coords / 4)
  const float fxf = floorf(x), fyf = floorf(y);
  b.dx = (float)(_Float16)(x - fxf);              // correlation_kernel.cu:223-224
  b.dy = (float)(_Float16)(y - fyf);
  b.ixm = (int)fminf(fmaxf(fxf, -1.0e6f), 1.0e6f);
  b.iym = (int)fminf(fmaxf(fyf, -1.0e6f), 1.0e6f);
  b.x0 = __builtin_amdgcn_readfirstlane(min16(b.ixm) - 3);
  b.y0 = __builtin_amdgcn_readfirstlane(min16(b.iym) - 3);
  b.Wb = __builtin_amdgcn_readfirstlane(max16(b.ixm) + 4 - b.x0 + 1);
  b.Hb = __builtin_amdgcn_readfirstlane(max16(b.iym) + 4 - b.y0 + 1);
  b.fast = (b.Wb <= 16) && (b.Hb <= RAW_ROWS);
  return b;
}

// one window row (16 pixels x 8 channels per lane group) of the channels-last map -> MFMA A fragment
template <int KS>
__device__ __forceinline__ void load_row(const _Float16* __restrict__ fbase, const LevelParams& LP, const Box& b,
                                         int t, int n, int g, int C, bool idx_ok, cdv_half8 (&v)[KS]) {
  const int py = b.y0 + t, px = b.x0 + n;
  const bool ok = idx_ok && t < b.Hb && n < b.Wb && py >= 0 && py < LP.H && px >= 0 && px < LP.W;
#pragma unroll
  for (int s = 0; s < KS; s++) {
    cdv_half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    if (ok && (32 * s + 8 * g) < C)
      z = *reinterpret_cast<const cdv_half8*>(fbase + ((size_t)py * LP.W + px) * C + 32 * s + 8 * g);
    v[s] = z;
  }
}

// D[window pixel n][patch pixel m] = sum_c window[n][c] * patch[m][c]; lane (m = lane & 15, g = lane >> 4)
// receives the 4 consecutive window pixels n = 4g .. 4g+3 of row t: one 8-byte LDS store of 4 halfs.
template <int KS>
__device__ __forceinline__ void mfma_row_store(const cdv_half8 (&win)[KS], const cdv_half8 (&pat)[KS],
                                               _Float16* __restrict__ raw, int t, int lane) {
  cdv_float4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < KS; s++) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(win[s], pat[s], acc, 0, 0, 0);
  const int m = lane & 15, g = lane >> 4;
  if (m < 9) {
    cdv_half4 h = {(_Float16)acc[0], (_Float16)acc[1], (_Float16)acc[2], (_Float16)acc[3]};
    *reinterpret_cast<cdv_half4*>(raw + m * RAW_MSH + t * 16 + 4 * g) = h;
  }
}

// 8x8 -> 7x7 bilinear blend of patch pixel m at x offset xo (lane = 7 m + xo), separable; res[yo]
__device__ __forceinline__ void blend_level(const _Float16* __restrict__ raw, const Box& b, int lane,
                                            float (&res)[7]) {
  const int m = lane / 7, xo = lane - 7 * m;
  const int msrc = m < 9 ? m : 0;
  const int bx = b.fast ? (__shfl(b.ixm, msrc) - 3 - b.x0) : 0;
  const int by = b.fast ? (__shfl(b.iym, msrc) - 3 - b.y0) : 0;
  const float dxm = __shfl(b.dx, msrc), dym = __shfl(b.dy, msrc);
  const _Float16* rp = raw + msrc * RAW_MSH + by * 16 + bx + xo;
  float h[8];
#pragma unroll
  for (int r = 0; r < 8; r++) {
    const float c0 = (float)rp[r * 16], c1 = (float)rp[r * 16 + 1];
    h[r] = (1.f - dxm) * c0 + dxm * c1;
  }
#pragma unroll
  for (int yo = 0; yo < 7; yo++) res[yo] = (1.f - dym) * h[yo] + dym * h[yo + 1];
}

// wide reprojection footprint (strong zoom / rotation): every patch pixel gets its own 8x8 window; two
// window rows share one MFMA (pixels 0-7 | 8-15), only column m of D is kept.  Loads are not prefetched.
template <int KS>
__device__ __forceinline__ void slow_level(const _Float16* __restrict__ fbase, const LevelParams& LP, const Box& b,
                                           const cdv_half8 (&pat)[KS], _Float16* __restrict__ raw, int lane, int C,
                                           bool idx_ok) {
  const int n = lane & 15, g = lane >> 4;
  for (int m = 0; m < 9; m++) {
    const int xm = __shfl(b.ixm, m) - 3, ym = __shfl(b.iym, m) - 3;
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
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(v, pat[s], acc, 0, 0, 0);
      }
      // lane (col = lane & 15 = patch pixel, g): window pixels 4g .. 4g+3 of the 16 (two rows of 8)
      if (n == m) {
        const int row = 2 * t2 + (g >> 1), col = 4 * (g & 1);
        cdv_half4 h = {(_Float16)acc[0], (_Float16)acc[1], (_Float16)acc[2], (_Float16)acc[3]};
        *reinterpret_cast<cdv_half4*>(raw + m * RAW_MSH + row * 16 + col) = h;
      }
    }
  }
}

template <int KS>
__global__ __launch_bounds__(256) void corr_fused_kernel(const _Float16* __restrict__ gmap, LevelParams L0,
                                                         LevelParams L1, const float* __restrict__ coords,
                                                         const int64_t* __restrict__ kk,
                                                         const int64_t* __restrict__ jj,
                                                         const int32_t* __restrict__ order,
                                                         _Float16* __restrict__ out, int E, int64_t Ng, int64_t slots,
                                                         int C, int nlev, int64_t kmod, int64_t jmod, int exp) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int p = blockIdx.x * 4 + wave;
  if (p >= E) return;  // no block-wide barriers below: waves are independent
  const int e = order ? order[p] : p;
  CDV_STAMP(corr, p, 0);
  _Float16* raw = reinterpret_cast<_Float16*>(smem_raw + (size_t)wave * WAVE_LDS_BYTES);
  _Float16* outT = raw + RAW_HALFS;

  // ---- round trip 1: indices and the 18 coordinates ------------------------------------------------
  int64_t kpatch = kk[e], jslot = jj[e];
  const int mm = lane < 9 ? lane : 0;  // lanes 0..8 own patch pixel m = lane; the others mirror pixel 0
  const float cxm = coords[(int64_t)e * 18 + mm];
  const float cym = coords[(int64_t)e * 18 + 9 + mm];
  if (kmod > 0) kpatch %= kmod;
  if (jmod > 0) jslot %= jmod;
  const bool idx_ok0 = kpatch >= 0 && kpatch < Ng && jslot >= 0 && jslot < slots;
  const bool idx_ok = idx_ok0 && !(exp & 1);   // experiment bit 0: no window loads
  const int n = lane & 15, g = lane >> 4;

  const Box b0 = make_box(cxm, cym, L0.scale);
  const Box b1 = make_box(cxm, cym, nlev == 2 ? L1.scale : L0.scale);
  CDV_STAMP(corr, p, 1);
  const _Float16* f0 = L0.fmap + (size_t)(idx_ok ? jslot : 0) * L0.H * L0.W * C;
  const _Float16* f1 = nlev == 2 ? L1.fmap + (size_t)(idx_ok ? jslot : 0) * L1.H * L1.W * C : f0;

  // ---- round trip 2: patch tile (MFMA B operand) and, in the common case, EVERY window row of both
  //      levels -- all loads are in flight before the first MFMA -----------------------------------------
  cdv_half8 pat[KS];
#pragma unroll
  for (int s = 0; s < KS; s++) {
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int ch = 32 * s + 8 * g + j;
      _Float16 v = (_Float16)0.f;
      if (idx_ok0 && !(exp & 8) && n < 9 && ch < C) v = gmap[(kpatch * C + ch) * 9 + n];
      pat[s][j] = v;
    }
  }

  float res0[7], res1[7];
  if constexpr (KS == 1) {
    cdv_half8 w0[RAW_ROWS][1], w1[RAW_ROWS][1];
    if (b0.fast) {
#pragma unroll
      for (int t = 0; t < RAW_ROWS; t++) load_row<1>(f0, L0, b0, t, n, g, C, idx_ok, w0[t]);
    }
    if (nlev == 2 && b1.fast) {
#pragma unroll
      for (int t = 0; t < RAW_ROWS; t++) load_row<1>(f1, L1, b1, t, n, g, C, idx_ok, w1[t]);
    }
    CDV_STAMP(corr, p, 2);
    // level 0
    if (b0.fast) {
#pragma unroll
      for (int t = 0; t < RAW_ROWS; t++)
        if (t < b0.Hb) mfma_row_store<1>(w0[t], pat, raw, t, lane);
    } else {
      slow_level<1>(f0, L0, b0, pat, raw, lane, C, idx_ok);
    }
    wave_lds_sync();
    CDV_STAMP(corr, p, 3);
    blend_level(raw, b0, lane, res0);
    CDV_STAMP(corr, p, 4);
    if (nlev == 2) {
      wave_lds_sync();
      if (b1.fast) {
#pragma unroll
        for (int t = 0; t < RAW_ROWS; t++)
          if (t < b1.Hb) mfma_row_store<1>(w1[t], pat, raw, t, lane);
      } else {
        slow_level<1>(f1, L1, b1, pat, raw, lane, C, idx_ok);
      }
      wave_lds_sync();
      CDV_STAMP(corr, p, 5);
      blend_level(raw, b1, lane, res1);
      CDV_STAMP(corr, p, 6);
    }
  } else {
    // wide feature vectors (DPVO, C = 128): rows in batches of 2 to stay inside the register file
    for (int lev = 0; lev < nlev; lev++) {
      const LevelParams& LP = lev == 0 ? L0 : L1;
      const Box& b = lev == 0 ? b0 : b1;
      const _Float16* fb = lev == 0 ? f0 : f1;
      if (lev == 1) wave_lds_sync();
      if (b.fast) {
        for (int tb = 0; tb < b.Hb; tb += 2) {
          cdv_half8 wa[KS], wb[KS];
          load_row<KS>(fb, LP, b, tb, n, g, C, idx_ok, wa);
          load_row<KS>(fb, LP, b, tb + 1, n, g, C, idx_ok, wb);
          mfma_row_store<KS>(wa, pat, raw, tb, lane);
          if (tb + 1 < b.Hb) mfma_row_store<KS>(wb, pat, raw, tb + 1, lane);
        }
      } else {
        slow_level<KS>(fb, LP, b, pat, raw, lane, C, idx_ok);
      }
      wave_lds_sync();
      if (lev == 0) blend_level(raw, b, lane, res0); else blend_level(raw, b, lane, res1);
    }
  }

  // ---- stage the edge's output row [x][y][m][lev] in LDS, then 256-byte coalesced stores ------------
  {
    const int m = lane / 7, xo = lane - 7 * m;
    if (lane < 63) {
      if (nlev == 2) {
        uint32_t* o32 = reinterpret_cast<uint32_t*>(outT);
#pragma unroll
        for (int yo = 0; yo < 7; yo++) {
          const _Float16 h0 = (_Float16)res0[yo], h1 = (_Float16)res1[yo];
          const uint32_t lo = __builtin_bit_cast(unsigned short, h0), hi = __builtin_bit_cast(unsigned short, h1);
          o32[(xo * 7 + yo) * 9 + m] = lo | (hi << 16);
        }
      } else {
#pragma unroll
        for (int yo = 0; yo < 7; yo++) outT[(xo * 7 + yo) * 9 + m] = (_Float16)res0[yo];
      }
    }
  }
  wave_lds_sync();
  CDV_STAMP(corr, p, 7);
  if (exp & 2) return;                          // experiment bit 1: no global store
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
  CDV_STAMP(corr, p, 8);
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
  static const int exp = getenv("CDV_CORR_EXP") ? atoi(getenv("CDV_CORR_EXP")) : 0;  // diagnostics only
  hipStream_t s = (hipStream_t)stream;
  if (C <= 32)
    hipLaunchKernelGGL(corr_fused_kernel<1>, dim3(blocks), dim3(256), smem, s, (const _Float16*)gmap, L0, L1, coords,
                       kk, jj, order, (_Float16*)out, (int)E, Ng, slots, C, nlev, kmod, jmod, exp);
  else
    hipLaunchKernelGGL(corr_fused_kernel<4>, dim3(blocks), dim3(256), smem, s, (const _Float16*)gmap, L0, L1, coords,
                       kk, jj, order, (_Float16*)out, (int)E, Ng, slots, C, nlev, kmod, jmod, exp);
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
