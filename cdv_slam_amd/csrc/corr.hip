// corr.hip -- altcorr: patch-to-frame local correlation + bilinear blend, gfx950.
//
// Reference: cdvslam/altcorr/correlation_kernel.cu:82-136 (27.5 M threads per level, each doing 24
// strided 2-byte loads from channel-planar maps), :213-232 (raw 8x8 volume written to HBM and re-read
// by four ATen slice-multiply-add passes), then torch.stack of the two levels (slam.py:323).
//
// Here (cdv_corr_fused): ONE launch for both pyramid levels.  One wave per edge:
//   * the union window of the 9 patch pixels (<= 16 x 12 feature pixels, typically 10 x 11) is packed densely into the
//     16 pixel slots of the MFMA A operand, loaded straight from a CHANNELS-LAST padded feature ring (one pixel = C
//     contiguous halves, so a lane's 8 k-values are one 16-byte buffer load; the descriptor's range check and the zero
//     margins give the reference's out-of-bounds rule -- no LDS staging of the inputs, no masks);
//   * the 3x3xC patch tile is the B operand (9 of 16 columns used), one 16-byte load from the pixel-major tile;
//   * v_mfma_f32_16x16x32_f16 produces 16 window pixels x 9 patch pixels per instruction (f32 accumulate);
//   * every window group of a level is requested before its first MFMA, the level-1 request flies under the level-0
//     blend (two dependent memory round trips per edge: indices + coordinates, then patch tile + windows);
//   * the raw volume lives only in LDS (f16 like the reference's, 3.7 KB per wave, conflict-free row stride); the
//     8x8 -> 7x7 bilinear blend of every patch pixel reads it back with its own sub-pixel offset, separably;
//   * the [882]-half row of the edge is staged in LDS (over the raw volume) and leaves as 16-byte-per-lane stores.
// corr_fused2_kernel is the product kernel (C <= 32: CDV-SLAM's DIMF = 24); corr_wide_kernel serves C up to 128 (DPVO).
// Algorithmic HBM bytes per edge: 1764 out + 72 coords + 16 idx (+ the feature maps once): DESIGN.md.
#include <math.h>
#include <stdlib.h>

#include "cdv_common.h"
#include "cdv_graph.h"
#include "cdv_parts.h"

CDV_STAMP_TU(corr)

// ablation switches of the diagnostic build (make STAMPS=1, env CDV_CORR_EXP): compiled out of the product library
#ifdef CDV_STAMPS
#define CDV_EXP(bit) ((exp & (bit)) != 0)
#else
#define CDV_EXP(bit) false
#endif

namespace {

// HBM layout of the feature rings the fused kernel gathers from ("padded channels-last"):
//   [slot][H + 2*PADY][W + 2*PADX][C] f16, zero margins.  A 16 x 12 window box whose origin is clamped to
//   [-PADX, W] x [-PADY, H] never leaves the allocation, and everything it reads outside the image is the
//   zero the reference's out-of-bounds rule asks for (correlation_kernel.cu:122) -- no per-lane bounds
//   tests, no masks in the kernel.
constexpr int PADX = CDV_FMAP_PADX;             // 16
constexpr int PADY = CDV_FMAP_PADY;             // 12

// Per-wave LDS: the raw correlation volume of ONE level in f16 (the reference's raw volume is f16 too,
// correlation_kernel.cu:207) + the staged output row of the edge.
constexpr int RAW_ROWS = 12;                    // window rows the fast path holds (typical: 10-11 / 8-9)
constexpr int NQ_MAX = 8;                       // 16-pixel groups of the densely packed window kept in registers at a time (144 pixels; typical
                                                // windows: 110-121 / 81 pixels; larger ones take a second round): 78 VGPRs,
                                                // 6 waves per SIMD
// halfs per patch pixel: 102 dwords = 6 (mod 32), so the 16 lanes of a ds_write_b64 group (patch pixels n = 0..15 of
// one MFMA D tile, 2 banks each) cover all 32 banks exactly once (with the earlier 100 dwords the stores were 3-way
// conflicted: 112 of 296 LDS cycles per edge)
constexpr int RAW_MSH = RAW_ROWS * 16 + 12;
constexpr int RAW_HALFS = 9 * RAW_MSH;          // 1836
#ifndef CDV_CORR_WAVES
#define CDV_CORR_WAVES 4        // waves (= edges) per workgroup of the product kernel; the waves are independent (no barrier)
#endif
#ifndef CDV_CORR_EPW
#define CDV_CORR_EPW 1          // edges per wave (packed-stream instance only): > 1 requests the next record under the current edge
#endif
#ifndef CDV_CORR_OUT_POLICY
#define CDV_CORR_OUT_POLICY 16   // cache policy of the product kernel's output stores: 0 plain, 16 sc1 (write-through), 2 nt
#endif
constexpr int OUT_HALFS = 896;                  // the staged output row: 882 halfs, linear (the copy-out needs no index math)
constexpr int WAVE_LDS_BYTES = RAW_HALFS * 2 + OUT_HALFS * 2;  // wide kernel: raw volume + staged row
// product kernel: the staged row re-uses the raw volume (written after the last blend has read it); the D-tile lanes
// that are not patch pixels (n >= 9) store into a 512-byte dump on the banks the same lanes of a full 16-row layout
// would use
constexpr int DUMP_OFF_B = 3712;                // >= RAW_HALFS * 2, a multiple of 128 B (bank 0)
constexpr int WAVE_LDS2_BYTES = DUMP_OFF_B + 512;
static_assert(RAW_HALFS * 2 <= DUMP_OFF_B && OUT_HALFS * 2 <= RAW_HALFS * 2, "LDS layout");

typedef _Float16 cdv_half4 __attribute__((ext_vector_type(4)));
typedef _Float16 cdv_half2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void wave_lds_sync() {
  // LDS operations of one wave execute in order; this only has to stop the compiler from moving the
  // reads above the writes and to wait for outstanding DS ops.  (Not a workgroup-scope release fence: that one
  // also drains vmcnt, i.e. waits for the window loads of the NEXT level that are meant to fly under the blend.)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

// all-reduce (min or max) of an int over the 16 lanes of a DPP row by rotation (row_ror 8,4,2,1): the DPP operand is
// folded into the min / max itself (one VALU instruction per step; integers need no NaN canonicalisation).  The
// s_nop covers the two wait states between a VALU write and a DPP read of the same register.
template <bool IS_MIN>
__device__ __forceinline__ int row16_reduce_i32(int v) {
#define CDV_ROR(n)                                                                                             \
  if (IS_MIN) asm volatile("s_nop 1\n\tv_min_i32_dpp %0, %1, %1 row_ror:" #n " row_mask:0xf bank_mask:0xf" : "=v"(v) : "0"(v)); \
  else asm volatile("s_nop 1\n\tv_max_i32_dpp %0, %1, %1 row_ror:" #n " row_mask:0xf bank_mask:0xf" : "=v"(v) : "0"(v));
  CDV_ROR(8) CDV_ROR(4) CDV_ROR(2) CDV_ROR(1)
#undef CDV_ROR
  return v;
}

// the four extremes the window boxes need (min / max of x and of y) with the four rotation chains interleaved: every
// DPP read is three instructions behind the write of its register, so no wait states (s_nop) are needed at all
__device__ __forceinline__ void row16_minmax4(int x, int y, int& xmin, int& xmax, int& ymin, int& ymax) {
  int a = x, b = x, c = y, d = y;
#define CDV_STEP(n)                                                                                  \
  asm volatile("v_min_i32_dpp %0, %0, %0 row_ror:" #n " row_mask:0xf bank_mask:0xf\n\t"               \
               "v_max_i32_dpp %1, %1, %1 row_ror:" #n " row_mask:0xf bank_mask:0xf\n\t"               \
               "v_min_i32_dpp %2, %2, %2 row_ror:" #n " row_mask:0xf bank_mask:0xf\n\t"               \
               "v_max_i32_dpp %3, %3, %3 row_ror:" #n " row_mask:0xf bank_mask:0xf"                    \
               : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
  asm volatile("s_nop 1" ::: "memory");   // the copies above may be the instructions right before the first DPP read
  CDV_STEP(8) CDV_STEP(4) CDV_STEP(2) CDV_STEP(1)
#undef CDV_STEP
  xmin = a; xmax = b; ymin = c; ymax = d;
}

struct LevelParams {
  const _Float16* fmap;  // [slots][H + 2 PADY][W + 2 PADX][C]
  int H, W;
  float inv_scale;       // 1 / scale, scale a power of two: x * inv_scale == x / scale exactly (slam.py:321-322)
  int shift;             // log2(scale / scale of level 0) >= 0: floor(x / scale) == floor(x / scale0) >> shift
};

// wave-uniform window box of one level (all members live in SGPRs)
struct Box {
  int x0, y0;        // unclamped origin of the union window of the 9 patch pixels
  int Wb, Hb;        // extent
  bool fast;         // fits the 16 x RAW_ROWS tile
  int stride;        // halfs between window rows of the raw volume in LDS (dense: Wb; per-pixel path: 16)
  bool outside;      // fast && entirely outside the padded image: the level is all zeros
  int x0c, y0c;      // origin clamped into the padded map
};

__device__ __forceinline__ int floor_clamped(float v) {
  return (int)fminf(fmaxf(floorf(v), -1.0e6f), 1.0e6f);
}

// ixmin .. iymax: extreme floor(coordinate / scale) over the 9 patch pixels (floor is monotone, so the extreme pixels
// give the extreme integer coordinates)
__device__ __forceinline__ Box make_box(int ixmin, int ixmax, int iymin, int iymax, const LevelParams& LP) {
  Box b;
  b.x0 = ixmin - 3;   // slam.py:321-322, correlation_kernel.cu:107-110
  b.y0 = iymin - 3;
  b.Wb = ixmax + 4 - b.x0 + 1;
  b.Hb = iymax + 4 - b.y0 + 1;
  b.fast = (b.Wb <= 16) && (b.Hb <= RAW_ROWS);
  b.stride = b.fast ? b.Wb : 16;
  b.x0c = min(max(b.x0, -PADX), LP.W);
  b.y0c = min(max(b.y0, -PADY), LP.H);
  b.outside = (b.x0c != b.x0) || (b.y0c != b.y0);  // a clamped <=16 x <=12 box lies entirely in the margin
  return b;
}

// D[window pixel n][patch pixel m] = sum_c window[n][c] * patch[m][c]; lane (m = lane & 15, g = lane >> 4)
// receives the 4 consecutive window pixels n = 4g .. 4g+3 of row t: one 8-byte LDS store of 4 halfs.
template <int KS>
__device__ __forceinline__ void mfma_row_store(const cdv_half8 (&win)[KS], const cdv_half8 (&pat)[KS],
                                               _Float16* __restrict__ raw_lane, int t) {
  cdv_float4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < KS; s++) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(win[s], pat[s], acc, 0, 0, 0);
  cdv_half4 h = {(_Float16)acc[0], (_Float16)acc[1], (_Float16)acc[2], (_Float16)acc[3]};
  *reinterpret_cast<cdv_half4*>(raw_lane + t * 16) = h;
}

// per-lane blend geometry of one level: lane = 7 m + xo
struct BlendGeo {
  int hb;         // BYTE offset of the half raw[m][by][bx + xo]
  int stride_b;   // bytes between window rows
  cdv_half2 wx;   // (1 - dx, dx): horizontal taps, dx rounded to f16 as the reference casts it before blending
  float dy;       // vertical sub-pixel offset (rounded to f16 likewise)
};

__device__ __forceinline__ BlendGeo blend_geo(float xm, float ym, int m, int xo, const LevelParams& LP,
                                              const Box& b, bool per_pixel) {
  BlendGeo g;
  const float x = xm * LP.inv_scale, y = ym * LP.inv_scale;
  const float fxf = floorf(x), fyf = floorf(y);
  const float dx = (float)(_Float16)(x - fxf);   // correlation_kernel.cu:223-224
  g.wx = cdv_half2{(_Float16)(1.0f - dx), (_Float16)dx};
  g.dy = (float)(_Float16)(y - fyf);
  const int bx = per_pixel ? 0 : floor_clamped(x) - 3 - b.x0;
  const int by = per_pixel ? 0 : floor_clamped(y) - 3 - b.y0;
  g.hb = 2 * (m * RAW_MSH + __mul24(by, b.stride) + bx + xo);   // |by| is small: not the quarter-rate v_mul_lo_u32
  g.stride_b = 2 * b.stride;
  return g;
}

// 8x8 -> 7x7 bilinear blend, separable.  The pair (c0, c1) of a window row starts at an arbitrary half offset; a
// 4-byte LDS read at an odd half offset is an UNALIGNED access (very slow on the LDS), so the two aligned dwords
// around it are read (one ds_read2_b32) and funnel-shifted (v_alignbit) into the pair; the horizontal tap is one
// v_dot2_f32_f16 (exact f16 products, f32 sum).  5 VALU + 1 LDS instruction per window row.
__device__ __forceinline__ float dot2_f16(uint32_t pair, uint32_t w) {
  // v_dot2_f32_f16 with a literal zero accumulator (the compiler's pick, v_dot2c, needs a v_mov 0 every time)
  float r;
  asm("v_dot2_f32_f16 %0, %1, %2, 0" : "=v"(r) : "v"(pair), "v"(w));
  return r;
}

// The vertical tap and the rounding of the result to f16 are written as one expression, (f16)fma(dy, h1 - h0, h0): the
// compiler emits v_fma_mixlo/hi_f16 (f32 arithmetic, ONE rounding to f16) in every kernel that inlines this.
// (The wide kernel keeps float results and rounds when it packs them: its conditionally blended, zero-initialised f16
// results were mis-assembled into v_fma_mixhi_f16 chains by the compiler -- every second value wrong.)
template <typename TR>
__device__ __forceinline__ void blend_level(const _Float16* __restrict__ raw, const BlendGeo& g, TR (&res)[7]) {
  const char* rb = reinterpret_cast<const char*>(raw);   // raw is 16-byte aligned
  float h[8];
  // even and odd rows keep their own aligned LDS pointer and funnel shift: two rows further down the byte offset
  // moves by 2 stride_b = 4 stride, a multiple of 4, so neither alignment changes -- 3 VALU per row
  // (pointer add, v_alignbit, v_dot2)
  const int hb1 = g.hb + g.stride_b;
  const char* p0 = rb + (g.hb & ~3);
  const char* p1 = rb + (hb1 & ~3);
  const unsigned s0 = (unsigned)g.hb << 3, s1 = (unsigned)hb1 << 3;   // shift = low 5 bits: 0 or 16
  const int step2 = 2 * g.stride_b;
  const uint32_t wxb = __builtin_bit_cast(uint32_t, g.wx);
#pragma unroll
  for (int r = 0; r < 8; r++) {
    const uint32_t* pd = reinterpret_cast<const uint32_t*>((r & 1) ? p1 : p0);
    const uint32_t lo = pd[0], hi = pd[1];           // hi is unused when the pair is dword aligned (stays in the wave's LDS)
    const uint32_t pr = __builtin_amdgcn_alignbit(hi, lo, (r & 1) ? s1 : s0);
    h[r] = dot2_f16(pr, wxb);
    if (r & 1) p1 += step2; else p0 += step2;
  }
#pragma unroll
  for (int yo = 0; yo < 7; yo++) res[yo] = (TR)__builtin_fmaf(g.dy, h[yo + 1] - h[yo], h[yo]);
}

// wide reprojection footprint (strong zoom / rotation): every patch pixel gets its own 8x8 window; two
// window rows share one MFMA (pixels 0-7 | 8-15), only column m of D is kept.  Rare: compact, not tuned.
// (xv, yv): the coordinates of patch pixel m are in lane LM * m
template <int KS, int LM = 1>
__device__ __forceinline__ void slow_level(const LevelParams& LP, int64_t jslot, float xv, float yv,
                                           const cdv_half8 (&pat)[KS], _Float16* __restrict__ raw, int lane, int C) {
  const int n = lane & 15, g = lane >> 4;
  const int Wp = LP.W + 2 * PADX, Hp = LP.H + 2 * PADY;
  const _Float16* fbase = LP.fmap + (size_t)jslot * Hp * Wp * C;
#pragma unroll 1   // rare path: kept rolled, so that it adds no registers to the kernel (occupancy of the common path)
  for (int m = 0; m < 9; m++) {
    const float xm = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xv), LM * m));
    const float ym = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(yv), LM * m));
    const int x0 = floor_clamped(xm * LP.inv_scale) - 3, y0 = floor_clamped(ym * LP.inv_scale) - 3;
    const int x0c = min(max(x0, -PADX), LP.W), y0c = min(max(y0, -PADY), LP.H);
    const bool outside = (x0c != x0) || (y0c != y0);
#pragma unroll 1
    for (int t2 = 0; t2 < 4; t2++) {
      const int py = y0c + PADY + 2 * t2 + (n >> 3), px = x0c + PADX + (n & 7);
      cdv_float4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < KS; s++) {
        const int chg = min(32 * s + 8 * g, C - 8);  // lanes of the k padding re-read real (finite) data
        const cdv_half8 v = *reinterpret_cast<const cdv_half8*>(fbase + ((size_t)py * Wp + px) * C + chg);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(v, pat[s], acc, 0, 0, 0);
      }
      if (n == m) {
        const int row = 2 * t2 + (g >> 1), col = 4 * (g & 1);
        cdv_half4 h = {(_Float16)acc[0], (_Float16)acc[1], (_Float16)acc[2], (_Float16)acc[3]};
        if (outside) h = cdv_half4{0, 0, 0, 0};
        *reinterpret_cast<cdv_half4*>(raw + m * RAW_MSH + row * 16 + col) = h;
      }
    }
  }
}

// a level that contributes nothing (window entirely off the padded map, invalid indices): the raw volume is zeroed and
// the blend runs as always -- no per-value selects, no zero-initialised result registers on the common path
__device__ __forceinline__ void zero_raw(_Float16* __restrict__ raw, int lane) {
  typedef uint32_t cdv_u32x4z __attribute__((ext_vector_type(4)));
  const cdv_u32x4z z = {0u, 0u, 0u, 0u};
  for (int i = lane; i < RAW_HALFS / 8; i += 64) reinterpret_cast<cdv_u32x4z*>(raw)[i] = z;
}

// ---- wide feature vectors (DPVO, C = 128; KS = C / 32 k-steps): one 16-pixel tile per window row ---------------
template <int KS>
__global__ __launch_bounds__(256) void corr_wide_kernel(const _Float16* __restrict__ gmap, LevelParams L0,
                                                         LevelParams L1, const float* __restrict__ coords,
                                                         const int64_t* __restrict__ kk,
                                                         const int64_t* __restrict__ jj,
                                                         const int32_t* __restrict__ order,
                                                         _Float16* __restrict__ out, int E, int64_t Ng, int64_t slots,
                                                         int C, int nlev, int64_t kmod, int64_t jmod, uint32_t kmagic,
                                                         uint32_t jmagic, int gmap_pm,
                                                         int exp) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  _Float16* raw = reinterpret_cast<_Float16*>(smem_raw + (size_t)wave * WAVE_LDS_BYTES);
  _Float16* outT = raw + RAW_HALFS;
  const int p = blockIdx.x * 4 + wave;
  if (p >= E) return;  // no block-wide barriers below: waves are independent
  const int e = order ? __builtin_amdgcn_readfirstlane(order[p]) : p;   // wave-uniform: scalar loads below
  CDV_STAMP(corr, p, 0);

  // ---- round trip 1: indices (scalar) and the 18 coordinates ----------------------------------------
  int64_t kpatch = kk[e], jslot = jj[e];
  // index % modulus (slam.py:319-320) with the host's reciprocal: q = mulhi(x, ceil(2^32 / d)) is x / d or one more
  // for 0 <= x < 2^31, so one correction step; the 64-bit division the compiler would emit is ~150 scalar instructions
  if (kmod > 0 && kpatch >= 0 && kpatch < ((int64_t)1 << 31)) {
    const uint32_t x = (uint32_t)kpatch, d = (uint32_t)kmod;
    const uint32_t q = __builtin_amdgcn_readfirstlane((int)__umulhi(x, kmagic));
    const int32_t r = (int32_t)(x - q * d);
    kpatch = (d == 1u) ? 0 : (r < 0 ? r + (int32_t)d : r);
  }
  if (jmod > 0 && jslot >= 0 && jslot < ((int64_t)1 << 31)) {
    const uint32_t x = (uint32_t)jslot, d = (uint32_t)jmod;
    const uint32_t q = __builtin_amdgcn_readfirstlane((int)__umulhi(x, jmagic));
    const int32_t r = (int32_t)(x - q * d);
    jslot = (d == 1u) ? 0 : (r < 0 ? r + (int32_t)d : r);
  }
  const bool idx_ok = kpatch >= 0 && kpatch < Ng && jslot >= 0 && jslot < slots;
  if (!idx_ok) { kpatch = 0; jslot = 0; }  // reference behaviour is undefined here; stay in bounds
  if (CDV_EXP(16)) jslot = 0;                 // experiment bit 4: every edge reads map slot 0 (L2-resident)
  const float* cptr = coords + (size_t)e * 18;
  const int mm = lane < 9 ? lane : 0;       // row 0 of the wave: lanes 0..8 own patch pixel m, 9..15 mirror 0
  const int bm = min(lane / 7, 8), bxo = lane - 7 * (lane / 7);  // blend role of this lane: (m, x offset)
  // ONE vector load for the 18 coordinates (lane l < 18 holds value l), handed to the lanes that need them through
  // the LDS crossbar (ds_bpermute): the vector-memory pipe is the busy one in this kernel
  const int cval = __float_as_int(cptr[min(lane, 17)]);
  const float xv = __int_as_float(__builtin_amdgcn_ds_bpermute(4 * mm, cval));
  const float yv = __int_as_float(__builtin_amdgcn_ds_bpermute(4 * (9 + mm), cval));
  const float xb = __int_as_float(__builtin_amdgcn_ds_bpermute(4 * bm, cval));
  const float yb = __int_as_float(__builtin_amdgcn_ds_bpermute(4 * (9 + bm), cval));
  // integer pixel coordinates at level 0 per lane, their extremes over the patch (DPP row reductions), and from
  // those -- all scalar from here -- the window boxes of both levels (level 1: an arithmetic shift)
  const int ixl = floor_clamped(xv * L0.inv_scale), iyl = floor_clamped(yv * L0.inv_scale);
  const int ixmin = __builtin_amdgcn_readfirstlane(row16_reduce_i32<true>(ixl));
  const int ixmax = __builtin_amdgcn_readfirstlane(row16_reduce_i32<false>(ixl));
  const int iymin = __builtin_amdgcn_readfirstlane(row16_reduce_i32<true>(iyl));
  const int iymax = __builtin_amdgcn_readfirstlane(row16_reduce_i32<false>(iyl));
  const Box b0 = make_box(ixmin, ixmax, iymin, iymax, L0);
  const int sh1 = nlev == 2 ? L1.shift : 0;
  const Box b1 = make_box(ixmin >> sh1, ixmax >> sh1, iymin >> sh1, iymax >> sh1, nlev == 2 ? L1 : L0);
  CDV_STAMP(corr, p, 1);

  const int n = lane & 15, g = lane >> 4;
  // per-lane byte offset inside a window row: pixel n, channel group g (the k-padding group re-reads real,
  // finite data: its products meet the zero k-slots of the patch fragment)
  // ---- round trip 2: patch tile (MFMA B operand) ------------------------------------------------------
  cdv_half8 pat[KS];
  if (gmap_pm) {
    // pixel-major tiles [9][C]: one 16-byte buffer load per k-step; the descriptor covers exactly this tile, so the
    // lanes that are not patch pixels (n >= 9) and the k padding (forced offset) read zeros from the range check
    typedef int cdv_i32x4p __attribute__((ext_vector_type(4)));
    const unsigned tile_b = 9u * (unsigned)C * 2u;
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(gmap) + (size_t)kpatch * tile_b), (short)0, (int)tile_b, 0x00020000);
#pragma unroll
    for (int s = 0; s < KS; s++) {
      const int chg = 32 * s + 8 * g;
      const unsigned voff = (chg < C) ? (unsigned)(n * C + chg) * 2u : 0x40000000u;
      pat[s] = __builtin_bit_cast(cdv_half8, (cdv_i32x4p)__builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, 0, 0));
    }
  } else {
    const _Float16* gp = gmap + (size_t)kpatch * C * 9 + (n < 9 ? n : 0);
#pragma unroll
    for (int s = 0; s < KS; s++) {
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const int ch = 32 * s + 8 * g + j;
        const _Float16 v = gp[min(ch, C - 1) * 9];
        pat[s][j] = (n < 9 && ch < C) ? v : (_Float16)0.f;
      }
    }
  }

  float res0[7], res1[7];
#pragma unroll
  for (int i = 0; i < 7; i++) { res0[i] = 0.f; res1[i] = 0.f; }
  // lanes n >= 9 of the D tile are not patch pixels: they store into the (not yet used) staging area instead of
  // being masked off (no EXEC save/restore around every store)
  _Float16* raw_lane = (n < 9) ? raw + n * RAW_MSH + 4 * g : outT + ((n - 9) * 4 + g) * 4;

  // wide feature vectors (DPVO, C = 128): rows in batches of 2 to stay inside the register file
  for (int lev = 0; lev < nlev; lev++) {
    const LevelParams& LP = lev == 0 ? L0 : L1;
    Box b = lev == 0 ? b0 : b1;
    b.stride = 16;  // this path stores one 16-pixel tile per window row
    const int Wp = LP.W + 2 * PADX, Hp = LP.H + 2 * PADY;
    if (lev == 1) wave_lds_sync();
    if (b.fast && !b.outside) {
      const _Float16* fb = LP.fmap + (((size_t)jslot * Hp + (b.y0c + PADY)) * Wp + (b.x0c + PADX)) * C;
      for (int t = 0; t < b.Hb; t++) {
        cdv_half8 wa[KS];
#pragma unroll
        for (int s = 0; s < KS; s++)
          wa[s] = *reinterpret_cast<const cdv_half8*>(fb + ((size_t)t * Wp + n) * C + min(32 * s + 8 * g, C - 8));
        cdv_float4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; s++) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[s], pat[s], acc, 0, 0, 0);
        if (n < 9) {
          cdv_half4 h = {(_Float16)acc[0], (_Float16)acc[1], (_Float16)acc[2], (_Float16)acc[3]};
          *reinterpret_cast<cdv_half4*>(raw_lane + t * 16) = h;
        }
      }
    } else if (!b.fast) {
      slow_level<KS>(LP, jslot, xv, yv, pat, raw, lane, C);
    }
    wave_lds_sync();
    if (!(b.fast && b.outside)) {
      const BlendGeo gg = blend_geo(xb, yb, bm, bxo, LP, b, !b.fast);
      if (lev == 0) blend_level(raw, gg, res0); else blend_level(raw, gg, res1);
    }
  }

  // ---- stage the edge's output row [x][y][m][lev] in LDS, then 256-byte coalesced stores ------------
  if (lane < 63 && !CDV_EXP(512)) {
    if (nlev == 2) {
      uint32_t* o32 = reinterpret_cast<uint32_t*>(outT) + bxo * 63 + bm;   // dword (x, y, m) = 63 x + 9 y + m
#pragma unroll
      for (int yo = 0; yo < 7; yo++) {
        const _Float16 h0 = idx_ok ? (_Float16)res0[yo] : (_Float16)0.f;
        const _Float16 h1 = idx_ok ? (_Float16)res1[yo] : (_Float16)0.f;
        const uint32_t lo = __builtin_bit_cast(unsigned short, h0), hi = __builtin_bit_cast(unsigned short, h1);
        o32[yo * 9] = lo | (hi << 16);
      }
    } else {
#pragma unroll
      for (int yo = 0; yo < 7; yo++) outT[(bxo * 7 + yo) * 9 + bm] = idx_ok ? (_Float16)res0[yo] : (_Float16)0.f;
    }
  }
  wave_lds_sync();
  CDV_STAMP(corr, p, 7);
  if (!CDV_EXP(2)) {                             // experiment bit 1: no global store
  if (nlev == 2) {
    // 441 dwords: two 16-byte-per-lane stores (256 + 184 dwords) and one last dword.  The row starts on a 4-byte
    // boundary only (1764 B per edge); global memory takes the unaligned 16-byte accesses.
    const uint32_t* src = reinterpret_cast<const uint32_t*>(outT);
    uint32_t* dst = reinterpret_cast<uint32_t*>(out) + (size_t)e * 441;
    typedef uint32_t cdv_u32x4 __attribute__((ext_vector_type(4)));
    typedef uint32_t cdv_u32x4u __attribute__((ext_vector_type(4), aligned(4)));
    const cdv_u32x4 v0 = *reinterpret_cast<const cdv_u32x4*>(src + 4 * lane);
    *reinterpret_cast<cdv_u32x4u*>(dst + 4 * lane) = v0;
    if (lane < 46) {
      const cdv_u32x4 v1 = *reinterpret_cast<const cdv_u32x4*>(src + 256 + 4 * lane);
      *reinterpret_cast<cdv_u32x4u*>(dst + 256 + 4 * lane) = v1;
    }
    if (lane == 63) dst[440] = src[440];
  } else {
    _Float16* dst = out + (size_t)e * 441;
#pragma unroll
    for (int i = 0; i < 7; i++) {
      const int t = i * 64 + lane;
      if (t < 441) dst[t] = outT[t];
    }
  }
  }
  CDV_STAMP(corr, p, 8);
}

// ---- the product kernel (C <= 32) ------------------------------------------------------
// One wave per edge, both levels (header of this file); built around what bounds it on MI355X -- not bytes and not one saturated unit, but the serial chain of one wave (6 waves per SIMD;
// ~14,000 cycles per edge of which ~1,600 are its VALU issue) and the scalar unit the four SIMDs of a CU share:
//   * every argument the first instructions need sits at the front of ONE argument struct (one scalar load, not six
//     dependent ones); strides, pitches and the (biased) ring bases are precomputed by the host;
//   * the coordinate load is issued before the index loads are waited for (one memory round trip, not two);
//   * indices are handled in 32-bit scalar arithmetic (a 64-bit compare is a VALU instruction on this ISA);
//   * window addressing in float arithmetic, 3 VALU per load (see CDV_LD1);
//   * no result selects: a level that contributes nothing gets a zeroed raw volume, both levels of one output dword
//     are ONE v_cvt_pk_f16_f32;
//   * workgroup b runs on XCD b % 8: XCD x takes the x-th contiguous eighth of the edge list (the edges of one target
//     frame are neighbours in the list, so a frame's windows meet in one L2 instead of eight).
struct CorrLevel {
  const char* base_m;   // ring base MINUS 0x4B000000 bytes (the float-bit-pattern offsets of the window loads carry that bias)
  int H, W;
  float inv_scale;
  int shift;
  uint32_t pitch;       // bytes per padded row
  uint32_t slot_bytes;  // bytes per slot
};
struct CorrArgs2 {
  const float* coords;
  const int64_t* kk;
  const int64_t* jj;
  const int32_t* order;
  int E;
  uint32_t kmod, jmod, kmagic, jmagic;
  uint32_t Ng, slots;
  const char* gmap;     // tiles, f16: pixel-major [Ng][9][C] (gmap_pm) or the reference's planar [Ng][C][3][3]
  _Float16* out;
  CorrLevel L0, L1;
  int C;
  int gmap_pm;
  int exp;
  // one-level launches only: output element (e, t) at out[e * out_pitch + t * out_stride + out_off] (inside the row of a
  // two-level result), and edges whose coordinates equal coords_ref * ref_mul bit for bit are skipped (their values are
  // in place already: cdv_corr_level_checked)
  int out_stride, out_off, out_pitch;   // out_pitch: halves from one edge's row to the next
  const float* coords_ref;
  float ref_mul;
  // two-level launches: != 0 keeps the levels apart, [E][2][442] halves (row 884 B: each level a contiguous run of 441 --
  // what two separate one-level results look like to torch.stack) instead of [E][441][2]
  int split;
  // packed per-edge input stream in PROCESSING order (written by the index build next to `order`): record p =
  // {18 coords, edge id, patch-ring index, frame-ring index (0xFFFFFFFF: invalid), ...} -- cdv_graph.h CORR_REC_WORDS
  const uint32_t* rec;
  // != NULL: the number of edges is *dynE (include/cdvslam_hip.h CDV_DYN_E); E above bounds it and sized the launch
  const int32_t* dynE;
};

constexpr uint32_t FBIAS = 0x4B000000u;   // bit pattern of 2^23

__device__ __forceinline__ LevelParams level_params(const CorrLevel& L) {
  return LevelParams{reinterpret_cast<const _Float16*>(L.base_m + (size_t)FBIAS), L.H, L.W, L.inv_scale, L.shift};
}

// one edge; its first round trip (coordinates: lane l < 18 holds value l; ring indices already reduced and checked)
// was issued by the caller
// what the caller's first round trip brought: either the 18 coordinates spread over the lanes (cval: lane l < 18 holds
// value l; the box is then found here), or -- from the packed input stream -- this lane's blend coordinates directly and
// the wave-uniform extremes of floor(x), floor(y) over the patch (level 0) as scalars
struct EdgeCoords {
  int cval;
  float xb, yb;
  int ixmin, ixmax, iymin, iymax;
};

// the wave's NEXT record, requested inside corr_edge behind the level-0 window loads (experiment builds with more than one
// edge per wave, CDV_CORR_EPW): two gathers for this lane's blend coordinates, the record's scalar words as scalar loads
struct NextRec {
  float xb, yb;
  uint32_t w18, w19, w20, w21, w22;
};

template <int CC, int NLEV, bool SPLIT, bool REC>
__device__ __forceinline__ void corr_edge(const CorrArgs2& a, int p, int e, const EdgeCoords& ec, uint32_t kq, uint32_t jq,
                                          bool idx_ok, int lane, _Float16* __restrict__ raw, _Float16* __restrict__ outT,
                                          const uint32_t* __restrict__ rnext = nullptr, NextRec* nx = nullptr) {
  CDV_STAMP(corr, p, 0);
  const int C = CC ? CC : a.C;
#ifdef CDV_STAMPS
  const int exp = a.exp;
#endif
  const int mm = lane < 9 ? lane : 0;       // row 0 of the wave: lanes 0..8 own patch pixel m, 9..15 mirror 0
  const int bm = min(lane / 7, 8), bxo = lane - 7 * (lane / 7);  // blend role of this lane: (m, x offset)
  const int n = lane & 15, g = lane >> 4;
  if (!idx_ok) { kq = 0u; jq = 0u; }  // reference behaviour is undefined here; stay in bounds, emit zeros
  if (CDV_EXP(16)) jq = 0u;
  const int64_t jslot = (int64_t)jq;

  // ---- patch tile (MFMA B operand): one 16-byte buffer load; the descriptor covers exactly this tile, so the lanes
  // that are not patch pixels (n >= 9) and the k padding (forced offset) read zeros from the range check
  typedef int cdv_i32x4 __attribute__((ext_vector_type(4)));
  cdv_half8 pat[1];
  if (a.gmap_pm) {
    const unsigned tile_b = 9u * (unsigned)C * 2u;
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(a.gmap) + (size_t)(kq * tile_b), (short)0,
                                                      (int)tile_b, 0x00020000);   // tiles are below 4 GB (host check)
    const unsigned voff = (8 * g < C) ? (unsigned)(n * C + 8 * g) * 2u : 0x40000000u;
    pat[0] = __builtin_bit_cast(cdv_half8, (cdv_i32x4)__builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, 0, 0));
  } else {
    // the reference's planar tiles [Ng][C][3][3] (cdv_corr_fused on tensors nobody converted): eight strided gathers
    const _Float16* gp = reinterpret_cast<const _Float16*>(a.gmap) + (size_t)kq * C * 9 + (n < 9 ? n : 0);
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int ch = 8 * g + j;
      const _Float16 v = gp[min(ch, C - 1) * 9];
      pat[0][j] = (n < 9 && ch < C) ? v : (_Float16)0.f;
    }
  }

  const LevelParams L0 = level_params(a.L0), L1 = level_params(NLEV == 2 ? a.L1 : a.L0);
  float xv, yv, xb, yb;
  int ixmin, ixmax, iymin, iymax;
  if (REC) {
    // the stream brought this lane's blend coordinates and the box extremes: nothing to exchange between lanes.  (The
    // per-pixel path below -- rare -- is the only user of (xv, yv) and picks pixel m's pair from lane 7 m.)
    xb = ec.xb; yb = ec.yb;
    xv = xb; yv = yb;      // patch pixel m sits in lane 7 m (slow_level<1, 7>)
    ixmin = ec.ixmin; ixmax = ec.ixmax; iymin = ec.iymin; iymax = ec.iymax;
  } else {
    const int cval = ec.cval;
    xv = __int_as_float(__builtin_amdgcn_ds_bpermute(4 * mm, cval));
    yv = __int_as_float(__builtin_amdgcn_ds_bpermute(4 * (9 + mm), cval));
    xb = __int_as_float(__builtin_amdgcn_ds_bpermute(4 * bm, cval));
    yb = __int_as_float(__builtin_amdgcn_ds_bpermute(4 * (9 + bm), cval));
    const int ixl = floor_clamped(xv * L0.inv_scale), iyl = floor_clamped(yv * L0.inv_scale);
    int vxmin, vxmax, vymin, vymax;
    row16_minmax4(ixl, iyl, vxmin, vxmax, vymin, vymax);
    ixmin = __builtin_amdgcn_readfirstlane(vxmin); ixmax = __builtin_amdgcn_readfirstlane(vxmax);
    iymin = __builtin_amdgcn_readfirstlane(vymin); iymax = __builtin_amdgcn_readfirstlane(vymax);
  }
  const Box b0 = make_box(ixmin, ixmax, iymin, iymax, L0);
  const int sh1 = NLEV == 2 ? L1.shift : 0;
  const Box b1 = make_box(ixmin >> sh1, ixmax >> sh1, iymin >> sh1, iymax >> sh1, L1);
  CDV_STAMP(corr, p, 1);

  _Float16 res0[7], res1[7];
  // lanes n >= 9 of the D tile are not patch pixels: they store into the dump area instead of being masked off
  // (no EXEC save/restore around every store)
  _Float16* raw_lane = (n < 9) ? raw + n * RAW_MSH + 4 * g
                               : reinterpret_cast<_Float16*>(reinterpret_cast<char*>(raw) + DUMP_OFF_B +
                                                             4 * ((n * (RAW_MSH / 2) + 2 * g) & 31));

  // ---- round trip 2: EVERY window group of a level is requested before its first MFMA; the level-1 request goes out
  // as soon as level 0 has left the registers and flies under the level-0 blend -------------------------------------
  const unsigned CB = (unsigned)C * 2u;                              // bytes per pixel
  const char* r0 = a.L0.base_m + (size_t)(jq * a.L0.slot_bytes + (unsigned)(b0.y0c + PADY) * a.L0.pitch +
                                          (unsigned)(b0.x0c + PADX) * CB);   // rings are below 4 GB (host check)
  const CorrLevel& A1 = NLEV == 2 ? a.L1 : a.L0;
  const char* r1 = A1.base_m + (size_t)(jq * A1.slot_bytes + (unsigned)(b1.y0c + PADY) * A1.pitch +
                                        (unsigned)(b1.x0c + PADX) * CB);
  const unsigned pitch0 = a.L0.pitch, pitch1 = A1.pitch;
  cdv_half8 w[NQ_MAX][1];
  const bool do0 = b0.fast && !b0.outside && idx_ok;
  const bool do1 = NLEV == 2 && b1.fast && !b1.outside && idx_ok;
  const int nq0 = (b0.Wb * b0.Hb + 15) >> 4, nq1 = (b1.Wb * b1.Hb + 15) >> 4;
  // Window addressing in float arithmetic (exact: every value is an integer below 2^24).  The union window is packed
  // DENSELY into the 16 pixel slots of the MFMA A operand: slot n of group q is window pixel P = 16 q + n, in window
  // row floor((P + 0.5) / Wb) -- (P + 0.5) / Wb stays 0.5 / 16 away from every integer, far beyond the rounding of the
  // reciprocal -- at byte offset row * (pitch - Wb CB) + P CB (+ channel group).  (t, b) = ((P + 0.5) / Wb,
  // P CB + group + 2^23) advance together with one v_pk_add_f32 per group; the 2^23 makes the low mantissa bits of
  // fma(row, wrap, b) the integer offset itself, so the float's BIT PATTERN is the buffer offset (no conversion): the
  // descriptor base is biased by -0x4B000000 (host) and its size by +0x4B000000.  The range check of the descriptor
  // returns zeros past the last window row; the k-padding lanes start from 2^25 (always out of range).
  typedef float cdv_f32x2 __attribute__((ext_vector_type(2)));
  const float nhalf = (float)n + 0.5f;
  const float lanebase = (8 * g < C) ? (float)((unsigned)n * CB + (unsigned)(8 * g) * 2u) + 8388608.0f : 33554432.0f;
  // Groups per level: a FIXED number is requested and multiplied without any test -- 8 at level 0 (windows up to 128
  // pixels: 10 x 10 ... 11 x 11, what a patch at roughly its own scale gives), 6 at level 1 (up to 96: 8 x 8 ... 9 x 10).
  // Groups past the window lie past its last row, where the descriptor's range check returns zeros without touching
  // memory, and their products land in raw-volume rows nobody reads.  With the count known at compile time the waits
  // between the loads and the matrix instructions are progressive (vmcnt(n) instead of vmcnt(0): the first MFMA starts
  // when the first group has arrived), and the 2 x 2 tested chains of the earlier version (a compare and a branch per
  // group, four times per edge) are gone.  Larger windows (strong zoom) finish in a rolled tail loop, one group at a
  // time through one register set.
  constexpr int NQF0 = 8, NQF1 = 6;
#define CDV2_LOAD_SETUP(BX, RB, PITCH)                                                             \
    const unsigned Wb_ = (unsigned)(BX).Wb, pitch_ = (PITCH);                                      \
    const auto rsrc_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(RB), (short)0,          \
                                                         (int)((unsigned)(BX).Hb * pitch_ + FBIAS), 0x00020000); \
    const float invW_ = __builtin_amdgcn_rcpf((float)Wb_);                                         \
    const float wrapf_ = (float)(pitch_ - Wb_ * CB);                                               \
    const cdv_f32x2 dtb_ = {16.0f * invW_, (float)(16u * CB)};
#define CDV2_LD1(q)                                                                                \
    {                                                                                              \
      const float vf_ = __builtin_fmaf(__builtin_floorf(tb_[0]), wrapf_, tb_[1]);                  \
      const cdv_i32x4 v_ = __builtin_amdgcn_raw_buffer_load_b128(rsrc_, __float_as_int(vf_), 0, 0); \
      w[q][0] = __builtin_bit_cast(cdv_half8, v_);                                                 \
      tb_ += dtb_;                                                                                 \
    }
  // NQF - 1 groups unconditionally, the NQF-th if the window has it (level 0: 7 groups for 93 % of the edges of the
  // synthetic stream, 8 for 4 %; level 1: 4 or 5 for 72 %, 6 for 28 %)
#define CDV2_LOAD_FIXED(BX, RB, PITCH, NQF, NQ)                                                    \
  {                                                                                                \
    CDV2_LOAD_SETUP(BX, RB, PITCH)                                                                 \
    cdv_f32x2 tb_ = {nhalf * invW_, lanebase};                                                     \
    CDV2_LD1(0) CDV2_LD1(1) CDV2_LD1(2) CDV2_LD1(3) CDV2_LD1(4)                                    \
    if (NQF > 6) { CDV2_LD1(5) CDV2_LD1(6) }                                                       \
    if ((NQ) >= (NQF)) { CDV2_LD1(NQF - 1) }                                                       \
  }
#define CDV2_MF1(WQ, q)                                                                            \
    {                                                                                              \
      cdv_float4 acc = {0.f, 0.f, 0.f, 0.f};                                                       \
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(WQ, pat[0], acc, 0, 0, 0);                      \
      cdv_half4 h = {(_Float16)acc[0], (_Float16)acc[1], (_Float16)acc[2], (_Float16)acc[3]};      \
      *reinterpret_cast<cdv_half4*>(raw_lane + (q) * 16) = h;                                      \
    }
  // (scheduling barriers between pairs of groups: left alone, the scheduler runs all matrix instructions first and keeps
  // eight accumulators alive)
#define CDV2_MF2(qa, qb) CDV2_MF1(w[qa][0], qa) CDV2_MF1(w[qb][0], qb) __builtin_amdgcn_sched_barrier(0);
#define CDV2_MFMA_FIXED(NQF, NQ)                                                                   \
  {                                                                                                \
    CDV2_MF2(0, 1) CDV2_MF2(2, 3)                                                                  \
    if (NQF > 6) { CDV2_MF2(4, 5) CDV2_MF1(w[6][0], 6) } else { CDV2_MF1(w[4][0], 4) }             \
    if ((NQ) >= (NQF)) { CDV2_MF1(w[NQF - 1][0], NQF - 1) }                                        \
  }
  // groups NQF .. NQ - 1 of a large window, rolled (rare)
#define CDV2_TAIL(BX, RB, PITCH, NQF, NQ)                                                          \
  if ((NQ) > (NQF)) {                                                                              \
    CDV2_LOAD_SETUP(BX, RB, PITCH)                                                                 \
    cdv_f32x2 tb_ = {(nhalf + (float)(16 * (NQF))) * invW_, lanebase + (float)(16u * (unsigned)(NQF) * CB)}; \
    _Pragma("unroll 1") for (int q_ = (NQF); q_ < (NQ); q_++) {                                    \
      const float vf_ = __builtin_fmaf(__builtin_floorf(tb_[0]), wrapf_, tb_[1]);                  \
      const cdv_i32x4 v_ = __builtin_amdgcn_raw_buffer_load_b128(rsrc_, __float_as_int(vf_), 0, 0); \
      tb_ += dtb_;                                                                                 \
      CDV2_MF1(__builtin_bit_cast(cdv_half8, v_), q_)                                              \
    }                                                                                              \
  }
  if (do0 && !CDV_EXP(1)) CDV2_LOAD_FIXED(b0, r0, pitch0, NQF0, nq0)
  if (rnext) {   // (wave-uniform) behind the window request: the waits for the window data leave these in flight
    nx->xb = __int_as_float((int)rnext[bm]);
    nx->yb = __int_as_float((int)rnext[9 + bm]);
    nx->w18 = rnext[18]; nx->w19 = rnext[19]; nx->w20 = rnext[20]; nx->w21 = rnext[21]; nx->w22 = rnext[22];
  }
  const BlendGeo g0 = blend_geo(xb, yb, bm, bxo, L0, b0, !b0.fast);
  const BlendGeo g1 = blend_geo(xb, yb, bm, bxo, L1, b1, !b1.fast);
  CDV_STAMP(corr, p, 2);
  if (do0 && !CDV_EXP(256)) {
    CDV2_MFMA_FIXED(NQF0, nq0)
    CDV2_TAIL(b0, r0, pitch0, NQF0, nq0)
  } else if (!b0.fast && idx_ok) {
    slow_level<1, REC ? 7 : 1>(L0, jslot, xv, yv, pat, raw, lane, C);
  } else {
    zero_raw(raw, lane);
  }
  __builtin_amdgcn_sched_barrier(0);   // the level-1 request reuses the window registers: not before level 0 has left them
  if (do1 && !CDV_EXP(1)) CDV2_LOAD_FIXED(b1, r1, pitch1, NQF1, nq1)
  wave_lds_sync();
  CDV_STAMP(corr, p, 3);
  if (!CDV_EXP(128)) blend_level(raw, g0, res0);
  CDV_STAMP(corr, p, 4);
  if (NLEV == 2) {
    wave_lds_sync();
    if (do1 && !CDV_EXP(256)) {
      CDV2_MFMA_FIXED(NQF1, nq1)
      CDV2_TAIL(b1, r1, pitch1, NQF1, nq1)
    } else if (!b1.fast && idx_ok) {
      slow_level<1, REC ? 7 : 1>(L1, jslot, xv, yv, pat, raw, lane, C);
    } else {
      zero_raw(raw, lane);
    }
    wave_lds_sync();
    CDV_STAMP(corr, p, 5);
    if (!CDV_EXP(128)) blend_level(raw, g1, res1);
    CDV_STAMP(corr, p, 6);
  }
#undef CDV2_LOAD_FIXED
#undef CDV2_LOAD_SETUP
#undef CDV2_LD1
#undef CDV2_MF1
#undef CDV2_MF2
#undef CDV2_MFMA_FIXED
#undef CDV2_TAIL

  // ---- stage the edge's output row [x][y][m][lev] in LDS, then 16-byte-per-lane stores ------------------------------
  wave_lds_sync();   // the row overwrites the raw volume: keep the stores behind the last blend's reads
  if (lane < 63 && !CDV_EXP(512)) {
    if (NLEV == 2 && SPLIT) {
      _Float16* o0 = outT + __mul24(bxo, 63) + bm;   // half (x, y, m) = 63 x + 9 y + m; level 1 starts 442 halfs on
#pragma unroll
      for (int yo = 0; yo < 7; yo++) { o0[yo * 9] = res0[yo]; o0[442 + yo * 9] = res1[yo]; }
    } else if (NLEV == 2) {
      uint32_t* o32 = reinterpret_cast<uint32_t*>(outT) + __mul24(bxo, 63) + bm;   // dword (x, y, m) = 63 x + 9 y + m (24-bit multiply: full rate)
#pragma unroll
      for (int yo = 0; yo < 7; yo++) {
        const cdv_half2 h = {res0[yo], res1[yo]};
        o32[yo * 9] = __builtin_bit_cast(uint32_t, h);
      }
    } else {
#pragma unroll
      for (int yo = 0; yo < 7; yo++) outT[(bxo * 7 + yo) * 9 + bm] = res0[yo];
    }
  }
  wave_lds_sync();
  CDV_STAMP(corr, p, 7);
  if (!CDV_EXP(2)) {
    if (NLEV == 2) {
      // 441 dwords: two 16-byte-per-lane stores (256 + 184 dwords) and one last dword.  The row starts on a 4-byte
      // boundary only (1764 B per edge); global memory takes the unaligned 16-byte accesses.
      const uint32_t* src = reinterpret_cast<const uint32_t*>(outT);
      uint32_t* dst = reinterpret_cast<uint32_t*>(a.out) + (size_t)e * (SPLIT ? 442 : 441);
      typedef uint32_t cdv_u32x4 __attribute__((ext_vector_type(4)));
      typedef uint32_t cdv_u32x4u __attribute__((ext_vector_type(4), aligned(4)));
      const cdv_u32x4 v0 = *reinterpret_cast<const cdv_u32x4*>(src + 4 * lane);
#if CDV_CORR_OUT_POLICY
      // the output row leaves written through (sc1: the line is dropped from this XCD's L2 instead of staying there -- 84 MB of
      // output per launch that nobody of this launch reads again pushed the feature maps out of the 4 MB L2s; round 5, measured
      // back to back on the default / stress workloads: 38.4-38.9 -> 37.6 us, 70.5-71.2 -> 69.2-69.4 us; nt (2): the same)
      typedef int cdv_i32x4s __attribute__((ext_vector_type(4)));
      const auto rso = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(dst), (short)0, 1768, 0x00020000);
      __builtin_amdgcn_raw_buffer_store_b128((cdv_i32x4s)v0, rso, 16 * lane, 0, CDV_CORR_OUT_POLICY);
      if (lane < 46) {
        const cdv_u32x4 v1 = *reinterpret_cast<const cdv_u32x4*>(src + 256 + 4 * lane);
        __builtin_amdgcn_raw_buffer_store_b128((cdv_i32x4s)v1, rso, 1024 + 16 * lane, 0, CDV_CORR_OUT_POLICY);
      }
      if (lane == 63) __builtin_amdgcn_raw_buffer_store_b32((int)src[440], rso, 1760, 0, CDV_CORR_OUT_POLICY);
      if (lane == 62 && SPLIT) __builtin_amdgcn_raw_buffer_store_b32((int)src[441], rso, 1764, 0, CDV_CORR_OUT_POLICY);
#else
      *reinterpret_cast<cdv_u32x4u*>(dst + 4 * lane) = v0;
      if (lane < 46) {
        const cdv_u32x4 v1 = *reinterpret_cast<const cdv_u32x4*>(src + 256 + 4 * lane);
        *reinterpret_cast<cdv_u32x4u*>(dst + 256 + 4 * lane) = v1;
      }
      if (lane == 63) dst[440] = src[440];
      if (lane == 62 && SPLIT) dst[441] = src[441];
#endif
    } else {
      _Float16* dst = a.out + (size_t)e * a.out_pitch + a.out_off;
#pragma unroll
      for (int i = 0; i < 7; i++) {
        const int t = i * 64 + lane;
        if (t < 441) dst[t * a.out_stride] = outT[t];
      }
    }
  }
  CDV_STAMP(corr, p, 8);
}

// index % modulus (slam.py:319-320) with the host's reciprocal: q = mulhi(x, ceil(2^32 / d)) is x / d or one more
// for 0 <= x < 2^31, so one correction step.  Returns false for an index outside its ring (or not a 31-bit value).
__device__ __forceinline__ bool ring_index(int64_t v64, uint32_t mod, uint32_t magic, uint32_t limit, uint32_t& q) {
  q = (uint32_t)v64;
  const bool small = (uint32_t)(v64 >> 32) == 0u && q < 0x80000000u;
  if (mod > 1u) {
    const int32_t r = (int32_t)(q - __umulhi(q, magic) * mod);
    q = (uint32_t)(r < 0 ? r + (int32_t)mod : r);
  } else if (mod == 1u) {
    q = 0u;
  }
  return small && q < limit;
}

// The four leading arguments are what a wave needs to find its edge and request its record: one scalar load, and no second
// one for the grid size (gridDim.x lives in the hidden arguments; `eighth` = gridDim.x / 8 comes from the host).  (Round 5:
// with -amdgpu-kernarg-preload-count=4 these arguments arrive in scalar registers at wave launch and the record request
// goes out with no kernel-argument load in front of it at all -- measured next to this build and to the one before it
// (two dependent scalar round trips): 36.3-37.6 us back to back for all three, no difference.  The kernel is not bound by
// the length of one wave's chain of round trips; the preload flag is not used.)
constexpr int CW = CDV_CORR_WAVES;
constexpr int EPW = CDV_CORR_EPW;
template <int CC, int NLEV, bool SPLIT = false, bool REC = false>
__global__ __launch_bounds__(64 * CW) void corr_fused2_kernel(const uint32_t* __restrict__ rec_h, const int32_t* __restrict__ dynE_h,
                                                          int E_h, int eighth_h, const CorrArgs2 a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  _Float16* raw = reinterpret_cast<_Float16*>(smem_raw + (size_t)wave * WAVE_LDS2_BYTES);
  _Float16* outT = raw;   // the staged output row takes the place of the raw volume once both blends are done
  // workgroup b runs on XCD b % 8 and takes CW consecutive edges of the b % 8-th contiguous eighth of the list.  (Round 5,
  // CDV_CORR_EPW = 2 / 3 edges per wave with the next record requested behind the current edge's window loads: 77 registers
  // instead of 62 (70 without machine LICM), six (seven) waves per SIMD instead of eight: 40.5 (39.3) against 37.7 us back to
  // back -- the hidden round trip buys what the lost waves cost, no more; forced to 64 registers it spills 43 values: 79 us.
  // Workgroups of 8 / 16 waves instead of 4: 1 % / 5 % slower.  Kept as build switches)
  constexpr int KE = REC ? EPW : 1;
  int E = E_h, eighth = eighth_h;
  if (REC && dynE_h) {    // sizes on the device: the contiguous eighths are those of the ACTUAL list, not of the launch
    E = min(__builtin_amdgcn_readfirstlane(*dynE_h), E_h);
    eighth = (E + 8 * CW * KE - 1) / (8 * CW * KE);
    if (((int)blockIdx.x >> 3) >= eighth) return;
  }
  const int p0 = (((int)blockIdx.x & 7) * eighth + ((int)blockIdx.x >> 3)) * (CW * KE) + wave;
  if (p0 >= E) return;  // no block-wide barriers below: waves are independent
  if (REC && KE > 1) {
    // several edges per wave: positions p0, p0 + CW, ...; the next record is requested under the current edge
    const int bm = min(lane / 7, 8);
    const uint32_t* r = rec_h + (size_t)p0 * cdv::CORR_REC_WORDS;
    EdgeCoords ec;
    ec.cval = 0;
    ec.xb = __int_as_float((int)r[bm]);
    ec.yb = __int_as_float((int)r[9 + bm]);
    uint32_t w18 = r[18], w19 = r[19], w20 = r[20], w21 = r[21], w22 = r[22];
    int p = p0;
#pragma unroll 1
    for (int k = 0; k < KE; k++) {
      const int pn = p + CW;
      const bool has_next = k + 1 < KE && pn < E;      // wave-uniform
      ec.ixmin = (int)(short)(w21 & 0xffff); ec.ixmax = (int)w21 >> 16;
      ec.iymin = (int)(short)(w22 & 0xffff); ec.iymax = (int)w22 >> 16;
      NextRec nx;
      int lane_it = lane;      // opaque per trip: what the edge derives from the lane index is recomputed, not kept across the loop
      asm volatile("" : "+v"(lane_it));
      corr_edge<CC, NLEV, SPLIT, true>(a, p, (int)w18, ec, w19, w20, w20 != 0xFFFFFFFFu, lane_it, raw, outT,
                                       has_next ? rec_h + (size_t)pn * cdv::CORR_REC_WORDS : nullptr, &nx);
      if (!has_next) break;
      ec.xb = nx.xb; ec.yb = nx.yb;
      w18 = __builtin_amdgcn_readfirstlane(nx.w18); w19 = __builtin_amdgcn_readfirstlane(nx.w19);
      w20 = __builtin_amdgcn_readfirstlane(nx.w20); w21 = __builtin_amdgcn_readfirstlane(nx.w21);
      w22 = __builtin_amdgcn_readfirstlane(nx.w22);
      p = pn;
    }
    return;
  }
  if (REC) {
    // ---- ONE round trip: record p0 of the packed input stream the index build wrote in processing order -- the 18
    // coordinates as one vector load, edge id and ring indices as scalar loads of the same line (no order[] -> coords /
    // kk / jj indirection: one dependent memory round trip less per wave, no index arithmetic)
    const uint32_t* r = rec_h + (size_t)p0 * cdv::CORR_REC_WORDS;
    const int bm = min(lane / 7, 8);
    EdgeCoords ec;
    ec.cval = 0;
    ec.xb = __int_as_float((int)r[bm]);          // two gathers from the record's one or two cache lines: each lane gets
    ec.yb = __int_as_float((int)r[9 + bm]);      // the coordinates of ITS patch pixel, no cross-lane exchange
    const int e = (int)r[18];
    const uint32_t kq = r[19], jq = r[20];
    const int bx = (int)r[21], by = (int)r[22];   // extremes of floor(x), floor(y): (max << 16) | (min & 0xffff)
    ec.ixmin = (int)(short)(bx & 0xffff); ec.ixmax = bx >> 16;
    ec.iymin = (int)(short)(by & 0xffff); ec.iymax = by >> 16;
    corr_edge<CC, NLEV, SPLIT, true>(a, p0, e, ec, kq, jq, jq != 0xFFFFFFFFu, lane, raw, outT);
    return;
  }
  // ---- round trip 1: the 18 coordinates (one vector load) and the two indices (scalar loads) -----------------------
  const int e = a.order ? __builtin_amdgcn_readfirstlane(a.order[p0]) : p0;   // wave-uniform
  const int cval = __float_as_int(a.coords[(size_t)e * 18 + min(lane, 17)]);
  if (NLEV == 1 && a.coords_ref) {   // this edge's level is in place if it was computed from these very coordinates
    const float r = a.coords_ref[(size_t)e * 18 + min(lane, 17)] * a.ref_mul;
    if (__all(lane >= 18 || __float_as_int(r) == cval)) return;
  }
  uint32_t kq, jq;
  const bool k_ok = ring_index(a.kk[e], a.kmod, a.kmagic, a.Ng, kq);
  const bool j_ok = ring_index(a.jj[e], a.jmod, a.jmagic, a.slots, jq);
  EdgeCoords ec;
  ec.cval = cval;
  corr_edge<CC, NLEV, SPLIT, false>(a, p0, e, ec, kq, jq, k_ok && j_ok, lane, raw, outT);
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
// planar [N][C][H][W] -> padded channels-last [N][H+2PADY][W+2PADX][C] (interior only; the margins are
// zeroed once at allocation); one thread per (pixel, 8-channel group): 8 strided 2-byte reads (coalesced
// across the wave along W), one 16-byte write.
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
    *reinterpret_cast<cdv_half8*>(dst + ((nslot * (H + 2 * PADY) + yh + PADY) * (W + 2 * PADX) + xw + PADX) * C +
                                  8 * gq) = v;
  }
}

// ---- a planar ring whose writer is somebody else (the reference's slam.py writes fmap1_[:, n % mem] with torch ops) kept
// in step with its channels-last shadow WITHOUT converting all of it every frame: pass 1 fingerprints every slot of the
// planar ring (a read of the ring: 21 MB at level 0), pass 2 converts only the slots whose fingerprint differs from
// the one taken at the previous sync (normally ONE).  Fingerprint = FP_PARTS position-keyed 64-bit sums per slot.
constexpr int FP_PARTS = 16;      // workgroups (and partial sums) per slot

__device__ __forceinline__ void fingerprint_body(const uint32_t* __restrict__ src, int64_t words_per_slot,
                                                 uint64_t* __restrict__ fp, int bid) {
  const int slot = bid / FP_PARTS, part = bid % FP_PARTS;
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const u32x4* p = reinterpret_cast<const u32x4*>(src + (size_t)slot * words_per_slot);
  const int64_t n4 = words_per_slot / 4;        // slots are multiples of 16 bytes (C % 8 == 0)
  uint64_t acc = 0;
  for (int64_t i = (int64_t)part * 256 + threadIdx.x; i < n4; i += (int64_t)FP_PARTS * 256) {
    const u32x4 v = p[i];
    const uint64_t key = 0x9E3779B97F4A7C15ull + 2ull * (uint64_t)i;           // odd, different for every position
    acc += ((uint64_t)v[0] | ((uint64_t)v[1] << 32)) * key;
    acc += ((uint64_t)v[2] | ((uint64_t)v[3] << 32)) * (key ^ 0xD6E8FEB86659FD92ull);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  __shared__ uint64_t sw[4];
  if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) fp[(size_t)slot * FP_PARTS + part] = (sw[0] + sw[1]) + (sw[2] + sw[3]) + 1ull;   // never 0 = "no fingerprint yet"
}

__global__ __launch_bounds__(256) void fmap_fingerprint_kernel(const uint32_t* __restrict__ src, int64_t words_per_slot,
                                                               uint64_t* __restrict__ fp) {
  fingerprint_body(src, words_per_slot, fp, (int)blockIdx.x);
}

__device__ __forceinline__ void dirty_body(const _Float16* __restrict__ src, _Float16* __restrict__ dst, int C, int H, int W,
                                           const uint64_t* __restrict__ fp_new, const uint64_t* __restrict__ fp_old,
                                           int wg_per_slot, int32_t* __restrict__ n_dirty, int bid) {
  const int64_t nslot = bid / wg_per_slot;
  const int wg = bid % wg_per_slot;
  bool same = true;
#pragma unroll
  for (int i = 0; i < FP_PARTS; i++) same = same && fp_new[nslot * FP_PARTS + i] == fp_old[nslot * FP_PARTS + i];
  if (same) return;                                  // workgroup-uniform
  if (wg == 0 && threadIdx.x == 0 && n_dirty) atomicAdd(n_dirty, 1);
  const int G = C / 8;
  const int64_t total = (int64_t)H * W * G;
  for (int64_t idx = (int64_t)wg * 256 + threadIdx.x; idx < total; idx += (int64_t)wg_per_slot * 256) {
    int64_t t = idx;
    const int xw = (int)(t % W); t /= W;
    const int gq = (int)(t % G); t /= G;
    const int yh = (int)t;
    cdv_half8 v;
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = src[((nslot * C + 8 * gq + j) * H + yh) * W + xw];
    *reinterpret_cast<cdv_half8*>(dst + ((nslot * (H + 2 * PADY) + yh + PADY) * (W + 2 * PADX) + xw + PADX) * C +
                                  8 * gq) = v;
  }
}

__global__ __launch_bounds__(256) void nchw_to_nhwc_dirty_kernel(const _Float16* __restrict__ src,
                                                                 _Float16* __restrict__ dst, int C, int H, int W,
                                                                 const uint64_t* __restrict__ fp_new,
                                                                 const uint64_t* __restrict__ fp_old, int wg_per_slot,
                                                                 int32_t* __restrict__ n_dirty) {
  dirty_body(src, dst, C, H, W, fp_new, fp_old, wg_per_slot, n_dirty, (int)blockIdx.x);
}

// ---- the same two passes for SEVERAL rings at once, the tiles' conversion riding the second (cdv_shadows_sync): what an
// unchanged slam.py needs in front of its correlation -- both pyramid levels' shadows and the tile shadow in step -- is five
// launches through the single-ring entry points and two here.  Same bodies, same bytes.
struct ShadowJob {
  const _Float16* src; _Float16* dst;
  uint64_t *fp_new; const uint64_t* fp_old;
  int32_t* n_dirty;
  int64_t words_per_slot;
  int C, H, W, wg_per_slot, fp_blocks, cv_blocks;
};
struct ShadowJobs {
  ShadowJob j[2];
  int n;
  const _Float16* g_src; _Float16* g_dst;     // tiles (NULL: none)
  int64_t g_count;
  int g_C, g_blocks;
};

__global__ __launch_bounds__(256) void shadows_fingerprint_kernel(const ShadowJobs J) {
  int b = (int)blockIdx.x;
  for (int q = 0; q < J.n; q++) {               // workgroup-uniform
    if (b < J.j[q].fp_blocks) {
      fingerprint_body(reinterpret_cast<const uint32_t*>(J.j[q].src), J.j[q].words_per_slot, J.j[q].fp_new, b);
      return;
    }
    b -= J.j[q].fp_blocks;
  }
}

__global__ __launch_bounds__(256) void shadows_convert_kernel(const ShadowJobs J) {
  int b = (int)blockIdx.x;
  for (int q = 0; q < J.n; q++) {
    const ShadowJob& s = J.j[q];
    if (b < s.cv_blocks) {
      dirty_body(s.src, s.dst, s.C, s.H, s.W, s.fp_new, s.fp_old, s.wg_per_slot, s.n_dirty, b);
      return;
    }
    b -= s.cv_blocks;
  }
  if (J.g_src) cdv::gmap_pm_convert(J.g_src, J.g_dst, 0, J.g_count, J.g_C, (int64_t)b * 256 + threadIdx.x, (int64_t)J.g_blocks * 256);
}

__global__ __launch_bounds__(256) void gmap_pm_kernel(const _Float16* __restrict__ src, _Float16* __restrict__ dst,
                                                      int64_t first, int64_t count, int C) {
  cdv::gmap_pm_convert(src, dst, first, count, C, (int64_t)blockIdx.x * blockDim.x + threadIdx.x,
                       (int64_t)gridDim.x * blockDim.x);
}

// feature-map ring write + 4x4 pool (+ the frame's patch tiles): body in cdv_parts.h
__global__ __launch_bounds__(256) void fmap_ingest_kernel(cdv::IngestArgs a) {
  cdv::ingest_body(a, (int)blockIdx.x, (int)blockDim.x, (int)threadIdx.x);
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

// altcorr.patchify(net, coords, radius, mode) (correlation.py:51-71) in one pass: mode 1 = 'bilinear' (the (2r+2)^2
// gather of patchify_forward blended to (2r+1)^2 with the sub-pixel offset of the patch centre, in the reference's
// operation order x00 + x01 + x10 + x11), mode 2 = 'upperleft' (the 1x1 corner tile).  Out-of-image taps are zero.
// (the blend multiplies float32 offsets into the tile, so torch's type promotion makes the 'bilinear' result float32
// whatever the map's dtype; 'upperleft' is a slice and keeps the dtype)
template <typename T>
__global__ __launch_bounds__(256) void patchify_blend_kernel(const T* __restrict__ net, const float* __restrict__ coords,
                                                             void* __restrict__ outv, int B, int64_t M, int C, int H,
                                                             int W, int R, int mode) {
  const int d = (mode == 2) ? 1 : 2 * R + 1;
  const int64_t total = (int64_t)B * M * C * d * d;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    int64_t t = idx;
    const int b2 = (int)(t % d); t /= d;
    const int a2 = (int)(t % d); t /= d;
    const int ch = (int)(t % C); t /= C;
    const int64_t m = t % M; t /= M;
    const int bb = (int)t;
    const float x = coords[(bb * M + m) * 2 + 0], y = coords[(bb * M + m) * 2 + 1];
    const float fxf = floorf(x), fyf = floorf(y);
    const int i0 = (int)fminf(fmaxf(fyf, -1.0e6f), 1.0e6f) + (a2 - R);
    const int j0 = (int)fminf(fmaxf(fxf, -1.0e6f), 1.0e6f) + (b2 - R);
    const T* np = net + ((int64_t)bb * C + ch) * H * W;
    auto tap = [&](int i, int j) -> T { return (i >= 0 && i < H && j >= 0 && j < W) ? np[(int64_t)i * W + j] : (T)0.f; };
    if (mode == 2) {
      reinterpret_cast<T*>(outv)[idx] = tap(i0, j0);
    } else {
      const float dx = x - fxf, dy = y - fyf;   // correlation.py:58-66, same operation order
      const float x00 = (1.0f - dy) * (1.0f - dx) * (float)tap(i0, j0);
      const float x01 = (1.0f - dy) * dx * (float)tap(i0, j0 + 1);
      const float x10 = dy * (1.0f - dx) * (float)tap(i0 + 1, j0);
      const float x11 = dy * dx * (float)tap(i0 + 1, j0 + 1);
      reinterpret_cast<float*>(outv)[idx] = ((x00 + x01) + x10) + x11;
    }
  }
}

// Several altcorr.patchify calls on ONE set of patch centres in one launch: a new frame's imap / gmap / colour / patch
// tiles (net_cdv.py:355-374).  Job j reads its own map at (coords + o_j) * s_j -- the scaling the reference applies with
// torch ops before each call (scale_f2i * coords, 4 * (coords + 0.5)), same two float operations -- with its own radius,
// mode and dtype; workgroups [first[j], first[j + 1]) belong to job j.
struct PatchifyJobs {
  cdv_patchify_job j[CDV_MAX_PATCHIFY_JOBS];
  int first[CDV_MAX_PATCHIFY_JOBS + 1];
  int n_jobs;
};

template <typename T>
__device__ __forceinline__ void patchify_job_body(const cdv_patchify_job& J, const float* __restrict__ coords, int64_t M,
                                                  int64_t idx0, int64_t stride) {
  const int R = J.radius, mode = J.mode, C = J.C, H = J.H, W = J.W;
  const int d = (mode == 2) ? 1 : 2 * R + 1;
  const int64_t total = M * C * d * d;
  const T* net = reinterpret_cast<const T*>(J.net);
  for (int64_t idx = idx0; idx < total; idx += stride) {
    int64_t t = idx;
    const int b2 = (int)(t % d); t /= d;
    const int a2 = (int)(t % d); t /= d;
    const int ch = (int)(t % C); t /= C;
    const int64_t m = t;
    const float x = (coords[m * 2 + 0] + J.ox) * J.sx, y = (coords[m * 2 + 1] + J.oy) * J.sy;
    const float fxf = floorf(x), fyf = floorf(y);
    const int i0 = (int)fminf(fmaxf(fyf, -1.0e6f), 1.0e6f) + (a2 - R);
    const int j0 = (int)fminf(fmaxf(fxf, -1.0e6f), 1.0e6f) + (b2 - R);
    const T* np = net + (int64_t)ch * H * W;
    auto tap = [&](int i, int j) -> T { return (i >= 0 && i < H && j >= 0 && j < W) ? np[(int64_t)i * W + j] : (T)0.f; };
    if (mode == 2) {
      reinterpret_cast<T*>(J.out)[idx] = tap(i0, j0);
    } else {
      const float dx = x - fxf, dy = y - fyf;   // correlation.py:58-66, same operation order
      const float x00 = (1.0f - dy) * (1.0f - dx) * (float)tap(i0, j0);
      const float x01 = (1.0f - dy) * dx * (float)tap(i0, j0 + 1);
      const float x10 = dy * (1.0f - dx) * (float)tap(i0 + 1, j0);
      const float x11 = dy * dx * (float)tap(i0 + 1, j0 + 1);
      reinterpret_cast<float*>(J.out)[idx] = ((x00 + x01) + x10) + x11;
    }
  }
}

__global__ __launch_bounds__(256) void patchify_multi_kernel(const PatchifyJobs P, const float* __restrict__ coords,
                                                             int64_t M) {
  int ji = 0;
  while (ji + 1 < P.n_jobs && (int)blockIdx.x >= P.first[ji + 1]) ji++;
  const cdv_patchify_job& J = P.j[ji];
  const int nb = P.first[ji + 1] - P.first[ji];
  const int64_t idx0 = (int64_t)((int)blockIdx.x - P.first[ji]) * 256 + threadIdx.x, stride = (int64_t)nb * 256;
  if (J.dtype == CDV_F16) patchify_job_body<_Float16>(J, coords, M, idx0, stride);
  else patchify_job_body<float>(J, coords, M, idx0, stride);
}

}  // namespace

extern "C" int cdv_patchify_multi(const cdv_patchify_job* jobs, int n_jobs, const float* coords, int64_t M, void* stream) {
  CDV_REQUIRE(n_jobs >= 0 && n_jobs <= CDV_MAX_PATCHIFY_JOBS, CDV_ERR_ARG, "cdv_patchify_multi: too many jobs");
  if (n_jobs == 0 || M == 0) return CDV_OK;
  CDV_REQUIRE(jobs != nullptr && coords != nullptr && M > 0, CDV_ERR_ARG, "cdv_patchify_multi: NULL argument");
  PatchifyJobs P;
  P.n_jobs = n_jobs;
  P.first[0] = 0;
  for (int i = 0; i < n_jobs; i++) {
    const cdv_patchify_job& J = jobs[i];
    CDV_REQUIRE(J.dtype == CDV_F16 || J.dtype == CDV_F32, CDV_ERR_UNSUPPORTED, "cdv_patchify_multi: dtype must be f16 or f32");
    CDV_REQUIRE(J.mode == 1 || J.mode == 2, CDV_ERR_ARG, "cdv_patchify_multi: mode 1 (bilinear) or 2 (upperleft)");
    CDV_REQUIRE(J.net && J.out && J.C > 0 && J.H > 0 && J.W > 0 && J.radius >= 0, CDV_ERR_ARG, "cdv_patchify_multi: bad job");
    const int d = (J.mode == 2) ? 1 : 2 * J.radius + 1;
    const int64_t total = M * J.C * d * d;
    P.j[i] = J;
    P.first[i + 1] = P.first[i] + (int)(cdv_div_up(total, 256) < 4096 ? cdv_div_up(total, 256) : 4096);
  }
  hipLaunchKernelGGL(patchify_multi_kernel, dim3(P.first[n_jobs]), dim3(256), 0, (hipStream_t)stream, P, coords, M);
  CDV_LAUNCH_CHECK();
  return CDV_OK;
}

extern "C" int cdv_patchify_blend(const void* net, const float* coords, void* out, int B, int64_t M, int C, int H, int W,
                                  int radius, int mode, int dtype, void* stream) {
  CDV_REQUIRE(dtype == CDV_F16 || dtype == CDV_F32, CDV_ERR_UNSUPPORTED, "cdv_patchify_blend: dtype must be f16 or f32");
  CDV_REQUIRE(mode == 1 || mode == 2, CDV_ERR_ARG, "cdv_patchify_blend: mode 1 (bilinear) or 2 (upperleft)");
  const int d = (mode == 2) ? 1 : 2 * radius + 1;
  const int64_t total = (int64_t)B * M * C * d * d;
  if (total == 0) return CDV_OK;
  const int blocks = cdv_div_up(total, 256) < 16384 ? cdv_div_up(total, 256) : 16384;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == CDV_F16)
    hipLaunchKernelGGL(patchify_blend_kernel<_Float16>, dim3(blocks), dim3(256), 0, s, (const _Float16*)net, coords, out,
                       B, M, C, H, W, radius, mode);
  else
    hipLaunchKernelGGL(patchify_blend_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)net, coords, out, B, M,
                       C, H, W, radius, mode);
  CDV_LAUNCH_CHECK();
  return CDV_OK;
}

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

extern "C" size_t cdv_fmap_padded_elems(int64_t slots, int C, int H, int W) {
  return (size_t)slots * (size_t)(H + 2 * PADY) * (size_t)(W + 2 * PADX) * (size_t)C;
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

extern "C" size_t cdv_fmap_sync_workspace_bytes(int64_t N) {
  return (size_t)(2 * (N > 0 ? N : 1) * FP_PARTS) * sizeof(uint64_t) + 64;
}

extern "C" int cdv_fmap_sync_nhwc(const void* src_nchw, void* dst_nhwc, int64_t N, int C, int H, int W, void* ws,
                                  int parity, void* stream) {
  CDV_REQUIRE(C % 8 == 0 && C > 0, CDV_ERR_ARG, "cdv_fmap_sync_nhwc: C must be a multiple of 8");
  CDV_REQUIRE(src_nchw && dst_nhwc && ws && N >= 0 && H > 0 && W > 0, CDV_ERR_ARG, "cdv_fmap_sync_nhwc: bad argument");
  if (N == 0) return CDV_OK;
  uint64_t* fp = (uint64_t*)ws;
  uint64_t* fp_new = fp + (size_t)(parity & 1) * N * FP_PARTS;
  const uint64_t* fp_old = fp + (size_t)((parity & 1) ^ 1) * N * FP_PARTS;
  int32_t* n_dirty = (int32_t*)(fp + 2 * (size_t)N * FP_PARTS);
  const int64_t words = (int64_t)C * H * W / 2;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(fmap_fingerprint_kernel, dim3((unsigned)(N * FP_PARTS)), dim3(256), 0, s, (const uint32_t*)src_nchw,
                     words, fp_new);
  const int wg_per_slot = (int)(cdv_div_up((int64_t)H * W * (C / 8), 256) < 64 ? cdv_div_up((int64_t)H * W * (C / 8), 256) : 64);
  hipLaunchKernelGGL(nchw_to_nhwc_dirty_kernel, dim3((unsigned)(N * wg_per_slot)), dim3(256), 0, s, (const _Float16*)src_nchw,
                     (_Float16*)dst_nhwc, C, H, W, fp_new, fp_old, wg_per_slot, n_dirty);
  CDV_LAUNCH_CHECK();
  return CDV_OK;
}

extern "C" int cdv_shadows_sync(const cdv_shadow_ring* rings, int n_rings, const void* gmap_planar, void* gmap_pm, int64_t Ng,
                                int C_tiles, void* stream) {
  CDV_REQUIRE(n_rings >= 0 && n_rings <= 2 && (n_rings == 0 || rings != nullptr), CDV_ERR_ARG, "cdv_shadows_sync: 0 to 2 rings");
  const bool do_g = gmap_planar != nullptr && gmap_pm != nullptr && Ng > 0;
  CDV_REQUIRE(!do_g || (C_tiles % 8 == 0 && C_tiles > 0), CDV_ERR_ARG, "cdv_shadows_sync: C of the tiles must be a multiple of 8");
  ShadowJobs J;
  J.n = 0;
  int fp_total = 0, cv_total = 0;
  for (int q = 0; q < n_rings; q++) {
    const cdv_shadow_ring& r = rings[q];
    CDV_REQUIRE(r.C % 8 == 0 && r.C > 0 && r.src_nchw && r.dst_nhwc && r.ws && r.N >= 0 && r.H > 0 && r.W > 0, CDV_ERR_ARG,
                "cdv_shadows_sync: bad ring");
    if (r.N == 0) continue;
    ShadowJob& s = J.j[J.n++];
    uint64_t* fp = (uint64_t*)r.ws;
    s.src = (const _Float16*)r.src_nchw; s.dst = (_Float16*)r.dst_nhwc;
    s.fp_new = fp + (size_t)(r.parity & 1) * r.N * FP_PARTS;
    s.fp_old = fp + (size_t)((r.parity & 1) ^ 1) * r.N * FP_PARTS;
    s.n_dirty = (int32_t*)(fp + 2 * (size_t)r.N * FP_PARTS);
    s.words_per_slot = (int64_t)r.C * r.H * r.W / 2;
    s.C = r.C; s.H = r.H; s.W = r.W;
    const int64_t per = cdv_div_up((int64_t)r.H * r.W * (r.C / 8), 256);
    s.wg_per_slot = (int)(per < 64 ? per : 64);
    s.fp_blocks = (int)(r.N * FP_PARTS);
    s.cv_blocks = (int)(r.N * s.wg_per_slot);
    fp_total += s.fp_blocks; cv_total += s.cv_blocks;
  }
  J.g_src = do_g ? (const _Float16*)gmap_planar : nullptr;
  J.g_dst = (_Float16*)gmap_pm;
  J.g_count = Ng; J.g_C = C_tiles;
  const int64_t gtotal = do_g ? Ng * 9 * (C_tiles / 8) : 0;
  J.g_blocks = (int)(cdv_div_up(gtotal, 256) < 16384 ? cdv_div_up(gtotal, 256) : 16384);
  if (!do_g) J.g_blocks = 0;
  hipStream_t s = (hipStream_t)stream;
  if (fp_total > 0) hipLaunchKernelGGL(shadows_fingerprint_kernel, dim3((unsigned)fp_total), dim3(256), 0, s, J);
  if (cv_total + J.g_blocks > 0)
    hipLaunchKernelGGL(shadows_convert_kernel, dim3((unsigned)(cv_total + J.g_blocks)), dim3(256), 0, s, J);
  CDV_LAUNCH_CHECK();
  return CDV_OK;
}

extern "C" int cdv_frame_ingest(const void* fmap_chw, void* fmap1_nhwc, void* fmap2_nhwc, void* fmap1_nchw,
                                void* fmap2_nchw, int slot, int C, int H, int W, const void* gmap_planar, void* gmap_pm,
                                int64_t Ng, int64_t gmap_first, int64_t gmap_count, void* stream) {
  CDV_REQUIRE(C % 8 == 0 && C > 0, CDV_ERR_ARG, "cdv_fmap_ingest: C must be a multiple of 8");
  CDV_REQUIRE(H % 4 == 0 && W % 4 == 0, CDV_ERR_ARG, "cdv_fmap_ingest: H and W must be multiples of 4");
  CDV_REQUIRE(slot >= 0, CDV_ERR_ARG, "cdv_fmap_ingest: slot");
  const bool do_g = gmap_planar != nullptr && gmap_pm != nullptr && gmap_count > 0;
  CDV_REQUIRE(!do_g || (gmap_first >= 0 && gmap_first + gmap_count <= Ng), CDV_ERR_ARG, "cdv_frame_ingest: tile range");
  const int64_t total = (int64_t)(H / 4) * (W / 4) * (C / 8) * 16;   // one thread per pixel and 8-channel group
  const int blocks = cdv_div_up(total, 256);
  const int gblocks = do_g ? (int)cdv_div_up(gmap_count * 9 * (C / 8), 256) : 0;
  const cdv::IngestArgs a{(const _Float16*)fmap_chw, (_Float16*)fmap1_nhwc, (_Float16*)fmap2_nhwc, (_Float16*)fmap1_nchw,
                          (_Float16*)fmap2_nchw, slot, C, H, W, (const _Float16*)gmap_planar, (_Float16*)gmap_pm,
                          gmap_first, gmap_count, blocks, gblocks};
  hipLaunchKernelGGL(fmap_ingest_kernel, dim3(blocks + gblocks), dim3(256), 0, (hipStream_t)stream, a);
  CDV_LAUNCH_CHECK();
  return CDV_OK;
}

extern "C" int cdv_fmap_ingest(const void* fmap_chw, void* fmap1_nhwc, void* fmap2_nhwc, void* fmap1_nchw,
                               void* fmap2_nchw, int slot, int C, int H, int W, void* stream) {
  return cdv_frame_ingest(fmap_chw, fmap1_nhwc, fmap2_nhwc, fmap1_nchw, fmap2_nchw, slot, C, H, W, nullptr, nullptr, 0,
                          0, 0, stream);
}

extern "C" int cdv_gmap_to_pixel_major(const void* gmap_planar, void* gmap_pm, int64_t Ng, int C, int64_t first,
                                       int64_t count, void* stream) {
  CDV_REQUIRE(C % 8 == 0 && C > 0, CDV_ERR_ARG, "cdv_gmap_to_pixel_major: C must be a multiple of 8");
  CDV_REQUIRE(first >= 0 && count >= 0 && first + count <= Ng, CDV_ERR_ARG, "cdv_gmap_to_pixel_major: tile range");
  if (count == 0) return CDV_OK;
  const int64_t total = count * 9 * (C / 8);
  const int blocks = cdv_div_up(total, 256) < 16384 ? cdv_div_up(total, 256) : 16384;
  hipLaunchKernelGGL(gmap_pm_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const _Float16*)gmap_planar,
                     (_Float16*)gmap_pm, first, count, C);
  CDV_LAUNCH_CHECK();
  return CDV_OK;
}

static int corr_fused_impl(const void* gmap, const void* fmap0_nhwc, const void* fmap1_nhwc, const float* coords,
                           const int64_t* kk, const int64_t* jj, const int32_t* order, void* out, int64_t E,
                           int64_t Ng, int64_t slots, int C, int H0, int W0, int H1, int W1, float scale0,
                           float scale1, int nlev, int64_t kmod, int64_t jmod, int gmap_pixel_major, void* stream,
                           int out_stride, int out_off, int out_pitch, const float* coords_ref, float ref_mul, int split,
                           const uint32_t* rec = nullptr, const int32_t* dynE = nullptr) {
  CDV_REQUIRE(nlev == 1 || nlev == 2, CDV_ERR_ARG, "cdv_corr_fused: nlev must be 1 or 2");
  CDV_REQUIRE(C % 8 == 0 && C > 0 && C <= 128, CDV_ERR_UNSUPPORTED, "cdv_corr_fused: C must be a multiple of 8, <= 128");
  CDV_REQUIRE(E >= 0 && E < ((int64_t)1 << 31), CDV_ERR_ARG, "cdv_corr_fused: E out of range");
  int ex0 = 0, ex1 = 0;
  CDV_REQUIRE(scale0 > 0.f && frexpf(scale0, &ex0) == 0.5f && (nlev == 1 || (scale1 > 0.f && frexpf(scale1, &ex1) == 0.5f)),
              CDV_ERR_UNSUPPORTED, "cdv_corr_fused: pyramid scales must be powers of two (1 and 4 in SLAM.corr)");
  CDV_REQUIRE(fmap0_nhwc != nullptr && (nlev == 1 || fmap1_nhwc != nullptr), CDV_ERR_ARG, "cdv_corr_fused: NULL map");
  CDV_REQUIRE(cdv_fmap_padded_elems(slots, C, H0, W0) * 2 < ((size_t)1 << 32) &&
                  (nlev == 1 || cdv_fmap_padded_elems(slots, C, H1, W1) * 2 < ((size_t)1 << 32)),
              CDV_ERR_UNSUPPORTED, "cdv_corr_fused: a feature ring of 4 GB or more");
  CDV_REQUIRE(kmod >= 0 && jmod >= 0 && kmod < ((int64_t)1 << 31) && jmod < ((int64_t)1 << 31), CDV_ERR_ARG,
              "cdv_corr_fused: kmod / jmod out of range");
  // ceil(2^32 / d): mulhi(x, magic) is x / d or x / d + 1 for 0 <= x < 2^31
  const uint32_t kmagic = kmod > 1 ? (uint32_t)((((uint64_t)1 << 32) + (uint64_t)kmod - 1) / (uint64_t)kmod) : 0u;
  const uint32_t jmagic = jmod > 1 ? (uint32_t)((((uint64_t)1 << 32) + (uint64_t)jmod - 1) / (uint64_t)jmod) : 0u;
  if (E == 0) return CDV_OK;
  CDV_REQUIRE(nlev == 1 || ex1 >= ex0, CDV_ERR_UNSUPPORTED, "cdv_corr_fused: level 1 must not be finer than level 0");
  const size_t smem = 4 * (size_t)(C <= 32 ? WAVE_LDS2_BYTES : WAVE_LDS_BYTES);
  static const int exp = getenv("CDV_CORR_EXP") ? atoi(getenv("CDV_CORR_EXP")) : 0;  // diagnostics only
  hipStream_t s = (hipStream_t)stream;
  if (C <= 32) {
    CDV_REQUIRE(Ng < ((int64_t)1 << 31) && slots < ((int64_t)1 << 31) && (size_t)Ng * 9 * (size_t)C * 2 < ((size_t)1 << 32),
                CDV_ERR_UNSUPPORTED, "cdv_corr_fused: 4 GB or more of patch tiles");
    auto level = [&](const void* ring, int H, int W, float scale, int shift) {
      const uint32_t pitch = (uint32_t)(W + 2 * PADX) * (uint32_t)C * 2u;
      return CorrLevel{(const char*)ring - (size_t)FBIAS, H, W, 1.0f / scale, shift, pitch, pitch * (uint32_t)(H + 2 * PADY)};
    };
    const CorrLevel A0 = level(fmap0_nhwc, H0, W0, scale0, 0);
    const CorrArgs2 a{coords, kk, jj, order, (int)E, (uint32_t)kmod, (uint32_t)jmod, kmagic, jmagic, (uint32_t)Ng,
                      (uint32_t)slots, (const char*)gmap, (_Float16*)out, A0,
                      nlev == 2 ? level(fmap1_nhwc, H1, W1, scale1, ex1 - ex0) : A0, C, gmap_pixel_major, exp,
                      out_stride, out_off, out_pitch, coords_ref, ref_mul, split, rec, dynE};
    const int blocks = 8 * (int)cdv_div_up(E, 8 * CW * (rec ? EPW : 1));   // a multiple of 8: the kernel deals contiguous eighths to the XCDs
    const size_t smem2 = (size_t)CW * WAVE_LDS2_BYTES;
    if (smem2 > 48 * 1024) {   // (experiment builds with more than 11 waves per workgroup)
      static const hipError_t attr = [&] {
        hipError_t e = hipSuccess, x;
#define CDV_CATTR(...) if ((x = hipFuncSetAttribute((const void*)corr_fused2_kernel<__VA_ARGS__>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem2)) != hipSuccess) e = x;
        CDV_CATTR(24, 2, false, true) CDV_CATTR(0, 2, false, true) CDV_CATTR(24, 2, true) CDV_CATTR(0, 2, true)
        CDV_CATTR(24, 2) CDV_CATTR(0, 2) CDV_CATTR(24, 1) CDV_CATTR(0, 1)
#undef CDV_CATTR
        return e;
      }();
      CDV_HIP_CHECK(attr);
    }
    if (rec) {   // packed input stream in processing order (cdv_corr_fused_stream)
      CDV_REQUIRE(nlev == 2 && !split && coords_ref == nullptr, CDV_ERR_UNSUPPORTED, "cdv_corr_fused_stream: two fused levels only");
      if (C == 24) hipLaunchKernelGGL((corr_fused2_kernel<24, 2, false, true>), dim3(blocks), dim3(64 * CW), smem2, s, rec, dynE, (int)E, blocks >> 3, a);
      else hipLaunchKernelGGL((corr_fused2_kernel<0, 2, false, true>), dim3(blocks), dim3(64 * CW), smem2, s, rec, dynE, (int)E, blocks >> 3, a);
    } else if (nlev == 2 && split) {   // levels kept apart (cdv_corr_fused_split): a variant of its own, so that the main kernel
                              // keeps its 72 VGPRs (74 with the choice at run time: 6 instead of 7 waves per SIMD, +7 %)
      if (C == 24) hipLaunchKernelGGL((corr_fused2_kernel<24, 2, true>), dim3(blocks), dim3(64 * CW), smem2, s, rec, dynE, (int)E, blocks >> 3, a);
      else hipLaunchKernelGGL((corr_fused2_kernel<0, 2, true>), dim3(blocks), dim3(64 * CW), smem2, s, rec, dynE, (int)E, blocks >> 3, a);
    } else if (nlev == 2 && C == 24)
      hipLaunchKernelGGL((corr_fused2_kernel<24, 2>), dim3(blocks), dim3(64 * CW), smem2, s, rec, dynE, (int)E, blocks >> 3, a);
    else if (nlev == 2)
      hipLaunchKernelGGL((corr_fused2_kernel<0, 2>), dim3(blocks), dim3(64 * CW), smem2, s, rec, dynE, (int)E, blocks >> 3, a);
    else if (C == 24)   // one level per call: what an unchanged slam.py issues (slam.py:316-323), twice per update
      hipLaunchKernelGGL((corr_fused2_kernel<24, 1>), dim3(blocks), dim3(64 * CW), smem2, s, rec, dynE, (int)E, blocks >> 3, a);
    else
      hipLaunchKernelGGL((corr_fused2_kernel<0, 1>), dim3(blocks), dim3(64 * CW), smem2, s, rec, dynE, (int)E, blocks >> 3, a);
  } else {
    CDV_REQUIRE(out_stride == 1 && out_off == 0 && coords_ref == nullptr && !split, CDV_ERR_UNSUPPORTED,
                "cdv_corr_fused: split levels / checked calls need C <= 32");
    LevelParams L0{(const _Float16*)fmap0_nhwc, H0, W0, 1.0f / scale0, 0};
    LevelParams L1{(const _Float16*)fmap1_nhwc, H1, W1, nlev == 2 ? 1.0f / scale1 : 1.0f, nlev == 2 ? ex1 - ex0 : 0};
    hipLaunchKernelGGL(corr_wide_kernel<4>, dim3(cdv_div_up(E, 4)), dim3(256), smem, s, (const _Float16*)gmap, L0, L1, coords,
                       kk, jj, order, (_Float16*)out, (int)E, Ng, slots, C, nlev, kmod, jmod, kmagic, jmagic,
                       gmap_pixel_major, exp);
  }
  CDV_LAUNCH_CHECK();
  return CDV_OK;
}

extern "C" int cdv_corr_fused(const void* gmap, const void* fmap0_nhwc, const void* fmap1_nhwc, const float* coords,
                              const int64_t* kk, const int64_t* jj, const int32_t* order, void* out, int64_t E,
                              int64_t Ng, int64_t slots, int C, int H0, int W0, int H1, int W1, float scale0,
                              float scale1, int nlev, int64_t kmod, int64_t jmod, int gmap_pixel_major, void* stream) {
  return corr_fused_impl(gmap, fmap0_nhwc, fmap1_nhwc, coords, kk, jj, order, out, E, Ng, slots, C, H0, W0, H1, W1, scale0,
                         scale1, nlev, kmod, jmod, gmap_pixel_major, stream, 1, 0, 441, nullptr, 1.0f, 0);
}

extern "C" int cdv_corr_fused_stream(const void* gmap, const void* fmap0_nhwc, const void* fmap1_nhwc, const void* records,
                                     void* out, int64_t E, int64_t Ng, int64_t slots, int C, int H0, int W0, int H1, int W1,
                                     float scale0, float scale1, int gmap_pixel_major, void* stream) {
  CDV_REQUIRE(records != nullptr, CDV_ERR_ARG, "cdv_corr_fused_stream: NULL record stream");
  CDV_REQUIRE(C <= 32, CDV_ERR_UNSUPPORTED, "cdv_corr_fused_stream: C must be <= 32");
  return corr_fused_impl(gmap, fmap0_nhwc, fmap1_nhwc, nullptr, nullptr, nullptr, nullptr, out, E, Ng, slots, C, H0, W0, H1,
                         W1, scale0, scale1, 2, 0, 0, gmap_pixel_major, stream, 1, 0, 441, nullptr, 1.0f, 0,
                         (const uint32_t*)records);
}

// cdv_corr_fused_stream with the number of edges on the device: *dyn_E edges (at most E_bound, which sizes the launch)
extern "C" int cdv_corr_fused_stream_dyn(const void* gmap, const void* fmap0_nhwc, const void* fmap1_nhwc, const void* records,
                                         void* out, int64_t E_bound, const int32_t* dyn, int64_t Ng, int64_t slots, int C, int H0,
                                         int W0, int H1, int W1, float scale0, float scale1, int gmap_pixel_major, void* stream) {
  CDV_REQUIRE(records != nullptr && dyn != nullptr, CDV_ERR_ARG, "cdv_corr_fused_stream_dyn: NULL record stream / dynamic block");
  CDV_REQUIRE(C <= 32, CDV_ERR_UNSUPPORTED, "cdv_corr_fused_stream_dyn: C must be <= 32");
  return corr_fused_impl(gmap, fmap0_nhwc, fmap1_nhwc, nullptr, nullptr, nullptr, nullptr, out, E_bound, Ng, slots, C, H0, W0,
                         H1, W1, scale0, scale1, 2, 0, 0, gmap_pixel_major, stream, 1, 0, 441, nullptr, 1.0f, 0,
                         (const uint32_t*)records, dyn + CDV_DYN_E);
}

extern "C" int cdv_corr_fused_split(const void* gmap, const void* fmap0_nhwc, const void* fmap1_nhwc, const float* coords,
                                    const int64_t* kk, const int64_t* jj, const int32_t* order, void* out2, int64_t E,
                                    int64_t Ng, int64_t slots, int C, int H0, int W0, int H1, int W1, float scale0,
                                    float scale1, int64_t kmod, int64_t jmod, int gmap_pixel_major, void* stream) {
  CDV_REQUIRE(C <= 32, CDV_ERR_UNSUPPORTED, "cdv_corr_fused_split: C must be <= 32");
  return corr_fused_impl(gmap, fmap0_nhwc, fmap1_nhwc, coords, kk, jj, order, out2, E, Ng, slots, C, H0, W0, H1, W1, scale0,
                         scale1, 2, kmod, jmod, gmap_pixel_major, stream, 1, 0, 884, nullptr, 1.0f, 1);
}

extern "C" int cdv_corr_level_checked(const void* gmap, const void* fmap_nhwc, const float* coords, const float* coords_ref,
                                      float ref_mul, const int64_t* kk, const int64_t* jj, void* out2, int level, int64_t E,
                                      int64_t Ng, int64_t slots, int C, int H, int W, float scale, int64_t kmod,
                                      int64_t jmod, int gmap_pixel_major, void* stream) {
  CDV_REQUIRE(level == 0 || level == 1, CDV_ERR_ARG, "cdv_corr_level_checked: level must be 0 or 1");
  CDV_REQUIRE(coords_ref != nullptr && out2 != nullptr, CDV_ERR_ARG, "cdv_corr_level_checked: NULL argument");
  CDV_REQUIRE(C <= 32, CDV_ERR_UNSUPPORTED, "cdv_corr_level_checked: C must be <= 32");
  return corr_fused_impl(gmap, fmap_nhwc, nullptr, coords, kk, jj, nullptr, out2, E, Ng, slots, C, H, W, 0, 0, scale, 1.0f, 1,
                         kmod, jmod, gmap_pixel_major, stream, 1, 442 * level, 884, coords_ref, ref_mul, 0);
}

// cdv_corr_level_checked into the INTERLEAVED two-level result of cdv_corr_fused ([E][441][2] halves: what SLAM.corr's
// torch.stack(..., -1) produces): level `level` of element t of edge e lives at out[e * 882 + 2 t + level]
extern "C" int cdv_corr_level_checked_interleaved(const void* gmap, const void* fmap_nhwc, const float* coords,
                                                  const float* coords_ref, float ref_mul, const int64_t* kk, const int64_t* jj,
                                                  void* out, int level, int64_t E, int64_t Ng, int64_t slots, int C, int H, int W,
                                                  float scale, int64_t kmod, int64_t jmod, int gmap_pixel_major, void* stream) {
  CDV_REQUIRE(level == 0 || level == 1, CDV_ERR_ARG, "cdv_corr_level_checked_interleaved: level must be 0 or 1");
  CDV_REQUIRE(coords_ref != nullptr && out != nullptr, CDV_ERR_ARG, "cdv_corr_level_checked_interleaved: NULL argument");
  CDV_REQUIRE(C <= 32, CDV_ERR_UNSUPPORTED, "cdv_corr_level_checked_interleaved: C must be <= 32");
  return corr_fused_impl(gmap, fmap_nhwc, nullptr, coords, kk, jj, nullptr, out, E, Ng, slots, C, H, W, 0, 0, scale, 1.0f, 1,
                         kmod, jmod, gmap_pixel_major, stream, 2, level, 882, coords_ref, ref_mul, 0);
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
