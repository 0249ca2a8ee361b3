// cdv_common.h -- shared host/device helpers for the cdvslam HIP library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "../../include/cdvslam_hip.h"

#define CDV_WAVE 64

#define CDV_HIP_CHECK(expr)                              \
  do {                                                   \
    hipError_t _e = (expr);                              \
    if (_e != hipSuccess) {                              \
      cdv_set_error(CDV_ERR_HIP, hipGetErrorString(_e)); \
      return CDV_ERR_HIP;                                \
    }                                                    \
  } while (0)

#define CDV_REQUIRE(cond, code, msg) \
  do {                               \
    if (!(cond)) {                   \
      cdv_set_error((code), (msg));  \
      return (code);                 \
    }                                \
  } while (0)

// after a kernel launch: catches launch-configuration errors without synchronising
#define CDV_LAUNCH_CHECK() CDV_HIP_CHECK(hipGetLastError())

void cdv_set_error(int code, const char* msg);

static inline int cdv_div_up(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

typedef _Float16 cdv_half8 __attribute__((ext_vector_type(8)));
typedef float cdv_float4 __attribute__((ext_vector_type(4)));

// ---- in-kernel cycle stamps (diagnostic builds only: make STAMPS=1 -> libcdvslam_hip_stamps.so) ------------
// Each translation unit that wants stamps defines CDV_STAMP_TU(name): a __device__ buffer pointer and an
// exported setter cdv_set_stamps_<name>(ptr).  Layout: [wave slot][16] of s_memtime values; lane 0 of a wave
// writes.  No stamp executes in the product library.
#ifdef CDV_STAMPS
#define CDV_STAMP_TU(name)                                                                      \
  __device__ unsigned long long* g_stamps_##name = nullptr;                                     \
  extern "C" int cdv_set_stamps_##name(void* p) {                                               \
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_stamps_##name), &p, sizeof(p));                  \
  }
__device__ __forceinline__ unsigned long long cdv_now() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#define CDV_STAMP(name, slot, id)                                                               \
  do {                                                                                          \
    const unsigned long long _t = cdv_now();                                                    \
    if (g_stamps_##name && (threadIdx.x & 63) == 0) g_stamps_##name[(size_t)(slot) * 16 + (id)] = _t; \
  } while (0)
#define CDV_STAMP_RT(name, slot, id)                                                            \
  do {                                                                                          \
    const unsigned long long _t = __builtin_amdgcn_s_memrealtime();                             \
    if (g_stamps_##name && (threadIdx.x & 63) == 0) g_stamps_##name[(size_t)(slot) * 16 + (id)] = _t; \
  } while (0)
#define CDV_STAMP_VAL(name, slot, id, val)                                                      \
  do {                                                                                          \
    if (g_stamps_##name && (threadIdx.x & 63) == 0) g_stamps_##name[(size_t)(slot) * 16 + (id)] = (val); \
  } while (0)
#define CDV_IF_STAMPS(...) __VA_ARGS__
#else
#define CDV_STAMP_TU(name)
#define CDV_STAMP(name, slot, id) do { } while (0)
#define CDV_STAMP_VAL(name, slot, id, val) do { } while (0)
#define CDV_STAMP_RT(name, slot, id) do { } while (0)
#define CDV_IF_STAMPS(...)
#endif
