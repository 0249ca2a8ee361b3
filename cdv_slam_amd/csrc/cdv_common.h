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
