// core.cpp -- error reporting + version of libcdvslam_hip.so
#include <string.h>

#include "../../include/cdvslam_hip.h"

static thread_local char g_err[512] = "";

void cdv_set_error(int code, const char* msg) {
  (void)code;
  strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
  g_err[sizeof(g_err) - 1] = 0;
}

extern "C" const char* cdv_last_error(void) { return g_err; }

extern "C" const char* cdv_version(void) { return "cdvslam_hip 0.1 (gfx950, ROCm " CDV_ROCM_VERSION ")"; }
