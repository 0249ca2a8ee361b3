// Microbenchmark: what one step of a cross-lane dependent chain costs on gfx950, by the way a value travels from one lane to all:
//   x <- fma(x, bcast(x, lane k), c), 512 dependent steps, one wave.      hipcc --offload-arch=gfx950 -O3 lane_bcast.hip -o lane_bcast
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ unsigned long long now() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
__device__ __forceinline__ float rl(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

template <int MODE>
__global__ void k(float* out, unsigned long long* cyc, float c) {
  __shared__ float sh[64];
  const int lane = threadIdx.x;
  float x = 1.0f + 1e-3f * lane;
  const unsigned long long t0 = now();
#pragma unroll
  for (int i = 0; i < 512; i++) {
    const int kk = (i * 7 + 3) & 63;
    float b;
    if (MODE == 0) b = rl(x, kk);                                                                   // v_readlane -> SGPR -> VALU
    else if (MODE == 1) b = __int_as_float(__builtin_amdgcn_ds_bpermute(4 * kk, __float_as_int(x)));   // LDS crossbar
    else if (MODE == 2) { sh[lane] = x; asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); b = sh[kk]; }   // LDS write + broadcast read
    else if (MODE == 3) b = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x155, 0xf, 0xf, false));   // row_newbcast:5 (within 16 lanes)
    else if (MODE == 4) b = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xAA, 0xf, 0xf, false));    // quad_perm:[2,2,2,2] (within 4 lanes)
    else if (MODE == 5) b = x;                                                                         // no broadcast: the FMA chain alone
    else if (MODE == 6) { b = rl(x, kk); b = rl(b * x, (kk + 1) & 63); }                               // two readlanes in the step (the Cholesky chain)
    else b = __builtin_amdgcn_rsqf(x);                                                                 // MODE 7: a transcendental in the chain
    x = fmaf(x, b * 1e-3f, c);
  }
  const unsigned long long t1 = now();
  out[lane] = x;
  if (lane == 0) cyc[MODE] = t1 - t0;
}

int main() {
  float* out; unsigned long long* cyc;
  (void)hipMalloc(&out, 256); (void)hipMalloc(&cyc, 64 * 8); (void)hipMemset(cyc, 0, 64 * 8);
  for (int rep = 0; rep < 2; rep++) {
    hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, out, cyc, 0.5f);
    hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, out, cyc, 0.5f);
    hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, out, cyc, 0.5f);
    hipLaunchKernelGGL(k<3>, dim3(1), dim3(64), 0, 0, out, cyc, 0.5f);
    hipLaunchKernelGGL(k<4>, dim3(1), dim3(64), 0, 0, out, cyc, 0.5f);
    hipLaunchKernelGGL(k<5>, dim3(1), dim3(64), 0, 0, out, cyc, 0.5f);
    hipLaunchKernelGGL(k<6>, dim3(1), dim3(64), 0, 0, out, cyc, 0.5f);
    hipLaunchKernelGGL(k<7>, dim3(1), dim3(64), 0, 0, out, cyc, 0.5f);
    (void)hipDeviceSynchronize();
  }
  unsigned long long h[8];
  (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[8] = {"v_readlane (SGPR) + mul + fma", "ds_bpermute + mul + fma", "LDS write, broadcast read + mul + fma", "dpp row_newbcast (16 lanes) + mul + fma",
                          "dpp quad_perm (4 lanes) + mul + fma", "mul + fma alone", "two v_readlane + 2 mul + fma", "v_rsq + mul + fma"};
  for (int m = 0; m < 8; m++) printf("%-44s %6.1f cycles per dependent step\n", names[m], h[m] / 512.0);
  return 0;
}
