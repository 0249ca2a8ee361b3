// Microbenchmark: the 64 x 64 Cholesky factorisation of one diagonal block inside one workgroup, by variant.
//   hipcc --offload-arch=gfx950 -O3 chol64.hip -o chol64 && ./chol64
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int NB = 64, LD = 68;

__device__ __forceinline__ unsigned long long now() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
__device__ __forceinline__ float rl(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ int lget(int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lset(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// ---- variant 0: one wave, whole rows (ba_factor.hip factor_rows) ----
__device__ __forceinline__ void factor_rows(const float* Db, float* colb, int lane, f2 (&a2)[NB / 2]) {
#pragma unroll
  for (int c4 = 0; c4 < NB / 4; c4++) {
    const f4 q = *reinterpret_cast<const f4*>(&Db[lane * LD + 4 * c4]);
    a2[2 * c4] = f2{q[0], q[1]};
    a2[2 * c4 + 1] = f2{q[2], q[3]};
  }
  float Lk;
  {
    const float piv = rl(a2[0][0], 0);
    Lk = a2[0][0] * __builtin_amdgcn_rsqf(piv);
    a2[0][0] = Lk;
    colb[lane] = Lk;
  }
  f2 bcur[NB / 2], bnxt[NB / 2];
#pragma unroll
  for (int c4 = 0; c4 < NB / 4; c4++) {
    const f4 v = *reinterpret_cast<const f4*>(&colb[4 * c4]);
    bcur[2 * c4] = f2{v[0], v[1]};
    bcur[2 * c4 + 1] = f2{v[2], v[3]};
  }
#pragma unroll
  for (int k = 0; k < NB; k++) {
    float Ln = 0.f;
    if (k + 1 < NB) {
      const float an = fmaf(-Lk, rl(Lk, k + 1), a2[(k + 1) >> 1][(k + 1) & 1]);
      const float piv = rl(an, k + 1);
      Ln = an * __builtin_amdgcn_rsqf(piv);
      a2[(k + 1) >> 1][(k + 1) & 1] = Ln;
      colb[lane] = Ln;
#pragma unroll
      for (int c4 = (k + 2) / 4; c4 < NB / 4; c4++) {
        const f4 v = *reinterpret_cast<const f4*>(&colb[4 * c4]);
        bnxt[2 * c4] = f2{v[0], v[1]};
        bnxt[2 * c4 + 1] = f2{v[2], v[3]};
      }
    }
    if (((k + 2) & 1) && k + 2 < NB) a2[(k + 2) >> 1][1] = fmaf(-Lk, bcur[(k + 2) >> 1][1], a2[(k + 2) >> 1][1]);
    const f2 nLk = {-Lk, -Lk};
#pragma unroll
    for (int pp = (k + 3) >> 1; pp < NB / 2; pp++) a2[pp] = __builtin_elementwise_fma(nLk, bcur[pp], a2[pp]);
    Lk = Ln;
#pragma unroll
    for (int pp = (k + 2) >> 1; pp < NB / 2; pp++) bcur[pp] = bnxt[pp];
  }
}

// ---- variants 1..: four waves, 16 columns each (compile-time wave number: lane indices are immediates) ----
// AH: columns updated ahead through v_readlane; the rest of a column's rank-1 update from its LDS read-back, applied DL steps later
template <int W, int AH, bool FOLLOW, int SLEEP>
__device__ __forceinline__ void factor_cols(f2 (&a)[8], float* Lc, int* prog, int lane) {
  constexpr int r0 = 16 * W;
  constexpr int DL = AH - 1 < 15 ? AH - 1 : 0;
  if (FOLLOW && W > 0) {
    int k = 0;
    while (k < r0) {
      int have = lget(prog);
      while (have <= k) { if (SLEEP) __builtin_amdgcn_s_sleep(SLEEP); have = lget(prog); }
      asm volatile("" ::: "memory");
      const int n = (have < r0 ? have : r0) - k;
      for (int j = 0; j < n; j++) {
        const int kc = k + j;
        const float lrow = Lc[kc * NB + lane];
        const f4 q0 = *reinterpret_cast<const f4*>(&Lc[kc * NB + r0]);
        const f4 q1 = *reinterpret_cast<const f4*>(&Lc[kc * NB + r0 + 4]);
        const f4 q2 = *reinterpret_cast<const f4*>(&Lc[kc * NB + r0 + 8]);
        const f4 q3 = *reinterpret_cast<const f4*>(&Lc[kc * NB + r0 + 12]);
        const f2 nl = {-lrow, -lrow};
        a[0] = __builtin_elementwise_fma(nl, f2{q0[0], q0[1]}, a[0]); a[1] = __builtin_elementwise_fma(nl, f2{q0[2], q0[3]}, a[1]);
        a[2] = __builtin_elementwise_fma(nl, f2{q1[0], q1[1]}, a[2]); a[3] = __builtin_elementwise_fma(nl, f2{q1[2], q1[3]}, a[3]);
        a[4] = __builtin_elementwise_fma(nl, f2{q2[0], q2[1]}, a[4]); a[5] = __builtin_elementwise_fma(nl, f2{q2[2], q2[3]}, a[5]);
        a[6] = __builtin_elementwise_fma(nl, f2{q3[0], q3[1]}, a[6]); a[7] = __builtin_elementwise_fma(nl, f2{q3[2], q3[3]}, a[7]);
      }
      k += n;
    }
  }
  float Lh[16];
  f4 qb[16][4];
  {
    const float piv = rl(a[0][0], r0);
    Lh[0] = a[0][0] * __builtin_amdgcn_rsqf(piv);
    a[0][0] = Lh[0];
    Lc[r0 * NB + lane] = Lh[0];
    if (FOLLOW) lset(prog, r0 + 1);
  }
#pragma unroll
  for (int kk = 0; kk < 16; kk++) {
    const int k = r0 + kk;
    const float Lk = Lh[kk];
    if (kk + 1 < 16) {
      const float an = fmaf(-Lk, rl(Lk, k + 1), a[(kk + 1) >> 1][(kk + 1) & 1]);
      const float piv = rl(an, k + 1);
      const float Ln = an * __builtin_amdgcn_rsqf(piv);
      a[(kk + 1) >> 1][(kk + 1) & 1] = Ln;
      Lh[kk + 1] = Ln;
      Lc[(k + 1) * NB + lane] = Ln;
      if (FOLLOW) lset(prog, k + 2);
    }
#pragma unroll
    for (int d = 2; d <= AH; d++)
      if (kk + d < 16) a[(kk + d) >> 1][(kk + d) & 1] = fmaf(-Lk, rl(Lk, k + d), a[(kk + d) >> 1][(kk + d) & 1]);
    if (kk + AH + 1 < 16) {
#pragma unroll
      for (int c4 = (kk + AH + 1) / 4; c4 < 4; c4++) qb[kk][c4] = *reinterpret_cast<const f4*>(&Lc[k * NB + r0 + 4 * c4]);
    }
    if (kk >= DL && (kk - DL) + AH + 1 < 16) {
      const int j = kk - DL;
#pragma unroll
      for (int c = j + AH + 1; c < 16; c++) a[c >> 1][c & 1] = fmaf(-Lh[j], qb[j][c >> 2][c & 3], a[c >> 1][c & 1]);
    }
  }
}

template <int VAR>
__global__ __launch_bounds__(256) void k(const float* A, float* Lout, unsigned long long* cyc) {
  __shared__ __attribute__((aligned(16))) float Db[NB * LD];
  __shared__ __attribute__((aligned(16))) float Lc[NB * NB];
  __shared__ __attribute__((aligned(16))) float colb[NB];
  __shared__ int prog[4];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  for (int i = t; i < NB * NB; i += 256) Db[(i >> 6) * LD + (i & 63)] = A[i];
  if (t == 0) prog[0] = 0;
  __syncthreads();
  const unsigned long long t0 = now();
  if (VAR == 0) {
    if (wave == 0) {
      f2 a2[NB / 2];
      factor_rows(Db, colb, lane, a2);
      const unsigned long long t1e = now();
      if (lane == 0) cyc[4] = t1e - t0;
#pragma unroll
      for (int c = 0; c < NB; c++) Lout[lane * NB + c] = c <= lane ? a2[c >> 1][c & 1] : 0.f;
    }
  } else {
    f2 a[8];
#pragma unroll
    for (int c4 = 0; c4 < 4; c4++) {
      const f4 q = *reinterpret_cast<const f4*>(&Db[lane * LD + 16 * wave + 4 * c4]);
      a[2 * c4] = f2{q[0], q[1]};
      a[2 * c4 + 1] = f2{q[2], q[3]};
    }
    constexpr int AH = VAR == 1 ? 1 : VAR == 2 ? 4 : VAR == 3 ? 16 : VAR == 4 ? 4 : VAR == 5 ? 2 : 4;
    constexpr bool FOLLOW = VAR != 4;     // variant 4: wave 0's panel alone, no progress word, nobody follows
    constexpr int SLEEP = VAR == 6 ? 1 : VAR == 7 ? 3 : 0;
    if (wave == 0) factor_cols<0, AH, FOLLOW, SLEEP>(a, Lc, prog, lane);
    else if (FOLLOW && wave == 1) factor_cols<1, AH, FOLLOW, SLEEP>(a, Lc, prog, lane);
    else if (FOLLOW && wave == 2) factor_cols<2, AH, FOLLOW, SLEEP>(a, Lc, prog, lane);
    else if (FOLLOW) factor_cols<3, AH, FOLLOW, SLEEP>(a, Lc, prog, lane);
    const unsigned long long t1e = now();
    if (lane == 0) cyc[4 + wave] = t1e - t0;
#pragma unroll
    for (int c = 0; c < 16; c++) Lout[lane * NB + 16 * wave + c] = 16 * wave + c <= lane ? a[c >> 1][c & 1] : 0.f;
  }
  const unsigned long long t1 = now();
  if (lane == 0) cyc[wave] = t1 - t0;
}

int main() {
  std::vector<float> M(NB * NB), A(NB * NB), L(NB * NB), Lg(NB * NB);
  srand(7);
  for (auto& v : M) v = (rand() / (float)RAND_MAX - 0.5f);
  for (int i = 0; i < NB; i++)
    for (int j = 0; j < NB; j++) {
      double s = 0;
      for (int q = 0; q < NB; q++) s += (double)M[i * NB + q] * M[j * NB + q];
      A[i * NB + j] = (float)s + (i == j ? 8.f : 0.f);
    }
  std::vector<double> Ld(NB * NB, 0.0);
  for (int j = 0; j < NB; j++) {
    double d = A[j * NB + j];
    for (int q = 0; q < j; q++) d -= Ld[j * NB + q] * Ld[j * NB + q];
    Ld[j * NB + j] = sqrt(d);
    for (int i = j + 1; i < NB; i++) {
      double s = A[i * NB + j];
      for (int q = 0; q < j; q++) s -= Ld[i * NB + q] * Ld[j * NB + q];
      Ld[i * NB + j] = s / Ld[j * NB + j];
    }
  }
  float *dA, *dL; unsigned long long* dc;
  (void)hipMalloc(&dA, NB * NB * 4); (void)hipMalloc(&dL, NB * NB * 4); (void)hipMalloc(&dc, 64);
  (void)hipMemcpy(dA, A.data(), NB * NB * 4, hipMemcpyHostToDevice);
  const char* names[8] = {"one wave, whole rows (factor_rows)", "four waves, read-back one column late (AH 1)", "four waves, 4 ahead by readlane, read-back 3 late",
                          "four waves, everything by readlane (AH 16)", "wave 0's panel alone (AH 4), nobody follows", "four waves, 2 ahead, read-back 1 late",
                          "four waves, AH 4, followers sleep 1 between polls", "four waves, AH 4, followers sleep 3 between polls"};
  for (int v = 0; v < 8; v++) {
    unsigned long long hc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    (void)hipMemset(dc, 0, 64);
    for (int rep = 0; rep < 3; rep++) {
      (void)hipMemset(dL, 0, NB * NB * 4);
      switch (v) {
        case 0: hipLaunchKernelGGL(k<0>, dim3(1), dim3(256), 0, 0, dA, dL, dc); break;
        case 1: hipLaunchKernelGGL(k<1>, dim3(1), dim3(256), 0, 0, dA, dL, dc); break;
        case 2: hipLaunchKernelGGL(k<2>, dim3(1), dim3(256), 0, 0, dA, dL, dc); break;
        case 3: hipLaunchKernelGGL(k<3>, dim3(1), dim3(256), 0, 0, dA, dL, dc); break;
        case 4: hipLaunchKernelGGL(k<4>, dim3(1), dim3(256), 0, 0, dA, dL, dc); break;
        case 5: hipLaunchKernelGGL(k<5>, dim3(1), dim3(256), 0, 0, dA, dL, dc); break;
        case 6: hipLaunchKernelGGL(k<6>, dim3(1), dim3(256), 0, 0, dA, dL, dc); break;
        case 7: hipLaunchKernelGGL(k<7>, dim3(1), dim3(256), 0, 0, dA, dL, dc); break;
      }
      (void)hipDeviceSynchronize();
    }
    (void)hipMemcpy(hc, dc, 64, hipMemcpyDeviceToHost);
    (void)hipMemcpy(Lg.data(), dL, NB * NB * 4, hipMemcpyDeviceToHost);
    double err = 0;
    const int cols = v == 4 ? 16 : NB;
    for (int i = 0; i < NB; i++)
      for (int j = 0; j <= i && j < cols; j++) err = fmax(err, fabs(Lg[i * NB + j] - Ld[i * NB + j]));
    printf("%-52s factor done at %6llu %6llu %6llu %6llu cycles (with the stores %6llu)   max |L - L64| %.2e\n", names[v], hc[4], hc[5], hc[6], hc[7], hc[3] > hc[0] ? hc[3] : hc[0], err);
  }
  return 0;
}
