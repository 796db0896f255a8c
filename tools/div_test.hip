// div_test.hip -- is a scale-free Newton/Markstein fp64 division bit-identical to IEEE '/' on the operand ranges of the
// sweep?  (a) Log2Hot's t = (z-m)/(z+m): |num| < 2^-10, den in [2,4); (b) 1/D with D in [1e-6,1e12];
// (c) generic quotients.  Also reports the accuracy of v_rcp_f64.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>

__device__ __forceinline__ uint64_t sm64(uint64_t x) {
  uint64_t z = x + 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
__device__ __forceinline__ double unit(uint64_t h) { return (double)(h >> 11) * 0x1.0p-53; }

template <int NEWTON>
__device__ __forceinline__ double div_nr(double n, double d) {
  double r = __builtin_amdgcn_rcp(d);
#pragma unroll
  for (int i = 0; i < NEWTON; i++) {
    const double e = fma(-d, r, 1.0);
    r = fma(r, e, r);
  }
  const double q0 = n * r;
  const double rem = fma(-d, q0, n);
  return fma(rem, r, q0);
}

template <int NEWTON>
__device__ __forceinline__ double rcp_nr(double d) {
  double r = __builtin_amdgcn_rcp(d);
#pragma unroll
  for (int i = 0; i < NEWTON; i++) {
    const double e = fma(-d, r, 1.0);
    r = fma(r, e, r);
  }
  return r;
}

__global__ void test(unsigned long long *bad, double *maxRcpErr, int mode, uint64_t seed, int iters) {
  const uint64_t gid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  unsigned long long b1 = 0, b2 = 0, b3 = 0, br2 = 0, br3 = 0;
  double worst = 0;
  for (int it = 0; it < iters; it++) {
    const uint64_t h1 = sm64(seed + gid * 1315423911ULL + it * 2654435761ULL), h2 = sm64(h1);
    double n, d;
    if (mode == 0) {  // Log2Hot: z in [1,2), m = bucket midpoint
      const uint64_t uz = 0x3FF0000000000000ULL | (h1 >> 12);
      const double z = __longlong_as_double(uz);
      const double m = __longlong_as_double((1ULL << 41) | (uz & ~((1ULL << 42) - 1)));
      n = z - m;
      d = z + m;
    } else if (mode == 1) {  // 1/D
      n = 1.0;
      d = exp2(unit(h1) * 60.0 - 20.0) * (1.0 + unit(h2));
    } else {  // generic
      n = exp2(unit(h1) * 200.0 - 100.0) * (1.0 + unit(sm64(h2)));
      d = exp2(unit(h2) * 200.0 - 100.0) * (1.0 + unit(sm64(h1 ^ h2)));
    }
    const double ref = n / d;
    if (div_nr<1>(n, d) != ref) b1++;
    if (div_nr<2>(n, d) != ref) b2++;
    if (div_nr<3>(n, d) != ref) b3++;
    if (mode == 1) {
      if (rcp_nr<2>(d) != ref) br2++;
      if (rcp_nr<3>(d) != ref) br3++;
    }
    const double r0 = __builtin_amdgcn_rcp(d);
    const double err = fabs(fma(-d, r0, 1.0));
    worst = err > worst ? err : worst;
  }
  atomicAdd(&bad[0], b1);
  atomicAdd(&bad[1], b2);
  atomicAdd(&bad[2], b3);
  atomicAdd(&bad[3], br2);
  atomicAdd(&bad[4], br3);
  // max via atomicMax on the bits (positive doubles order as integers)
  atomicMax((unsigned long long *)maxRcpErr, (unsigned long long)__double_as_longlong(worst));
}

int main() {
  unsigned long long *bad;
  double *mx;
  hipMalloc(&bad, 5 * 8);
  hipMalloc(&mx, 8);
  for (int mode = 0; mode < 3; mode++) {
    hipMemset(bad, 0, 40);
    hipMemset(mx, 0, 8);
    const int blocks = 4096, threads = 256, iters = 1000;
    hipLaunchKernelGGL(test, dim3(blocks), dim3(threads), 0, 0, bad, mx, mode, 12345ULL + mode, iters);
    hipDeviceSynchronize();
    unsigned long long h[5];
    double hm;
    hipMemcpy(h, bad, 40, hipMemcpyDeviceToHost);
    hipMemcpy(&hm, mx, 8, hipMemcpyDeviceToHost);
    printf("mode %d: %.3g samples; mismatches vs IEEE '/': newton1=%llu newton2=%llu newton3=%llu  rcp-only newton2=%llu newton3=%llu; "
           "max |1-d*rcp(d)| = %.3g (2^%.1f)\n",
           mode, (double)blocks * threads * iters, h[0], h[1], h[2], h[3], h[4], hm, log2(hm));
  }
  return 0;
}
