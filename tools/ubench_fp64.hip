// ubench_fp64.hip -- per-instruction issue cost of the fp64 operations the sweep is made of, on gfx950.
// Build+run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench_fp64.hip -o /tmp/ub && /tmp/ub
// Each test runs ITER x UNROLL independent-chain ops per lane in `waves` waves on ONE SIMD-worth of a CU (block of
// 64*waves threads, 1 block) and reports cycles per wave-instruction from s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define ITER 2000

template <int OP>
__device__ __forceinline__ double op(double a, double b, double c) {
  if constexpr (OP == 0) return fma(a, b, c);
  if constexpr (OP == 1) return a + b;
  if constexpr (OP == 2) return a * b;
  if constexpr (OP == 3) return __builtin_amdgcn_rcp(a);
  if constexpr (OP == 4) return b / a;                       // IEEE division sequence
  if constexpr (OP == 5) return __builtin_amdgcn_div_fixup(a, b, c);
  if constexpr (OP == 6) return (double)__double2int_rn(a) + b;
  if constexpr (OP == 7) return __shfl_xor(a, 16, 64);
  if constexpr (OP == 8) return __shfl_xor(a, 1, 64);
  if constexpr (OP == 9) return sqrt(a);
  if constexpr (OP == 10) return __builtin_amdgcn_ldexp(a, 3);
  if constexpr (OP == 11) return __builtin_amdgcn_div_scale(a, b, true, nullptr);
  return a;
}

template <int OP, int CHAINS>
__global__ void bench(double *out, long long *cycles, double seed) {
  double x[CHAINS];
#pragma unroll
  for (int i = 0; i < CHAINS; i++) x[i] = seed + threadIdx.x * 1e-3 + i;
  const double b = 1.0000001, c = 1e-9;
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int i = 0; i < CHAINS; i++) x[i] = op<OP>(x[i], b, c);
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
#pragma unroll
  for (int i = 0; i < CHAINS; i++) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x % 64 == 0) cycles[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int OP, int CHAINS>
void run(const char *name, int wavesPerSimd) {
  double *out;
  long long *cyc;
  const int threads = 64 * 4 * wavesPerSimd;  // a block's waves are spread over the 4 SIMDs of one CU
  hipMalloc(&out, threads * sizeof(double));
  hipMalloc(&cyc, 64 * sizeof(long long));
  hipLaunchKernelGGL((bench<OP, CHAINS>), dim3(1), dim3(threads), 0, 0, out, cyc, 1.5);
  hipDeviceSynchronize();
  long long h[64];
  hipMemcpy(h, cyc, (threads / 64) * sizeof(long long), hipMemcpyDeviceToHost);
  long long mx = 0;
  for (int i = 0; i < threads / 64; i++) mx = h[i] > mx ? h[i] : mx;
  // s_memtime counts at a fixed 100 MHz on gfx9?  report raw ticks per op and per-SIMD issue cost assuming the
  // waves of one SIMD serialise: ticks * / (ITER*CHAINS*wavesPerSimd)
  printf("%-14s chains=%d waves/SIMD=%d  ticks/instr/wave=%.3f  ticks/instr (SIMD-serialised)=%.3f\n", name, CHAINS,
         wavesPerSimd, (double)mx / (ITER * CHAINS), (double)mx / ((double)ITER * CHAINS * wavesPerSimd));
  hipFree(out);
  hipFree(cyc);
}

__global__ void clock_ratio(long long *o) {
  const long long m0 = __builtin_amdgcn_s_memtime();
  const long long c0 = clock64();
  for (volatile int i = 0; i < 100000; i++) {}
  const long long m1 = __builtin_amdgcn_s_memtime();
  const long long c1 = clock64();
  o[0] = m1 - m0;
  o[1] = c1 - c0;
}

int main() {
  long long *o;
  hipMalloc(&o, 16);
  hipLaunchKernelGGL(clock_ratio, dim3(1), dim3(1), 0, 0, o);
  long long h[2];
  hipMemcpy(h, o, 16, hipMemcpyDeviceToHost);
  printf("memtime ticks=%lld clock64 ticks=%lld\n", h[0], h[1]);
  for (int w : {1, 2, 4}) {
    run<0, 8>("fma", w);
    run<1, 8>("add", w);
    run<2, 8>("mul", w);
    run<3, 8>("rcp", w);
    run<4, 8>("div(IEEE)", w);
    run<5, 8>("div_fixup", w);
    run<6, 8>("cvt i32<->f64", w);
    run<7, 8>("shfl_xor 16", w);
    run<8, 8>("shfl_xor 1", w);
    run<9, 8>("sqrt", w);
    run<10, 8>("ldexp", w);
    run<11, 8>("div_scale", w);
  }
  run<0, 1>("fma dep", 1);
  run<1, 1>("add dep", 1);
  run<4, 1>("div dep", 1);
  run<7, 1>("shfl16 dep", 1);
  run<3, 1>("rcp dep", 1);
  return 0;
}
