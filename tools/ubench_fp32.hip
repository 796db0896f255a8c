// ubench_fp32.hip -- per-instruction issue cost of the fp32 operations the batched sweep is made of, on gfx950.
// Build+run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench_fp32.hip -o /tmp/ub32 && /tmp/ub32
// ITER x CHAINS independent-chain ops per lane, `waves` waves per SIMD of ONE CU; cycles per wave-instruction from s_memtime
// (100 MHz -> shader cycles by the measured clock ratio of a known 4-cycle op is avoided: s_memtime counts shader cycles here).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define ITER 4000
typedef float f2 __attribute__((ext_vector_type(2)));

template <int OP>
__device__ __forceinline__ float op(float a, float b, float c) {
  if constexpr (OP == 0) return fmaf(a, b, c);
  if constexpr (OP == 1) return __builtin_amdgcn_logf(a) + 2.0f;     // v_log_f32 (+ an add to keep the chain in range)
  if constexpr (OP == 2) return __builtin_amdgcn_rcpf(a) + 1.0f;     // v_rcp_f32 (+ add)
  if constexpr (OP == 3) return a + b;                               // the add alone
  if constexpr (OP == 4) return __builtin_amdgcn_fmed3f(a, b, c) + 1e-3f;
  if constexpr (OP == 5) return __builtin_amdgcn_exp2f(a * 1e-3f);
  if constexpr (OP == 6) return __builtin_amdgcn_sqrtf(a) + 1.0f;
  return a;
}

template <int OP, int CHAINS>
__global__ void bench(float *out, long long *cycles, float seed) {
  float x[CHAINS];
#pragma unroll
  for (int i = 0; i < CHAINS; i++) x[i] = seed + threadIdx.x * 1e-3f + i;
  const float b = 1.0000001f, c = 1e-9f;
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int i = 0; i < CHAINS; i++) x[i] = op<OP>(x[i], b, c);
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
#pragma unroll
  for (int i = 0; i < CHAINS; i++) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x % 64 == 0) cycles[threadIdx.x / 64] = t1 - t0;
}

template <int CHAINS>
__global__ void bench_pk(f2 *out, long long *cycles, float seed) {
  f2 x[CHAINS];
#pragma unroll
  for (int i = 0; i < CHAINS; i++) x[i] = f2{seed + threadIdx.x * 1e-3f + i, seed + i};
  const f2 b = {1.0000001f, 0.9999999f}, c = {1e-9f, 2e-9f};
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int i = 0; i < CHAINS; i++) x[i] = __builtin_elementwise_fma(x[i], b, c);   // v_pk_fma_f32
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  f2 s = {0, 0};
#pragma unroll
  for (int i = 0; i < CHAINS; i++) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x % 64 == 0) cycles[threadIdx.x / 64] = t1 - t0;
}

static double report(const char *name, long long *cyc, int threads, int chains, int extraOps) {
  hipDeviceSynchronize();
  long long h[64];
  hipMemcpy(h, cyc, (threads / 64) * sizeof(long long), hipMemcpyDeviceToHost);
  long long mx = 0;
  for (int i = 0; i < threads / 64; i++) mx = h[i] > mx ? h[i] : mx;
  const int wavesPerSimd = threads / 256;
  const double perInstr = (double)mx / ((double)ITER * chains * wavesPerSimd);
  printf("%-34s waves/SIMD %d chains %2d : %.2f memtime-ticks per wave-op group (%d op(s) each)\n", name, wavesPerSimd, chains, perInstr, 1 + extraOps);
  return perInstr;
}

template <int OP, int CHAINS>
void run(const char *name, int wavesPerSimd, int extraOps = 0) {
  float *out;
  long long *cyc;
  const int threads = 64 * 4 * wavesPerSimd;
  hipMalloc(&out, threads * sizeof(float));
  hipMalloc(&cyc, 64 * sizeof(long long));
  hipLaunchKernelGGL((bench<OP, CHAINS>), dim3(1), dim3(threads), 0, 0, out, cyc, 1.5f);
  report(name, cyc, threads, CHAINS, extraOps);
  hipFree(out);
  hipFree(cyc);
}
template <int CHAINS>
void run_pk(int wavesPerSimd) {
  f2 *out;
  long long *cyc;
  const int threads = 64 * 4 * wavesPerSimd;
  hipMalloc(&out, threads * sizeof(f2));
  hipMalloc(&cyc, 64 * sizeof(long long));
  hipLaunchKernelGGL((bench_pk<CHAINS>), dim3(1), dim3(threads), 0, 0, out, cyc, 1.5f);
  report("v_pk_fma_f32", cyc, threads, CHAINS, 0);
  hipFree(out);
  hipFree(cyc);
}

// tick rate of s_memtime: a kernel that spins for a fixed number of ticks, timed by HIP events
__global__ void spin_ticks(long long n, long long *out) {
  const long long t0 = __builtin_amdgcn_s_memtime();
  long long t = t0;
  while (t - t0 < n) { __builtin_amdgcn_s_sleep(8); t = __builtin_amdgcn_s_memtime(); }
  out[0] = t - t0;
}
static void calibrate() {
  long long *d;
  hipMalloc(&d, 8);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int rep = 0; rep < 2; rep++) {
    hipEventRecord(a);
    hipLaunchKernelGGL(spin_ticks, dim3(1), dim3(64), 0, 0, 100000000LL, d);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    long long h = 0; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("s_memtime: %lld ticks in %.3f ms -> %.1f MHz\n", h, ms, h / (ms * 1e3));
  }
  hipFree(d);
}

int main() {
  calibrate();
  for (int w : {1, 2, 4}) {
    if (w == 1) { run<0, 16>("v_fma_f32", 1); run<3, 16>("v_add_f32", 1); run<1, 16>("v_log_f32 + add", 1, 1); run<2, 16>("v_rcp_f32 + add", 1, 1); run<4, 16>("v_med3_f32 + add", 1, 1); run<5, 16>("mul + v_exp_f32", 1, 1); run<6, 16>("v_sqrt_f32 + add", 1, 1); run_pk<16>(1); }
    if (w == 2) { run<0, 16>("v_fma_f32", 2); run<1, 16>("v_log_f32 + add", 2, 1); run<2, 16>("v_rcp_f32 + add", 2, 1); run_pk<16>(2); }
    if (w == 4) { run<0, 8>("v_fma_f32", 4); run<1, 8>("v_log_f32 + add", 4, 1); run<2, 8>("v_rcp_f32 + add", 4, 1); run_pk<8>(4); }
  }
  return 0;
}
