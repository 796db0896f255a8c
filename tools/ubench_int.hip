// ubench_int.hip -- issue cost of 32-bit VALU ops (wave64) on gfx950, same method as ubench_fp64.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 4000
template <int OP>
__device__ __forceinline__ unsigned op(unsigned a, unsigned b) {
  unsigned r;
  if constexpr (OP == 0) asm volatile("v_and_b32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  if constexpr (OP == 1) asm volatile("v_lshrrev_b32 %0, 3, %1" : "=v"(r) : "v"(a));
  if constexpr (OP == 2) asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(a));
  if constexpr (OP == 3) asm volatile("v_add_u32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  if constexpr (OP == 4) asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(b));
  if constexpr (OP == 5) asm volatile("v_bfe_u32 %0, %1, 10, 10" : "=v"(r) : "v"(a));
  if constexpr (OP == 6) asm volatile("v_fma_f32 %0, %1, %2, %2" : "=v"(r) : "v"(a), "v"(b));
  if constexpr (OP == 7) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(r) : "v"(a), "v"(b));
  if constexpr (OP == 8) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(a));
  if constexpr (OP == 9) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(*(double*)nullptr) : "v"(a));
  return r;
}
template <int OP, int CH>
__global__ void bench(unsigned *out, long long *cyc) {
  unsigned x[CH];
#pragma unroll
  for (int i = 0; i < CH; i++) x[i] = threadIdx.x + i;
  const unsigned b = 0x12345u + threadIdx.x;
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int i = 0; i < CH; i++) x[i] = op<OP>(x[i], b);
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  unsigned s = 0;
#pragma unroll
  for (int i = 0; i < CH; i++) s += x[i];
  out[threadIdx.x] = s;
  if (threadIdx.x % 64 == 0) cyc[threadIdx.x / 64] = t1 - t0;
}
template <int OP>
void run(const char *name, int wps) {
  unsigned *out; long long *cyc;
  const int threads = 256 * wps;
  hipMalloc(&out, threads * 4); hipMalloc(&cyc, 64 * 8);
  hipLaunchKernelGGL((bench<OP, 8>), dim3(1), dim3(threads), 0, 0, out, cyc);
  hipDeviceSynchronize();
  long long h[64]; hipMemcpy(h, cyc, (threads / 64) * 8, hipMemcpyDeviceToHost);
  long long mx = 0; for (int i = 0; i < threads / 64; i++) mx = h[i] > mx ? h[i] : mx;
  printf("%-16s waves/SIMD=%d cycles/instr (SIMD-serialised)=%.3f\n", name, wps, (double)mx / ((double)ITER * 8 * wps));
  hipFree(out); hipFree(cyc);
}
int main() {
  for (int w : {1, 4}) {
    run<0>("v_and_b32", w); run<1>("v_lshrrev_b32", w); run<2>("v_mov_b32", w); run<3>("v_add_u32", w);
    run<4>("v_and_or_b32", w); run<5>("v_bfe_u32", w); run<6>("v_fma_f32", w); run<7>("v_cndmask_b32", w);
    run<8>("v_mov_b32_dpp", w);
  }
  return 0;
}
