// xcc_test.hip -- which XCD does workgroup i of a 1-D grid run on (HW_REG_XCC_ID), and what does one L2 invalidate cost?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(unsigned *out, unsigned long long *cost) {
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  if (threadIdx.x == 0) out[blockIdx.x] = xcc;
  if (blockIdx.x < 8 && threadIdx.x < 64) {
    const unsigned long long t0 = wall_clock64();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    const unsigned long long t1 = wall_clock64();
    asm volatile("buffer_inv sc0" ::: "memory");
    const unsigned long long t2 = wall_clock64();
    if (threadIdx.x == 0) { cost[2 * blockIdx.x] = t1 - t0; cost[2 * blockIdx.x + 1] = t2 - t1; }
  }
}
int main() {
  const int grid = 768;
  unsigned *d; unsigned long long *c;
  hipMalloc(&d, grid * 4); hipMalloc(&c, 16 * 8);
  for (int rep = 0; rep < 2; rep++) {
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, d, c);
    hipDeviceSynchronize();
  }
  std::vector<unsigned> h(grid); std::vector<unsigned long long> hc(16);
  hipMemcpy(h.data(), d, grid * 4, hipMemcpyDeviceToHost); hipMemcpy(hc.data(), c, 16 * 8, hipMemcpyDeviceToHost);
  int mism = 0;
  for (int i = 0; i < grid; i++) if ((h[i] & 0xF) != (unsigned)(i % 8)) mism++;
  printf("raw xcc of blocks 0..15:"); for (int i = 0; i < 16; i++) printf(" %x", h[i]); printf("\n");
  printf("blocks whose XCC_ID[3:0] != block %% 8: %d of %d\n", mism, grid);
  for (int i = 0; i < 8; i++) printf("block %d: buffer_inv sc1 %llu ticks (10 ns), buffer_inv sc0 %llu\n", i, hc[2 * i], hc[2 * i + 1]);
  return 0;
}
