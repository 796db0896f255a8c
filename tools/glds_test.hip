// glds_test.hip -- checks the LDS-DMA (global_load_lds_dwordx4) addressing used by the sweep's landing ring on gfx950:
// destination = M0 (wave-uniform LDS byte address, beyond 64 KiB too) + lane*16, in-order completion under vmcnt(N).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/glds_test tools/glds_test.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ void glds16(const void *gsrc, unsigned ldsDst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(ldsDst) : "memory");
}

// 512 threads, `slots` chunks of 8 KiB each; chunk c of the input goes to slot c, wave w's KiB at +w*1024
__global__ __launch_bounds__(512) void k(const double2 *in, double2 *out, int slots) {
  extern __shared__ double2 ring[];
  const int tid = threadIdx.x, wave = tid / 64;
  const unsigned base = (unsigned)(uintptr_t)ring;
  for (int c = 0; c < slots; c++) {
    const unsigned dst = __builtin_amdgcn_readfirstlane(base + c * 8192 + wave * 1024);
    glds16(in + (size_t)c * 512 + tid, dst);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (int c = 0; c < slots; c++) out[(size_t)c * 512 + tid] = ring[c * 512 + tid];
}

// Streaming ring: nChunks chunks of 8 KiB flow through C slots with the sweep's protocol -- wait vmcnt(C-1), read the
// slot, re-arm it with chunk s+C.  SAFE additionally waits for the ds_read to return before re-arming.
template <int C, bool SAFE>
__global__ __launch_bounds__(512) void ring_k(const double2 *in, double *out, int nChunks, size_t strideChunks) {
  extern __shared__ double2 ring[];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid / 64);
  const unsigned base = (unsigned)(uintptr_t)ring + wave * 1024;
  const double2 *mine = ring + tid;
  const double2 *src = in + (size_t)blockIdx.x * nChunks * strideChunks * 512 + tid;
  for (int c = 0; c < C; c++) glds16(src + (size_t)c * strideChunks * 512, base + c * 8192);
  double sx = 0, sy = 0;
  int slot = 0;
  for (int s = 0; s < nChunks; s++) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C - 1) : "memory");
    const double2 v = mine[slot * 512];
    if (SAFE) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int nx = s + C < nChunks ? s + C : nChunks - 1;   // tail: harmless re-reads
    glds16(src + (size_t)nx * strideChunks * 512, base + slot * 8192);
    sx += v.x * (double)(s + 1);
    sy += v.y;
    slot = slot + 1 == C ? 0 : slot + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  out[((size_t)blockIdx.x * 512 + tid) * 2] = sx;
  out[((size_t)blockIdx.x * 512 + tid) * 2 + 1] = sy;
}

template <int C, bool SAFE>
int run_ring(int nBlocks, int nChunks, size_t strideChunks) {
  const size_t n = (size_t)nBlocks * nChunks * strideChunks * 512;
  std::vector<double2> h(n);
  for (size_t i = 0; i < n; i++) h[i] = make_double2((double)(i % 1000003), (double)(i % 7919) * 0.25);
  double2 *din;
  double *dout;
  hipMalloc(&din, n * sizeof(double2));
  hipMalloc(&dout, (size_t)nBlocks * 1024 * sizeof(double));
  hipMemcpy(din, h.data(), n * sizeof(double2), hipMemcpyHostToDevice);
  const size_t shmem = (size_t)C * 8192;
  auto kern = ring_k<C, SAFE>;
  hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
  size_t bad = 0;
  for (int rep = 0; rep < 5; rep++) {
    hipLaunchKernelGGL(kern, dim3(nBlocks), dim3(512), shmem, 0, din, dout, nChunks, strideChunks);
    if (hipDeviceSynchronize() != hipSuccess) { printf("ring launch failed\n"); return 2; }
    std::vector<double> r((size_t)nBlocks * 1024);
    hipMemcpy(r.data(), dout, r.size() * sizeof(double), hipMemcpyDeviceToHost);
    for (int b = 0; b < nBlocks; b++)
      for (int t = 0; t < 512; t++) {
        double sx = 0, sy = 0;
        for (int s2 = 0; s2 < nChunks; s2++) {
          const double2 v = h[((size_t)b * nChunks + s2) * strideChunks * 512 + t];
          sx += v.x * (double)(s2 + 1);
          sy += v.y;
        }
        if (r[((size_t)b * 512 + t) * 2] != sx || r[((size_t)b * 512 + t) * 2 + 1] != sy) bad++;
      }
  }
  printf("ring C=%d safe=%d blocks=%d chunks=%d stride=%zu: %zu bad lanes\n", C, (int)SAFE, nBlocks, nChunks, strideChunks, bad);
  hipFree(din);
  hipFree(dout);
  return bad ? 1 : 0;
}

int main() {
  int rc = 0;
  rc |= run_ring<18, false>(256, 200, 1);
  rc |= run_ring<18, true>(256, 200, 1);
  rc |= run_ring<18, false>(64, 100, 7);
  rc |= run_ring<4, false>(256, 200, 1);
  rc |= run_ring<4, true>(256, 200, 1);
  const int slots = 19;
  const size_t n = (size_t)slots * 512;
  std::vector<double2> h(n), r(n);
  for (size_t i = 0; i < n; i++) h[i] = make_double2((double)i, -(double)i - 0.5);
  double2 *din, *dout;
  hipMalloc(&din, n * sizeof(double2));
  hipMalloc(&dout, n * sizeof(double2));
  hipMemcpy(din, h.data(), n * sizeof(double2), hipMemcpyHostToDevice);
  hipMemset(dout, 0, n * sizeof(double2));
  const size_t shmem = (size_t)slots * 8192;
  if (hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem) != hipSuccess) {
    printf("cannot get %zu B of LDS\n", shmem);
    return 2;
  }
  hipLaunchKernelGGL(k, dim3(4), dim3(512), shmem, 0, din, dout, slots);
  if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(hipGetLastError())); return 2; }
  hipMemcpy(r.data(), dout, n * sizeof(double2), hipMemcpyDeviceToHost);
  size_t bad = 0, firstBad = 0;
  for (size_t i = 0; i < n; i++)
    if (r[i].x != h[i].x || r[i].y != h[i].y) { if (!bad) firstBad = i; bad++; }
  printf("slots=%d (%zu KiB LDS): %zu mismatches%s\n", slots, shmem / 1024, bad, bad ? "" : " -- LDS-DMA addressing OK");
  if (bad) printf("first mismatch at %zu (slot %zu): got %g %g\n", firstBad, firstBad / 512, r[firstBad].x, r[firstBad].y);
  return (bad ? 1 : 0) | rc;
}
