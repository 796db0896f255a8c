// fetch_calib.hip -- calibrates rocprofv3's FETCH_SIZE for the access widths the batched sweep uses (MI355X_MICROARCH.md: the x2
// correction is calibrated for 16 B/lane streaming reads only).  Each kernel streams the same 4 GiB buffer once:
//   read4  : 4 B per lane per load (dword, coalesced 256 B per wave)  -- cube staging and prior loads of eval_batch_kernel
//   read16 : 16 B per lane per load (dwordx4)                          -- the single-quiz sweep's row loads
// Build + run:  hipcc --offload-arch=gfx950 -O3 tools/fetch_calib.hip -o tools/fetch_calib
//               rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -o p -- tools/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void read4(const float *p, size_t n, float *out) {
  float s = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += p[i];
  if (s == 12345.678f) out[0] = s;
}
__global__ void read16(const float4 *p, size_t n, float *out) {
  float s = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float4 v = p[i];
    s += v.x + v.y + v.z + v.w;
  }
  if (s == 12345.678f) out[0] = s;
}

int main() {
  const size_t bytes = 4ull << 30;
  float *buf, *out;
  hipMalloc(&buf, bytes);
  hipMalloc(&out, 4);
  hipMemset(buf, 0, bytes);
  hipDeviceSynchronize();
  for (int rep = 0; rep < 3; rep++) {
    hipLaunchKernelGGL(read4, dim3(256 * 8), dim3(256), 0, 0, buf, bytes / 4, out);
    hipLaunchKernelGGL(read16, dim3(256 * 8), dim3(256), 0, 0, (const float4 *)buf, bytes / 16, out);
  }
  hipDeviceSynchronize();
  printf("streamed %zu bytes per kernel launch\n", bytes);
  return 0;
}
