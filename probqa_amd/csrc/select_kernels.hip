// select_kernels.hip -- picking the next question from the priority vector (gfx950).
//   * argmax (the north-star selector): lowest index among the maximal priorities of eligible questions;
//   * sampled (the reference's selector, PqaCore/CpuEngine.cpp:362-400): per-subtask Kahan run lengths,
//     Kahan grand totals, one uniform number, two upper_bounds -- reproduced step for step, so with the same
//     priorities, subtask count and random number it returns the reference's question.
// Both are single-workgroup kernels over Q doubles: latency-bound, nothing to tile.
#include "pqa_device.h"
#include "pqa_kernels.h"

namespace pqa {

namespace {

struct Best {
  double p;
  int64_t i;
};

__device__ __forceinline__ bool better(const Best &a, const Best &b) {  // is a better than b
  if (b.i < 0) return a.i >= 0;
  if (a.i < 0) return false;
  return (a.p > b.p) || (a.p == b.p && a.i < b.i);
}

__global__ __launch_bounds__(1024) void select_argmax_kernel(const double *__restrict__ priority,
                                                             const uint32_t *__restrict__ qgap,
                                                             const uint32_t *__restrict__ asked, int64_t qFirst,
                                                             int64_t n, int64_t outBase, SelectResult *out, uint64_t *flag,
                                                             uint64_t flagValue) {
  __shared__ double sp[16];
  __shared__ int64_t si[16];
  Best b = {0.0, -1};
  for (int64_t j = threadIdx.x; j < n; j += blockDim.x) {
    const int64_t q = qFirst + j;
    if (bit_test(qgap, q) || bit_test(asked, q)) continue;
    double p = priority[j];
    if (p != p) p = -__builtin_huge_val();  // NaN never wins over a number
    const Best c = {p, q};
    if (better(c, b)) b = c;
  }
#pragma unroll
  for (int m = kWave / 2; m >= 1; m >>= 1) {
    Best o;
    o.p = __shfl_xor(b.p, m, kWave);
    o.i = __shfl_xor(b.i, m, kWave);
    if (better(o, b)) b = o;
  }
  const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
  if (lane == 0) {
    sp[wave] = b.p;
    si[wave] = b.i;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    Best r = {sp[0], si[0]};
    const int nw = (blockDim.x + kWave - 1) / kWave;
    for (int w = 1; w < nw; w++) {
      const Best c = {sp[w], si[w]};
      if (better(c, r)) r = c;
    }
    out->priority = r.p;
    out->index = r.i < 0 ? -1 : r.i - qFirst + outBase;
    if (flag != nullptr) {  // `out` and `flag` in host-coherent memory: the host polls instead of copying + synchronising
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
      __hip_atomic_store(flag, flagValue, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

__global__ __launch_bounds__(1024) void select_sampled_kernel(const double *__restrict__ priority,
                                                              const uint32_t *__restrict__ qgap,
                                                              const uint32_t *__restrict__ asked, int64_t qFirst,
                                                              int64_t n, int64_t nWorkers, uint64_t rnd,
                                                              double *__restrict__ runLength, SelectResult *out, uint64_t *flag, uint64_t flagValue) {
  extern __shared__ double grand[];  // nSubtasks doubles
  const SampledPick r = select_sampled_wg_impl<false>(priority, qgap, asked, qFirst, n, nWorkers, rnd, runLength, grand);
  if (threadIdx.x == 0) {
    out->priority = r.priority;
    out->index = r.index;
    if (flag != nullptr) {  // `out` and `flag` in host-coherent memory: the host polls instead of copying + synchronising
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
      __hip_atomic_store(flag, flagValue, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// The same selection with priorities, run lengths and totals staged in LDS (select_sampled_wg_lds): every global load of the
// kernel is issued in one parallel round instead of one per item of a subtask's dependent chain.  For question counts whose
// staging fits 64 KiB.
__global__ __launch_bounds__(256) void select_sampled_lds_kernel(const double *__restrict__ priority,
                                                                 const uint32_t *__restrict__ qgap,
                                                                 const uint32_t *__restrict__ asked, int64_t qFirst, int64_t n,
                                                                 int64_t nWorkers, uint64_t rnd, SelectResult *out, uint64_t *flag,
                                                                 uint64_t flagValue) {
  extern __shared__ double lds[];
  const SampledPick r = select_sampled_wg_lds<false>(priority, qgap, asked, qFirst, n, nWorkers, rnd, lds);
  if (threadIdx.x == 0) {
    out->priority = r.priority;
    out->index = r.index;
    if (flag != nullptr) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
      __hip_atomic_store(flag, flagValue, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

}  // namespace

hipError_t LaunchSelectArgmax(const double *priority, const uint32_t *qgap, const uint32_t *asked, int64_t qFirst,
                              int64_t n, int64_t outBase, SelectResult *out, uint64_t *flag, uint64_t flagValue,
                              hipStream_t stream) {
  const unsigned threads = n >= 1024 ? 1024 : (unsigned)(((n + 63) / 64) * 64 ? ((n + 63) / 64) * 64 : 64);
  hipLaunchKernelGGL(select_argmax_kernel, dim3(1), dim3(threads), 0, stream, priority, qgap, asked, qFirst, n,
                     outBase, out, flag, flagValue);
  return hipGetLastError();
}

hipError_t LaunchSelectSampled(const double *priority, const uint32_t *qgap, const uint32_t *asked, int64_t qFirst,
                               int64_t n, int64_t nSubtasks, uint64_t rnd, double *runLength, SelectResult *out,
                               uint64_t *flag, uint64_t flagValue, hipStream_t stream) {
  if (n <= 0 || nSubtasks <= 0) return hipErrorInvalidValue;
  const size_t staged = (size_t)select_sampled_lds_doubles(n, nSubtasks) * sizeof(double);
  if (staged <= 64 * 1024) {
    hipLaunchKernelGGL(select_sampled_lds_kernel, dim3(1), dim3(256), staged, stream, priority, qgap, asked, qFirst, n, nSubtasks,
                       rnd, out, flag, flagValue);
    return hipGetLastError();
  }
  const size_t shmem = (size_t)nSubtasks * sizeof(double);
  if (shmem > 64 * 1024) return hipErrorInvalidValue;
  hipLaunchKernelGGL(select_sampled_kernel, dim3(1), dim3(1024), shmem, stream, priority, qgap, asked, qFirst, n, nSubtasks,
                     rnd, runLength, out, flag, flagValue);
  return hipGetLastError();
}

}  // namespace pqa
