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
                                                             int64_t n, int64_t outBase, SelectResult *out) {
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
  }
}

// std::upper_bound: first element strictly greater than v
__device__ __forceinline__ int64_t upper_bound_d(const double *a, int64_t n, double v) {
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    const int64_t mid = lo + ((hi - lo) >> 1);
    if (!(v < a[mid])) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// SRPoolRunner::CalcSplit bound i (reference: SRPlatform/Interface/SRPoolRunner.h:96-110) in closed form:
// the first `rem` subtasks get quot+1 items.
__device__ __forceinline__ int64_t split_bound(int64_t i, int64_t quot, int64_t rem) {  // end of subtask i
  const int64_t n1 = (i + 1 < rem) ? (i + 1) : rem;
  return (i + 1) * quot + n1;
}

__global__ __launch_bounds__(1024) void select_sampled_kernel(const double *__restrict__ priority,
                                                              const uint32_t *__restrict__ qgap,
                                                              const uint32_t *__restrict__ asked, int64_t qFirst,
                                                              int64_t n, int64_t nWorkers, uint64_t rnd,
                                                              double *__restrict__ runLength, SelectResult *out, uint64_t *flag, uint64_t flagValue) {
  extern __shared__ double grand[];  // nSubtasks doubles
  const int64_t quot = n / nWorkers, rem = n % nWorkers;
  const int64_t nSubtasks = (quot == 0) ? rem : nWorkers;  // CalcSplit stops once the items run out
  // per-subtask inclusive Kahan running sums (PqaCore/CEEvalQsSubtaskConsider.cpp:52,212-214)
  for (int64_t s = threadIdx.x; s < nSubtasks; s += blockDim.x) {
    const int64_t first = (s == 0) ? 0 : split_bound(s - 1, quot, rem), limit = split_bound(s, quot, rem);
    Kahan1 acc;
    acc.init(0.0);
    for (int64_t i = first; i < limit; i++) {
      // gap / asked questions only copy the running sum (:54-58); evaluated ones are Kahan-added (:212)
      if (!(bit_test(qgap, qFirst + i) || bit_test(asked, qFirst + i))) acc.add(priority[i]);
      runLength[i] = acc.get();
    }
    grand[s] = acc.get();
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    Kahan1 accTotG;
    accTotG.init(0.0);                                         // PqaCore/CpuEngine.cpp:362
    for (int64_t s = 0; s < nSubtasks; s++) {
      accTotG.add(grand[s]);                                   // :366-367
      grand[s] = accTotG.get();                                // :368
    }
    const double totG = grand[nSubtasks - 1];                  // :375
    // SRDoubleNumber::MakeRandom (SRPlatform/Interface/SRDoubleNumber.h:35-39)
    const double selRunLen = totG * (double)rnd / 18446744073709551615.0;  // :379
    int64_t sel;
    const int64_t iWorker = upper_bound_d(grand, nSubtasks, selRunLen);    // :380-381
    if (iWorker >= nSubtasks) {
      sel = n - 1;                                             // :384
    } else {
      const double inWorkerRunLen = selRunLen - ((iWorker == 0) ? 0.0 : grand[iWorker - 1]);  // :388
      const int64_t first = (iWorker == 0) ? 0 : split_bound(iWorker - 1, quot, rem);         // :389
      const int64_t limit = split_bound(iWorker, quot, rem);                                   // :390
      sel = first + upper_bound_d(runLength + first, limit - first, inWorkerRunLen);           // :391
      if (sel >= limit) sel = limit - 1;                       // :392-400
    }
    out->priority = totG;
    out->index = sel;
    if (flag != nullptr) {  // `out` and `flag` in host-coherent memory: the host polls instead of copying + synchronising
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
      __hip_atomic_store(flag, flagValue, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

}  // namespace

hipError_t LaunchSelectArgmax(const double *priority, const uint32_t *qgap, const uint32_t *asked, int64_t qFirst,
                              int64_t n, int64_t outBase, SelectResult *out, hipStream_t stream) {
  const unsigned threads = n >= 1024 ? 1024 : (unsigned)(((n + 63) / 64) * 64 ? ((n + 63) / 64) * 64 : 64);
  hipLaunchKernelGGL(select_argmax_kernel, dim3(1), dim3(threads), 0, stream, priority, qgap, asked, qFirst, n,
                     outBase, out);
  return hipGetLastError();
}

hipError_t LaunchSelectSampled(const double *priority, const uint32_t *qgap, const uint32_t *asked, int64_t qFirst,
                               int64_t n, int64_t nSubtasks, uint64_t rnd, double *runLength, SelectResult *out,
                               uint64_t *flag, uint64_t flagValue, hipStream_t stream) {
  if (n <= 0 || nSubtasks <= 0) return hipErrorInvalidValue;
  const size_t shmem = (size_t)nSubtasks * sizeof(double);
  if (shmem > 64 * 1024) return hipErrorInvalidValue;
  hipLaunchKernelGGL(select_sampled_kernel, dim3(1), dim3(1024), shmem, stream, priority, qgap, asked, qFirst, n, nSubtasks,
                     rnd, runLength, out, flag, flagValue);
  return hipGetLastError();
}

}  // namespace pqa
