// prior_device.h -- the posterior update of one answered question as a device function (RecordAnswer: reference
// PqaCore/CERecordAnswerSubtaskMul.cpp:15-42, PqaCore/Summator.h:11-21, PqaCore/CEDivTargPriorsSubtask.h:12-30), shared by
// the one-quiz and the batched kernels of prior_kernels.hip.  Bit-identical to the CPU engine: see prior_kernels.hip for the
// summation order that is reproduced here.
#pragma once
#include "pqa_device.h"

namespace pqa {

__device__ __forceinline__ int64_t prior_split_bound(int64_t i, int64_t quot, int64_t rem) {  // end of subtask i
  const int64_t n1 = (i + 1 < rem) ? (i + 1) : rem;
  return (i + 1) * quot + n1;
}

// Sum v[0 .. 4*nVects) exactly as the CPU engine does and return the total to every thread.
// lds: 8*nSubtasks + 1 doubles.  Must be called by all threads; contains barriers.
// AHEAD: request sixteen elements of a chain at once (long rows, values in global memory); the 256-thread forms that run beside
// the resident sweep have their values in LDS and must stay within the registers it leaves them.
template <bool AHEAD = true>
__device__ double reference_order_sum(const double *__restrict__ v, int64_t nVects, int64_t nWorkers, double *lds) {
  const int64_t quot = nVects / nWorkers, rem = nVects % nWorkers;
  const int64_t nSubtasks = (quot == 0) ? rem : nWorkers;
  __syncthreads();  // v was written by other threads of this workgroup
  for (int64_t ch = threadIdx.x; ch < nSubtasks * 4; ch += blockDim.x) {
    const int64_t s = ch >> 2;
    const int c = (int)(ch & 3);
    const int64_t first = (s == 0) ? 0 : prior_split_bound(s - 1, quot, rem), limit = prior_split_bound(s, quot, rem);
    double sum = 0, corr = 0;  // SRAccumVectDbl256::Add, SRPlatform/Interface/SRAccumVectDbl256.h:40-46
    // The chain's additions depend on each other, its loads do not: sixteen elements are requested at once and then added in
    // order (a chain is T / (4 nWorkers) elements long -- 1667 at 100000 targets and 15 subtasks -- and one L2 round trip per
    // element, taken one after the other, was 0.4 us each: 736 us of StartQuiz, 356 us of RecordAnswer there).
    int64_t j = first;
    if constexpr (AHEAD)
    for (; j + 16 <= limit; j += 16) {
      double x[16];
#pragma unroll
      for (int e = 0; e < 16; e++) x[e] = v[4 * (j + e) + c];
#pragma unroll
      for (int e = 0; e < 16; e++) {
        const double y = x[e] - corr;
        const double t = sum + y;
        corr = (t - sum) - y;
        sum = t;
      }
    }
    for (; j < limit; j++) {
      const double y = v[4 * j + c] - corr;
      const double t = sum + y;
      corr = (t - sum) - y;
      sum = t;
    }
    lds[8 * s + c] = sum;
    lds[8 * s + 4 + c] = corr;
  }
  __syncthreads();
  // PreciseSum of each subtask's four lanes in parallel (one subtask per thread; the result replaces the subtask's first
  // slot), then the serial Kahan over the subtasks in order: the same operations in the same order as one thread doing
  // both, 1.4 us sooner at 16 workers
  for (int64_t s2 = threadIdx.x; s2 < nSubtasks; s2 += blockDim.x) {
    const double ps = precise_sum4(lds + 8 * s2, lds + 8 * s2 + 4);
    lds[8 * s2] = ps;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    Kahan1 acc;  // Summator::ForPriors, PqaCore/Summator.h:14-19
    acc.init(0.0);
    int64_t s2 = 0;
    for (; s2 + 8 <= nSubtasks; s2 += 8) {   // (eight values requested at once, added in order)
      double x[8];
#pragma unroll
      for (int e = 0; e < 8; e++) x[e] = lds[8 * (s2 + e)];
#pragma unroll
      for (int e = 0; e < 8; e++) acc.add(x[e]);
    }
    for (; s2 < nSubtasks; s2++) acc.add(lds[8 * s2]);
    lds[8 * nSubtasks] = acc.get();
  }
  __syncthreads();
  return lds[8 * nSubtasks];
}

struct PriorArgs {
  const void *cube;
  int elem;
  const double *vB;
  const uint32_t *tgap;
  double *prior;
  int64_t K, T, ldT, nWorkers;
  // 1: the un-normalised values are staged in LDS (ldT doubles behind the summation's 8 * nWorkers + 1) instead of in `prior`:
  // the chains of the reference-order sum walk their elements one dependent read after the other, ~0.3 us each from L2 (17 per
  // chain at 1000 targets and 16 workers: 5 of the kernel's 12.7 us), ~0.03 us from LDS
  int stage;
};
__device__ __forceinline__ double *prior_stage(const PriorArgs &a, double *lds) { return a.stage ? lds + 8 * a.nWorkers + 1 : a.prior; }

// `top` (optional): the call that follows RecordAnswer in every quiz loop is ListTopTargets (PqaClient.cpp:185, the website,
// DichotomyTest.cpp:91), and a dependent launch costs ~8 us of dispatch whatever its size -- so the new posterior's top
// targets are listed here, into host-coherent memory, and ListTopTargets finds them waiting.
struct TopRequest {
  TopOut *out;
  int64_t *nOut;
  uint64_t *flag;
  uint64_t flagValue;
  int64_t count;       // 0: no listing
};

// COH: the caller is the resident kernel -- the old posterior may have been written by a kernel on another XCD (read past the
// non-coherent cache levels), and the new one must reach memory before the sweep's workgroups on other XCDs read it.
// rowsElsewhere (optional): the answered question belongs to ANOTHER shard of the question axis (sharded_engine.cpp) -- its rows
// sA[q][a][.] and mD[q][.] are read where they are (that shard's cube over peer access, or a staged copy) and there is no bit of
// this shard's bitmap to set; every shard then computes the posterior itself, the same bits everywhere.
struct RowPair { const void *a, *d; };
template <bool SMALL, bool COH>
__device__ __forceinline__ void record_answer_body(PriorArgs a, int64_t iQuestion, int64_t iAnswer, uint32_t *asked, TopRequest top,
                                                   double *lds, TopScratch *topScratch, RowPair rowsElsewhere = RowPair{nullptr, nullptr}) {
  // CEQuiz::RecordAnswer marks the question as asked (PqaCore/CEQuiz.h:92); done here, in stream order with the sweeps
  // that read the bitmap, so that the host call needs neither a copy nor a synchronisation
  if (threadIdx.x == 0 && rowsElsewhere.a == nullptr) {
    const uint32_t w = COH ? __hip_atomic_load(asked + (iQuestion >> 5), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : asked[iQuestion >> 5];
    asked[iQuestion >> 5] = w | (1u << (iQuestion & 31));
  }
  const int64_t nVects = (a.T + 3) >> 2;
  int64_t rowA = (iQuestion * (a.K + 1) + iAnswer) * a.ldT;  // CERecordAnswerSubtaskMul.cpp:25
  int64_t rowD = (iQuestion * (a.K + 1) + a.K) * a.ldT;      // :26
  if (rowsElsewhere.a != nullptr) {   // (wave-uniform; offsets in elements from the A row, which stands in for the cube's base)
    a.cube = rowsElsewhere.a;
    rowA = 0;
    rowD = (int64_t)((static_cast<const char *>(rowsElsewhere.d) - static_cast<const char *>(rowsElsewhere.a)) / a.elem);
  }
  double *stage = prior_stage(a, lds);
  // (four elements per thread and round: their twelve loads are requested together -- a long row is ~100 rounds of one L2 round
  //  trip each otherwise; the element arithmetic and its order per element are unchanged)
  const int64_t step = blockDim.x;
  int64_t t = threadIdx.x;
  if constexpr (!SMALL)
  for (; t + 3 * step < a.ldT; t += 4 * step) {
    double av[4], dv[4], old[4];
#pragma unroll
    for (int e = 0; e < 4; e++) {
      av[e] = cube_ld(a.cube, a.elem, rowA + t + e * step);
      dv[e] = cube_ld(a.cube, a.elem, rowD + t + e * step);
      old[e] = COH ? __hip_atomic_load(a.prior + t + e * step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : a.prior[t + e * step];
    }
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const double product = old[e] * (av[e] / dv[e]);         // :31, :34
      stage[t + e * step] = bit_test(a.tgap, t + e * step) ? 0.0 : product;   // :35-37
    }
  }
  for (; t < a.ldT; t += step) {
    const double pQaGivenT = cube_ld(a.cube, a.elem, rowA + t) / cube_ld(a.cube, a.elem, rowD + t);   // :31
    const double old = COH ? __hip_atomic_load(a.prior + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : a.prior[t];
    const double product = old * pQaGivenT;                    // :34
    stage[t] = bit_test(a.tgap, t) ? 0.0 : product;            // :35-37
  }
  const double total = reference_order_sum<!SMALL>(stage, nVects, a.nWorkers, lds);
  t = threadIdx.x;
  if constexpr (!SMALL)
  for (; t + 3 * step < a.ldT; t += 4 * step) {
    double x[4];
#pragma unroll
    for (int e = 0; e < 4; e++) x[e] = stage[t + e * step];
#pragma unroll
    for (int e = 0; e < 4; e++) a.prior[t + e * step] = t + e * step < 4 * nVects ? x[e] / total : x[e];
  }
  for (; t < a.ldT; t += step) a.prior[t] = t < 4 * nVects ? stage[t] / total : stage[t];
  if constexpr (COH) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // posterior and asked bit out of this XCD's L2
  if (top.count > 0) {
    __syncthreads();  // (a thread lists exactly the targets it has just written; the barrier is for the shared LDS rows)
    top_targets_publish<SMALL>(a.prior, a.tgap, a.T, top.count, top.out, top.nOut, top.flag, top.flagValue, topScratch);
  }
}

}  // namespace pqa
