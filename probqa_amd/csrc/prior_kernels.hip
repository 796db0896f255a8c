// prior_kernels.hip -- Bayes prior / posterior updates of a quiz on CDNA4 (gfx950).
//
//   StartQuiz    : CESetPriorsSubtaskSum (reference: PqaCore/CESetPriorsSubtaskSum.cpp:17-40)
//   RecordAnswer : CERecordAnswerSubtaskMul (PqaCore/CERecordAnswerSubtaskMul.cpp:15-42)
//   ResumeQuiz   : CEUpdatePriorsSubtaskMul (PqaCore/CEUpdatePriorsSubtaskMul.cpp:16-114) +
//                  CpuEngine::NormalizePriors (PqaCore/CpuEngine.cpp:284-335) with CENormPriorsSubtaskMax /
//                  CENormPriorsSubtaskCorrSum
//   all followed by Summator::ForPriors (PqaCore/Summator.h:11-21) and CEDivTargPriorsSubtask
//   (PqaCore/CEDivTargPriorsSubtask.h:12-30).
//
// These are O(T) vector updates (8-80 KB): one workgroup, latency-bound, 0.04 % of the reference's profile.  They are
// written to be BIT-IDENTICAL to the CPU engine instead of fast: element-wise steps use IEEE division / multiplication
// and the exponent surgery of the reference, and the normalising sum reproduces the reference's summation order --
// `nWorkers` subtasks over ceil(T/4) AVX vectors (SRPoolRunner::CalcSplit), 4 Kahan lanes per subtask
// (lane c sums targets == c mod 4), PreciseSum per subtask, serial Kahan over the subtasks.  One GPU lane runs one
// (subtask, AVX-lane) chain.
#include "pqa_device.h"
#include "pqa_kernels.h"
#include "prior_device.h"

namespace pqa {

namespace {

constexpr int kThreads = 1024;
// While the resident sweep (eval_kernels.hip: eval_server_f64) holds three workgroups of 154 (allocated: 160) VGPRs on every CU, a
// 1024-thread workgroup of these kernels fits nowhere and would wait for the sweep to leave.  With <= 1024 targets they are
// launched with 256 threads instead (one wave per SIMD, within the 32 VGPRs that are left: 24 - 30): same arithmetic, same
// summation order, without the load batching of the long-row forms.
constexpr int kSmallThreads = 256;
static inline bool small_launch(const KbView &kb) { return kb.smallLaunches && kb.T <= 4 * kSmallThreads; }

// (clears the new quiz's asked bitmap as well: a memset beside it is a second launch)
// SMALL: the 256-thread form that runs beside the resident sweep (within the 32 registers it leaves per lane)
template <bool SMALL>
__global__ __launch_bounds__(SMALL ? kSmallThreads : kThreads) void start_quiz_kernel(PriorArgs a, uint32_t *__restrict__ asked, int64_t askedWords) {
  extern __shared__ double lds[];
  for (int64_t i = threadIdx.x; i < askedWords; i += blockDim.x) asked[i] = 0;
  const int64_t nVects = (a.T + 3) >> 2;
  double *stage = prior_stage(a, lds);
  for (int64_t t = threadIdx.x; t < a.ldT; t += blockDim.x)
    stage[t] = bit_test(a.tgap, t) ? 0.0 : a.vB[t];            // CESetPriorsSubtaskSum.cpp:28-31
  const double total = reference_order_sum<!SMALL>(stage, nVects, a.nWorkers, lds);
  for (int64_t t = threadIdx.x; t < a.ldT; t += blockDim.x) a.prior[t] = t < 4 * nVects ? stage[t] / total : stage[t];  // CEDivTargPriors :19
}

__global__ __launch_bounds__(kThreads) void start_quiz_batch_kernel(PriorArgs a, StartBatchInline batch) {
  extern __shared__ double lds[];
  uint32_t *asked = batch.asked[blockIdx.x];
  a.prior = batch.prior[blockIdx.x];
  for (int64_t i = threadIdx.x; i < batch.askedWords; i += blockDim.x) asked[i] = 0;
  const int64_t nVects = (a.T + 3) >> 2;
  double *stage = prior_stage(a, lds);
  for (int64_t t = threadIdx.x; t < a.ldT; t += blockDim.x)
    stage[t] = bit_test(a.tgap, t) ? 0.0 : a.vB[t];            // CESetPriorsSubtaskSum.cpp:28-31
  const double total = reference_order_sum(stage, nVects, a.nWorkers, lds);
  for (int64_t t = threadIdx.x; t < a.ldT; t += blockDim.x) a.prior[t] = t < 4 * nVects ? stage[t] / total : stage[t];  // CEDivTargPriors :19
}

template <bool SMALL>
__global__ __launch_bounds__(SMALL ? kSmallThreads : kThreads) void record_answer_kernel(PriorArgs a, int64_t iQuestion, int64_t iAnswer,
                                                                                         uint32_t *asked, TopRequest top, RowPair rowsElsewhere) {
  extern __shared__ double lds[];
  __shared__ TopScratch topScratch;
  record_answer_body<SMALL, false>(a, iQuestion, iAnswer, asked, top, lds, &topScratch, rowsElsewhere);
}

// grid.x = update: workgroup i runs quiz i's RecordAnswer exactly as record_answer_kernel would
template <bool SMALL>
__global__ __launch_bounds__(SMALL ? kSmallThreads : kThreads) void record_answer_batch_kernel(PriorArgs a, RecordBatchInline batch) {
  extern __shared__ double lds[];
  __shared__ TopScratch topScratch;
  const RecordSlot &s = batch.s[blockIdx.x];
  a.prior = s.prior;
  TopOut *topOut = reinterpret_cast<TopOut *>(s.pin);
  int64_t *topN = reinterpret_cast<int64_t *>(topOut + kQuizTopDev);
  const TopRequest top{topOut, topN, reinterpret_cast<uint64_t *>(topN + 1), s.topFlagValue, s.pin ? (int64_t)batch.topCount : 0};
  record_answer_body<SMALL, false>(a, s.iQuestion, s.iAnswer, s.asked, top, lds, &topScratch, RowPair{s.rowA, s.rowD});
}

__device__ __forceinline__ int ceil_log2_u64(uint64_t val) {  // SRPlatform/Interface/SRMath.h:46-51
  if (!val) return 0;
  const int index = 63 - __clzll((long long)val);
  return index + ((val & (val - 1)) ? 1 : 0);
}

// rows: for answered question i the rows sA[q_i][a_i][.] (rows[2i]) and mD[q_i][.] (rows[2i + 1]) -- pointers instead of
// indices, so that the rows of a question held by ANOTHER device of the process (a shard of the question axis,
// sharded_engine.cpp) are read in place over xGMI peer access.
__global__ __launch_bounds__(kThreads) void resume_quiz_kernel(PriorArgs a, int64_t *__restrict__ exps,
                                                               const void *const *__restrict__ rows, int64_t nAnswered,
                                                               int bugCompat, int64_t *status) {
  extern __shared__ double lds[];
  __shared__ long long sMax[kThreads / kWave];
  const int nWaves = (int)(blockDim.x / kWave);
  __shared__ long long sCorr;
  const int64_t nVects = (a.T + 3) >> 2;
  // a8: every target runs its own chain of products over the answered questions, mantissa and exponent kept apart.
  long long myMax = INT64_MIN;
  for (int64_t t = threadIdx.x; t < a.ldT; t += blockDim.x) {
    double mant;
    int64_t ex;
    {
      const double pQaGivenT = cube_ld(rows[0], a.elem, t) / cube_ld(rows[1], a.elem, t);
      const double oldMant = bugCompat ? a.vB[t & 3] : a.vB[t];  // CEUpdatePriorsSubtaskMul.cpp:53 loads pvB, not pvB+j
      const uint64_t up = d2u(oldMant * pQaGivenT);              // :54
      mant = u2d(kExp0Up | (up & ~kExpMaskUp));                  // :56 MakeExponent0
      ex = (int64_t)((up & kExpMaskUp) >> 52);                   // :59 ExtractExponents64<false>
    }
    for (int64_t i = 1; i < nAnswered; i++) {
      const double pQaGivenT = cube_ld(rows[2 * i], a.elem, t) / cube_ld(rows[2 * i + 1], a.elem, t);
      const uint64_t up = d2u(mant * pQaGivenT);                 // :75
      mant = u2d(kExp0Up | (up & ~kExpMaskUp));                  // :77
      ex += (int64_t)((up & kExpMaskUp) >> 52);                  // :80-82
    }
    a.prior[t] = mant;
    exps[t] = ex;
    // a9 part 1: CENormPriorsSubtaskMax (gap lanes retain the running maximum)
    const long long totExp = ex + (long long)((d2u(mant) & kExpMaskUp) >> 52);
    if (t < 4 * nVects && !bit_test(a.tgap, t) && totExp > myMax) myMax = totExp;
  }
#pragma unroll
  for (int m = kWave / 2; m >= 1; m >>= 1) {
    const long long o = __shfl_xor(myMax, m, kWave);
    myMax = o > myMax ? o : myMax;
  }
  if (threadIdx.x % kWave == 0) sMax[threadIdx.x / kWave] = myMax;
  __syncthreads();
  if (threadIdx.x == 0) {
    long long fullMax = sMax[0];
    for (int w = 1; w < nWaves; w++) fullMax = sMax[w] > fullMax ? sMax[w] : fullMax;
    const long long highBound = 1023 + 1023 - ceil_log2_u64((uint64_t)a.T) - 2;  // CpuEngine.cpp:316
    const long long minAllowed = INT64_MIN + highBound + 1;                      // :317
    status[1] = fullMax;
    if (fullMax <= minAllowed) {                                                 // :318-321 I64Underflow
      status[0] = 16;
      sCorr = INT64_MIN;
    } else {
      status[0] = 0;
      sCorr = highBound - fullMax;                                               // :322
    }
  }
  __syncthreads();
  const long long corrExp = sCorr;
  if (corrExp == INT64_MIN) return;  // error: priors left un-normalised, the host destroys the quiz
  // a9 part 2: CENormPriorsSubtaskCorrSum::Process (CENormPriorsSubtaskCorrSum.cpp:25-40)
  for (int64_t t = threadIdx.x; t < a.ldT; t += blockDim.x) {
    const uint64_t um = d2u(a.prior[t]);
    const long long normExp = exps[t] + (long long)((um & kExpMaskUp) >> 52) + corrExp;
    const bool assume0 = (1 > normExp) || bit_test(a.tgap, t);
    a.prior[t] = assume0 ? 0.0 : u2d(((uint64_t)normExp << 52) | (um & ~kExpMaskUp));  // ReplaceExponents
    exps[t] = 0;
  }
  const double total = reference_order_sum(a.prior, nVects, a.nWorkers, lds);
  for (int64_t t = threadIdx.x; t < 4 * nVects; t += blockDim.x) a.prior[t] = a.prior[t] / total;
}

// ---- long rows (ldT > 16384: BASELINE configs[4]'s 100000 targets).  One workgroup per SUBTASK of the reference's sum instead of
// one for the row: the element-wise step of a subtask's ~T / nWorkers targets is one or two rounds of loads for 1024 threads
// (a hundred rounds for one workgroup over the whole row), its four Kahan chains -- the reference's order, so still one lane
// each -- walk values that are in LDS, and the chains of all subtasks run at the same time on different CUs.  The workgroup
// that finishes last adds the subtasks' sums (PreciseSum per subtask, serial Kahan over the subtasks: as reference_order_sum).
// The division by the total is a second launch over the whole row (long_row_divide_kernel): a dependent launch is ~8 us, a
// device-wide wait inside one kernel would need every workgroup resident at once.  Same operations on the same values in the
// same order as the one-workgroup kernels: bit-identical.
// scratch: [8 * kMaxWorkers] partial sums and corrections, then the total, then the arrival counter (a 32-bit word).
template <bool RECORD>
__global__ __launch_bounds__(kThreads) void long_row_stage_kernel(PriorArgs a, int64_t iQuestion, int64_t iAnswer, uint32_t *__restrict__ asked,
                                                                  int64_t askedWords, double *__restrict__ scratch, int valuesInLds) {
  extern __shared__ double lds[];   // 8 * nSubtasks + 1 doubles for the last workgroup's sum, then this subtask's values
  __shared__ int isLast;
  const int64_t nVects = (a.T + 3) >> 2;
  const int64_t quot = nVects / a.nWorkers, rem = nVects % a.nWorkers;
  const int64_t nSubtasks = gridDim.x, s = blockIdx.x;
  const int64_t first = (s == 0) ? 0 : prior_split_bound(s - 1, quot, rem), limit = prior_split_bound(s, quot, rem);
  const int64_t e0 = 4 * first, e1 = (s == nSubtasks - 1) ? a.ldT : 4 * limit;   // (the last subtask's workgroup also takes the row's padding)
  double *vals = lds + 8 * nSubtasks + 1;
  if constexpr (RECORD) {
    if (s == 0 && threadIdx.x == 0 && asked != nullptr) asked[iQuestion >> 5] |= 1u << (iQuestion & 31);   // CEQuiz::RecordAnswer, PqaCore/CEQuiz.h:92
  } else {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < askedWords; i += (int64_t)gridDim.x * blockDim.x) asked[i] = 0;
  }
  const int64_t rowA = RECORD ? (iQuestion * (a.K + 1) + iAnswer) * a.ldT : 0;  // CERecordAnswerSubtaskMul.cpp:25
  const int64_t rowD = RECORD ? (iQuestion * (a.K + 1) + a.K) * a.ldT : 0;      // :26
  const int64_t step = blockDim.x;
  int64_t t = e0 + threadIdx.x;
  for (; t + 3 * step < e1; t += 4 * step) {
    double x[4];
    if constexpr (RECORD) {
      double av[4], dv[4], old[4];
#pragma unroll
      for (int e = 0; e < 4; e++) {
        av[e] = cube_ld(a.cube, a.elem, rowA + t + e * step);
        dv[e] = cube_ld(a.cube, a.elem, rowD + t + e * step);
        old[e] = a.prior[t + e * step];
      }
#pragma unroll
      for (int e = 0; e < 4; e++) x[e] = old[e] * (av[e] / dv[e]);             // :31, :34
    } else {
#pragma unroll
      for (int e = 0; e < 4; e++) x[e] = a.vB[t + e * step];                    // CESetPriorsSubtaskSum.cpp:28-31
    }
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const double v = bit_test(a.tgap, t + e * step) ? 0.0 : x[e];             // CERecordAnswerSubtaskMul.cpp:35-37
      a.prior[t + e * step] = v;
      if (valuesInLds) vals[t + e * step - e0] = v;
    }
  }
  for (; t < e1; t += step) {
    double x;
    if constexpr (RECORD) x = a.prior[t] * (cube_ld(a.cube, a.elem, rowA + t) / cube_ld(a.cube, a.elem, rowD + t));
    else x = a.vB[t];
    const double v = bit_test(a.tgap, t) ? 0.0 : x;
    a.prior[t] = v;
    if (valuesInLds) vals[t - e0] = v;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    // one chain per lane (SRAccumVectDbl256::Add, SRPlatform/Interface/SRAccumVectDbl256.h:40-46), sixteen values requested ahead
    const int c = (int)threadIdx.x;
    // (element t at v[t - off]: an LDS pointer moved below its segment is no LDS pointer any more)
    const double *v = valuesInLds ? vals : a.prior;
    const int64_t off = valuesInLds ? e0 : 0;
    double sum = 0, corr = 0;
    int64_t j = first;
    for (; j + 16 <= limit; j += 16) {
      double x[16];
#pragma unroll
      for (int e = 0; e < 16; e++) x[e] = v[4 * (j + e) + c - off];
#pragma unroll
      for (int e = 0; e < 16; e++) {
        const double y = x[e] - corr;
        const double u = sum + y;
        corr = (u - sum) - y;
        sum = u;
      }
    }
    for (; j < limit; j++) {
      const double y = v[4 * j + c - off] - corr;
      const double u = sum + y;
      corr = (u - sum) - y;
      sum = u;
    }
    __hip_atomic_store(scratch + 8 * s + c, sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(scratch + 8 * s + 4 + c, corr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  unsigned *counter = reinterpret_cast<unsigned *>(scratch + 8 * kMaxWorkers + 1);
  if (threadIdx.x < kWave) {   // (the chains' wave: its release covers the four lanes' stores)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    if (threadIdx.x == 0) isLast = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(nSubtasks - 1);
  }
  __syncthreads();
  if (!isLast) return;
  for (int64_t i = threadIdx.x; i < 8 * nSubtasks; i += blockDim.x) lds[i] = __hip_atomic_load(scratch + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  for (int64_t s2 = threadIdx.x; s2 < nSubtasks; s2 += blockDim.x) {
    const double ps = precise_sum4(lds + 8 * s2, lds + 8 * s2 + 4);
    lds[8 * s2] = ps;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    Kahan1 acc;  // Summator::ForPriors, PqaCore/Summator.h:14-19
    acc.init(0.0);
    for (int64_t s2 = 0; s2 < nSubtasks; s2++) acc.add(lds[8 * s2]);
    scratch[8 * kMaxWorkers] = acc.get();
    __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // for the next launch on the stream
  }
}

__global__ __launch_bounds__(256) void long_row_divide_kernel(double *__restrict__ prior, const double *__restrict__ scratch, int64_t n4) {
  const double total = scratch[8 * kMaxWorkers];
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n4; t += (int64_t)gridDim.x * blockDim.x)
    prior[t] = prior[t] / total;   // CEDivTargPriors :19
}

size_t sum_lds_bytes(int64_t nWorkers) { return (size_t)(8 * nWorkers + 1) * sizeof(double); }
// (64 KB of dynamic LDS need no opt-in; 1000 targets at 16 workers: 9 KB)
bool stage_fits(const KbView &kb, int64_t nWorkers) { return sum_lds_bytes(nWorkers) + (size_t)kb.ldT * sizeof(double) <= 65536; }
size_t staged_lds_bytes(const KbView &kb, int64_t nWorkers) {
  return sum_lds_bytes(nWorkers) + (stage_fits(kb, nWorkers) ? (size_t)kb.ldT * sizeof(double) : 0);
}

PriorArgs make_args(const KbView &kb, double *prior, int64_t nWorkers, bool staged = false) {
  PriorArgs a;
  a.cube = kb.cube;
  a.elem = kb.elem;
  a.vB = kb.vB;
  a.tgap = kb.tgap;
  a.prior = prior;
  a.K = kb.K;
  a.T = kb.T;
  a.ldT = kb.ldT;
  a.nWorkers = nWorkers;
  a.stage = staged && stage_fits(kb, nWorkers) ? 1 : 0;
  return a;
}

}  // namespace

// The long-row form: its launches, or false if the row is short / the engine gave no scratch.
template <bool RECORD>
static bool launch_long_row(const KbView &kbLocal, double *prior, uint32_t *asked, int64_t askedWords, int64_t iQuestion, int64_t iAnswer,
                            int64_t nWorkers, hipStream_t stream, const void *cubeElsewhere = nullptr) {
  if (kbLocal.ldT <= 16384 || kbLocal.priorScratch == nullptr) return false;
  KbView kb = kbLocal;
  if (cubeElsewhere != nullptr) kb.cube = cubeElsewhere;   // (another shard's question block: same row length and layout)
  const int64_t nVects = (kb.T + 3) >> 2, quot = nVects / nWorkers, rem = nVects % nWorkers;
  const int64_t nSubtasks = quot == 0 ? rem : nWorkers;
  const size_t values = (size_t)(4 * (quot + 1) + (kb.ldT - 4 * nVects)) * sizeof(double);
  const bool inLds = sum_lds_bytes(nSubtasks) + values <= 65536;
  hipLaunchKernelGGL(long_row_stage_kernel<RECORD>, dim3((unsigned)nSubtasks), dim3(kThreads), sum_lds_bytes(nSubtasks) + (inLds ? values : 0), stream,
                     make_args(kb, prior, nWorkers), iQuestion, iAnswer, asked, askedWords, kb.priorScratch, inLds ? 1 : 0);
  const int64_t n4 = 4 * nVects;
  hipLaunchKernelGGL(long_row_divide_kernel, dim3((unsigned)std::min<int64_t>((n4 + 1023) / 1024, 1024)), dim3(256), 0, stream, prior, kb.priorScratch, n4);
  return true;
}

hipError_t LaunchStartQuiz(const KbView &kb, double *prior, uint32_t *asked, int64_t askedWords, int64_t nWorkers, hipStream_t stream) {
  if (nWorkers < 1 || nWorkers > kMaxWorkers) return hipErrorInvalidValue;
  if (launch_long_row<false>(kb, prior, asked, askedWords, 0, 0, nWorkers, stream)) return hipGetLastError();
  if (small_launch(kb))
    hipLaunchKernelGGL(start_quiz_kernel<true>, dim3(1), dim3(kSmallThreads), staged_lds_bytes(kb, nWorkers), stream,
                       make_args(kb, prior, nWorkers, true), asked, askedWords);
  else
    hipLaunchKernelGGL(start_quiz_kernel<false>, dim3(1), dim3(kThreads), staged_lds_bytes(kb, nWorkers), stream,
                       make_args(kb, prior, nWorkers, true), asked, askedWords);
  return hipGetLastError();
}

hipError_t LaunchRecordAnswer(const KbView &kb, double *prior, uint32_t *asked, int64_t iQuestion, int64_t iAnswer,
                              int64_t nWorkers, RatedTargetDev *topOut, int64_t *topN, uint64_t *topFlag,
                              uint64_t topFlagValue, int64_t topCount, hipStream_t stream, const void *rowA, const void *rowD) {
  if (nWorkers < 1 || nWorkers > kMaxWorkers || (rowA == nullptr) != (rowD == nullptr)) return hipErrorInvalidValue;
  const TopRequest top{reinterpret_cast<TopOut *>(topOut), topN, topFlag, topFlagValue, (topOut && kb.T <= 16384) ? topCount : 0};
  // (the long-row form addresses the rows through the cube: another shard's rows take the one-workgroup kernel)
  if (top.count == 0 && rowA == nullptr && launch_long_row<true>(kb, prior, asked, 0, iQuestion, iAnswer, nWorkers, stream)) return hipGetLastError();
  // another shard's rows in place (the D row of a question block lies K - iAnswer rows behind its A row): the block stands in for
  // a one-question cube; staged copies of the two rows take the one-workgroup kernel
  if (top.count == 0 && rowA != nullptr &&
      static_cast<const char *>(rowD) - static_cast<const char *>(rowA) == (ptrdiff_t)((kb.K - iAnswer) * kb.ldT * kb.elem) &&
      launch_long_row<true>(kb, prior, nullptr, 0, 0, iAnswer, nWorkers, stream, static_cast<const char *>(rowA) - (size_t)(iAnswer * kb.ldT * kb.elem)))
    return hipGetLastError();
  const RowPair rows{rowA, rowD};
  if (small_launch(kb))
    hipLaunchKernelGGL(record_answer_kernel<true>, dim3(1), dim3(kSmallThreads), staged_lds_bytes(kb, nWorkers), stream,
                       make_args(kb, prior, nWorkers, true), iQuestion, iAnswer, asked, top, rows);
  else
    hipLaunchKernelGGL(record_answer_kernel<false>, dim3(1), dim3(kThreads), staged_lds_bytes(kb, nWorkers), stream,
                       make_args(kb, prior, nWorkers, true), iQuestion, iAnswer, asked, top, rows);
  return hipGetLastError();
}

hipError_t LaunchStartQuizBatch(const KbView &kb, const StartBatchInline &batch, int64_t nWorkers, hipStream_t stream) {
  if (nWorkers < 1 || nWorkers > kMaxWorkers || batch.n < 1 || batch.n > kStartInline) return hipErrorInvalidValue;
  hipLaunchKernelGGL(start_quiz_batch_kernel, dim3((unsigned)batch.n), dim3(kb.T <= 4 * kSmallThreads ? kSmallThreads : kThreads),
                     staged_lds_bytes(kb, nWorkers), stream, make_args(kb, nullptr, nWorkers, true), batch);
  return hipGetLastError();
}

hipError_t LaunchRecordAnswerBatch(const KbView &kb, const RecordBatchInline &batch, int64_t nWorkers, hipStream_t stream) {
  if (nWorkers < 1 || nWorkers > kMaxWorkers || batch.n < 1 || batch.n > kRecordInline) return hipErrorInvalidValue;
  RecordBatchInline b = batch;
  if (kb.T > 16384) b.topCount = 0;
  // 256-thread workgroups whenever the rows are short: several quizzes' updates share a CU
  if (kb.T <= 4 * kSmallThreads)
    hipLaunchKernelGGL(record_answer_batch_kernel<true>, dim3((unsigned)b.n), dim3(kSmallThreads), staged_lds_bytes(kb, nWorkers), stream,
                       make_args(kb, nullptr, nWorkers, true), b);
  else
    hipLaunchKernelGGL(record_answer_batch_kernel<false>, dim3((unsigned)b.n), dim3(kThreads), staged_lds_bytes(kb, nWorkers), stream,
                       make_args(kb, nullptr, nWorkers, true), b);
  return hipGetLastError();
}

hipError_t LaunchResumeQuiz(const KbView &kb, double *prior, int64_t *exps, const void *const *rows, int64_t nAnswered,
                            int64_t nWorkers, int bugCompat, int64_t *status, hipStream_t stream) {
  if (nWorkers < 1 || nWorkers > kMaxWorkers || nAnswered < 1) return hipErrorInvalidValue;
  hipLaunchKernelGGL(resume_quiz_kernel, dim3(1), dim3(small_launch(kb) ? kSmallThreads : kThreads), sum_lds_bytes(nWorkers), stream,
                     make_args(kb, prior, nWorkers), exps, rows, nAnswered, bugCompat, status);
  return hipGetLastError();
}

}  // namespace pqa
