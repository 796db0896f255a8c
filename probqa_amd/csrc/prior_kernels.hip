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
                                                                                         uint32_t *asked, TopRequest top) {
  extern __shared__ double lds[];
  __shared__ TopScratch topScratch;
  record_answer_body<SMALL, false>(a, iQuestion, iAnswer, asked, top, lds, &topScratch);
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
  record_answer_body<SMALL, false>(a, s.iQuestion, s.iAnswer, s.asked, top, lds, &topScratch);
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

hipError_t LaunchStartQuiz(const KbView &kb, double *prior, uint32_t *asked, int64_t askedWords, int64_t nWorkers, hipStream_t stream) {
  if (nWorkers < 1 || nWorkers > kMaxWorkers) return hipErrorInvalidValue;
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
                              uint64_t topFlagValue, int64_t topCount, hipStream_t stream) {
  if (nWorkers < 1 || nWorkers > kMaxWorkers) return hipErrorInvalidValue;
  const TopRequest top{reinterpret_cast<TopOut *>(topOut), topN, topFlag, topFlagValue, (topOut && kb.T <= 16384) ? topCount : 0};
  if (small_launch(kb))
    hipLaunchKernelGGL(record_answer_kernel<true>, dim3(1), dim3(kSmallThreads), staged_lds_bytes(kb, nWorkers), stream,
                       make_args(kb, prior, nWorkers, true), iQuestion, iAnswer, asked, top);
  else
    hipLaunchKernelGGL(record_answer_kernel<false>, dim3(1), dim3(kThreads), staged_lds_bytes(kb, nWorkers), stream,
                       make_args(kb, prior, nWorkers, true), iQuestion, iAnswer, asked, top);
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
