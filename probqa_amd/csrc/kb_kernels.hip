// kb_kernels.hip -- knowledge-base construction / mutation and top-target listing on gfx950.
//   fill_fresh      : CpuEngine ctor (reference: PqaCore/CpuEngine.cpp:44-84)  A = init^2, D = init^2*K, B = init
//   fill_synthetic  : deterministic benchmark / test cube, bit-identical to probqa_amd/synth.py (numpy)
//   train           : CETrainOperation::Perform1 / Perform2 (PqaCore/CETrainOperation.cpp:15-83)
//   top_targets     : ListTopTargets (PqaCore/CEListTopTargetsAlgorithm.cpp:30-95): descending probability
#include "pqa_device.h"
#include "pqa_kernels.h"

namespace pqa {

namespace {

__global__ __launch_bounds__(256) void fill_fresh_kernel(void *__restrict__ cube, int elem, double *__restrict__ vB, int64_t K,
                                                         int64_t Q, int64_t T, int64_t ldT, double initAmount) {
  const double init1 = initAmount, initSqr = init1 * init1, initMD = initSqr * (double)K;  // CpuEngine.cpp:45-47
  const int64_t total = Q * (K + 1) * ldT;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t t = i % ldT, r = (i / ldT) % (K + 1);
    cube_st(cube, elem, i, (t < T) ? (r < K ? initSqr : initMD) : (r < K ? 0.0 : 1.0));
    if (i < ldT) vB[i] = (i < T) ? (elem == 4 ? (double)(float)init1 : init1) : 0.0;   // (Float engines: vB holds fp32 values)
  }
}

__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  uint64_t z = x + 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
__device__ __forceinline__ double hash_unit(uint64_t seed, uint64_t idx) {  // uniform in [0,1), 53 bits
  return (double)(splitmix64(seed + idx * 0x9E3779B97F4A7C15ULL) >> 11) * 0x1.0p-53;
}

// One thread per (question, target): loops over the answers so that D = sum_k A is accumulated in k order.
__global__ __launch_bounds__(256) void fill_synth_kernel(void *__restrict__ cube, int elem, double *__restrict__ vB, int64_t K,
                                                         int64_t Q, int64_t T, int64_t ldT, int64_t qOffset,
                                                         int64_t qTotal, double initAmount, double nTrain,
                                                         double noiseAmp, uint64_t seed) {
  const int64_t total = Q * ldT;
  const int64_t w = (32 * T) / 1000 > 1 ? (32 * T) / 1000 : 1;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t q = i / ldT, t = i % ldT;
    const int64_t col = q * (K + 1) * ldT + t;
    if (t >= T) {
      for (int64_t k = 0; k < K; k++) cube_st(cube, elem, col + k * ldT, 0.0);
      cube_st(cube, elem, col + K * ldT, 1.0);
    } else {
      const int64_t qg = qOffset + q;
      const int64_t x = (qg * T) / qTotal;
      int64_t ans;  // the +-32 "binary search" answer rule of PqaCoreTests/DichotomyTest.cpp:50-64, scaled by T/1000
      if (t < x - w) ans = 0; else if (t < x) ans = 1; else if (t == x) ans = 2; else if (t <= x + w) ans = 3; else ans = 4;
      if (ans > K - 1) ans = K - 1;
      double d = 0.0;
      for (int64_t k = 0; k < K; k++) {
        double a = initAmount;
        if (k == ans) a = a + nTrain;
        a = a + noiseAmp * hash_unit(seed, (uint64_t)((qg * K + k) * T + t));
        double a2 = a * a;  // the cube stores squares (PqaCore/CETrainOperation.cpp:15-25)
        if (elem == 4) a2 = (double)(float)a2;   // Float cube: D is the sum of the ROUNDED squares, as training would leave it
        cube_st(cube, elem, col + k * ldT, a2);
        d = d + a2;
      }
      cube_st(cube, elem, col + K * ldT, d);
    }
    if (q == 0) {
      const double b = (t < T) ? (initAmount + nTrain) + noiseAmp * hash_unit(seed ^ 0x5851F42D4C957F2DULL, (uint64_t)t) : 0.0;
      vB[t] = elem == 4 ? (double)(float)b : b;
    }
  }
}

// Training (reference PqaCore/CETrainOperation.cpp:15-83).  The host turns the call's answered questions into steps in the
// reference's own pairing order (hip_engine_update.cpp: BuildTrainSteps) and groups the steps by question: steps on different
// questions touch different cells and run in parallel, one thread per question; a question's steps run in order.
//   kind 1  Perform1 (:28-30, and either half of a Perform2 over two different questions, :56-82): a = sqrt(A); A, D += 2ab + b^2
//   kind 2  Perform2, same question and answer (:34-35): ONE step of 2b -- A, D += 4ab + 4b^2 (_inc4B, _incSquare2B)
//   kind 3  Perform2, same question, answers a1 != a2 (:37-54): each cell its own addend; D += TWICE THE FIRST cell's addend (:45-46)
__device__ __forceinline__ void train_steps_body(void *__restrict__ cube, int elem, double *__restrict__ vB, int64_t K, int64_t ldT,
                                                 const TrainStep *__restrict__ steps, const int64_t *__restrict__ chainStart, int64_t nChains,
                                                 int64_t iTarget, double amount) {
  const double twoB = 2 * amount, bSquare = amount * amount;  // CETrainTaskNumSpec.h:24-32
  const double fourB = 4 * amount, square2B = 4 * bSquare;
  for (int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; c < nChains; c += (int64_t)gridDim.x * blockDim.x) {
    for (int64_t i = chainStart[c]; i < chainStart[c + 1]; i++) {
      const TrainStep st = steps[i];
      const int64_t iA = (st.q * (K + 1) + st.a1) * ldT + iTarget, iD = (st.q * (K + 1) + K) * ldT + iTarget;
      const double oldA = cube_ld(cube, elem, iA);
      const double a = sqrt(oldA);                               // CETrainOperation.cpp:18
      if (st.kind == 3) {
        const int64_t iA2 = (st.q * (K + 1) + st.a2) * ldT + iTarget;
        const double oldA2 = cube_ld(cube, elem, iA2);
        const double add1 = a * twoB + bSquare, add2 = sqrt(oldA2) * twoB + bSquare;   // :42-44
        cube_st(cube, elem, iA, oldA + add1);                    // :47-53
        cube_st(cube, elem, iA2, oldA2 + add2);
        cube_st(cube, elem, iD, cube_ld(cube, elem, iD) + (add1 + add1));
      } else {
        const double addend = st.kind == 2 ? a * fourB + square2B : a * twoB + bSquare;   // :19
        cube_st(cube, elem, iA, oldA + addend);                  // :23-24
        cube_st(cube, elem, iD, cube_ld(cube, elem, iD) + addend);   // :25
      }
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const double b = vB[iTarget] + amount;                   // PqaCore/CpuEngine.cpp:172, :462
    vB[iTarget] = elem == 4 ? (double)(float)b : b;
  }
}
__global__ void train_steps_kernel(void *__restrict__ cube, int elem, double *__restrict__ vB, int64_t K, int64_t ldT,
                                   const TrainStep *__restrict__ steps, const int64_t *__restrict__ chainStart, int64_t nChains,
                                   int64_t iTarget, double amount) {
  train_steps_body(cube, elem, vB, K, ldT, steps, chainStart, nChains, iTarget, amount);
}
// The same with the steps in the kernel's arguments: the training call at the end of a quiz has a handful of them, and two
// staged copies plus the wait for them (the host vectors are their sources) cost four times what the launch does.
__global__ void train_steps_inline_kernel(void *__restrict__ cube, int elem, double *__restrict__ vB, int64_t K, int64_t ldT,
                                          TrainStepsInline in, int64_t iTarget, double amount) {
  train_steps_body(cube, elem, vB, K, ldT, in.steps, in.chainStart, in.nChains, iTarget, amount);
}

// Several calls' steps, one workgroup per call (the calls' targets differ: no cell belongs to two of them).  The arithmetic of a step
// is train_steps_body's.
__global__ void train_batch_inline_kernel(void *__restrict__ cube, int elem, double *__restrict__ vB, int64_t K, int64_t ldT, TrainBatchInline in) {
  const TrainBatchCall call = in.calls[blockIdx.x];
  const int64_t iTarget = call.iTarget;
  const double amount = call.amount;
  const double twoB = 2 * amount, bSquare = amount * amount;  // CETrainTaskNumSpec.h:24-32
  const double fourB = 4 * amount, square2B = 4 * bSquare;
  // chain c of the call: steps [chainStart[firstChain + callIndex + c], chainStart[firstChain + callIndex + c + 1]) -- every call's
  // chain starts are followed by one end marker, so call b's entries begin at firstChain + b
  const uint16_t *cs = in.chainStart + call.firstChain + blockIdx.x;
  for (int c = threadIdx.x; c < call.nChains; c += blockDim.x) {
    for (int i = cs[c]; i < cs[c + 1]; i++) {
      const TrainBatchStep st = in.steps[i];
      const int64_t iA = ((int64_t)st.q * (K + 1) + st.a1) * ldT + iTarget, iD = ((int64_t)st.q * (K + 1) + K) * ldT + iTarget;
      const double oldA = cube_ld(cube, elem, iA);
      const double a = sqrt(oldA);                               // CETrainOperation.cpp:18
      if (st.kind == 3) {
        const int64_t iA2 = ((int64_t)st.q * (K + 1) + st.a2) * ldT + iTarget;
        const double oldA2 = cube_ld(cube, elem, iA2);
        const double add1 = a * twoB + bSquare, add2 = sqrt(oldA2) * twoB + bSquare;   // :42-44
        cube_st(cube, elem, iA, oldA + add1);                    // :47-53
        cube_st(cube, elem, iA2, oldA2 + add2);
        cube_st(cube, elem, iD, cube_ld(cube, elem, iD) + (add1 + add1));
      } else {
        const double addend = st.kind == 2 ? a * fourB + square2B : a * twoB + bSquare;   // :19
        cube_st(cube, elem, iA, oldA + addend);                  // :23-24
        cube_st(cube, elem, iD, cube_ld(cube, elem, iD) + addend);   // :25
      }
    }
  }
  if (threadIdx.x == 0) {
    const double b = vB[iTarget] + amount;                   // PqaCore/CpuEngine.cpp:172, :462
    vB[iTarget] = elem == 4 ? (double)(float)b : b;
  }
}

// ListTopTargets on the device (top_targets_publish in pqa_device.h); T <= 16384.
static_assert(sizeof(TopOut) == sizeof(RatedTargetDev), "same record");
template <bool SMALL>
__global__ __launch_bounds__(SMALL ? 256 : 1024) void top_targets_kernel(const double *__restrict__ prior,
                                                                         const uint32_t *__restrict__ tgap, int64_t T,
                                                                         int64_t maxCount, RatedTargetDev *out, int64_t *nOut,
                                                                         uint64_t *flag, uint64_t flagValue) {
  __shared__ TopScratch scratch;
  top_targets_publish<SMALL>(prior, tgap, T, maxCount, reinterpret_cast<TopOut *>(out), nOut, flag, flagValue, &scratch);
}

unsigned grid_for(int64_t n, int threads) {
  int64_t b = (n + threads - 1) / threads;
  if (b > 256 * 32) b = 256 * 32;
  if (b < 1) b = 1;
  return (unsigned)b;
}

// ---- maintenance (reference PqaCore/CpuEngine.cpp:468-658) ---------------------------------------------------------
// (Re)initialise whole questions: A = init^2, D = init^2 * K on real targets; padding columns A = 0, D = 1.
__global__ __launch_bounds__(256) void fill_questions_kernel(void *__restrict__ cube, int elem, int64_t K, int64_t T, int64_t ldT,
                                                             const int64_t *__restrict__ qs,
                                                             const double *__restrict__ inits, int64_t n) {
  const int64_t per = (K + 1) * ldT, total = n * per;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t iq = i / per, r = (i % per) / ldT, t = i % ldT;
    const double initSqr = inits[iq] * inits[iq], initMD = initSqr * (double)K;  // :503-510 / :548-555
    cube_st(cube, elem, qs[iq] * per + r * ldT + t, (t < T) ? (r < K ? initSqr : initMD) : (r < K ? 0.0 : 1.0));
  }
}

// (Re)initialise target columns over questions [0,nQ) except those flagged in skipQ (already initialised as questions).
__global__ __launch_bounds__(256) void fill_targets_kernel(void *__restrict__ cube, int elem, double *__restrict__ vB, int64_t K,
                                                           int64_t ldT, int64_t nQ, const uint32_t *__restrict__ skipQ,
                                                           const int64_t *__restrict__ ts,
                                                           const double *__restrict__ inits, int64_t n) {
  const int64_t total = nQ * n;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t q = i / n, it = i % n;
    if (q == 0) vB[ts[it]] = elem == 4 ? (double)(float)inits[it] : inits[it];   // :532 / :566
    if (skipQ && bit_test(skipQ, q)) continue;                                   // :558-560
    const double initSqr = inits[it] * inits[it], initMD = initSqr * (double)K;  // :517,:523 / :556-557
    const int64_t col = q * (K + 1) * ldT + ts[it];
    for (int64_t k = 0; k < K; k++) cube_st(cube, elem, col + k * ldT, initSqr);
    cube_st(cube, elem, col + K * ldT, initMD);
  }
}

// Compaction of the target axis (:619-647): column dst <- column src for every kept question and for vB.
__global__ __launch_bounds__(256) void move_targets_kernel(void *__restrict__ cube, int elem, double *__restrict__ vB, int64_t K,
                                                           int64_t ldT, int64_t nQ, const int64_t *__restrict__ moves,
                                                           int64_t n) {
  const int64_t rows = nQ * (K + 1) + 1, total = rows * n;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / n, m = i % n;
    // (src, dst) pairs; sources are never destinations (:604-618)
    if (row < nQ * (K + 1)) cube_st(cube, elem, row * ldT + moves[2 * m + 1], cube_ld(cube, elem, row * ldT + moves[2 * m]));
    else vB[moves[2 * m + 1]] = vB[moves[2 * m]];
  }
}

// Rebuild of a shard after a change of the dimensions (sharded_engine.cpp): question q of the new cube takes the rows of the
// block src[q] -- which may live on another device of the process (peer access) -- with its columns picked by colMap (new
// target t <- old target colMap[t]; -1: a new column, left as the fresh fill made it and initialised afterwards).
__global__ __launch_bounds__(256) void adopt_rows_kernel(void *__restrict__ dst, int elem, int64_t K, int64_t nQ, int64_t Tn, int64_t ldTn,
                                                         const void *const *__restrict__ src, int64_t ldTs,
                                                         const int64_t *__restrict__ colMap) {
  const int64_t per = (K + 1) * Tn, total = nQ * per;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t q = i / per, r = (i % per) / Tn, t = i % Tn;
    const void *s = src[q];
    const int64_t c = colMap[t];
    if (s != nullptr && c >= 0) cube_st(dst, elem, (q * (K + 1) + r) * ldTn + t, cube_ld(s, elem, r * ldTs + c));
  }
}
__global__ __launch_bounds__(256) void adopt_vb_kernel(double *__restrict__ dst, const double *__restrict__ src, const int64_t *__restrict__ colMap,
                                                       int64_t Tn) {
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < Tn; t += (int64_t)gridDim.x * blockDim.x)
    if (colMap[t] >= 0) dst[t] = src[colMap[t]];
}

}  // namespace

hipError_t LaunchAdoptRows(void *dst, int elem, double *dstVB, int64_t K, int64_t nQ, int64_t Tn, int64_t ldTn, const void *const *src,
                           int64_t ldTs, const double *srcVB, const int64_t *colMap, hipStream_t stream) {
  if (nQ <= 0 || Tn <= 0) return hipSuccess;
  hipLaunchKernelGGL(adopt_rows_kernel, dim3(grid_for(nQ * (K + 1) * Tn, 256)), dim3(256), 0, stream, dst, elem, K, nQ, Tn, ldTn, src, ldTs, colMap);
  hipLaunchKernelGGL(adopt_vb_kernel, dim3(grid_for(Tn, 256)), dim3(256), 0, stream, dstVB, srcVB, colMap, Tn);
  return hipGetLastError();
}

hipError_t LaunchFillQuestions(void *cube, int elem, int64_t K, int64_t T, int64_t ldT, const int64_t *qs, const double *inits,
                               int64_t n, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(fill_questions_kernel, dim3(grid_for(n * (K + 1) * ldT, 256)), dim3(256), 0, stream, cube, elem, K, T, ldT,
                     qs, inits, n);
  return hipGetLastError();
}

hipError_t LaunchFillTargets(void *cube, int elem, double *vB, int64_t K, int64_t ldT, int64_t nQ, const uint32_t *skipQ,
                             const int64_t *ts, const double *inits, int64_t n, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(fill_targets_kernel, dim3(grid_for(nQ * n, 256)), dim3(256), 0, stream, cube, elem, vB, K, ldT, nQ, skipQ, ts,
                     inits, n);
  return hipGetLastError();
}

hipError_t LaunchMoveTargets(void *cube, int elem, double *vB, int64_t K, int64_t ldT, int64_t nQ, const int64_t *moves,
                             int64_t n, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(move_targets_kernel, dim3(grid_for((nQ * (K + 1) + 1) * n, 256)), dim3(256), 0, stream, cube, elem, vB, K,
                     ldT, nQ, moves, n);
  return hipGetLastError();
}

hipError_t LaunchFillFresh(void *cube, int elem, double *vB, int64_t K, int64_t Q, int64_t T, int64_t ldT, double initAmount,
                           hipStream_t stream) {
  hipLaunchKernelGGL(fill_fresh_kernel, dim3(grid_for(Q * (K + 1) * ldT, 256)), dim3(256), 0, stream, cube, elem, vB, K, Q, T,
                     ldT, initAmount);
  return hipGetLastError();
}

hipError_t LaunchFillSynthetic(void *cube, int elem, double *vB, int64_t K, int64_t Q, int64_t T, int64_t ldT, int64_t qOffset,
                               int64_t qTotal, double initAmount, double nTrain, double noiseAmp, uint64_t seed,
                               hipStream_t stream) {
  hipLaunchKernelGGL(fill_synth_kernel, dim3(grid_for(Q * ldT, 256)), dim3(256), 0, stream, cube, elem, vB, K, Q, T, ldT,
                     qOffset, qTotal, initAmount, nTrain, noiseAmp, seed);
  return hipGetLastError();
}

hipError_t LaunchTrainSteps(void *cube, int elem, double *vB, int64_t K, int64_t ldT, const TrainStep *steps,
                            const int64_t *chainStart, int64_t nChains, int64_t iTarget, double amount, hipStream_t stream) {
  hipLaunchKernelGGL(train_steps_kernel, dim3(grid_for(nChains, 64)), dim3(64), 0, stream, cube, elem, vB, K, ldT, steps, chainStart,
                     nChains, iTarget, amount);
  return hipGetLastError();
}

hipError_t LaunchTrainStepsInline(void *cube, int elem, double *vB, int64_t K, int64_t ldT, const TrainStepsInline &in,
                                  int64_t iTarget, double amount, hipStream_t stream) {
  hipLaunchKernelGGL(train_steps_inline_kernel, dim3(1), dim3(64), 0, stream, cube, elem, vB, K, ldT, in, iTarget, amount);
  return hipGetLastError();
}

hipError_t LaunchTrainBatchInline(void *cube, int elem, double *vB, int64_t K, int64_t ldT, const TrainBatchInline &in, hipStream_t stream) {
  if (in.nCalls < 1 || in.nCalls > kTrainBatchCalls) return hipErrorInvalidValue;
  hipLaunchKernelGGL(train_batch_inline_kernel, dim3((unsigned)in.nCalls), dim3(64), 0, stream, cube, elem, vB, K, ldT, in);
  return hipGetLastError();
}

hipError_t LaunchTopTargets(const KbView &kb, const double *prior, int64_t maxCount, RatedTargetDev *out,
                            int64_t *nOut, uint64_t *flag, uint64_t flagValue, hipStream_t stream) {
  if (kb.T > 16384) return hipErrorInvalidValue;  // LaunchTopTargetsBatch takes over (hip_engine_update.cpp)
  if (kb.smallLaunches && kb.T <= 1024)   // beside the resident sweep (prior_kernels.hip: kSmallThreads)
    hipLaunchKernelGGL(top_targets_kernel<true>, dim3(1), dim3(256), 0, stream, prior, kb.tgap, kb.T, maxCount, out, nOut, flag,
                       flagValue);
  else
    hipLaunchKernelGGL(top_targets_kernel<false>, dim3(1), dim3(1024), 0, stream, prior, kb.tgap, kb.T, maxCount, out, nOut, flag,
                       flagValue);
  return hipGetLastError();
}

// ---- ListTopTargets over rows of any length, for many quizzes at once ----------------------------------------------------
// Level 0: every WAVE holds 1024 targets of a quiz's posterior in registers (16 per lane) and lists ITS best maxCount by itself -- no
// LDS, no barrier: a round is two wave all-reduces, and the lane that held the winner looks at its 16 again.  The best maxCount of the
// whole row under the order (probability descending, target ascending) are among the waves' bests under the same order.  Then merges
// of up to 16384 / maxCount candidate lists per workgroup until one list per quiz is left; the last level writes the caller's records
// and counts.  What crosses to the host is nQuizzes x maxCount records -- the reference's GPU engine copies all T posteriors per quiz
// and heapifies on the host (PqaCore/CudaEngine.cpp:251-289).
namespace {
constexpr int kTopChunkThreads = 256, kTopWaveTargets = 16 * kWave;
static_assert(kTopChunkThreads / kWave * kTopWaveTargets == kTopChunkTargets, "four waves of 1024 targets per workgroup");

__global__ __launch_bounds__(kTopChunkThreads) void top_chunks_kernel(TopBatchPriors priors, const uint32_t *__restrict__ tgap, int64_t T,
                                                                      int64_t maxCount, TopOut *lists) {
  const int lane = threadIdx.x % kWave, wave = threadIdx.x / kWave;
  const int64_t quiz = blockIdx.y, nLists = (int64_t)gridDim.x * (kTopChunkThreads / kWave);
  const int64_t list = (int64_t)blockIdx.x * (kTopChunkThreads / kWave) + wave;
  const int64_t tFirst = list * kTopWaveTargets;
  const double *prior = priors.prior[quiz];
  double p[16];
  int t[16];
#pragma unroll
  for (int e = 0; e < 16; e++) {
    const int64_t tt = tFirst + lane + e * kWave;
    const bool ok = tt < T && !bit_test(tgap, tt);
    p[e] = ok ? prior[tt] : -1.0;
    if (!(p[e] > 0.0)) p[e] = -1.0;       // gaps and probabilities <= 0 are no candidates (pqa_device.h)
    t[e] = (int)tt;
  }
  TopOut *mine = lists + (quiz * nLists + list) * maxCount;
  const int64_t listed = top_rounds_wave<16>(p, t, maxCount, mine);
  for (int64_t i = listed + lane; i < maxCount; i += kWave) mine[i] = TopOut{-1, -1.0};   // the rest of the list says "no candidate"
}

__device__ __forceinline__ void top_scratch_init(TopScratch *scratch) {
  if (threadIdx.x < 32) {   // waves that do not exist never win a round
    scratch->sp[threadIdx.x >> 4][threadIdx.x & 15] = -1.0;
    scratch->st[threadIdx.x >> 4][threadIdx.x & 15] = kTopNone;
  }
  __syncthreads();
}
template <int E>
__device__ __forceinline__ int64_t top_merge_rounds(const TopOut *cand, int64_t n, int64_t maxCount, TopScratch *scratch) {
  double p[E];
  int t[E];
#pragma unroll
  for (int e = 0; e < E; e++) {
    const int64_t i = threadIdx.x + (int64_t)e * blockDim.x;
    const TopOut c = i < n ? cand[i] : TopOut{-1, -1.0};
    p[e] = c.prob > 0.0 ? c.prob : -1.0;
    t[e] = (int)c.iTarget;
  }
  return top_rounds<E>(p, t, maxCount, scratch->staged, scratch->sp, scratch->st);
}
// workgroup (group, quiz): lists [group * fanIn, ...) of the quiz's nLists lists of maxCount records -> list `group` of the next level
// (256 threads for up to 4096 candidates, 1024 beyond: a round costs every wave its all-reduces, whatever it holds).  The winners are
// staged in LDS and leave in one burst: the last level's list goes to host-coherent memory, where a store per round would cost more
// than the round.
__global__ __launch_bounds__(1024) void top_merge_kernel(const TopOut *__restrict__ lists, int64_t nLists, int64_t fanIn, int64_t maxCount,
                                                         TopOut *outLists, int64_t *nOut, uint64_t *flag, uint64_t flagValue) {
  __shared__ TopScratch scratch;
  top_scratch_init(&scratch);
  const int64_t group = blockIdx.x, quiz = blockIdx.y;
  const int64_t first = group * fanIn, limit = first + fanIn < nLists ? first + fanIn : nLists;
  const TopOut *cand = lists + (quiz * nLists + first) * maxCount;
  const int64_t n = (limit - first) * maxCount;
  int64_t listed;
  if (n <= blockDim.x) listed = top_merge_rounds<1>(cand, n, maxCount, &scratch);
  else if (n <= 4 * (int64_t)blockDim.x) listed = top_merge_rounds<4>(cand, n, maxCount, &scratch);
  else listed = top_merge_rounds<16>(cand, n, maxCount, &scratch);
  __syncthreads();
  TopOut *list = outLists + (quiz * gridDim.x + group) * maxCount;
  for (int64_t i = threadIdx.x; i < maxCount; i += blockDim.x) list[i] = i < listed ? scratch.staged[i] : TopOut{-1, -1.0};
  if (nOut != nullptr || flag != nullptr) {   // (the last level)
    __syncthreads();
    if (threadIdx.x == 0) {
      if (nOut != nullptr) nOut[quiz] = listed;
      if (flag != nullptr) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        __hip_atomic_store(flag, flagValue, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}
}  // namespace

// ---- ListTopTargets in the REFERENCE'S order among equal probabilities (round 6) --------------------------------------------------
// Where the listing above shows equal probabilities among the listed targets, or between the last listed and the next, what the
// reference returns is decided by its heaps (PqaCore/CEListTopTargetsAlgorithm.cpp:30-95): the target axis in one piece per worker
// thread (CalcSplit), every piece's candidates -- no gaps, prob > 0 -- in index order made a heap (std::make_heap,
// CEHeapifyPriorsSubtaskMake.cpp:56-88), then a head heap over the pieces' tops that pops maxCount times.  Reproduced step for step:
//   * top_pieces_kernel, a workgroup per (piece, quiz): the candidates compacted in index order into LDS (a piece of more than 8192
//     candidates: into global scratch); std::make_heap level by level -- the library sifts the holes n/2 - 1 ... 0 in turn, deeper
//     levels before shallower ones, and the subtrees of one level are disjoint: a level's holes sift side by side, a barrier between
//     levels, the same heap --; then ONE thread pops the piece's first maxCount tops, which are what the piece yields whatever the
//     other pieces do (a piece's heap changes only when its own top is taken);
//   * top_heads_kernel, a thread per quiz: the head heap over the pieces' first tops (std::make_heap), maxCount times the top, the
//     piece's next key and SRHeapHelper::Down (SRPlatform/Interface/SRHeap.h:16-39) -- or std::pop_heap when the piece is exhausted.
// The heap steps are the algorithm MSVC's and libstdc++'s libraries share (the oracle's restatement is held to libstdc++'s).
namespace {
struct HeapRec { double prob; int64_t id; };
__device__ __forceinline__ void heap_push_by_index(HeapRec *first, int64_t hole, int64_t top, HeapRec val) {
  for (int64_t idx = (hole - 1) >> 1; top < hole && first[idx].prob < val.prob; idx = (hole - 1) >> 1) {
    first[hole] = first[idx];
    hole = idx;
  }
  first[hole] = val;
}
__device__ __forceinline__ void heap_adjust(HeapRec *first, int64_t hole, int64_t bottom, HeapRec val) {
  const int64_t top = hole;
  int64_t idx = hole;
  const int64_t maxNonLeaf = (bottom - 1) >> 1;
  while (idx < maxNonLeaf) {                         // the hole moves down to the larger child (the right one unless it is less)
    idx = 2 * idx + 2;
    if (first[idx].prob < first[idx - 1].prob) --idx;
    first[hole] = first[idx];
    hole = idx;
  }
  if (idx == maxNonLeaf && bottom % 2 == 0) {        // an only child at the bottom
    first[hole] = first[bottom - 1];
    hole = bottom - 1;
  }
  heap_push_by_index(first, hole, top, val);
}
__device__ __forceinline__ void heap_pop(HeapRec *first, int64_t n) {   // std::pop_heap: the top goes to first[n - 1]
  if (n < 2) return;
  const HeapRec val = first[n - 1];
  first[n - 1] = first[0];
  heap_adjust(first, 0, n - 1, val);
}
__device__ __forceinline__ void heap_down(HeapRec *first, int64_t n) {  // SRHeapHelper::Down
  int64_t cur = 0;
  for (;;) {
    const int64_t c1 = 2 * cur + 1;
    if (c1 >= n) return;
    const int64_t c2 = c1 + 1;
    if (c2 >= n) {
      if (first[cur].prob < first[c1].prob) { const HeapRec t = first[cur]; first[cur] = first[c1]; first[c1] = t; }
      return;
    }
    const int64_t hi = first[c2].prob < first[c1].prob ? c1 : c2;
    if (!(first[cur].prob < first[hi].prob)) return;
    const HeapRec t = first[cur]; first[cur] = first[hi]; first[hi] = t;
    cur = hi;
  }
}

constexpr int kTopPieceThreads = 256;
constexpr int64_t kTopPieceLds = 8192;               // candidates of a piece held in LDS (128 KB)
struct TopExactArgs {
  TopBatchPriors priors;
  const uint32_t *tgap;
  int64_t T, quot, rem, nSub, maxCount;              // CalcSplit(T, nWorkers): piece i = [i quot + min(i, rem), ...) of quot + (i < rem) targets
  HeapRec *heaps;                                    // [quiz][T]: the pieces that do not fit LDS (null: all fit)
  TopOut *lists;                                     // [quiz][nSub][maxCount]: a piece's first tops, in the order it yields them
  int32_t *counts;                                   // [quiz][nSub]: min(candidates of the piece, maxCount + 1)
  TopOut *out;                                       // [quiz][maxCount]
  int64_t *nOut;                                     // [quiz]
  uint64_t *flag;                                    // (one quiz) set to flagValue behind the results
  uint64_t flagValue;
};

__global__ __launch_bounds__(kTopPieceThreads) void top_pieces_kernel(TopExactArgs a) {
  extern __shared__ double smem[];
  __shared__ int64_t waveCount[kTopPieceThreads / kWave];
  const int tid = threadIdx.x, lane = tid % kWave, wave = tid / kWave;
  const int64_t piece = blockIdx.x, quiz = blockIdx.y;
  const int64_t first = piece * a.quot + (piece < a.rem ? piece : a.rem), size = a.quot + (piece < a.rem ? 1 : 0), limit = first + size;
  const double *prior = a.priors.prior[quiz];
  HeapRec *h = a.heaps == nullptr ? reinterpret_cast<HeapRec *>(smem) : a.heaps + quiz * a.T + first;   // (the launcher's choice, by the largest piece)
  // ---- the candidates, in index order (CEHeapifyPriorsSubtaskMake.cpp:42-52, :66-83)
  int64_t m = 0;
  for (int64_t base = first; base < limit; base += kTopPieceThreads) {
    const int64_t t = base + tid;
    double p = 0.0;
    bool ok = t < limit && !bit_test(a.tgap, t);
    if (ok) { p = prior[t]; ok = p > 0.0; }
    const unsigned long long mask = __ballot(ok);
    const int rank = __popcll(mask & ((1ull << lane) - 1ull));
    if (lane == 0) waveCount[wave] = __popcll(mask);
    __syncthreads();
    int64_t off = m;
    for (int w = 0; w < wave; w++) off += waveCount[w];
    if (ok) h[off + rank] = HeapRec{p, t};
    int64_t all = 0;
    for (int w = 0; w < kTopPieceThreads / kWave; w++) all += waveCount[w];
    m += all;
    __syncthreads();
  }
  // ---- std::make_heap (:87), level by level
  if (m >= 2) {
    const int64_t lastInternal = m / 2 - 1;
    int L = 0;
    while (((int64_t)2 << L) - 1 <= lastInternal) L++;          // the deepest level with an internal node: 2^L - 1 <= lastInternal
    for (; L >= 0; L--) {
      const int64_t lo = ((int64_t)1 << L) - 1, hiNode = ((int64_t)2 << L) - 2 < lastInternal ? ((int64_t)2 << L) - 2 : lastInternal;
      for (int64_t j = lo + tid; j <= hiNode; j += kTopPieceThreads) heap_adjust(h, j, m, h[j]);
      __syncthreads();
    }
  }
  // ---- what the piece yields, in turn: its top, then std::pop_heap (CEListTopTargetsAlgorithm.cpp:74-91)
  if (tid == 0) {
    const int64_t cnt = m < a.maxCount ? m : a.maxCount;
    TopOut *mine = a.lists + (quiz * a.nSub + piece) * a.maxCount;
    int64_t n = m;
    for (int64_t r = 0; r < cnt; r++) {
      mine[r] = TopOut{h[0].id, h[0].prob};
      if (r + 1 < cnt) { heap_pop(h, n); n--; }
    }
    a.counts[quiz * a.nSub + piece] = (int32_t)(m < a.maxCount + 1 ? m : a.maxCount + 1);
  }
}

__global__ __launch_bounds__(kWave) void top_heads_kernel(TopExactArgs a) {
  extern __shared__ double smem[];
  HeapRec *head = reinterpret_cast<HeapRec *>(smem);           // [nSub]
  int32_t *taken = reinterpret_cast<int32_t *>(head + a.nSub); // [nSub]
  const int64_t quiz = blockIdx.x;
  if (threadIdx.x != 0) return;
  const TopOut *lists = a.lists + quiz * a.nSub * a.maxCount;
  const int32_t *counts = a.counts + quiz * a.nSub;
  int64_t nHh = 0;
  for (int64_t i = 0; i < a.nSub; i++) {                        // :58-66: the pieces that have a candidate, in piece order
    taken[i] = 0;
    if (counts[i] == 0) continue;
    head[nHh++] = HeapRec{lists[i * a.maxCount].prob, i};
  }
  for (int64_t hole = nHh >> 1; hole > 0;) { --hole; heap_adjust(head, hole, nHh, head[hole]); }   // std::make_heap :67
  TopOut *out = a.out + quiz * a.maxCount;
  int64_t listed = a.maxCount;
  for (int64_t i = 0; i < a.maxCount; i++) {                    // :69-95
    if (nHh == 0) { listed = i; break; }
    const int64_t piece = head[0].id;
    out[i] = lists[piece * a.maxCount + taken[piece]];
    taken[piece]++;
    if (taken[piece] == counts[piece]) {                        // the piece is exhausted (:81-88)
      heap_pop(head, nHh);
      nHh--;
      continue;
    }
    if (i + 1 == a.maxCount) break;                             // (the key behind the last listed target is never looked at)
    head[0].prob = lists[piece * a.maxCount + taken[piece]].prob;   // :93
    heap_down(head, nHh);                                       // :94
  }
  for (int64_t i = listed; i < a.maxCount; i++) out[i] = TopOut{-1, -1.0};
  a.nOut[quiz] = listed;
  if (a.flag != nullptr) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    __hip_atomic_store(a.flag, a.flagValue, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
}  // namespace

// scratch of the exact listing for nQuizzes quizzes: bytes of {the pieces' lists, their counts, the heaps of pieces beyond LDS}
size_t TopExactScratchBytes(int64_t T, int64_t nWorkers, int64_t maxCount, int64_t nQuizzes) {
  const int64_t quot = T / nWorkers, rem = T % nWorkers, nSub = quot == 0 ? rem : nWorkers;
  const bool big = quot + (rem ? 1 : 0) > kTopPieceLds;
  return (size_t)nQuizzes * ((size_t)nSub * (size_t)maxCount * sizeof(TopOut) + (size_t)nSub * sizeof(int32_t) + 16 + (big ? (size_t)T * sizeof(HeapRec) : 0));
}
hipError_t LaunchTopTargetsExact(const KbView &kb, const TopBatchPriors &priors, int64_t nQuizzes, int64_t nWorkers, int64_t maxCount, void *scratch,
                                 RatedTargetDev *out, int64_t *nOut, uint64_t *flag, uint64_t flagValue, hipStream_t stream) {
  if (nQuizzes <= 0 || nQuizzes > kTopBatchQuizzes || maxCount <= 0 || nWorkers < 1 || scratch == nullptr || (flag != nullptr && nQuizzes != 1)) return hipErrorInvalidValue;
  TopExactArgs a{};
  a.priors = priors; a.tgap = kb.tgap; a.T = kb.T; a.maxCount = maxCount;
  a.quot = kb.T / nWorkers; a.rem = kb.T % nWorkers; a.nSub = a.quot == 0 ? a.rem : nWorkers;   // SRPoolRunner::CalcSplit, SRPoolRunner.h:96-110
  const int64_t pieceMax = a.quot + (a.rem ? 1 : 0);
  const bool big = pieceMax > kTopPieceLds;
  char *p = static_cast<char *>(scratch);
  a.lists = reinterpret_cast<TopOut *>(p);
  p += (size_t)nQuizzes * a.nSub * maxCount * sizeof(TopOut);
  a.heaps = big ? reinterpret_cast<HeapRec *>(p) : nullptr;
  if (big) p += (size_t)nQuizzes * kb.T * sizeof(HeapRec);
  a.counts = reinterpret_cast<int32_t *>(p);
  a.out = reinterpret_cast<TopOut *>(out); a.nOut = nOut; a.flag = flag; a.flagValue = flagValue;
  const size_t shPieces = big ? 16 : (size_t)pieceMax * sizeof(HeapRec);
  static LaunchCache cache;
  const int dev = LaunchCache::Device();
  int dummy = 0;
  if (shPieces > 64 * 1024 && !cache.Get(dev, 1, &dummy)) {   // (the opt-in to more than 64 KiB of dynamic LDS: once per device, for the largest size)
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(top_pieces_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kTopPieceLds * sizeof(HeapRec)));
    if (e != hipSuccess) return e;
    cache.Put(dev, 1, 1);
  }
  hipLaunchKernelGGL(top_pieces_kernel, dim3((unsigned)a.nSub, (unsigned)nQuizzes), dim3(kTopPieceThreads), shPieces, stream, a);
  const size_t shHeads = (size_t)a.nSub * (sizeof(HeapRec) + sizeof(int32_t)) + 16;
  hipLaunchKernelGGL(top_heads_kernel, dim3((unsigned)nQuizzes), dim3(kWave), shHeads, stream, a);
  return hipGetLastError();
}

static int64_t TopBatchLists(int64_t T) { return (T <= 0 ? 1 : (T + kTopChunkTargets - 1) / kTopChunkTargets) * (kTopChunkThreads / kWave); }
static int64_t TopFanIn(int64_t maxCount) { const int64_t f = kTopMergeCapacity / maxCount; return f < 2 ? 2 : f; }
// records per quiz each of the two scratch buffers must hold
int64_t TopBatchScratchRecords(int64_t T, int64_t maxCount) { return TopBatchLists(T) * maxCount; }
hipError_t LaunchTopTargetsBatch(const KbView &kb, const TopBatchPriors &priors, int64_t nQuizzes, int64_t maxCount, RatedTargetDev *scratchA,
                                 RatedTargetDev *scratchB, RatedTargetDev *out, int64_t *nOut, uint64_t *flag, uint64_t flagValue,
                                 hipStream_t stream) {
  if (nQuizzes <= 0 || nQuizzes > kTopBatchQuizzes || maxCount <= 0 || maxCount > 256 || (flag != nullptr && nQuizzes != 1)) return hipErrorInvalidValue;
  int64_t nLists = TopBatchLists(kb.T);
  hipLaunchKernelGGL(top_chunks_kernel, dim3((unsigned)(nLists / (kTopChunkThreads / kWave)), (unsigned)nQuizzes), dim3(kTopChunkThreads), 0, stream, priors,
                     kb.tgap, kb.T, maxCount, reinterpret_cast<TopOut *>(scratchA));
  const int64_t fanIn = TopFanIn(maxCount);
  RatedTargetDev *src = scratchA, *dst = scratchB;
  for (;;) {
    const int64_t nGroups = (nLists + fanIn - 1) / fanIn;
    const bool last = nGroups == 1;
    const int threads = (nLists < fanIn ? nLists : fanIn) * maxCount <= 4096 ? 256 : 1024;
    hipLaunchKernelGGL(top_merge_kernel, dim3((unsigned)nGroups, (unsigned)nQuizzes), dim3(threads), 0, stream, reinterpret_cast<const TopOut *>(src), nLists,
                       fanIn, maxCount, reinterpret_cast<TopOut *>(last ? out : dst), last ? nOut : nullptr, last ? flag : nullptr, flagValue);
    if (last) break;
    nLists = nGroups;
    RatedTargetDev *t = src; src = dst; dst = t;
  }
  return hipGetLastError();
}

}  // namespace pqa
