// eval_kernels.hip -- the next-question priority sweep on CDNA4 (gfx950).
//
// Replaces CEEvalQsSubtaskConsider<SRDoubleNumber>::Run (reference: PqaCore/CEEvalQsSubtaskConsider.cpp:41-217) and,
// for the selection step, the tail of CpuEngine::NextQuestionSpec (PqaCore/CpuEngine.cpp:362-400).
//
// Shape: one workgroup of WPQ wavefronts per candidate question.  Lane `tid` owns the target pairs
// p = tid + j*(64*WPQ), j < NP, i.e. every global load is a fully coalesced 16 B/lane (1 KiB/wave) read down the
// target axis of one sA row.  The row's likelihoods and 1/D stay in registers between the two passes, so every byte
// of the cube is read from HBM exactly once: algorithmic traffic = Q*(K+1)*ldT*8 bytes per sweep.
// Reductions: compensated (TwoSum) butterfly for the answer weight W_k (it feeds a division whose result goes through
// log2), plain fp64 butterflies for the entropy / velocity / lack sums; cross-wave through 2 LDS slots.
// The epilogue (weighted averages over answers, velocity component, integer powers) is the reference's scalar code,
// run by one lane with the reference's Kahan lane order.
//
// This is a reduction with a nonlinear inner function (table log2 + two divisions per element): no MFMA.
#include "pqa_device.h"
#include "pqa_kernels.h"

namespace pqa {

static __device__ double gLog2Table[1024];

hipError_t UploadLog2Table(const double *hostTable) {
  return hipMemcpyToSymbol(HIP_SYMBOL(gLog2Table), hostTable, 1024 * sizeof(double));
}

struct EvalArgs {
  const double *cube;
  const double *prior;
  const uint32_t *tgap;
  const uint32_t *qgap;
  const uint32_t *asked;
  double *priority;
  int64_t K, ldT, qFirst, qLimit, nValidPlus1;
};

// Reference epilogue, PqaCore/CEEvalQsSubtaskConsider.cpp:134-207.  mW/mH/mV: per-answer weight, entropy, velocity^2.
__device__ __forceinline__ double eval_epilogue(const double *mW, const double *mH, const double *mV, int64_t K,
                                                double lackSum, int64_t nValidPlus1) {
  Kahan1 accTotW;
  accTotW.init(0.0);
  // 4-lane Kahan accumulators accAvgH / accAvgV (:139-172): answer k lands in lane k & 3, in k order
  double hS[4] = {0, 0, 0, 0}, hC[4] = {0, 0, 0, 0}, vS[4] = {0, 0, 0, 0}, vC[4] = {0, 0, 0, 0};
  const int64_t nVectorized = (K >> 2) << 2;
  for (int64_t k = 0; k < K; k++) {
    accTotW.add(mW[k]);                                        // :89
    const int c = (int)(k & 3);
    const double wh = mW[k] * mH[k];                           // :152 / :167
    const double wv = mW[k] * sqrt(mV[k]);                     // :156-157 / :165-167
    {
      const double y = wh - hC[c];
      const double t = hS[c] + y;
      hC[c] = (t - hS[c]) - y;
      hS[c] = t;
    }
    {
      const double y = wv - vC[c];
      const double t = vS[c] + y;
      vC[c] = (t - vS[c]) - y;
      vS[c] = t;
    }
    // A full vector Add (:153,:158) also Kahan-adds into lanes that received their value in the same instruction;
    // a tail scalar Add (:170-171) touches one lane only.  Per lane the sequence of adds is the same either way.
    (void)nVectorized;
  }
  const double totW = accTotW.get();                           // :134
  const double avgH = precise_sum4(hS, hC) / totW;             // :175-177 (PairSum == two PreciseSums side by side)
  const double avgV = precise_sum4(vS, vC) / totW;
  const double nExpectedTargets = exp2(avgH);                  // :181
  const double cLnMaxV = 0.34657359027997265470861606072909;   // SRMath::_cLnSqrt2
  const double lnV = (avgV == 0) ? -746.0 : log(avgV);         // :29
  const double nT = (double)nValidPlus1;
  const double vComp = 1 / (cLnMaxV - lnV + cLnMaxV / (nT * nT));  // :30-32
  const double lack = -lackSum;                                // :201
  const double v2 = vComp * vComp, v4 = v2 * v2, v8 = v4 * v4, v9 = v8 * vComp;  // :207 with integer powers (:206)
  return lack * v9 * (1.0 / (nExpectedTargets * nExpectedTargets));
}

// ------------------------------------------------------------------------------------------------------------------
// Register-resident sweep: WPQ waves per question, NP target pairs per lane.  Requires ldT <= 128*WPQ*NP.
// ------------------------------------------------------------------------------------------------------------------
template <int WPQ, int NP>
__global__ __launch_bounds__(WPQ * 64) void eval_questions_f64(EvalArgs a) {
  constexpr int kThreads = WPQ * kWave;
  extern __shared__ double smem[];
  double *tbl = smem;                              // 1024 doubles
  double *red = tbl + 1024;                        // 2 * WPQ * 3 doubles of reduction scratch
  double *mets = red + 2 * WPQ * 3;                // 3 * K doubles: W_k, H_k, V2_k
  const int tid = threadIdx.x;
  for (int i = tid; i < 1024; i += kThreads) tbl[i] = gLog2Table[i];

  const int64_t ldT = a.ldT, K = a.K;
  const int64_t nPairs = ldT >> 1;
  // Per-lane constants of the sweep: masked priors and gap flags of the lane's targets.
  double2 pr[NP];
  uint32_t gapBits = 0;  // bit 2j / 2j+1 : target pair j element 0 / 1 is a gap (or beyond the row)
  int pidx[NP];
#pragma unroll
  for (int j = 0; j < NP; j++) {
    const int p = tid + j * kThreads;
    const bool inRow = p < nPairs;
    pidx[j] = inRow ? p : (int)(nPairs - 1);       // clamped: out-of-row lanes re-read the last pair and are masked
    const int64_t t0 = 2 * (int64_t)pidx[j];
    const bool g0 = !inRow || bit_test(a.tgap, t0), g1 = !inRow || bit_test(a.tgap, t0 + 1);
    gapBits |= (g0 ? 1u : 0u) << (2 * j) | (g1 ? 1u : 0u) << (2 * j + 1);
    const double2 pv = reinterpret_cast<const double2 *>(a.prior)[pidx[j]];
    pr[j].x = g0 ? 0.0 : pv.x;                     // :103 andnot(gapMask, prior)
    pr[j].y = g1 ? 0.0 : pv.y;
  }
  __syncthreads();

  int phase = 0;
  const int64_t qStride = (K + 1) * ldT;
  for (int64_t q = a.qFirst + blockIdx.x; q < a.qLimit; q += gridDim.x) {
    if (bit_test(a.qgap, q) || bit_test(a.asked, q)) {         // :54
      if (tid == 0) a.priority[q - a.qFirst] = 0;
      continue;
    }
    const double *qBase = a.cube + q * qStride;
    const double2 *rowD = reinterpret_cast<const double2 *>(qBase + K * ldT);
    const double2 *rowA = reinterpret_cast<const double2 *>(qBase);
    double2 invD[NP], aCur[NP];
#pragma unroll
    for (int j = 0; j < NP; j++) {
      const double2 d = rowD[pidx[j]];
      aCur[j] = rowA[pidx[j]];
      invD[j].x = ((gapBits >> (2 * j)) & 1) ? 0.0 : 1.0 / d.x;        // :74 andnot(gapMask, 1/D)
      invD[j].y = ((gapBits >> (2 * j + 1)) & 1) ? 0.0 : 1.0 / d.y;
    }
    double accL = 0;
    for (int64_t k = 0; k < K; k++) {
      // ---- pass 1 (:66-87): likelihoods into registers, W_k
      double2 lh[NP];
      Comp w = {0.0, 0.0};
#pragma unroll
      for (int j = 0; j < NP; j++) {
        lh[j].x = (aCur[j].x * invD[j].x) * pr[j].x;           // :81-82 (gap lanes: invD = 0 and prior = 0)
        lh[j].y = (aCur[j].y * invD[j].y) * pr[j].y;
        comp_add(w, lh[j].x);
        comp_add(w, lh[j].y);
      }
      // prefetch the next answer's row while this one is reduced and log2'ed
      if (k + 1 < K) {
        const double2 *rowN = reinterpret_cast<const double2 *>(qBase + (k + 1) * ldT);
#pragma unroll
        for (int j = 0; j < NP; j++) aCur[j] = rowN[pidx[j]];
      }
      const double Wk = block_sum_comp<WPQ>(w, red, phase);    // :88
      const double invWk = 1.0 / Wk;                           // :91
      // ---- pass 2 (:95-128)
      double hv[2] = {0, 0};  // entropy sum, velocity sum
#pragma unroll
      for (int j = 0; j < NP; j++) {
        {
          const double p = lh[j].x * invWk;                    // :97
          const double l2 = log2hot(p, tbl);                   // :106 (gap lanes: p = 0 -> -1023, contributes -0)
          hv[0] = fma(p, l2, hv[0]);                           // :113-114
          accL += (invD[j].x * invD[j].x) / l2;                // :117
          const double d = p - pr[j].x;                        // :119
          hv[1] = fma(d, d, hv[1]);                            // :126-127
        }
        {
          const double p = lh[j].y * invWk;
          const double l2 = log2hot(p, tbl);
          hv[0] = fma(p, l2, hv[0]);
          accL += (invD[j].y * invD[j].y) / l2;
          const double d = p - pr[j].y;
          hv[1] = fma(d, d, hv[1]);
        }
      }
      block_sum<WPQ, 2>(hv, red, phase);
      if (tid == 0) {
        mets[k] = Wk;                                          // :90
        mets[K + k] = -hv[0];                                  // :130-131
        mets[2 * K + k] = hv[1];                               // :132
      }
    }
    double lv[1] = {accL};
    block_sum<WPQ, 1>(lv, red, phase);
    if (tid == 0) a.priority[q - a.qFirst] = eval_epilogue(mets, mets + K, mets + 2 * K, K, lv[0], a.nValidPlus1);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Streaming fallback for rows too long to keep on chip (ldT > 16384): one 256-thread workgroup per question, pass 2
// re-reads the sA row (served by L2 / Infinity Cache when it can).  Same arithmetic per element.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void eval_questions_f64_stream(EvalArgs a) {
  constexpr int WPQ = 4, kThreads = 256;
  extern __shared__ double smem[];
  double *tbl = smem;
  double *red = tbl + 1024;
  double *mets = red + 2 * WPQ * 3;
  const int tid = threadIdx.x;
  for (int i = tid; i < 1024; i += kThreads) tbl[i] = gLog2Table[i];
  __syncthreads();
  const int64_t ldT = a.ldT, K = a.K;
  const int64_t qStride = (K + 1) * ldT;
  int phase = 0;
  for (int64_t q = a.qFirst + blockIdx.x; q < a.qLimit; q += gridDim.x) {
    if (bit_test(a.qgap, q) || bit_test(a.asked, q)) {
      if (tid == 0) a.priority[q - a.qFirst] = 0;
      continue;
    }
    const double *qBase = a.cube + q * qStride;
    const double *rowD = qBase + K * ldT;
    double accL = 0;
    for (int64_t k = 0; k < K; k++) {
      const double *rowA = qBase + k * ldT;
      Comp w = {0.0, 0.0};
      for (int64_t t = tid; t < ldT; t += kThreads) {
        const bool g = bit_test(a.tgap, t);
        const double invD = g ? 0.0 : 1.0 / rowD[t];
        const double pri = g ? 0.0 : a.prior[t];
        comp_add(w, (rowA[t] * invD) * pri);
      }
      const double Wk = block_sum_comp<WPQ>(w, red, phase);
      const double invWk = 1.0 / Wk;
      double hv[2] = {0, 0};
      for (int64_t t = tid; t < ldT; t += kThreads) {
        const bool g = bit_test(a.tgap, t);
        const double invD = g ? 0.0 : 1.0 / rowD[t];
        const double pri = g ? 0.0 : a.prior[t];
        const double p = ((rowA[t] * invD) * pri) * invWk;
        const double l2 = log2hot(p, tbl);
        hv[0] = fma(p, l2, hv[0]);
        accL += (invD * invD) / l2;
        const double d = p - pri;
        hv[1] = fma(d, d, hv[1]);
      }
      block_sum<WPQ, 2>(hv, red, phase);
      if (tid == 0) {
        mets[k] = Wk;
        mets[K + k] = -hv[0];
        mets[2 * K + k] = hv[1];
      }
    }
    double lv[1] = {accL};
    block_sum<WPQ, 1>(lv, red, phase);
    if (tid == 0) a.priority[q - a.qFirst] = eval_epilogue(mets, mets + K, mets + 2 * K, K, lv[0], a.nValidPlus1);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// launch
// ------------------------------------------------------------------------------------------------------------------
namespace {

struct Variant {
  int id, wpq, np;
  const char *name;
};
// capacity in targets = 128 * wpq * np
const Variant kVariants[] = {
    {1, 1, 8, "wave_per_question_np8"},   {2, 4, 2, "wg256_np2"},  {3, 4, 4, "wg256_np4"},   {4, 4, 8, "wg256_np8"},
    {5, 8, 8, "wg512_np8"},               {6, 16, 5, "wg1024_np5"}, {7, 16, 8, "wg1024_np8"}, {8, 2, 4, "wg128_np4"},
    {9, 8, 2, "wg512_np2"},               {99, 4, 0, "stream256"},
};

int pick_variant(int64_t ldT, int variant) {
  if (variant != 0) return variant;
  if (ldT <= 1024) return 2;
  if (ldT <= 2048) return 3;
  if (ldT <= 4096) return 4;
  if (ldT <= 8192) return 5;
  if (ldT <= 10240) return 6;
  if (ldT <= 16384) return 7;
  return 99;
}

template <int WPQ, int NP>
hipError_t launch_reg(const EvalArgs &args, int64_t nQ, hipStream_t stream) {
  const size_t shmem = (1024 + 2 * WPQ * 3 + 3 * (size_t)args.K) * sizeof(double);
  // Enough workgroups to fill 256 CUs several times over; questions beyond the grid are grid-strided.
  const int64_t maxBlocks = 256 * 16;
  const unsigned grid = (unsigned)(nQ < maxBlocks ? nQ : maxBlocks);
  hipLaunchKernelGGL((eval_questions_f64<WPQ, NP>), dim3(grid), dim3(WPQ * 64), shmem, stream, args);
  return hipGetLastError();
}

}  // namespace

const char *EvalVariantName(const KbView &kb, int variant) {
  const int v = pick_variant(kb.ldT, variant);
  for (const Variant &x : kVariants)
    if (x.id == v) return x.name;
  return "unknown";
}

hipError_t LaunchEvalQuestions(const KbView &kb, const double *prior, const uint32_t *asked, int64_t qFirst,
                               int64_t qLimit, double *priority, int variant, hipStream_t stream) {
  if (qLimit <= qFirst) return hipSuccess;
  EvalArgs args;
  args.cube = kb.cube;
  args.prior = prior;
  args.tgap = kb.tgap;
  args.qgap = kb.qgap;
  args.asked = asked;
  args.priority = priority;
  args.K = kb.K;
  args.ldT = kb.ldT;
  args.qFirst = qFirst;
  args.qLimit = qLimit;
  args.nValidPlus1 = kb.nValidTargets + 1;  // PqaCore/CEEvalQsSubtaskConsider.cpp:191
  const int64_t nQ = qLimit - qFirst;
  const int v = pick_variant(kb.ldT, variant);
  int wpq = 0, np = 0;
  for (const Variant &x : kVariants)
    if (x.id == v) { wpq = x.wpq; np = x.np; }
  if (v != 99 && (wpq == 0 || kb.ldT > (int64_t)128 * wpq * np)) return hipErrorInvalidValue;
  switch (v) {
    case 1: return launch_reg<1, 8>(args, nQ, stream);
    case 2: return launch_reg<4, 2>(args, nQ, stream);
    case 3: return launch_reg<4, 4>(args, nQ, stream);
    case 4: return launch_reg<4, 8>(args, nQ, stream);
    case 5: return launch_reg<8, 8>(args, nQ, stream);
    case 6: return launch_reg<16, 5>(args, nQ, stream);
    case 7: return launch_reg<16, 8>(args, nQ, stream);
    case 8: return launch_reg<2, 4>(args, nQ, stream);
    case 9: return launch_reg<8, 2>(args, nQ, stream);
    case 99: {
      const size_t shmem = (1024 + 2 * 4 * 3 + 3 * (size_t)args.K) * sizeof(double);
      const int64_t maxBlocks = 256 * 8;
      const unsigned grid = (unsigned)(nQ < maxBlocks ? nQ : maxBlocks);
      hipLaunchKernelGGL(eval_questions_f64_stream, dim3(grid), dim3(256), shmem, stream, args);
      return hipGetLastError();
    }
    default: return hipErrorInvalidValue;
  }
}

}  // namespace pqa
