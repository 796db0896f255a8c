// eval_device.h -- the per-question epilogue of the priority sweep (reference PqaCore/CEEvalQsSubtaskConsider.cpp:134-207) and its
// scalar helpers, shared by the single-quiz sweep (eval_kernels.hip) and the batched / fp32 sweeps (batch_kernels.hip).  Always
// fp64: one lane runs it per (question, quiz), whatever the precision the elements were accumulated in.
#pragma once
#include "pqa_device.h"

namespace pqa {

// Natural logarithm of a positive double for the epilogue: m in [sqrt(1/2), sqrt(2)), s = (m-1)/(m+1),
// log x = e ln2 + 2s (1 + s^2/3 + ... + s^20/21), |error| < 2 ulp.  A third of the instructions (and of the dependent
// latency) of the library routine, and none of its register footprint, which would cost the short-row sweep a wave of
// occupancy; the reference's std::log (MSVC CRT) is not pinned by any of its tests either.
__device__ __forceinline__ double log_pos(double x) {
  if (!(x < __builtin_huge_val())) return x;                   // +inf, nan
  int e = -1023;
  if (x < 2.2250738585072014e-308) {                           // subnormal: rescale by 2^54
    x *= 18014398509481984.0;
    e -= 54;
  }
  const uint64_t ux = d2u(x);
  e += (int)(ux >> 52);
  double m = u2d((ux & 0x000FFFFFFFFFFFFFULL) | kExp0Up);
  if (m > 1.4142135623730951) { m *= 0.5; e += 1; }
  const double s = div_nr(m - 1.0, m + 1.0);
  const double z = s * s;
  double p = 1.0 / 21;
  p = fma(p, z, 1.0 / 19);
  p = fma(p, z, 1.0 / 17);
  p = fma(p, z, 1.0 / 15);
  p = fma(p, z, 1.0 / 13);
  p = fma(p, z, 1.0 / 11);
  p = fma(p, z, 1.0 / 9);
  p = fma(p, z, 1.0 / 7);
  p = fma(p, z, 1.0 / 5);
  p = fma(p, z, 1.0 / 3);
  const double s2 = s + s;
  const double lm = fma(s2 * z, p, s2);
  const double de = (double)e;
  return fma(de, 6.93147180369123816490e-01, fma(de, 1.90821492927058770002e-10, lm));
}

// exact quotient (div_nr) when the divisor is an ordinary number, the hardware's IEEE sequence otherwise
__device__ __forceinline__ double div_fast(double n, double d) {
  const double ad = __builtin_fabs(d);
  return (ad > 1e-290 && ad < 1e290) ? div_nr(n, d) : n / d;
}

// Reference epilogue, PqaCore/CEEvalQsSubtaskConsider.cpp:134-207.  mW: per-answer weights W_k; mWV: W_k * sqrt(V2_k)
// (:156-157 / :165-167; the callers form these products, lane-parallel over k where they can);
// whSum = sum_k W_k * H_k, which the sweep accumulates directly as -sum_{k,t} l_kt * log2(p_kt) (W_k * p_kt == l_kt up to
// the rounding of p = l * (1/W_k)), so the per-answer entropies H_k are never materialised.
// vCompTail = ln(sqrt 2) / (nValidTargets + 1)^2 (:191), computed once on the host with the same two operations.
// One lane runs this per question, so it is written for latency: exact-quotient divisions, the short logarithm above.
// stride: distance in doubles between consecutive answers' entries of mW / mWV (the batched sweep keeps them one lane apart).
__device__ __forceinline__ double eval_epilogue_strided(const double *mW, double whSum, const double *mWV, int64_t K,
                                                        double lackSum, double vCompTail, int64_t stride) {
  Kahan1 accTotW;
  accTotW.init(0.0);
  // 4-lane Kahan accumulator accAvgV (:140-172): answer k lands in lane k & 3, in k order.  A full vector Add (:158) and
  // a tail scalar Add (:171) give every lane the same sequence of Kahan steps.
  double vS[4] = {0, 0, 0, 0}, vC[4] = {0, 0, 0, 0};
  for (int64_t k = 0; k < K; k++) {
    accTotW.add(mW[k * stride]);                                        // :89
    const int c = (int)(k & 3);
    const double y = mWV[k * stride] - vC[c];                           // :158 / :171
    const double t = vS[c] + y;
    vC[c] = (t - vS[c]) - y;
    vS[c] = t;
  }
  const double totW = accTotW.get();                           // :134
  const double avgH = div_fast(whSum, totW);                   // :175-177
  const double avgV = div_fast(precise_sum4(vS, vC), totW);
  const double nExpectedTargets = exp2(avgH);                  // :181
  const double cLnMaxV = 0.34657359027997265470861606072909;   // SRMath::_cLnSqrt2
  const double lnV = (avgV == 0) ? -746.0 : log_pos(avgV);     // :29
  const double vComp = div_fast(1.0, cLnMaxV - lnV + vCompTail);   // :30-32
  const double lack = -lackSum;                                // :201
  const double v2 = vComp * vComp, v4 = v2 * v2, v8 = v4 * v4, v9 = v8 * vComp;  // :207 with integer powers (:206)
  return lack * v9 * div_fast(1.0, nExpectedTargets * nExpectedTargets);
}
__device__ __forceinline__ double eval_epilogue(const double *mW, double whSum, const double *mWV, int64_t K,
                                                double lackSum, double vCompTail) {
  return eval_epilogue_strided(mW, whSum, mWV, K, lackSum, vCompTail, 1);
}

}  // namespace pqa
