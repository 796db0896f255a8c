// pole_device.h -- rows at the pole of the lack term, re-evaluated the reference's way (shared by the single-quiz sweep,
// eval_kernels.hip: pole_fix, and the batched sweeps, batch_kernels.hip).  See the comment above pole_fix for what, why and what
// it costs; reference: PqaCore/CEEvalQsSubtaskConsider.cpp:62-132, SRPlatform/Interface/SRAccumVectDbl256.h:40-46, :62-92,
// SRPlatform/Interface/SRVectMath.h:87-135.
#pragma once
#include "pqa_device.h"
#include "eval_device.h"

namespace pqa {

constexpr uint32_t kNearOneHi = 0x3FEFFFF0u;   // high word of 1 - 2^-17

// SRVectMath.h:87-135, operation for operation (oracle: orc_log2hot).  tbl: the Log2Hot table in global memory ({log2 midpoint,
// 1 / (2 midpoint)} per bucket), entry0: its entry 0 as the REFERENCE has it (SRVectMath.cpp:31,42 -- the table's own is re-seated
// for the division-free log2hot, hip_engine.cpp).
__device__ __forceinline__ double log2hot_ref(double x, const double *tbl, double entry0) {
  const uint64_t ux = d2u(x);
  const double z = u2d((ux & ~kExpMaskUp) | kExp0Up);          // :88-89
  const int32_t high32 = (int32_t)(ux >> 32);                  // :92-94
  const int32_t normExps = (high32 >> 20) - 1023;              // :97-98
  const int32_t idx = (high32 >> 10) & 1023;                   // :101-102
  const double y = idx == 0 ? entry0 : tbl[2 * idx];           // :105-106
  const double m = u2d((1ULL << 41) | (d2u(z) & ~((1ULL << 42) - 1)));   // :108
  const double t = div_fast(z - m, z + m);                     // :111-114 (the exact quotient: div_nr, pqa_device.h)
  const double t2 = t * t;                                     // :115
  const double t3 = t * t2;                                    // :117
  const double terms01 = fma(1.0 / 3, t3, t);                  // :118
  const double log2z = fma(terms01, 2.8853900817779268147198493620038, y);   // :122
  return log2z + (double)normExps;                             // :131-133
}

struct PoleRows {
  const double *cube, *prior;    // [Q][K+1][ldT]; the quiz's posterior (gap targets are masked here)
  const uint32_t *tgap;
  int64_t K, T, ldT;
  const double *tblGlobal;       // log2hot_ref's table
  double entry0Ref;
};

// One question (index q of the cube) whose largest posterior element is within 2^-17 of 1.  rec: its sums as the sweep formed them
// -- W_k [K] | W_k sqrt(V_k) or V_k [K] (secondIsWV) | sum l log2 p | lack sum.  All threads of the workgroup; the Log2Hot table
// must be at LDS address 0 (log2hot); red: LDS, 4 x waves (at least 8) doubles; stage: LDS, 4 ceil(T / 4) doubles.  Thread 0 puts
// the reference-order W_k of the rows at the pole into rec and returns in dH / dL what their near-1 elements change in the
// entropy and lack sums.
template <bool COH>
__device__ __forceinline__ void pole_fix_question(const PoleRows &g, int64_t q, double *rec, bool secondIsWV, double *red, double *stage,
                                                  double &dH, double &dL) {
  const int tid = threadIdx.x, nThreads = blockDim.x, lane = tid % kWave, wave = tid / kWave, nWaves = nThreads / kWave;
  const int64_t K = g.K, ldT = g.ldT, nT = 4 * ((g.T + 3) >> 2);
  auto prior_at = [&](int64_t t) { return COH ? __hip_atomic_load(g.prior + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : g.prior[t]; };
  const double *qBase = g.cube + q * (K + 1) * ldT, *rowD = qBase + K * ldT;
  for (int64_t k = 0; k < K; k++) {
    const double *rowA = qBase + k * ldT;
    Comp c{0.0, 0.0};
    double mx = 0.0, mxId = 0.0;
    for (int64_t tb = tid; tb < nT; tb += 4 * nThreads) {    // (four targets per thread and round, their loads requested together)
      double av[4], dv[4], pv[4];
      bool live[4];
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const int64_t t = tb + e * nThreads;
        live[e] = t < nT && !bit_test(g.tgap, t);
        const int64_t tc = live[e] ? t : 0;
        av[e] = rowA[tc];
        dv[e] = rowD[tc];
        pv[e] = prior_at(tc);
      }
#pragma unroll
      for (int e = 0; e < 4; e++) {
        if (!live[e]) continue;
        const double id = div_nr(1.0, dv[e]);                // :74
        const double l = (av[e] * id) * pv[e];               // :81-82, as pass 1 forms it
        c = comp_merge(c, Comp{l, 0.0});
        if (l > mx) { mx = l; mxId = id; }
      }
    }
    c = wave_sum_comp(c);
    double wmx = mx;
    for (int m = kWave / 2; m >= 1; m >>= 1) wmx = fmax(wmx, __shfl_xor(wmx, m, kWave));
    if (lane == 0) { red[2 * wave] = c.s; red[2 * wave + 1] = c.c; red[2 * nWaves + 2 * wave] = 0.0; red[2 * nWaves + 2 * wave + 1] = 0.0; }
    if (mx == wmx && mx > 0.0) { red[2 * nWaves + 2 * wave] = mx; red[2 * nWaves + 2 * wave + 1] = mxId; }   // (behind lane 0's zeros)
    __syncthreads();
    // is THIS row at the pole?  (every thread decides, from the same numbers)
    Comp tot{red[0], red[1]};
    double cand = red[2 * nWaves], candId = red[2 * nWaves + 1];
    for (int w = 1; w < nWaves; w++) {
      tot = comp_merge(tot, Comp{red[2 * w], red[2 * w + 1]});
      if (red[2 * nWaves + 2 * w] > cand) { cand = red[2 * nWaves + 2 * w]; candId = red[2 * nWaves + 2 * w + 1]; }
    }
    const double Wc = tot.s + tot.c;                         // the row's sum to the last place or one short of the reference's
    const bool atPole = cand > 0.0 && (uint32_t)(d2u(cand * div_nr(1.0, Wc)) >> 32) >= kNearOneHi - 1;
    __syncthreads();                                         // (red is written again below)
    if (atPole) {
      // W_k in the REFERENCE'S ORDER (:66-88): the row's likelihoods into LDS, every thread its share; then one lane per Kahan
      // lane c takes the targets 4j + c in order (SRAccumVectDbl256.h:40-46) and PreciseSum (:62-92) folds the four.  The
      // correctly rounded sum would do in 70 - 90 % of such rows (tests/test_oracle.py), not in all: the compensation of a lane
      // that meets the large element after smaller ones is itself rounded.
      for (int64_t tb = tid; tb < nT; tb += 4 * nThreads) {
        double av[4], dv[4], pv[4];
        bool in[4], gap[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const int64_t t = tb + e * nThreads;
          in[e] = t < nT;
          const int64_t tc = in[e] ? t : 0;
          gap[e] = bit_test(g.tgap, tc);
          av[e] = rowA[tc];
          dv[e] = rowD[tc];
          pv[e] = prior_at(tc);
        }
#pragma unroll
        for (int e = 0; e < 4; e++)
          if (in[e]) stage[tb + e * nThreads] = gap[e] ? 0.0 : (av[e] * div_nr(1.0, dv[e])) * pv[e];   // :72-82
      }
      __syncthreads();
      if (tid < 4) {
        double sum = 0.0, corr = 0.0;
        const double *src = stage + tid;
        int64_t j = 0;
        for (; j + 8 <= nT / 4; j += 8) {                    // (eight elements requested at once, added in order)
          double x[8];
#pragma unroll
          for (int e = 0; e < 8; e++) x[e] = src[4 * (j + e)];
#pragma unroll
          for (int e = 0; e < 8; e++) {
            const double y = x[e] - corr;
            const double u = sum + y;
            corr = (u - sum) - y;
            sum = u;
          }
        }
        for (; j < nT / 4; j++) {
          const double y = src[4 * j] - corr;
          const double u = sum + y;
          corr = (u - sum) - y;
          sum = u;
        }
        red[tid] = sum;
        red[4 + tid] = corr;
      }
      __syncthreads();
      if (tid == 0) {
        const double Wx = precise_sum4(red, red + 4);        // :88
        const double invWx = div_nr(1.0, Wx);                // :91
        if ((uint32_t)(d2u(cand * invWx) >> 32) >= kNearOneHi) {
          const double Wf = rec[k];                          // the sweep's W_k
          const double lFast = log2hot(cand * div_nr(1.0, Wf), nullptr);   // what pass 2 took for this element (the table is at LDS address 0)
          const double lRef = log2hot_ref(cand * invWx, g.tblGlobal, g.entry0Ref);     // :106
          dH += cand * lRef - cand * lFast;                  // :113-114
          const double id2 = candId * candId;
          dL += div_fast(id2, lRef) - div_fast(id2, lFast);  // :117 (pass 2's quotient was within 2^-48.8 of the second one)
          if (secondIsWV) rec[K + k] = Wx * div_fast(rec[K + k], Wf);   // W_k sqrt(V_k): the velocity sum stays the sweep's
          rec[k] = Wx;
        }
      }
      __syncthreads();
    }
  }
}

}  // namespace pqa
