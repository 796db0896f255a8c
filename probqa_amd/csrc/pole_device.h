// pole_device.h -- rows at the pole of the lack term, re-evaluated the reference's way (shared by the single-quiz sweep,
// eval_kernels.hip: pole_fix, and the batched sweeps, batch_kernels.hip).  See the comment above pole_fix for what, why and what
// it costs; reference: PqaCore/CEEvalQsSubtaskConsider.cpp:62-132, SRPlatform/Interface/SRAccumVectDbl256.h:40-46, :62-92,
// SRPlatform/Interface/SRVectMath.h:87-135.
#pragma once
#include "pqa_device.h"
#include "eval_device.h"
#include "pqa_kernels.h"

namespace pqa {

// A row is redone in the reference's order when its largest posterior element is at least 1 - 2^-10.  Two mechanisms set the bar.
// The lack term's pole: the summation order of W_k moves the priority by ~1e-16 / (1 - p): 1e-9 at p = 1 - 1e-7.  The velocity
// term reaches further out: (p - prior)^2 of a target that holds nearly all the mass BEFORE and AFTER the answer is the square of
// a difference of two numbers next to 1 -- |p - prior| ~ (1 - p) x (how much the answer tells this target from the rest) -- and
// p's last place (1.1e-16) is 1e-8 of it at p = 1 - 4e-5, where the likeliest answer's velocity is a sixth of the question's:
// 1.3e-9 of the priority (tests/test_gpu_late.py: the first cases that found it).  At 1 - 2^-10 both are below 1e-11.
constexpr uint32_t kNearOneHi = 0x3FEFF800u;   // high word of 1 - 2^-10
// The velocity term again, further out still: an answer that tells a question's targets apart by less than one part in 10^5 -- the
// near-certain answer of a late quiz, or of a question with few answers -- leaves the posterior where the prior was, and the
// row's velocity sum V_k = sum (p - prior)^2 is then the square of ONE difference |d| ~ sqrt(V_k) carrying p's last place:
// 2.2e-16 / sqrt(V_k) relative.  Rows with V_k <= kSmallV whose largest element holds at least a quarter of the mass (below that no
// single difference makes up the sum) are redone as well: what is left is below 1.1e-11 of a row's velocity.
constexpr double kSmallV = 4e-10;
constexpr uint32_t kQuarterHi = 0x3FCE0000u;   // high word of 0.234: the element of a listed row whose terms are corrected

// a sweep's entry for a question that passed its watch (pqa_kernels.h: PoleHeader; one thread)
__device__ __forceinline__ uint32_t pole_list_append(PoleHeader *list, uint32_t q, uint32_t rowMask, uint32_t b) {
  const uint32_t at = atomicAdd(&list->count, 1u);
  reinterpret_cast<PoleEntry *>(list + 1)[at] = PoleEntry{q, rowMask, b, 0u};
  return at;
}

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

// the wave's maximum in every lane (the DPP / permlane steps of wave_sum: six ds_bpermute round trips otherwise)
__device__ __forceinline__ double wave_max_d(double v) {
  v = fmax(v, mov_dpp<kDppXor1>(v));
  v = fmax(v, mov_dpp<kDppXor2>(v));
  v = fmax(v, mov_dpp<kDppHalfMirror>(v));
  v = fmax(v, mov_dpp<kDppMirror>(v));
  Pair p = swap16(v);
  v = fmax(p.a, p.b);
  p = swap32(v);
  return fmax(p.a, p.b);
}

struct PoleRows {
  const double *cube, *prior;    // [Q][K+1][ldT]; the quiz's posterior (gap targets are masked here)
  const uint32_t *tgap;
  int64_t K, T, ldT;
  const double *tblGlobal;       // log2hot_ref's table
  double entry0Ref;
};

// One question (index q of the cube) whose largest posterior element is within 2^-17 of 1.  rec: its sums as the sweep formed them
// -- W_k [K] | W_k sqrt(V_k) or V_k [K] (secondIsWV) | sum l log2 p | lack sum.  rowMask: the answer rows in which the sweep saw such
// an element (bit k), or 0: not known -- then every row is looked at first (a compensated sum and its largest likelihood).  All
// threads of the workgroup; the Log2Hot table must be at LDS address 0 (log2hot); red: LDS, redDoubles >= 6 x waves + 8 doubles;
// stage: LDS, stageDoubles >= 4 ceil(T / 4).  Thread 0 puts the reference-order W_k of the rows at the pole into rec and returns in
// dH / dL what their near-1 elements change in the entropy and lack sums.
// In a late quiz -- the posterior on one target -- EVERY answer row of a question is at the pole (the target's likelihood is all of
// W_k whatever the answer), and a row's reference-order sum is T / 4 DEPENDENT Kahan steps on four lanes: so the rows at the pole
// are staged side by side -- as many as stage and red have room for -- and their chains run at the same time, four lanes each, in
// one wave: five rows cost one chain's time, not five.
template <bool COH>
__device__ __forceinline__ void pole_fix_question(const PoleRows &g, int64_t q, double *rec, bool secondIsWV, uint32_t rowMask, double *red,
                                                  int redDoubles, double *stage, int stageDoubles, double &dH, double &dL) {
  const int tid = threadIdx.x, nThreads = blockDim.x, lane = tid % kWave, wave = tid / kWave, nWaves = nThreads / kWave;
  const int64_t K = g.K, ldT = g.ldT, nT = 4 * ((g.T + 3) >> 2);
  auto prior_at = [&](int64_t t) { return COH ? __hip_atomic_load(g.prior + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : g.prior[t]; };
  const double *qBase = g.cube + q * (K + 1) * ldT, *rowD = qBase + K * ldT;
  // red: per row of a batch the waves' largest likelihood and its 1/D [2 x waves] and the chains' results [8]
  const int perRow = 2 * nWaves + 8;
  int maxRows = (int)min((int64_t)(stageDoubles / nT), (int64_t)(redDoubles / perRow));
  maxRows = maxRows > 16 ? 16 : maxRows;                     // (the chains of a batch: four lanes each, one wave)
  if (maxRows < 1 || redDoubles < 4 * nWaves) return;        // (no room: the sweep's own sums stand)
  // W_k of a batch of rows in the REFERENCE'S ORDER (:66-88): the rows' likelihoods into LDS, every thread its share (and the
  // largest of them with its 1/D: the element whose terms are replaced); then four lanes per row take the targets 4j + c in order
  // (SRAccumVectDbl256.h:40-46) and PreciseSum (:62-92) folds the four.  The correctly rounded sum would do in 70 - 90 % of such
  // rows (tests/test_oracle.py), not in all: the compensation of a lane that meets the large element after smaller ones is itself
  // rounded.
  auto run_batch = [&](int64_t base, uint32_t batch, int nb) __attribute__((always_inline)) {   // rows base + (the bits of batch)
    int r = 0;
    for (uint32_t rest = batch; rest != 0; rest &= rest - 1, r++) {
      const int64_t k = base + __builtin_ctz(rest);
      const double *rowA = qBase + k * ldT;
      double *dst = stage + (int64_t)r * nT;
      double mx = 0.0, mxId = 0.0;
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
          if (in[e]) {
            const double id = div_nr(1.0, dv[e]);            // :74
            const double l = gap[e] ? 0.0 : (av[e] * id) * pv[e];   // :72-82, as pass 1 forms it
            dst[tb + e * nThreads] = l;
            if (l > mx) { mx = l; mxId = id; }
          }
      }
      const double wmx = wave_max_d(mx);
      double *cw = red + r * perRow + 2 * wave;
      if (lane == 0) { cw[0] = 0.0; cw[1] = 0.0; }
      if (mx == wmx && mx > 0.0) { cw[0] = mx; cw[1] = mxId; }   // (behind lane 0's zeros; lanes that tie hold the same element's values or an equal one's)
    }
    __syncthreads();
    if (tid < 4 * nb) {
      double sum = 0.0, corr = 0.0;
      const double *src = stage + (int64_t)(tid >> 2) * nT + (tid & 3);
      int64_t j = 0;
      for (; j + 8 <= nT / 4; j += 8) {                      // (eight elements requested at once, added in order)
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
      double *out = red + (tid >> 2) * perRow + 2 * nWaves;
      out[tid & 3] = sum;
      out[4 + (tid & 3)] = corr;
    }
    __syncthreads();
    if (tid == 0) {
      int rr = 0;
      for (uint32_t rest = batch; rest != 0; rest &= rest - 1, rr++) {
        const int64_t k = base + __builtin_ctz(rest);
        const double *cw = red + rr * perRow;
        double cand = cw[0], candId = cw[1];
        for (int w = 1; w < nWaves; w++)
          if (cw[2 * w] > cand) { cand = cw[2 * w]; candId = cw[2 * w + 1]; }
        const double Wx = precise_sum4(cw + 2 * nWaves, cw + 2 * nWaves + 4);   // :88
        const double invWx = div_nr(1.0, Wx);                // :91
        if (cand > 0.0 && (uint32_t)(d2u(cand * invWx) >> 32) >= kNearOneHi) {
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
    }
    __syncthreads();
  };
  uint32_t poleMask = K <= 31 ? rowMask : 0u;
  if (poleMask == 0) {
    // ---- which rows are at the pole is not known: a compensated sum of every row and its largest likelihood say
    for (int64_t k = 0; k < K; k++) {
      const double *rowA = qBase + k * ldT;
      Comp c{0.0, 0.0};
      double mx = 0.0;
      for (int64_t tb = tid; tb < nT; tb += 4 * nThreads) {  // (four targets per thread and round, their loads requested together)
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
          const double l = (av[e] * div_nr(1.0, dv[e])) * pv[e];   // :74, :81-82
          c = comp_merge(c, Comp{l, 0.0});
          mx = fmax(mx, l);
        }
      }
      c = wave_sum_comp(c);
      const double wmx = wave_max_d(mx);
      if (lane == 0) { red[3 * wave] = c.s; red[3 * wave + 1] = c.c; red[3 * wave + 2] = wmx; }
      __syncthreads();
      Comp tot{red[0], red[1]};                              // (every thread decides, from the same numbers)
      double cand = red[2];
      for (int w = 1; w < nWaves; w++) {
        tot = comp_merge(tot, Comp{red[3 * w], red[3 * w + 1]});
        cand = fmax(cand, red[3 * w + 2]);
      }
      const double Wc = tot.s + tot.c;                       // the row's sum to the last place or one short of the reference's
      const bool atPole = cand > 0.0 && (uint32_t)(d2u(cand * div_nr(1.0, Wc)) >> 32) >= kNearOneHi - 1;
      __syncthreads();                                       // (red is written again)
      if (atPole) {
        if (maxRows == 1 || K > 31) run_batch(k, 1u, 1);     // (one row's room: now, while its cache lines are warm; dozens of answers: no mask of them)
        else poleMask |= 1u << k;
      }
    }
  }
  while (poleMask != 0) {
    uint32_t batch = 0;
    int nb = 0;
    for (uint32_t rest = poleMask; rest != 0 && nb < maxRows; rest &= rest - 1, nb++) batch |= rest & (0u - rest);
    poleMask &= ~batch;
    run_batch(0, batch, nb);
  }
}

}  // namespace pqa
