// batch_kernels.hip -- the priority sweep for MANY quizzes at once, and the fp32 sweeps, on CDNA4 (gfx950).
//
// Same arithmetic per (question, answer, target) element as CEEvalQsSubtaskConsider<>::Run (reference:
// PqaCore/CEEvalQsSubtaskConsider.cpp:41-217) -- B independent NextQuestion sweeps over one knowledge base (the reference
// serves concurrent quizzes one sweep each under a shared read lock, PqaCore/CpuEngine.cpp:357-361; BASELINE configs[4]:
// "256 concurrent quizzes batched along a leading dim").
//
// Shape.  The single-quiz sweep (eval_kernels.hip) maps lanes to TARGETS and pays a wave reduction per row; run once per
// quiz it also reads the cube once per quiz.  Here a lane is a QUIZ: wave w, lane l of a workgroup owns quiz 64 w + l for
// the whole launch, so every sum over targets is a private serial sum in that lane's registers -- no cross-lane traffic
// at all -- and a cube element, fetched once, serves all B quizzes:
//   * the workgroup stages a tile of the cube in LDS -- for QB questions, KG answers and TC targets the products
//     c = A * (1/D) (the reference's first multiplication, :81, shared by every quiz) and 1/D^2 (:117) -- computed once
//     per batch instead of once per quiz;
//   * every lane walks the tile target by target: the tile values are wave-uniform (one LDS address for all 64 lanes:
//     a broadcast read, no bank conflicts), the lane's own operand is its quiz's masked prior, read from a transposed
//     staging matrix PT[target][quiz] (one coalesced 256-byte row per wave and target);
//   * pass 1 (W_k = sum_t c * prior, :66-88) and pass 2 (posterior, log2, entropy / lack / velocity sums, :95-128) each walk
//     the row chunk by chunk; a row that fits one tile (ldT <= TC) is staged once for both passes.
// Cube traffic per batch: Q (K+1) ldT s bytes once (twice for rows longer than a tile: pass 2 re-stages), whatever B is --
// the previous batched launch (grid.y = quiz) read it B times.  The work is B Q K T element evaluations and for B >= 64 the
// sweep is VALU-bound: ~31 fp64 / ~15 fp32 VALU slots per element against 9.6 / 4.8 bytes per B elements.
//
// Precision (R = double | float): R is the type of the cube, of the staged tile, of the transposed priors and of the
// per-element arithmetic.  fp64 uses the reference's Log2Hot (pqa_device.h).  fp32 uses the hardware's v_log_f32 (1 ulp)
// clamped to [-127, -2^-25/ln 2]: the Float analogue of Log2Hot(0) = -1023 (biased exponent field 0, SRVectMath.h:96-98)
// and of its deliberately negative value at 1 (SRVectMath.cpp:41-43; the lack term divides by it).  Sums over targets run
// in R over one chunk (<= TC terms) and are folded into fp64 totals per chunk; the per-question epilogue (:134-207: weighted
// averages, exp2, log, ninth power) is always fp64 (eval_device.h) -- vComp^9 alone overflows fp32 for 10^5 targets.
//
// No MFMA: a reduction with a table / transcendental inner function.  (Pass 1 alone is a contraction, but it is 1 of ~15
// slots per element.)
#include <algorithm>

#include "eval_device.h"
#include "pole_device.h"
#include "pqa_device.h"
#include "pqa_kernels.h"

namespace pqa {

static __device__ double gLog2TableB[kLog2TableDoubles];  // this translation unit's copy of the Log2Hot table

hipError_t UploadLog2TableBatch(const double *hostTable) {
  return hipMemcpyToSymbol(HIP_SYMBOL(gLog2TableB), hostTable, kLog2TableDoubles * sizeof(double));
}

namespace {

constexpr int kTileThreads = 256;   // prep kernel

// ---- PT[t][Bp] = quiz b's prior at target t, masked by the target gaps (:103), in R; columns b >= nSlots are zero ------
template <typename R>
__global__ __launch_bounds__(kTileThreads) void batch_prep_kernel(const QuizSlot *__restrict__ slots, int nSlots, int Bp,
                                                                  const uint32_t *__restrict__ tgap, int64_t ldT, R *__restrict__ PT) {
  __shared__ double tile[64][65];
  const int64_t t0 = (int64_t)blockIdx.x * 64;
  const int b0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;   // 4 rows of 64 per pass
  for (int r = ty; r < 64; r += 4) {                        // r: quiz within the tile, tx: target (coalesced along t)
    const int b = b0 + r;
    const int64_t t = t0 + tx;
    double v = 0.0;
    if (b < nSlots && t < ldT && !bit_test(tgap, t)) v = slots[b].prior[t];
    tile[r][tx] = v;
  }
  __syncthreads();
  for (int r = ty; r < 64; r += 4) {                        // r: target within the tile, tx: quiz (coalesced along b)
    const int64_t t = t0 + r;
    if (t < ldT && b0 + tx < Bp) PT[t * Bp + b0 + tx] = (R)tile[tx][r];
  }
}

// ---- per-element arithmetic ------------------------------------------------------------------------------------------
template <typename R> struct Num;
template <> struct Num<double> {
  static constexpr bool kTable = true;
  static __device__ __forceinline__ double log2p(double p, const double *tbl) { return log2hot(p, tbl); }
  static __device__ __forceinline__ double rcp(double x) {    // 2^-48.8: below the rounding of the sum it feeds
    double r = __builtin_amdgcn_rcp(x);
    return fma(r, fma(-x, r, 1.0), r);
  }
  static __device__ __forceinline__ double inv(double x) { return div_nr(1.0, x); }   // exact quotient (:74, :91)
};
template <> struct Num<float> {
  static constexpr bool kTable = false;
  static __device__ __forceinline__ float log2p(float p, const double *) {
    // v_log_f32: log2, 1 ulp; 0 -> -inf, clamped to the Float analogue of Log2Hot's range (see the file comment)
    return __builtin_amdgcn_fmed3f(__builtin_amdgcn_logf(p), -127.0f, -4.2992253e-08f);
  }
  static __device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }   // 1 ulp
  static __device__ __forceinline__ float inv(float x) { return 1.0f / x; }
};

// The lane's priors, requested ahead of their use: body(tc, prior of target tc) for tc in [0, tcN), tcN even; ptc = the lane's
// column of PT at the tile's first target.  A load whose value is consumed an iteration later is placed by the scheduler right
// before that use (it shortens the live range), which puts a full L2 round trip in front of every target -- measured, the waves
// of the 12500 x 5 x 100000 sweep sat in s_waitcnt for 48 % of their cycles.  The scheduling barrier right behind each load
// keeps it where it is written: two targets (one loop iteration) ahead of its use; the compiler's own wait-count bookkeeping
// stays in charge of the waits.  The empty asm statements keep the vectorizer from pairing the two unrolled targets'
// operations lane by lane (both targets' values live at once: 256 VGPRs and spills).
template <typename R, typename F>
__device__ __forceinline__ void walk_targets(const R *ptc, int Bp, int tcN, F &&body) {
  R pA = ptc[0], pB = ptc[Bp];
#pragma unroll 1
  for (int tc = 0; tc < tcN; tc += 2) {
    const int nA = tc + 2 < tcN ? tc + 2 : tcN - 1, nB = tc + 3 < tcN ? tc + 3 : tcN - 1;   // (the tail re-reads the last target)
    const R a = pA;
    pA = ptc[(size_t)nA * Bp];
    __builtin_amdgcn_sched_barrier(0);
    body(tc, a);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    const R b = pB;
    pB = ptc[(size_t)nB * Bp];
    __builtin_amdgcn_sched_barrier(0);
    body(tc + 1, b);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  }
}

// The same walk with D targets' priors in flight: for pass 1, whose 20 multiply-adds per target do not cover a load's latency two
// targets ahead (12500 x 5 x 100000 fp32, 256 quizzes, same box: 538 ms per sweep two ahead, 526 eight ahead, 518 sixteen ahead;
// pass 2 with four ahead instead of two: 546 -- its loop unrolled by four is the worse code).
template <int D, typename R, typename F>
__device__ __forceinline__ void walk_targets_deep(const R *ptc, int Bp, int tcN, F &&body) {
  R p[D];
#pragma unroll
  for (int d = 0; d < D; d++) p[d] = ptc[(size_t)(d < tcN ? d : tcN - 1) * Bp];
#pragma unroll 1
  for (int tc = 0; tc < tcN; tc += D) {
#pragma unroll
    for (int d = 0; d < D; d++) {
      const int nx = tc + D + d < tcN ? tc + D + d : tcN - 1;     // (the tail re-reads the last target)
      const R v = p[d];
      p[d] = ptc[(size_t)nx * Bp];
      __builtin_amdgcn_sched_barrier(0);
      if (tc + d < tcN) body(tc + d, v);
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

struct BatchArgs {
  const void *cube;          // R [Q][K+1][ldT]
  const void *PT;            // R [ldT][Bp]
  const uint32_t *tgap, *qgap;
  const QuizSlot *slots;
  int nSlots, Bp;
  int Bq;                    // lanes per question group: lane = (group g = tid / Bq, quiz b = tid % Bq); Bq <= Bp, Bq divides the threads
  int64_t K, Q, ldT;
  int64_t qBegin, qEnd;      // the questions of THIS launch (a sweep is one launch, or a main launch and one for the last questions: LaunchEvalBatch)
  int TC;                    // targets per tile (even)
  double vCompTail;          // ln(sqrt 2) / (nValidTargets + 1)^2 (:191)
  double *acc;               // fp64 totals, [grid][threads][QB][2K+2]: W_k (K), V_k (K), sum W_k H_k, lack
  BatchRecord *recs;         // [grid][Bp]: every workgroup's best question per quiz
  double *priorityT;         // optional [Q][Bp]: the priorities themselves (tests, EvalPrioritiesBatch)
  // the pole watch (Double engines; pole_kernels.hip redoes the listed (question, quiz) pairs behind the sweep): the list, and
  // per entry the pair's sums as the epilogue left them -- W_k | W_k sqrt(V_k) | sum l log2 p | lack; null: no watch
  PoleHeader *poleList;
  double *poleSums;
};

// Tile in LDS: R tile[TC][G * QB][KG + 1] (G question groups of QB questions each -- see eval_batch_kernel); entry [tc][qi][k < KG] =
// A[q][kg + k][t] * invD[q][t], entry [tc][qi][KG] = invD^2.
// Gap targets hold zeros in both (:74, :79 andnot masks; the reference masks the lack term instead, :117).
template <typename R, int QB, int KG>
__device__ __forceinline__ void stage_tile(const BatchArgs &a, R *tile, int G, int64_t q0, int64_t kg, int kN, int64_t t0, int tcN) {
  const R *cube = static_cast<const R *>(a.cube);
  const int64_t ldT = a.ldT, K = a.K;
  const int QT = G * QB;
  for (int idx = threadIdx.x; idx < tcN * G; idx += blockDim.x) {
    const int gg = G == 1 ? 0 : idx / tcN, tc = idx - gg * tcN;   // (neighbouring threads: neighbouring targets of one question group)
    const int64_t t = t0 + tc;
    const bool gap = bit_test(a.tgap, t);
#pragma unroll
    for (int qi = 0; qi < QB; qi++) {
      const int64_t qq = q0 + gg * QB + qi;
      const int64_t q = qq < a.qEnd ? qq : a.qEnd - 1;        // (beyond the last question: a copy whose results are dropped)
      const R *qb = cube + q * (K + 1) * ldT;
      // (non-temporal: the cube streams through a workgroup once per pass.  12500 x 5 x 100000 fp32, 256 quizzes, same box: 556.7 ->
      //  545.4 ms per sweep; the L2s then keep less of what misses them -- FETCH_SIZE 168 -> 189 GB per sweep, served by the
      //  Infinity Cache, which holds the quizzes' 102 MB of priors whole)
      const R d = __builtin_nontemporal_load(qb + K * ldT + t);
      const R invD = gap ? (R)0 : Num<R>::inv(d);             // :74
      R *dst = tile + ((size_t)tc * QT + gg * QB + qi) * (KG + 1);
#pragma unroll
      for (int k = 0; k < KG; k++) dst[k] = k < kN ? __builtin_nontemporal_load(qb + (kg + k) * ldT + t) * invD : (R)0;   // :81 (A * invD)
      dst[KG] = invD * invD;                                   // :117
    }
  }
}

// One workgroup: lane tid <-> (question group g = tid / Bq, quiz b = tid % Bq).  A full batch (Bq = 256 lanes) has one group: a
// lane is a quiz.  A smaller batch would leave the workgroup with one or two waves and the chip with a fraction of its waves (32
// quizzes on 12500 x 5 x 100000, fp32: 379 ms per sweep where 256 quizzes take 552): the freed lanes take further questions --
// G = threads / Bq groups side by side, each with its own QB questions of the block's G * QB; the tile holds all of them, a lane
// reads its group's entries (one LDS address per group and wave), the lanes of a quiz read the same prior.
// Questions in blocks of G * QB consecutive local indices, answers in groups of KG (K <= KG: one group).
// EXACT: K == KG, so the one answer group is full and nothing in the element loops depends on a run-time answer count.
template <typename R, int QB, int KG, bool EXACT>
__global__ __launch_bounds__(256) void eval_batch_kernel(BatchArgs a) {
  extern __shared__ double smem[];
  const double *tbl = smem;
  R *tile = reinterpret_cast<R *>(smem + (Num<R>::kTable ? kLog2TableDoubles : 0));
  if constexpr (Num<R>::kTable) {
    if (!lds_table_at_zero(tbl)) __builtin_trap();            // log2hot addresses the table absolutely
    for (int i = threadIdx.x; i < kLog2TableDoubles; i += blockDim.x) smem[i] = gLog2TableB[i];
  }
  const int tid = threadIdx.x, nThreads = blockDim.x;
  const int Bq = a.Bq, G = nThreads / Bq;
  const int g = G == 1 ? 0 : tid / Bq, b = tid - g * Bq;       // question group and quiz of this lane
  const bool live = b < a.nSlots;
  const int64_t K = a.K, ldT = a.ldT;
  const int Bp = a.Bp, TC = a.TC;
  const int QT = G * QB;
  const R *tileLane = tile + (size_t)g * QB * (KG + 1);        // the lane's group within a target's entries
  const size_t tileStride = (size_t)QT * (KG + 1);
  const int nChunks = (int)((ldT + TC - 1) / TC);
  const R *pt = static_cast<const R *>(a.PT) + b;
  const uint32_t *asked = live ? a.slots[b].asked : a.qgap;   // (idle lanes: any valid words)
  const int nAcc = (int)(2 * K + 2);
  // entry (qi, r) of this lane at acc[qi * nAcc + r]: one contiguous record per lane, so that every entry is the lane's base
  // pointer plus a (wave-uniform, mostly compile-time) offset -- with entries nThreads apart the 48 addresses were hoisted out of
  // the tile loops as loop invariants and held 96 VGPRs through them
  double *acc = a.acc + ((size_t)blockIdx.x * nThreads + tid) * (size_t)(QB * nAcc);
  double bestP = 0.0;
  int64_t bestQ = -1;
  const int64_t nBlocks = (a.qEnd - a.qBegin + QT - 1) / QT;
  constexpr bool kPoleWatch = Num<R>::kTable;                  // (Double engines: the fp32 tolerance covers what the summation order moves)
  for (int64_t blk = blockIdx.x; blk < nBlocks; blk += gridDim.x) {
    const int64_t qBlk = a.qBegin + blk * QT, q0 = qBlk + (int64_t)g * QB;   // the block's first question; this lane's first question
    uint32_t hiMax[QB];                                        // the largest posterior element's high word, per question (pole_device.h)
#pragma unroll
    for (int qi = 0; qi < QB; qi++) hiMax[qi] = 0;
    bool staged = false;                                       // single-tile rows, single answer group: pass 2 reuses pass 1's tile
    for (int64_t kg = 0; kg < K; kg += KG) {
      const int kN = EXACT ? KG : (int)(K - kg < KG ? K - kg : KG);
      // ---- pass 1 (:66-88): W_k = sum_t (A * invD) * prior
      double Wd[QB][KG];
#pragma unroll
      for (int qi = 0; qi < QB; qi++)
#pragma unroll
        for (int k = 0; k < KG; k++) Wd[qi][k] = 0.0;
      for (int ch = 0; ch < nChunks; ch++) {
        const int64_t t0 = (int64_t)ch * TC;
        const int tcN = (int)(ldT - t0 < TC ? ldT - t0 : TC);
        __syncthreads();                                       // everybody is done with the previous tile
        stage_tile<R, QB, KG>(a, tile, G, qBlk, kg, kN, t0, tcN);
        __syncthreads();
        R W[QB][KG];
#pragma unroll
        for (int qi = 0; qi < QB; qi++)
#pragma unroll
          for (int k = 0; k < KG; k++) W[qi][k] = (R)0;
        walk_targets_deep<16, R>(pt + t0 * Bp, Bp, tcN, [&](int tc, R pi) __attribute__((always_inline)) {
          const R *c = tileLane + (size_t)tc * tileStride;
#pragma unroll
          for (int qi = 0; qi < QB; qi++)
#pragma unroll
            for (int k = 0; k < KG; k++)
              if (EXACT || k < kN) W[qi][k] = fma(c[qi * (KG + 1) + k], pi, W[qi][k]);   // :81-82, :85
        });
#pragma unroll
        for (int qi = 0; qi < QB; qi++)
#pragma unroll
          for (int k = 0; k < KG; k++) Wd[qi][k] += (double)W[qi][k];
      }
      staged = nChunks == 1;
      R invW[QB][KG];
#pragma unroll
      for (int qi = 0; qi < QB; qi++)
#pragma unroll
        for (int k = 0; k < KG; k++) {
          invW[qi][k] = (R)div_fast(1.0, Wd[qi][k]);            // :91
          if (EXACT || k < kN) acc[qi * nAcc + (int)kg + k] = Wd[qi][k];   // :90
        }
      // ---- pass 2 (:95-128)
      for (int ch = 0; ch < nChunks; ch++) {
        const int64_t t0 = (int64_t)ch * TC;
        const int tcN = (int)(ldT - t0 < TC ? ldT - t0 : TC);
        if (!staged) {
          __syncthreads();
          stage_tile<R, QB, KG>(a, tile, G, qBlk, kg, kN, t0, tcN);
          __syncthreads();
        }
        R v[QB][KG], hW[QB], accL[QB];
#pragma unroll
        for (int qi = 0; qi < QB; qi++) {
          hW[qi] = accL[qi] = (R)0;
#pragma unroll
          for (int k = 0; k < KG; k++) v[qi][k] = (R)0;
        }
        walk_targets<R>(pt + t0 * Bp, Bp, tcN, [&](int tc, R pi) __attribute__((always_inline)) {
          // the target's tile values -- QB x (KG + 1) wave-uniform numbers -- are all requested before the first of them is used:
          // behind the scheduling barrier of each question group they would be fetched group by group, an LDS round trip each
          const R *c = tileLane + (size_t)tc * tileStride;
          R cv[QB][KG + 1];
#pragma unroll
          for (int qi = 0; qi < QB; qi++)
#pragma unroll
            for (int k = 0; k <= KG; k++) cv[qi][k] = c[qi * (KG + 1) + k];
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int qi = 0; qi < QB; qi++) {
            const R id2 = cv[qi][KG];
            // one element: likelihood again (:81-82: cheaper than keeping it), posterior (:97), its log2 (:106), the entropy term
            // weighted by W_k (:113-114, eval_epilogue), the velocity term (:119, :126-127); returns log2 for the lack term
            auto element = [&](int k) __attribute__((always_inline)) {
              const R lh = cv[qi][k] * pi;
              const R p = lh * invW[qi][k];
              const R l2 = Num<R>::log2p(p, tbl);
              if constexpr (kPoleWatch) hiMax[qi] = max(hiMax[qi], (uint32_t)(d2u((double)p) >> 32));
              hW[qi] = fma(lh, l2, hW[qi]);
              const R d = p - pi;
              v[qi][k] = fma(d, d, v[qi][k]);
              return l2;
            };
            // :117 lack += invD^2 / log2(p): the answers of one question and target share invD^2, so they share ONE reciprocal --
            // sum_k 1/l_k = N / D with D = prod l_k and N = sum_k prod_{j != k} l_j, built up answer by answer (N <- N l + D,
            // D <- D l: two full-rate operations per answer for the quarter-rate instruction they replace).  All l_k < 0, so the
            // terms of N have one sign (no cancellation); in fp32 |l| is clamped to [4.3e-8, 127]: |D| in [1.5e-37, 3.3e10] for the
            // at most five answers of a group, in fp64 (|l| in [5.7e-20, 1023]) within [6e-97, 1.2e15].
            R accN = (R)1, accD = (R)1;
            bool any = false;
#pragma unroll
            for (int k = 0; k < KG; k++)
              if (EXACT || k < kN) {
                const R l = element(k);
                if (!any) { accD = l; any = true; }
                else { accN = fma(accN, l, accD); accD = accD * l; }
              }
            if (any) accL[qi] = fma(id2 * accN, Num<R>::rcp(accD), accL[qi]);
            __builtin_amdgcn_sched_barrier(0);   // one question's KG elements in flight at a time: enough independent chains to cover
                                                 // the transcendental latency, and a third of the registers of all QB x KG at once
          }
        });
        // fold the chunk's sums into the fp64 totals
        const bool first = ch == 0;
#pragma unroll
        for (int qi = 0; qi < QB; qi++) {
#pragma unroll
          for (int k = 0; k < KG; k++)
            if (EXACT || k < kN) {
              double *p = acc + qi * nAcc + (int)(K + kg) + k;
              *p = (first ? 0.0 : *p) + (double)v[qi][k];
            }
          double *ph = acc + qi * nAcc + (int)(2 * K), *pl = ph + 1;
          const bool firstOfQuestion = first && kg == 0;
          *ph = (firstOfQuestion ? 0.0 : *ph) + (double)hW[qi];
          *pl = (firstOfQuestion ? 0.0 : *pl) + (double)accL[qi];
        }
      }
    }
    // the pole watch's verdict per question: every row if an element came within 2^-10 of 1; the rows whose velocity sum all but
    // vanishes if one holds a quarter of the mass (pole_device.h)
    [[maybe_unused]] uint32_t listRows[QB];
    if constexpr (kPoleWatch) {
#pragma unroll
      for (int qi = 0; qi < QB; qi++) {
        listRows[qi] = 0;
        if (a.poleList != nullptr && hiMax[qi] >= kQuarterHi) {
          if (hiMax[qi] >= kNearOneHi) listRows[qi] = K >= 32 ? 0xFFFFFFFFu : (1u << K) - 1u;
          else
            for (int64_t k = 0; k < K; k++)
              if (acc[qi * nAcc + (int)(K + k)] <= kSmallV) listRows[qi] |= 1u << (k < 31 ? (int)k : 31);
        }
      }
    }
    // ---- epilogue (:134-207), one per (question, quiz), fp64
#pragma unroll 1
    for (int qi = 0; qi < QB; qi++) {
      const int64_t q = q0 + qi;
      if (q >= a.qEnd) break;
      const bool skip = bit_test(a.qgap, q) || ((asked[q >> 5] >> (q & 31)) & 1u);   // :54
      double pri = 0.0;
      if (!skip) {
        double *rec = acc + (size_t)qi * nAcc;
        // mWV[k] = W_k * sqrt(V_k) (:156-157) in place of V_k
        for (int64_t k = 0; k < K; k++) rec[K + k] = rec[k] * sqrt(rec[K + k]);
        pri = eval_epilogue(rec, -rec[2 * K], rec + K, K, rec[2 * K + 1], a.vCompTail);
      }
      if (live) {
        if (a.priorityT) a.priorityT[q * Bp + b] = pri;
        if (!skip) {
          const double cand = pri != pri ? -__builtin_huge_val() : pri;   // NaN never wins over a number
          if (bestQ < 0 || cand > bestP) { bestP = cand; bestQ = q; }     // (questions ascend: the lowest index wins a tie)
        }
      }
    }
    if constexpr (kPoleWatch) {
      if (a.poleList != nullptr) {
        // ---- the (question, quiz) pairs the watch has named: listed, with their sums, for the fix behind the sweep
#pragma unroll
        for (int qi = 0; qi < QB; qi++) {
          const int64_t q = q0 + qi;
          if (live && q < a.qEnd && listRows[qi] != 0 && !(bit_test(a.qgap, q) || ((asked[q >> 5] >> (q & 31)) & 1u))) {
            const uint32_t at = pole_list_append(a.poleList, (uint32_t)q, K <= 31 ? listRows[qi] : 0u, (uint32_t)b);
            const double *rec = acc + (size_t)qi * nAcc;       // (W_k sqrt(V_k) in place of V_k by now)
            double *dst = a.poleSums + (size_t)at * nAcc;
            for (int i = 0; i < nAcc; i++) dst[i] = rec[i];
          }
        }
      }
    }
  }
  a.recs[((size_t)blockIdx.x * G + g) * Bp + b] = BatchRecord{bestP, bestQ};   // (a record strip per workgroup and question group)
}

// ------------------------------------------------------------------------------------------------------------------
// The sweep for a FEW DOZEN quizzes at once, fp64, short rows (round 3).  Between the single-quiz sweep with grid.y = quiz
// (lanes over targets: a wave reduction per row and quiz, the cube re-read per quiz -- 11-14 us per quiz at 1000 x 5 x 1000) and
// the row-sharing sweep above (a lane is a quiz: no reductions, but 64 quizzes per wave or its lanes idle, and a lane walks the
// whole row serially: it needs ~200 quizzes to fill the chip) there was nothing for the batches a server with dozens of client
// threads produces (hip_engine_combine.cpp: Combine -- 5 to 60 quizzes per combined sweep).  Here a lane is a (quiz, chunk of the
// row): QS quiz slots x 64 / QS chunks per wave, four waves per workgroup, one QUESTION per workgroup at a time --
//   * the workgroup stages the question's whole row block in LDS once: c = A * (1/D) for every answer and 1/D^2 (the divisions
//     and the cube's bytes shared by all quizzes of the launch);
//   * lane (slot s, chunk c) walks targets c, c + nCh, c + 2 nCh ...: lanes of one chunk read one LDS address (broadcast), the 64 / QS
//     chunks of a wave read neighbouring entries (different banks); the lane's own operand is its quiz's masked prior out of
//     the transposed matrix PT[target][quiz] (batch_prep_kernel);
//   * every sum over targets is a serial sum over T / nCh targets in the lane's registers, folded over the chunks by one or two
//     permlane swaps inside the wave (the lanes of a quiz are 16 or 32 apart) and one LDS exchange between the four waves:
//     two short folds per question and quiz (W_k; then the K + 2 pass-2 sums) instead of a butterfly per row;
//   * one lane per quiz runs the fp64 epilogue and keeps the quiz's best question; priorities go to the priority matrix and /
//     or straight to the host as tagged records (QuizSlot::hostPriority), as the other sweeps deliver them.
// Per (question, quiz) ~31 VALU slots per element as the row-sharing sweep, against ~50 per element pair-half of the register
// form with its reductions.  Two to eight answers (five -- the configuration every benchmark of the reference uses -- with nothing depending on a
// run-time count); rows whose block fits the CU's LDS (2500 targets at five answers).
// ------------------------------------------------------------------------------------------------------------------
constexpr int kMidThreads = 256, kMidMaxK = 8;   // answers: 5 (exact) or any number up to 8

struct MidArgs {
  const double *cube;        // [Q][K+1][ldT]
  const double *PT;          // [ldT][Bp]
  const uint32_t *tgap, *qgap;
  const QuizSlot *slots;
  int nSlots, Bp;
  int64_t Q, ldT, K;
  double vCompTail;
  BatchRecord *recs;         // [grid.x][Bp]
  double *priorityT;         // optional [Q][Bp]
  uint64_t tag;              // launch tag of the host hand-over records
  PoleHeader *poleList;      // the pole watch (as BatchArgs'): the list and the listed pairs' sums; null: no watch
  double *poleSums;
};

template <int QS> __device__ __forceinline__ double chunk_fold(double v) {   // sum over the lanes of one quiz slot within a wave
  if constexpr (QS <= 8) v += mov_dpp<0x128>(v);             // row_ror:8 -- the lane eight away in the row of sixteen
  if constexpr (QS <= 16) { const Pair p = swap16(v); v = p.a + p.b; }
  if constexpr (QS <= 32) { const Pair p = swap32(v); v = p.a + p.b; }
  return v;
}

// The lane's priors of its targets chunk, chunk + nCh, ... -- `cnt` of them -- eight at a time, the NEXT eight requested before the
// current eight are worked on: the loads are L2 round trips (~0.4 us) and an iteration of pass 1 is five fused multiply-adds --
// one load ahead (the first version) made the sweep wait a round trip per target: 290 us per launch whatever the batch.
template <typename F>
__device__ __forceinline__ void walk_chunk(const double *pt, int Bp, int chunk, int nCh, int cnt, F &&body) {
  constexpr int U = 8;
  double cur[U], nxt[U];
  const size_t stride = (size_t)nCh * (size_t)Bp;
  const double *p = pt + (size_t)chunk * Bp;
#pragma unroll
  for (int e = 0; e < U; e++) cur[e] = p[(size_t)(e < cnt ? e : cnt - 1) * stride];
  for (int i0 = 0; i0 < cnt; i0 += U) {
    const bool more = i0 + U < cnt;
#pragma unroll
    for (int e = 0; e < U; e++) {
      const int i = i0 + U + e;
      nxt[e] = p[(size_t)(i < cnt ? i : cnt - 1) * stride];   // (the tail re-reads the lane's last target)
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < U; e++)
      if (i0 + e < cnt) body((i0 + e) * nCh + chunk, cur[e]);
    if (more) {
#pragma unroll
      for (int e = 0; e < U; e++) cur[e] = nxt[e];
    }
  }
}

// KM: the answers the arrays hold; EXACT: K == KM (nothing in the element loops depends on a run-time answer count)
template <int QS, int KM, bool EXACT>
__global__ __launch_bounds__(kMidThreads) void eval_midbatch_kernel(MidArgs a) {
  constexpr int K = KM, NW = kMidThreads / kWave, NSUB = kWave / QS, NCH = NW * NSUB;
  const int kN = EXACT ? KM : (int)a.K;                     // the answers there are
  extern __shared__ double smem[];
  const double *tbl = smem;
  if (!lds_table_at_zero(tbl)) __builtin_trap();            // log2hot addresses the table absolutely
  double *tile = smem + kLog2TableDoubles;                  // [ldT][K + 1]
  double *red = tile + (size_t)a.ldT * (K + 1);             // [NW][K + 2][QS]
  uint32_t *watchW = reinterpret_cast<uint32_t *>(red + (size_t)NW * (K + 2) * QS);   // [2][QS]: the rows of this question that passed the pole watch -- an element next to 1 | an element of a quarter -- per quiz slot
  for (int i = threadIdx.x; i < kLog2TableDoubles; i += kMidThreads) smem[i] = gLog2TableB[i];
  if (threadIdx.x < 2 * QS) watchW[threadIdx.x] = 0;
  const int tid = threadIdx.x, lane = tid % kWave, wave = tid / kWave;
  const int slot = lane % QS, chunk = wave * NSUB + lane / QS;
  const int b = blockIdx.y * QS + slot;                     // this lane's quiz
  const bool live = b < a.nSlots;
  const int64_t ldT = a.ldT;
  const int Bp = a.Bp;
  const double *pt = a.PT + (live ? b : 0);
  const QuizSlot qs = a.slots[live ? b : 0];
  const uint32_t *asked = live ? qs.asked : a.qgap;
  const bool head = wave == 0 && lane < QS;                 // the lane that runs this quiz's epilogues
  const int cnt = (int)((ldT - chunk + NCH - 1) / NCH);     // targets of this lane: chunk, chunk + NCH, ...
  double bestP = 0.0;
  int64_t bestQ = -1;
  for (int64_t q = blockIdx.x; q < a.Q; q += gridDim.x) {
    __syncthreads();                                        // everybody is done with the previous question's tile (and the table is in)
    {
      const double *qb = a.cube + q * (kN + 1) * ldT;
      for (int64_t t = tid; t < ldT; t += kMidThreads) {
        const double invD = bit_test(a.tgap, t) ? 0.0 : div_nr(1.0, qb[kN * ldT + t]);  // :74
        double *dst = tile + t * (K + 1);
#pragma unroll
        for (int k = 0; k < K; k++) dst[k] = (EXACT || k < kN) ? qb[k * ldT + t] * invD : 0.0;   // :81
        dst[K] = invD * invD;                                                             // :117
      }
    }
    __syncthreads();
    // ---- pass 1 (:66-88)
    double W[K];
#pragma unroll
    for (int k = 0; k < K; k++) W[k] = 0.0;
    walk_chunk(pt, Bp, chunk, NCH, cnt, [&](int tc, double pi) __attribute__((always_inline)) {
      const double *c = tile + (size_t)tc * (K + 1);
#pragma unroll
      for (int k = 0; k < K; k++) W[k] = fma(c[k], pi, W[k]);                             // :81-82, :85
    });
    double Wlane[K];                                        // (the pole watch: this lane's share of W_k)
#pragma unroll
    for (int k = 0; k < K; k++) {
      Wlane[k] = W[k];
      W[k] = chunk_fold<QS>(W[k]);
      if (lane < QS) red[(wave * (K + 2) + k) * QS + slot] = W[k];
    }
    __syncthreads();
    double invW[K];
    uint32_t poleRows = 0, quarterRows = 0;
#pragma unroll
    for (int k = 0; k < K; k++) {
      double w = 0.0;
#pragma unroll
      for (int w2 = 0; w2 < NW; w2++) w += red[(w2 * (K + 2) + k) * QS + slot];
      W[k] = w;                                                                           // :88-90 (the same bits in every lane of the quiz)
      invW[k] = (EXACT || k < kN) ? div_fast(1.0, w) : 0.0;                               // :91
      // an element within 2^-10 of 1 is nearly all of W_k, and so is the serial sum of the lane that holds it; likewise an element
      // of a quarter (eval_kernels.hip: the watch)
      if ((EXACT || k < kN) && Wlane[k] >= w * 0.25 && Wlane[k] > 0.0) {
        quarterRows |= 1u << k;
        if (Wlane[k] >= w * (1.0 - 0x1p-9)) poleRows |= 1u << k;
      }
    }
    if (quarterRows != 0 && a.poleList != nullptr) {                                        // (rare; read by the quiz's head lane behind the next barriers)
      atomicOr(&watchW[QS + slot], quarterRows);
      if (poleRows != 0) atomicOr(&watchW[slot], poleRows);
    }
    __syncthreads();                                        // (the exchange buffer is used again below)
    // ---- pass 2 (:95-128)
    double v[K], hW = 0.0, accL = 0.0;
#pragma unroll
    for (int k = 0; k < K; k++) v[k] = 0.0;
    walk_chunk(pt, Bp, chunk, NCH, cnt, [&](int tc, double pi) __attribute__((always_inline)) {
      const double *c = tile + (size_t)tc * (K + 1);
      double cv[K + 1];
#pragma unroll
      for (int k = 0; k <= K; k++) cv[k] = c[k];
      // :117 one reciprocal for the K lack terms of the target (see eval_batch_kernel: N / D built answer by answer)
      double accN = 1.0, accD = 1.0;
#pragma unroll
      for (int k = 0; k < K; k++) {
        if (!EXACT && k >= kN) break;
        const double lh = cv[k] * pi;                                                     // :81-82
        const double p = lh * invW[k];                                                    // :97
        const double l2 = Num<double>::log2p(p, tbl);                                     // :106
        hW = fma(lh, l2, hW);                                                             // :113-114 weighted by W_k (eval_epilogue)
        const double d = p - pi;                                                          // :119
        v[k] = fma(d, d, v[k]);                                                           // :126-127
        if (k == 0) accD = l2;
        else { accN = fma(accN, l2, accD); accD = accD * l2; }
      }
      accL = fma(cv[K] * accN, Num<double>::rcp(accD), accL);
    });
#pragma unroll
    for (int k = 0; k < K; k++) {
      v[k] = chunk_fold<QS>(v[k]);
      if (lane < QS) red[(wave * (K + 2) + k) * QS + slot] = v[k];
    }
    hW = chunk_fold<QS>(hW);
    accL = chunk_fold<QS>(accL);
    if (lane < QS) {
      red[(wave * (K + 2) + K) * QS + slot] = hW;
      red[(wave * (K + 2) + K + 1) * QS + slot] = accL;
    }
    __syncthreads();
    // ---- epilogue (:134-207), one lane per quiz
    if (head) {
      const bool skip = bit_test(a.qgap, q) || ((asked[q >> 5] >> (q & 31)) & 1u);        // :54
      double pri = 0.0;
      if (!skip) {
        double mW[K], mWV[K], sums[2];
#pragma unroll
        for (int r = 0; r < K + 2; r++) {
          double s2 = 0.0;
#pragma unroll
          for (int w2 = 0; w2 < NW; w2++) s2 += red[(w2 * (K + 2) + r) * QS + slot];
          if (r < K) { mW[r] = W[r]; mWV[r] = W[r] * sqrt(s2); }                          // :156-157
          else sums[r - K] = s2;
        }
        pri = eval_epilogue(mW, -sums[0], mWV, kN, sums[1], a.vCompTail);
        uint32_t listRows = watchW[slot];
        if (watchW[QS + slot] != 0) {
          // (rows with an element of a quarter whose velocity sum all but vanishes: mWV = W sqrt(V))
#pragma unroll
          for (int r = 0; r < K; r++)
            if ((EXACT || r < kN) && ((watchW[QS + slot] >> r) & 1u) && mWV[r] * mWV[r] <= kSmallV * (mW[r] * mW[r])) listRows |= 1u << r;
        }
        if (live && listRows != 0) {
          // the pair's sums as they are, for the fix behind the sweep (pole_kernels.hip)
          const uint32_t at = pole_list_append(a.poleList, (uint32_t)q, listRows, (uint32_t)b);
          double *dst = a.poleSums + (size_t)at * (2 * kN + 2);
#pragma unroll
          for (int r = 0; r < K; r++)
            if (EXACT || r < kN) { dst[r] = mW[r]; dst[kN + r] = mWV[r]; }
          dst[2 * kN] = sums[0];
          dst[2 * kN + 1] = sums[1];
        }
      }
      watchW[slot] = 0;                                       // (set again only behind the next question's barriers)
      watchW[QS + slot] = 0;
      if (live) {
        if (a.priorityT) a.priorityT[q * Bp + b] = pri;
        if (!skip) {
          if (qs.hostPriority != nullptr) {   // the host's selector: {priority, launch tag}, one write-through store (eval_kernels.hip: flush_pending)
            typedef unsigned int u4 __attribute__((ext_vector_type(4)));
            const uint64_t w0 = d2u(pri), w1 = a.tag;
            const u4 x = {(unsigned)w0, (unsigned)(w0 >> 32), (unsigned)w1, (unsigned)(w1 >> 32)};
            asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(qs.hostPriority + q), "v"(x) : "memory");
          }
          const double cand = pri != pri ? -__builtin_huge_val() : pri;                   // NaN never wins over a number
          if (bestQ < 0 || cand > bestP) { bestP = cand; bestQ = q; }                     // (questions ascend: the lowest index wins a tie)
        }
      }
    }
  }
  if (head && b < Bp) a.recs[(size_t)blockIdx.x * Bp + b] = BatchRecord{bestP, bestQ};
}

// ---- every quiz's winner over the workgroups' records; the result and then the flag go to host-coherent memory ----------
// One wave per quiz: lane l merges records l, l + 64, ... (each a workgroup's best for this quiz), then the lanes' bests are merged
// by shuffles.  (Until round 3 one THREAD per quiz walked all ~500 records, a dependent load each: 150 us behind every batched
// sweep -- a tenth of the 256-quiz sweep at 1000 x 5 x 1000, more than the whole sweep for the batches of a few dozen quizzes.)
// dirty (optional, [nSlots]): quizzes with a priority the fix behind the sweep has changed (pole_kernels.hip) -- their workgroups'
// records are stale, the winner comes out of the quiz's column of the priority matrix; the mark is cleared for the next launch.
__global__ __launch_bounds__(64) void batch_pick_kernel(const BatchRecord *__restrict__ recs, int nRecs, int Bp,
                                                        const QuizSlot *__restrict__ slots, int nSlots, int64_t outBase,
                                                        uint64_t flagValue, uint32_t *dirty, const double *__restrict__ priorityT,
                                                        int64_t Q, const uint32_t *__restrict__ qgap) {
  const int b = blockIdx.x, lane = threadIdx.x;
  if (b >= nSlots) return;
  double bp = 0.0;
  int64_t bq = -1;
  if (dirty != nullptr && dirty[b] != 0) {
    const uint32_t *asked = slots[b].asked;
    for (int64_t q = lane; q < Q; q += 64) {
      if (bit_test(qgap, q) || bit_test(asked, q)) continue;
      double p = priorityT[(size_t)q * Bp + b];
      if (p != p) p = -__builtin_huge_val();                  // NaN never wins over a number
      if (bq < 0 || p > bp) { bp = p; bq = q; }               // (q ascending: the first of equal values stays)
    }
    if (lane == 0) dirty[b] = 0;
  } else
  for (int g = lane; g < nRecs; g += 64) {
    const BatchRecord r = recs[(size_t)g * Bp + b];
    if (r.index >= 0 && (bq < 0 || r.priority > bp || (r.priority == bp && r.index < bq))) { bp = r.priority; bq = r.index; }
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    const double op = __shfl_xor(bp, m, 64);
    const int64_t oq = __shfl_xor(bq, m, 64);
    if (oq >= 0 && (bq < 0 || op > bp || (op == bp && oq < bq))) { bp = op; bq = oq; }
  }
  if (lane != 0) return;
  const QuizSlot s = slots[b];
  s.out->priority = bq < 0 ? 0.0 : bp;
  s.out->index = bq < 0 ? -1 : bq + outBase;
  if (s.seq != nullptr) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");             // system scope: the record before the flag
    __hip_atomic_store(s.seq, flagValue, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// ---- single-quiz sweep for Float engines: one 256-thread workgroup per question, lanes over targets (16 bytes per lane and
// load), the answer row read again by pass 2 (from L2: a row is 4 T bytes).  LDSROW (rows of up to kF32LdsTargets targets): the
// masked prior, converted to fp32 once per workgroup, and the question's 1/D, computed once per question (v_rcp_f32 + one
// Newton step: 2^-22.5 -> full fp32 precision), live in LDS; longer rows recompute both per element.  The fp32 twin of the
// streaming form of eval_kernels.hip -- Double engines have the register-resident shapes there; the Float configuration the
// work went into is the batched one above (BASELINE configs[4]).
constexpr int kF32LdsTargets = 16384;
__device__ __forceinline__ float wave_sum_f(float v) { return wave_sum_f32(v); }
__device__ __forceinline__ float rcp_f32_nr(float x) {
  const float r = __builtin_amdgcn_rcpf(x);
  return fmaf(r, fmaf(-x, r, 1.0f), r);
}
// NT threads per workgroup: 256 for short rows (several workgroups per CU), 1024 for long ones (the row's two LDS vectors allow one
// workgroup per CU: sixteen waves instead of four keep its SIMDs busy).
template <bool LDSROW, int NT>
__global__ __launch_bounds__(NT) void eval_questions_f32_stream(const float *__restrict__ cube, const double *__restrict__ prior,
                                                                 const uint32_t *__restrict__ tgap, const uint32_t *__restrict__ qgap,
                                                                 const uint32_t *__restrict__ asked, double *__restrict__ priority,
                                                                 int64_t K, int64_t Q, int64_t ldT, double vCompTail) {
  constexpr int NW = NT / kWave;
  extern __shared__ double smem[];      // W_k [K] | W_k sqrt(V_k) [K] | partials [4][16 floats] | LDSROW: prior [ldT] | 1/D [ldT] (floats)
  double *wk = smem, *wv = smem + K;
  float *part = reinterpret_cast<float *>(smem + 2 * K);
  float4 *prL = reinterpret_cast<float4 *>(part + 64), *idL = prL + (ldT >> 2);
  const int tid = threadIdx.x, lane = tid % kWave, wave = tid / kWave;
  const int64_t nQuads = ldT >> 2;      // ldT is a multiple of 32 floats
  auto prior4 = [&](int64_t i) __attribute__((always_inline)) {   // masked prior of targets 4i .. 4i+3 (:103)
    const uint32_t g = tgap[i >> 3] >> ((4 * i) & 31);
    const double2 a = reinterpret_cast<const double2 *>(prior)[2 * i], b = reinterpret_cast<const double2 *>(prior)[2 * i + 1];
    return make_float4((g & 1) ? 0.f : (float)a.x, (g & 2) ? 0.f : (float)a.y, (g & 4) ? 0.f : (float)b.x, (g & 8) ? 0.f : (float)b.y);
  };
  auto invd4 = [&](const float4 *rowD, int64_t i) __attribute__((always_inline)) {   // masked 1/D (:74)
    const uint32_t g = tgap[i >> 3] >> ((4 * i) & 31);
    const float4 d = rowD[i];
    return make_float4((g & 1) ? 0.f : rcp_f32_nr(d.x), (g & 2) ? 0.f : rcp_f32_nr(d.y), (g & 4) ? 0.f : rcp_f32_nr(d.z),
                       (g & 8) ? 0.f : rcp_f32_nr(d.w));
  };
  if constexpr (LDSROW)
    for (int64_t i = tid; i < nQuads; i += NT) prL[i] = prior4(i);
  for (int64_t q = blockIdx.x; q < Q; q += gridDim.x) {
    if (bit_test(qgap, q) || bit_test(asked, q)) {
      if (tid == 0) priority[q] = 0.0;
      continue;
    }
    const float *qb = cube + q * (K + 1) * ldT;
    const float4 *rowD = reinterpret_cast<const float4 *>(qb + K * ldT);
    __syncthreads();                                          // the previous question is done with the LDS rows
    if constexpr (LDSROW)
      for (int64_t i = tid; i < nQuads; i += NT) idL[i] = invd4(rowD, i);
    __syncthreads();
    double hWd = 0.0, accLd = 0.0;
    for (int64_t k = 0; k < K; k++) {
      const float4 *rowA = reinterpret_cast<const float4 *>(qb + k * ldT);
      float s = 0.f;
      for (int64_t i = tid; i < nQuads; i += NT) {
        const float4 a = rowA[i];
        float4 id, pr;
        if constexpr (LDSROW) { id = idL[i]; pr = prL[i]; } else { id = invd4(rowD, i); pr = prior4(i); }
        s += (a.x * id.x) * pr.x;                              // :81-82
        s += (a.y * id.y) * pr.y;
        s += (a.z * id.z) * pr.z;
        s += (a.w * id.w) * pr.w;
      }
      s = wave_sum_f(s);
      __syncthreads();                                        // the previous answer's partials have been read
      if (lane == 0) part[wave] = s;
      __syncthreads();
      float Wk = 0.f;
#pragma unroll
      for (int w = 0; w < NW; w++) Wk += part[w];
      const float invWk = 1.0f / Wk;
      float v = 0.f, hW = 0.f, accL = 0.f;
      auto element = [&](float a, float id, float pi) __attribute__((always_inline)) {
        const float lh = (a * id) * pi;
        const float p = lh * invWk;                            // :97
        const float l2 = Num<float>::log2p(p, nullptr);        // :106
        hW = fmaf(lh, l2, hW);                                 // :113-114 weighted by W_k
        const float d = p - pi;                                // :119
        v = fmaf(d, d, v);
        return l2;
      };
      for (int64_t i = tid; i < nQuads; i += NT) {
        const float4 a = rowA[i];
        float4 id, pr;
        if constexpr (LDSROW) { id = idL[i]; pr = prL[i]; } else { id = invd4(rowD, i); pr = prior4(i); }
        // :117 lack += invD^2 / log2 p, two targets per reciprocal: (ix^2 lb + iy^2 la) / (la lb)
        const float la = element(a.x, id.x, pr.x), lb = element(a.y, id.y, pr.y);
        accL = fmaf(fmaf(id.x * id.x, lb, (id.y * id.y) * la), Num<float>::rcp(la * lb), accL);
        const float lc = element(a.z, id.z, pr.z), ld = element(a.w, id.w, pr.w);
        accL = fmaf(fmaf(id.z * id.z, ld, (id.w * id.w) * lc), Num<float>::rcp(lc * ld), accL);
      }
      v = wave_sum_f(v);
      hW = wave_sum_f(hW);
      accL = wave_sum_f(accL);
      __syncthreads();
      if (lane == 0) { part[16 + wave] = v; part[32 + wave] = hW; part[48 + wave] = accL; }
      __syncthreads();
      float vS = 0.f, hS = 0.f, lS = 0.f;
#pragma unroll
      for (int w = 0; w < NW; w++) { vS += part[16 + w]; hS += part[32 + w]; lS += part[48 + w]; }
      if (tid == 0) {
        wk[k] = (double)Wk;
        wv[k] = (double)Wk * sqrt((double)vS);
      }
      hWd += (double)hS;
      accLd += (double)lS;
    }
    __syncthreads();
    if (tid == 0) priority[q] = eval_epilogue(wk, -hWd, wv, K, accLd, vCompTail);
  }
}

template <typename R, int QB, int KG, bool EXACT>
hipError_t launch_batch(const BatchArgs &args0, int nThreads, size_t *accBytesNeeded, int *gridOut, int *capacityOut, bool queryOnly,
                        hipStream_t stream) {
  BatchArgs args = args0;
  auto kern = eval_batch_kernel<R, QB, KG, EXACT>;
  const int G = nThreads / args.Bq;   // question groups side by side (eval_batch_kernel)
  const size_t tileBytes = (size_t)args.TC * G * QB * (KG + 1) * sizeof(R);
  const size_t shmem = (Num<R>::kTable ? kLog2TableDoubles * sizeof(double) : 0) + tileBytes;
  if (shmem > 160 * 1024) return hipErrorInvalidValue;
  static LaunchCache cache;   // (per instantiation and device; the occupancy also depends on the thread count: part of the key)
  const int devSlot = LaunchCache::Device();
  const size_t key = shmem * 2048 + (size_t)nThreads;
  int perCU = 0;
  if (!cache.Get(devSlot, key, &perCU)) {
    if (shmem > 64 * 1024) {
      const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
      if (e != hipSuccess) return e;
    }
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, kern, nThreads, shmem) != hipSuccess || perCU < 1) perCU = 1;
    cache.Put(devSlot, key, perCU);
  }
  const int nCU = cache.NumCUs(devSlot);
  const int64_t nBlocks = (args.qEnd - args.qBegin + G * QB - 1) / (G * QB);
  int64_t grid = (int64_t)nCU * perCU;
  if (grid > kBatchMaxGrid) grid = kBatchMaxGrid;
  if (capacityOut) *capacityOut = (int)grid;                  // (workgroups the device holds at once: LaunchEvalBatch plans the last round by it)
  if (grid > nBlocks) grid = nBlocks;
  *gridOut = (int)grid;
  *accBytesNeeded = (size_t)grid * QB * (2 * args.K + 2) * nThreads * sizeof(double);
  if (queryOnly) return hipSuccess;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(nThreads), shmem, stream, args);
  return hipGetLastError();
}

// The pole scratch of a batched sweep (BatchPlan::pole): marks of the quizzes the fix changed [256] | list header | entries [Q x Bp]
// | the listed pairs' sums [Q x Bp][2 K + 2].  The caller clears the first kBatchPoleClear bytes once; every launch leaves them cleared.
static_assert(kBatchPoleClear == 256 * sizeof(uint32_t) + sizeof(PoleHeader), "marks and list header");
size_t batch_pole_bytes(const KbView &kb, int Bp) {
  const size_t cap = (size_t)kb.Q * (size_t)Bp;
  return kBatchPoleClear + cap * sizeof(PoleEntry) + cap * (size_t)(2 * kb.K + 2) * sizeof(double);
}
struct BatchPole { PoleHeader *list; uint32_t *dirty; double *sums; };
BatchPole batch_pole(const KbView &kb, int Bp, void *pole) {
  char *p = static_cast<char *>(pole);
  const size_t cap = (size_t)kb.Q * (size_t)Bp;
  return BatchPole{reinterpret_cast<PoleHeader *>(p + 256 * sizeof(uint32_t)), reinterpret_cast<uint32_t *>(p),
                   reinterpret_cast<double *>(p + kBatchPoleClear + cap * sizeof(PoleEntry))};
}
// the fix between a batched sweep and its pick: the listed pairs, corrected in the priority matrix (and in the host's records)
hipError_t launch_batch_fixup(const KbView &kb, const QuizSlot *slots, int nSlots, int Bp, const BatchPole &bp, double *priorityT,
                              uint64_t hostTag, double vCompTail, hipStream_t stream) {
  PoleFix f{};
  f.cube = static_cast<const double *>(kb.cube); f.tgap = kb.tgap; f.qgap = kb.qgap; f.slots = slots;
  f.list = bp.list; f.dirty = bp.dirty; f.sums = bp.sums; f.sumsStride = 2 * kb.K + 2; f.bySlot = 1;
  f.wOff = 0; f.vOff = (int)kb.K; f.hOff = (int)(2 * kb.K); f.lOff = (int)(2 * kb.K + 1); f.secondIsWV = 1;
  f.priorityT = priorityT; f.Bp = Bp; f.hostTag = hostTag;
  f.K = kb.K; f.T = kb.T; f.ldT = kb.ldT; f.qFirst = 0; f.nQ = kb.Q; f.capacity = kb.Q * (int64_t)nSlots;
  f.vCompTail = vCompTail;
  return LaunchPoleFixup(f, stream);
}

}  // namespace

// Plan or run one batched sweep.  queryOnly: fill plan->{grid, accBytes, ptBytes, recBytes} for the caller to size its
// scratch; otherwise prep (transposed priors) + sweep + pick on `stream`.
hipError_t LaunchEvalBatch(const KbView &kb, const QuizSlot *slots, int nSlots, BatchPlan *plan, void *PT, double *acc,
                           BatchRecord *recs, double *priorityT, int64_t outBase, uint64_t flagValue, bool queryOnly, hipStream_t stream,
                           bool skipPick) {
  if (nSlots <= 0 || nSlots > 256 || plan == nullptr) return hipErrorInvalidValue;
  const bool f32 = kb.elem == 4;
  // lanes: Bq per question group (the batch rounded up to 32, 64, 128 or whole waves), 256 / Bq groups per workgroup where that
  // is a whole number -- a lane is a quiz, and lanes a small batch leaves over take further questions (eval_batch_kernel)
  const int Bp = ((nSlots + 63) / 64) * 64;
  const int Bq = nSlots <= 32 ? 32 : Bp;
  // questions per lane and block: the more, the fewer prior loads and tile reads per element -- and the more registers, the fewer
  // and longer lane-tasks.  Measured over batch sizes and cubes in round 6 (tools/batch_bench.py, one box per table): fp64 is
  // better off with ONE at every batch size (1000 x 5 x 1000: 32 quizzes 0.94 -> 0.60 ms, 256: 1.33 -> 1.16; 10000 x 5 x 10000, 256:
  // 117.6 -> 108.9 ms; 2000 x 5 x 100000: 32 quizzes 95.5 -> 56.3 ms, 256: 233 -> 199), fp32 with one up to 32 quizzes
  // (12500 x 5 x 100000: 125.5 -> 102.7 ms; 2000 x 5 x 100000: 74.1 -> 39.8), two up to 128 (64: 178.6 -> 150.6, 128: 287 -> 270)
  // and four beyond -- what rounds 3-5 had for every size was tuned at 256 quizzes of fp32
  const int qb = plan->questionsPerBlock > 0 ? plan->questionsPerBlock : !f32 ? 1 : nSlots <= 32 ? 1 : nSlots <= 128 ? 2 : 4;
  // groups: as many as the lanes allow while every CU still gets a workgroup (a small cube keeps its waves apart instead)
  static LaunchCache devInfo;
  const int nCUs = devInfo.NumCUs(LaunchCache::Device());
  int G = 256 % Bq == 0 ? 256 / Bq : 1;
  if (plan->questionGroups > 0) G = std::min(G, 1 << (31 - __builtin_clz((unsigned)plan->questionGroups)));
  else if (f32 && G >= 4) G >>= 1;   // (fp32, same tables: half the groups the lanes allow -- 32 quizzes 110.7 -> 102.7 ms with four instead of eight, 64: 161.4 -> 150.6 with two)
  else while (G > 1 && (kb.Q + (int64_t)G * qb - 1) / ((int64_t)G * qb) < (f32 ? nCUs : nCUs * 3 / 4)) G >>= 1;   // (measured: tools/batch_bench.py)
  const int nThreads = G * Bq;
  BatchArgs a{};
  a.cube = kb.cube; a.PT = PT; a.tgap = kb.tgap; a.qgap = kb.qgap; a.slots = slots; a.nSlots = nSlots; a.Bp = Bp; a.Bq = Bq;
  a.K = kb.K; a.Q = kb.Q; a.ldT = kb.ldT;
  // default tile (measured, 256 quizzes): fp64 256 targets; fp32 512 (2000 x 5 x 100000: 91.2 ms against 93.6 at 256 and 129.8 at
  // 1024), and a row of up to 1024 targets whole -- staged once for both passes (1000 x 5 x 1000: 341 k selections/s against 324 k)
  // (the tile holds G groups' questions: the targets per tile shrink by G, its bytes stay)
  int tc = plan->tileTargets > 0 ? plan->tileTargets : !f32 ? 256 : kb.ldT <= 1024 ? 1024 : 512;
  tc = std::max(64, ((tc / G + 63) / 64) * 64);
  if ((int64_t)tc > kb.ldT) tc = (int)kb.ldT;                  // (ldT is a multiple of 32)
  a.TC = tc;
  const double nT = (double)(kb.nValidTargets + 1);            // PqaCore/CEEvalQsSubtaskConsider.cpp:191
  a.vCompTail = 0.34657359027997265470861606072909 / (nT * nT);
  a.acc = acc; a.recs = recs; a.priorityT = priorityT;
  // the engine's option pole_fix (KbView::poleList), Double engines: the sweep lists the (question, quiz) pairs at the pole of the
  // lack term, pole_kernels.hip redoes them in the priority matrix -- which the caller then provides -- before the pick
  const bool watch = !f32 && kb.poleList != nullptr && !skipPick;
  plan->poleBytes = watch ? batch_pole_bytes(kb, Bp) : 0;
  BatchPole bpole{};
  if (watch && !queryOnly) {
    if (plan->pole == nullptr || priorityT == nullptr) return hipErrorInvalidValue;
    bpole = batch_pole(kb, Bp, plan->pole);
    a.poleList = bpole.list;
    a.poleSums = bpole.sums;
  }
  plan->ptBytes = (size_t)kb.ldT * Bp * (f32 ? 4 : 8);
  plan->Bp = Bp;
  hipError_t e;
  const bool k5 = kb.K == 5;
  // one launch of the shape with `qbSel` questions per group over the questions [a.qBegin, a.qEnd)
  auto run = [&](int qbSel, bool query, size_t *accBytes, int *grid, int *capacity) -> hipError_t {
    if (f32) {
      if (k5) return qbSel >= 4   ? launch_batch<float, 4, 5, true>(a, nThreads, accBytes, grid, capacity, query, stream)
                     : qbSel >= 2 ? launch_batch<float, 2, 5, true>(a, nThreads, accBytes, grid, capacity, query, stream)
                                  : launch_batch<float, 1, 5, true>(a, nThreads, accBytes, grid, capacity, query, stream);
      return qbSel >= 4   ? launch_batch<float, 4, 4, false>(a, nThreads, accBytes, grid, capacity, query, stream)
             : qbSel >= 2 ? launch_batch<float, 2, 4, false>(a, nThreads, accBytes, grid, capacity, query, stream)
                          : launch_batch<float, 1, 4, false>(a, nThreads, accBytes, grid, capacity, query, stream);
    }
    if (k5) return qbSel >= 2 ? launch_batch<double, 2, 5, true>(a, nThreads, accBytes, grid, capacity, query, stream)
                              : launch_batch<double, 1, 5, true>(a, nThreads, accBytes, grid, capacity, query, stream);
    return qbSel >= 2 ? launch_batch<double, 2, 4, false>(a, nThreads, accBytes, grid, capacity, query, stream)
                      : launch_batch<double, 1, 4, false>(a, nThreads, accBytes, grid, capacity, query, stream);
  };
  // The sweep is a persistent grid striding over blocks of G * qb questions: with nBlocks = rounds * grid + rest, `rest` workgroups
  // make one more round while the others idle.  Where that last round is a large part of the sweep -- ONE full round and a rest
  // (12500 questions, 64 quizzes: 782 blocks of 16 questions on 512 workgroups) -- the full round is one launch and the last
  // questions another, of the largest shape with fewer questions per group that fits the device in one round: 205 -> 182 ms.
  // Not beyond that: after several rounds the few workgroups of the last one have the chip to themselves and run faster than a
  // second launch of smaller blocks, whose time shrinks less than their questions (256 quizzes, six rounds + 53 blocks: 540 ms
  // as one launch, 545 - 554 ms split; 128 quizzes, three rounds: 293 against 311 ms).
  const int qbMain = f32 ? (qb >= 4 ? 4 : qb >= 2 ? 2 : 1) : (qb >= 2 ? 2 : 1);
  int gridMain = 0, capMain = 0, gridTail = 0, qbTail = 0;
  int64_t qSplit = kb.Q;
  a.qBegin = 0;
  a.qEnd = kb.Q;
  e = run(qbMain, true, &plan->accBytes, &gridMain, &capMain);
  if (e != hipSuccess) return e;
  // (fp32, 256 quizzes, many rounds of blocks: two questions per lane sweep 3 % faster -- 12500 x 5 x 100000 516 -> 501 ms -- and read the
  //  priors through the L2s three times as often, 643 GB against 215 per sweep: not taken.  With about one round of blocks four are
  //  faster anyway, 2000 x 5 x 100000 88 against 97 ms.)
  {
    const int64_t QT = (int64_t)G * qbMain, nBlocks = (kb.Q + QT - 1) / QT, rounds = nBlocks / std::max(1, gridMain), rest = nBlocks % std::max(1, gridMain);
    if (plan->splitTail && gridMain == capMain && rounds == 1 && rest > 0 && qbMain > 1) {
      const int64_t tailQ = kb.Q - rounds * gridMain * QT;
      for (int cand = qbMain / 2; cand >= 1; cand /= 2) {      // (the largest smaller shape that still fits: a block's time shrinks less than its questions)
        a.qBegin = kb.Q - tailQ;
        a.qEnd = kb.Q;
        size_t accT = 0;
        int gT = 0, capT = 0;
        if (run(cand, true, &accT, &gT, &capT) != hipSuccess) continue;
        if ((tailQ + (int64_t)G * cand - 1) / ((int64_t)G * cand) <= capT) {   // (one round of the smaller shape)
          qbTail = cand;
          gridTail = gT;
          qSplit = kb.Q - tailQ;
          plan->accBytes = std::max(plan->accBytes, accT);
          break;
        }
      }
      if (qbTail != 0) {   // (the main launch covers whole rounds only)
        a.qBegin = 0;
        a.qEnd = qSplit;
        size_t accM = 0;
        e = run(qbMain, true, &accM, &gridMain, &capMain);
        if (e != hipSuccess) return e;
        plan->accBytes = std::max(plan->accBytes, accM);
      }
    }
  }
  plan->grid = gridMain + gridTail;
  plan->recBytes = (size_t)(gridMain + gridTail) * G * Bp * sizeof(BatchRecord);
  if (queryOnly) return hipSuccess;
  if (PT == nullptr || acc == nullptr || recs == nullptr) return hipErrorInvalidValue;
  const dim3 pgrid((unsigned)((kb.ldT + 63) / 64), (unsigned)(Bp / 64));
  if (f32) hipLaunchKernelGGL(batch_prep_kernel<float>, pgrid, dim3(kTileThreads), 0, stream, slots, nSlots, Bp, kb.tgap, kb.ldT, static_cast<float *>(PT));
  else hipLaunchKernelGGL(batch_prep_kernel<double>, pgrid, dim3(kTileThreads), 0, stream, slots, nSlots, Bp, kb.tgap, kb.ldT, static_cast<double *>(PT));
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  size_t dummy = 0;
  int gDummy = 0;
  a.qBegin = 0;
  a.qEnd = qSplit;
  a.recs = recs;
  e = run(qbMain, false, &dummy, &gDummy, nullptr);
  if (e != hipSuccess) return e;
  if (qbTail != 0) {
    a.qBegin = qSplit;
    a.qEnd = kb.Q;
    a.recs = recs + (size_t)gridMain * G * Bp;
    e = run(qbTail, false, &dummy, &gDummy, nullptr);
    if (e != hipSuccess) return e;
  }
  if (skipPick) return hipSuccess;
  if (watch) {
    e = launch_batch_fixup(kb, slots, nSlots, Bp, bpole, priorityT, 0, a.vCompTail, stream);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(batch_pick_kernel, dim3((unsigned)nSlots), dim3(64), 0, stream, recs, (gridMain + gridTail) * G, Bp, slots, nSlots,
                     outBase, flagValue, watch ? bpole.dirty : nullptr, priorityT, kb.Q, kb.qgap);
  return hipGetLastError();
}

// ---- the sweep for a few dozen quizzes (eval_midbatch_kernel): Double engines, K == 5, rows up to kMidMaxTargets --------------
bool EvalMidBatchSupported(const KbView &kb) {
  const int64_t km = kb.K == 5 ? 5 : kMidMaxK;
  return kb.elem == 8 && kb.K >= 2 && kb.K <= kMidMaxK &&
         (size_t)(kLog2TableDoubles + kb.ldT * (km + 1) + (kMidThreads / kWave) * (km + 2) * 64 + 64) * sizeof(double) <= 160 * 1024;
}

// plan: out grid / Bp / ptBytes / recBytes (queryOnly), as LaunchEvalBatch; PT and recs from the caller.  Every quiz's winner goes to
// its slot's `out` and flagValue to its `seq`; slots with hostPriority get their priorities as tagged records (tag = flagValue).
hipError_t LaunchEvalMidBatch(const KbView &kb, const QuizSlot *slots, int nSlots, BatchPlan *plan, void *PT, BatchRecord *recs,
                              double *priorityT, int64_t outBase, uint64_t flagValue, bool queryOnly, hipStream_t stream) {
  if (!EvalMidBatchSupported(kb) || nSlots <= 0 || nSlots > 256 || plan == nullptr) return hipErrorInvalidValue;
  const int Bp = ((nSlots + 63) / 64) * 64;
  // quiz slots per wave: as few as hold the batch (more chunks per wave: shorter serial sums), 64 beyond 32 quizzes
  const int QS = nSlots <= 8 ? 8 : nSlots <= 16 ? 16 : nSlots <= 32 ? 32 : 64;
  const int groups = (nSlots + QS - 1) / QS;
  static LaunchCache cache;
  const int devSlot = LaunchCache::Device();
  const int nCU = cache.NumCUs(devSlot);
  const int KM = kb.K == 5 ? 5 : kMidMaxK;
  const size_t shmem = (size_t)(kLog2TableDoubles + kb.ldT * (KM + 1) + (kMidThreads / kWave) * (KM + 2) * QS + 64) * sizeof(double);   // (+ the watch words)
  if (shmem > 160 * 1024) return hipErrorInvalidValue;
  const int perCU = (int)std::max<size_t>(1, std::min<size_t>(3, (160 * 1024) / shmem));
  int64_t grid = std::min<int64_t>(kb.Q, std::max<int64_t>(1, (int64_t)nCU * perCU / groups));
  if (grid > kBatchMaxGrid) grid = kBatchMaxGrid;
  if (kb.maxGrid > 0 && grid > kb.maxGrid) grid = kb.maxGrid;
  plan->grid = (int)grid;
  plan->Bp = Bp;
  plan->ptBytes = (size_t)kb.ldT * Bp * sizeof(double);
  plan->accBytes = 0;
  plan->recBytes = (size_t)grid * Bp * sizeof(BatchRecord);
  const bool watch = kb.poleList != nullptr;                  // (as LaunchEvalBatch)
  plan->poleBytes = watch ? batch_pole_bytes(kb, Bp) : 0;
  if (queryOnly) return hipSuccess;
  if (PT == nullptr || recs == nullptr) return hipErrorInvalidValue;
  if (watch && (plan->pole == nullptr || priorityT == nullptr)) return hipErrorInvalidValue;
  const BatchPole bpole = watch ? batch_pole(kb, Bp, plan->pole) : BatchPole{};
  // the instantiation: quiz slots per wave x (five answers exactly | up to eight)
  void (*kern)(MidArgs) = nullptr;
  if (KM == 5) kern = QS == 8 ? eval_midbatch_kernel<8, 5, true> : QS == 16 ? eval_midbatch_kernel<16, 5, true> : QS == 32 ? eval_midbatch_kernel<32, 5, true> : eval_midbatch_kernel<64, 5, true>;
  else kern = QS == 8 ? eval_midbatch_kernel<8, kMidMaxK, false> : QS == 16 ? eval_midbatch_kernel<16, kMidMaxK, false> : QS == 32 ? eval_midbatch_kernel<32, kMidMaxK, false> : eval_midbatch_kernel<64, kMidMaxK, false>;
  int attr = 0;
  const size_t key = shmem * 1024 + (size_t)QS * 16 + (size_t)KM;
  if (shmem > 64 * 1024 && !cache.Get(devSlot, key, &attr)) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    if (e != hipSuccess) return e;
    cache.Put(devSlot, key, 1);
  }
  const dim3 pgrid((unsigned)((kb.ldT + 63) / 64), (unsigned)(Bp / 64));
  hipLaunchKernelGGL(batch_prep_kernel<double>, pgrid, dim3(kTileThreads), 0, stream, slots, nSlots, Bp, kb.tgap, kb.ldT, static_cast<double *>(PT));
  MidArgs a{};
  a.cube = static_cast<const double *>(kb.cube); a.PT = static_cast<const double *>(PT); a.tgap = kb.tgap; a.qgap = kb.qgap;
  a.slots = slots; a.nSlots = nSlots; a.Bp = Bp; a.Q = kb.Q; a.ldT = kb.ldT; a.K = kb.K;
  const double nT = (double)(kb.nValidTargets + 1);            // PqaCore/CEEvalQsSubtaskConsider.cpp:191
  a.vCompTail = 0.34657359027997265470861606072909 / (nT * nT);
  a.recs = recs; a.priorityT = priorityT; a.tag = flagValue;
  a.poleList = bpole.list; a.poleSums = bpole.sums;
  const dim3 g((unsigned)grid, (unsigned)groups);
  hipLaunchKernelGGL(kern, g, dim3(kMidThreads), shmem, stream, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  if (watch) {
    e = launch_batch_fixup(kb, slots, nSlots, Bp, bpole, priorityT, flagValue, a.vCompTail, stream);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(batch_pick_kernel, dim3((unsigned)nSlots), dim3(64), 0, stream, recs, (int)grid, Bp, slots, nSlots,
                     outBase, flagValue, watch ? bpole.dirty : nullptr, priorityT, kb.Q, kb.qgap);
  return hipGetLastError();
}

hipError_t LaunchEvalQuestionsF32(const KbView &kb, const double *prior, const uint32_t *asked, double *priority, hipStream_t stream) {
  if (kb.elem != 4) return hipErrorInvalidValue;
  const double nT = (double)(kb.nValidTargets + 1);
  const double vCompTail = 0.34657359027997265470861606072909 / (nT * nT);
  const bool ldsRow = kb.ldT <= kF32LdsTargets, big = kb.ldT >= 4096;
  const size_t shmem = (size_t)(2 * kb.K + 32) * sizeof(double) + (ldsRow ? (size_t)kb.ldT * 8 : 0);
  static LaunchCache cache;   // (per device)
  const int devSlot = LaunchCache::Device();
  int attrSet = 0;
  if (ldsRow && shmem > 64 * 1024 && !cache.Get(devSlot, 1, &attrSet)) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(eval_questions_f32_stream<true, 1024>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
    if (e != hipSuccess) return e;
    cache.Put(devSlot, 1, 1);
  }
  const int nCU = cache.NumCUs(devSlot);
  const int nt = big ? 1024 : 256;
  int perCU = (int)std::max<size_t>(1, std::min<size_t>((size_t)(2048 / nt), (160 * 1024) / std::max<size_t>(shmem, 1)));
  int64_t grid = std::min<int64_t>(kb.Q, (int64_t)nCU * perCU);
  if (kb.maxGrid > 0 && grid > kb.maxGrid) grid = kb.maxGrid;
  const float *cube = static_cast<const float *>(kb.cube);
#define PQA_F32_LAUNCH(L, N) hipLaunchKernelGGL((eval_questions_f32_stream<L, N>), dim3((unsigned)grid), dim3(N), shmem, stream, cube, prior, \
                                                kb.tgap, kb.qgap, asked, priority, kb.K, kb.Q, kb.ldT, vCompTail)
  if (ldsRow) { if (big) PQA_F32_LAUNCH(true, 1024); else PQA_F32_LAUNCH(true, 256); }
  else PQA_F32_LAUNCH(false, 1024);
#undef PQA_F32_LAUNCH
  return hipGetLastError();
}

}  // namespace pqa
