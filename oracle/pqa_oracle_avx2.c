/*
 * pqa_oracle_avx2.c -- AVX2/FMA + pthreads restatement of CEEvalQsSubtaskConsider<SRDoubleNumber>::Run, used as the
 * TIMED CPU BASELINE ("port" of the reference's AVX2 SRThreadPool path; the reference is MSVC/Win32-only and cannot
 * run here).  TEST INFRASTRUCTURE ONLY.  Same intrinsic sequence and operation order as
 * PqaCore/CEEvalQsSubtaskConsider.cpp:41-217; bit-identical to the scalar oracle (orc_eval_all), checked in tests.
 * Scheduling shape of PqaCore/CpuEngine.cpp:339,355-360 + SRPlatform/SRThreadPool.cpp:132-178: a fixed set of worker
 * threads pops contiguous question ranges from one FIFO.  Compile: gcc -O2 -mavx2 -mfma -ffp-contract=off -pthread.
 */
#include "pqa_oracle.h"

#include <immintrin.h>
#include <math.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdlib.h>
#include <string.h>

int orc_have_avx2(void) { return __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma"); }

typedef struct { __m256d sum, corr; } K4;

static inline void k4_reset(K4 *a) { a->sum = _mm256_setzero_pd(); a->corr = _mm256_setzero_pd(); }
static inline void k4_add(K4 *a, __m256d v) {                                   /* SRAccumVectDbl256.h:40-46 */
  const __m256d y = _mm256_sub_pd(v, a->corr);
  const __m256d t = _mm256_add_pd(a->sum, y);
  a->corr = _mm256_sub_pd(_mm256_sub_pd(t, a->sum), y);
  a->sum = t;
}
static inline void k4_export(const K4 *a, OrcKahan4 *o) { _mm256_storeu_pd(o->sum, a->sum); _mm256_storeu_pd(o->corr, a->corr); }

/* SRSimd::SetToBitQuadHot (SRPlatform/Interface/SRSimd.h:253-256): nibble -> 4x64-bit lane mask */
static inline __m256d gap_mask(uint8_t quad) {
  return _mm256_castsi256_pd(_mm256_set_epi64x(-(int64_t)(quad >> 3), -(int64_t)((quad >> 2) & 1),
                                               -(int64_t)((quad >> 1) & 1), -(int64_t)(quad & 1)));
}
static inline uint8_t get_quad(const uint8_t *bits, int64_t iQuad) {            /* SRBitArray.h:249-253 */
  return (uint8_t)((bits[iQuad >> 1] >> ((iQuad & 1) << 2)) & 0x0f);
}

static inline __m256d log2hot_v(const double *tbl, __m256d x) {                 /* SRVectMath.h:87-135 */
  const __m256d cNotExp = _mm256_castsi256_pd(_mm256_set1_epi64x((long long)~0x7FF0000000000000ULL));
  const __m256d cExp0 = _mm256_castsi256_pd(_mm256_set1_epi64x(0x3FF0000000000000LL));
  const __m256d z = _mm256_or_pd(_mm256_and_pd(cNotExp, x), cExp0);             /* :88-89 */
  const __m128 hiLane = _mm_castpd_ps(_mm256_extractf128_pd(x, 1));             /* :92 */
  const __m128 loLane = _mm_castpd_ps(_mm256_castpd256_pd128(x));               /* :93 */
  const __m128i high32 = _mm_castps_si128(_mm_shuffle_ps(loLane, hiLane, _MM_SHUFFLE(3, 1, 3, 1))); /* :94 */
  const __m128i exps32 = _mm_srai_epi32(high32, 52 - 32);                       /* :97 */
  const __m128i normExps = _mm_sub_epi32(exps32, _mm_set1_epi32(1023));         /* :98 */
  const __m128i idx = _mm_and_si128(_mm_set1_epi32(1023), _mm_srai_epi32(high32, 52 - 32 - 10)); /* :101-102 */
  uint32_t ix[4]; _mm_storeu_si128((__m128i *)ix, idx);
  const __m256d y = _mm256_set_pd(tbl[ix[3]], tbl[ix[2]], tbl[ix[1]], tbl[ix[0]]); /* :105-106 */
  const __m256d cMask = _mm256_castsi256_pd(_mm256_set1_epi64x((long long)~((1ULL << 42) - 1)));
  const __m256d cPlus = _mm256_castsi256_pd(_mm256_set1_epi64x(1LL << 41));
  const __m256d exp2Y = _mm256_or_pd(cPlus, _mm256_and_pd(z, cMask));           /* :108 */
  const __m256d tNum = _mm256_sub_pd(z, exp2Y);                                 /* :111 */
  const __m256d tDen = _mm256_add_pd(z, exp2Y);                                 /* :112 */
  const __m256d t = _mm256_div_pd(tNum, tDen);                                  /* :114 */
  const __m256d t2 = _mm256_mul_pd(t, t);                                       /* :115 */
  const __m256d t3 = _mm256_mul_pd(t, t2);                                      /* :117 */
  const __m256d terms01 = _mm256_fmadd_pd(_mm256_set1_pd(1.0 / 3), t3, t);      /* :118 */
  const __m256d log2_z = _mm256_fmadd_pd(terms01, _mm256_set1_pd(2.8853900817779268147198493620038), y); /* :122 */
  const __m256d leading = _mm256_cvtepi32_pd(normExps);                         /* :131 */
  return _mm256_add_pd(log2_z, leading);                                        /* :133 */
}

static inline __m256d load_stream(const double *p) {                            /* SRSimd::Load<false>, SRSimd.h:63-70 */
  return _mm256_castsi256_pd(_mm256_stream_load_si256((const __m256i *)p));
}

static inline int bit_test(const uint8_t *bits, int64_t i) { return (bits[i >> 3] >> (i & 7)) & 1; }

static void eval_subtask_avx2(const OrcKB *kb, const OrcQuiz *quiz, int64_t nValidTargets, int64_t iFirst,
                              int64_t iLimit, double *runLength, double *priority) {
  const double *tbl = orc_log2hot_table();
  const int64_t K = kb->nAnswers, ldT = kb->ldT;
  const int64_t nTargVects = (kb->nTargets + 3) >> 2;
  const double *pPriors = quiz->mants;
  double *invDi = (double *)aligned_alloc(32, (size_t)nTargVects * 32);         /* :49 (stack in the reference) */
  double *post = (double *)aligned_alloc(32, (size_t)nTargVects * 32);          /* :50 */
  double *mW = (double *)malloc((size_t)K * 8), *mH = (double *)malloc((size_t)K * 8), *mV = (double *)malloc((size_t)K * 8);
  const __m256d one = _mm256_set1_pd(1.0);
  const int rowsAligned = ((((uintptr_t)kb->A | (uintptr_t)kb->D) & 31) == 0) && ((ldT & 3) == 0);

  OrcKahan1 accRunLength; orc_k1_init(&accRunLength, 0.0);
  for (int64_t i = iFirst; i < iLimit; i++) {
    if (bit_test(kb->questionGaps, i) || bit_test(quiz->asked, i)) {
      runLength[i] = orc_k1_get(&accRunLength);
      if (priority) priority[i] = 0;
      continue;
    }
    const double *pmDi = kb->D + (size_t)i * ldT;
    OrcKahan1 accTotW; orc_k1_init(&accTotW, 0.0);
    K4 accL; k4_reset(&accL);
    for (int64_t k = 0; k < K; k++) {
      K4 accLhEnt; k4_reset(&accLhEnt);
      const double *psAik = kb->A + ((size_t)i * K + k) * ldT;
      for (int64_t j = 0; j < nTargVects; j++) {                                /* pass 1 :66-87 */
        const __m256d gm = gap_mask(get_quad(kb->targetGaps, j));
        const __m256d priors = _mm256_load_pd(pPriors + 4 * j);
        __m256d invCountTotal;
        if (k == 0) {
          const __m256d vDij = rowsAligned ? load_stream(pmDi + 4 * j) : _mm256_loadu_pd(pmDi + 4 * j);
          invCountTotal = _mm256_andnot_pd(gm, _mm256_div_pd(one, vDij));
          _mm256_store_pd(invDi + 4 * j, invCountTotal);
        } else {
          invCountTotal = _mm256_load_pd(invDi + 4 * j);
        }
        const __m256d a = rowsAligned ? load_stream(psAik + 4 * j) : _mm256_loadu_pd(psAik + 4 * j);
        const __m256d prQk = _mm256_mul_pd(a, invCountTotal);
        const __m256d likelihood = _mm256_andnot_pd(gm, _mm256_mul_pd(prQk, priors));
        _mm256_store_pd(post + 4 * j, likelihood);
        k4_add(&accLhEnt, likelihood);
      }
      OrcKahan4 tmp; k4_export(&accLhEnt, &tmp);
      const double Wk = orc_k4_precise_sum(&tmp);
      orc_k1_add(&accTotW, Wk);
      mW[k] = Wk;
      const __m256d invWk = _mm256_div_pd(one, _mm256_set1_pd(Wk));

      k4_reset(&accLhEnt);
      K4 accV; k4_reset(&accV);
      for (int64_t j = 0; j < nTargVects; j++) {                                /* pass 2 :95-128 */
        const __m256d posteriors = _mm256_mul_pd(_mm256_load_pd(post + 4 * j), invWk);
        const __m256d gm = gap_mask(get_quad(kb->targetGaps, j));
        const __m256d priors = _mm256_andnot_pd(gm, _mm256_load_pd(pPriors + 4 * j));
        const __m256d l2post = _mm256_andnot_pd(gm, log2hot_v(tbl, posteriors));
        const __m256d Hikj = _mm256_mul_pd(posteriors, l2post);
        k4_add(&accLhEnt, Hikj);
        const __m256d invDij = _mm256_load_pd(invDi + 4 * j);
        k4_add(&accL, _mm256_andnot_pd(gm, _mm256_div_pd(_mm256_mul_pd(invDij, invDij), l2post)));
        const __m256d diff = _mm256_sub_pd(posteriors, priors);
        k4_add(&accV, _mm256_mul_pd(diff, diff));
      }
      OrcKahan4 eH, eV; k4_export(&accLhEnt, &eH); k4_export(&accV, &eV);
      double velocity;
      mH[k] = -orc_k4_pair_sum(&eH, &eV, &velocity);
      mV[k] = velocity;
    }
    const double totW = orc_k1_get(&accTotW);
    OrcKahan4 accAvgH, accAvgV; orc_k4_reset(&accAvgH); orc_k4_reset(&accAvgV);
    const int64_t nVectorized = (K >> 2) << 2;
    for (int64_t k = 0; k < nVectorized; k += 4) {
      double wh[4], wv[4];
      for (int c = 0; c < 4; c++) { wh[c] = mW[k + c] * mH[k + c]; wv[c] = mW[k + c] * sqrt(mV[k + c]); }
      orc_k4_add(&accAvgH, wh); orc_k4_add(&accAvgV, wv);
    }
    for (int64_t k = nVectorized; k < K; k++) {
      const int at = (int)(k - nVectorized);
      orc_k4_add_at(&accAvgH, at, mW[k] * mH[k]);
      orc_k4_add_at(&accAvgV, at, mW[k] * sqrt(mV[k]));
    }
    double avgV;
    double avgH = orc_k4_pair_sum(&accAvgH, &accAvgV, &avgV);
    avgH = avgH / totW; avgV = avgV / totW;
    const double nExpectedTargets = exp2(avgH);
    const double cLnMaxV = 0.34657359027997265470861606072909;
    const double lnV = ((avgV == 0) ? -746.0 : log(avgV));
    const double nT = (double)(nValidTargets + 1);
    const double vComp = 1 / (cLnMaxV - lnV + cLnMaxV / (nT * nT));
    OrcKahan4 eL; k4_export(&accL, &eL);
    const double lack = -orc_k4_precise_sum(&eL);
    const double v2 = vComp * vComp, v4 = v2 * v2, v8 = v4 * v4, v9 = v8 * vComp;
    const double prio = lack * v9 * (1.0 / (nExpectedTargets * nExpectedTargets));
    orc_k1_add(&accRunLength, prio);
    runLength[i] = orc_k1_get(&accRunLength);
    if (priority) priority[i] = prio;
  }
  free(invDi); free(post); free(mW); free(mH); free(mV);
}

typedef struct {
  const OrcKB *kb; const OrcQuiz *quiz; int64_t nValid; const int64_t *bounds; int64_t nSubtasks;
  double *runLength, *priority; atomic_long next;
} Job;

static void run_job(Job *job) {
  for (;;) {
    const long s = atomic_fetch_add(&job->next, 1);                             /* FIFO pop, SRThreadPool.cpp:132-178 */
    if (s >= job->nSubtasks) break;
    eval_subtask_avx2(job->kb, job->quiz, job->nValid, s == 0 ? 0 : job->bounds[s - 1], job->bounds[s],
                      job->runLength, job->priority);
  }
}

/* Persistent worker pool (the reference keeps hardware_concurrency threads alive for the engine's lifetime,
 * PqaCore/BaseCpuEngine.cpp:19-22), so thread creation is not part of the timed sweep. */
static struct {
  pthread_mutex_t mu; pthread_cond_t cvWork, cvDone;
  pthread_t *threads; int64_t nThreads; Job *job; uint64_t generation; int64_t nRunning;
} gPool = { PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER, NULL, 0, NULL, 0, 0 };

static void *pool_worker(void *arg) {
  uint64_t seen = (uint64_t)(uintptr_t)arg;              /* the generation at which this worker was spawned */
  pthread_mutex_lock(&gPool.mu);
  for (;;) {
    while (gPool.generation == seen) pthread_cond_wait(&gPool.cvWork, &gPool.mu);
    seen = gPool.generation;
    Job *job = gPool.job;
    pthread_mutex_unlock(&gPool.mu);
    run_job(job);
    pthread_mutex_lock(&gPool.mu);
    if (--gPool.nRunning == 0) pthread_cond_signal(&gPool.cvDone);
  }
  return NULL;
}

/* The pool only grows: a request for more threads than exist spawns the difference, a request for fewer uses them all
 * (callers that compare pool sizes go from small to large). */
static void pool_ensure(int64_t nThreads) {
  pthread_mutex_lock(&gPool.mu);
  if (nThreads > gPool.nThreads) {
    gPool.threads = (pthread_t *)realloc(gPool.threads, (size_t)nThreads * sizeof(pthread_t));
    for (int64_t i = gPool.nThreads; i < nThreads; i++) {
      pthread_create(&gPool.threads[i], NULL, pool_worker, (void *)(uintptr_t)gPool.generation);
      pthread_detach(gPool.threads[i]);
    }
    gPool.nThreads = nThreads;
  }
  pthread_mutex_unlock(&gPool.mu);
}

void orc_eval_all_avx2_mt(const OrcKB *kb, const OrcQuiz *quiz, int64_t nThreads, int64_t nSubtasks,
                          double *runLength, double *priority) {
  orc_log2hot_table();                                                          /* init before threads start */
  int64_t *bounds = (int64_t *)malloc((size_t)nSubtasks * 8);
  Job job;
  job.kb = kb; job.quiz = quiz; job.nValid = kb->nTargets - kb->nTargetGaps; job.bounds = bounds;
  job.nSubtasks = orc_calc_split(kb->nQuestions, nSubtasks, bounds);
  job.runLength = runLength; job.priority = priority; atomic_init(&job.next, 0);
  if (nThreads <= 1) {
    run_job(&job);
  } else {
    pool_ensure(nThreads);
    pthread_mutex_lock(&gPool.mu);
    gPool.job = &job; gPool.nRunning = gPool.nThreads; gPool.generation++;
    pthread_cond_broadcast(&gPool.cvWork);
    while (gPool.nRunning != 0) pthread_cond_wait(&gPool.cvDone, &gPool.mu);
    pthread_mutex_unlock(&gPool.mu);
  }
  free(bounds);
}
