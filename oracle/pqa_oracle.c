/*
 * pqa_oracle.c -- scalar, lane-emulating CPU restatement of ProbQA's CpuEngine<SRDoubleNumber> hot path.
 * TEST INFRASTRUCTURE ONLY (see pqa_oracle.h).  Compile: gcc -O2 -std=c11 -ffp-contract=off -mfma.
 * Paths in comments are relative to /root/reference/ProbQA.
 */
#include "pqa_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------------------------
 * bit helpers (SRPlatform/Interface/SRNumTraits.h:10-32)
 * ---------------------------------------------------------------------------------------------------------------- */
#define EXP_MASK_UP 0x7FF0000000000000ULL
#define EXP0_UP     0x3FF0000000000000ULL
#define EXP_OFFS    52

static inline uint64_t d2u(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }
static inline double   u2d(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }
static inline int bit_test(const uint8_t *bits, int64_t i) { return (bits[i >> 3] >> (i & 7)) & 1; }
static inline void bit_set(uint8_t *bits, int64_t i, int v) {
  if (v) bits[i >> 3] |= (uint8_t)(1u << (i & 7)); else bits[i >> 3] &= (uint8_t)~(1u << (i & 7));
}

/* ------------------------------------------------------------------------------------------------------------------
 * a2: Log2Hot.  SRPlatform/Interface/SRVectMath.h:87-135; table SRPlatform/SRVectMath.cpp:30-44.
 * ---------------------------------------------------------------------------------------------------------------- */
static double gLog2Tbl[1024];
static int gLog2TblInit = 0;

const double *orc_log2hot_table(void) {
  if (!gLog2TblInit) {
    for (uint32_t i = 0; i < 1024; i++) {
      const uint64_t iZ = EXP0_UP | ((uint64_t)i << (52 - 10));                 /* SRVectMath.cpp:33-34 */
      const uint64_t iZp = iZ | (1ULL << (52 - 10 - 1));                        /* :37 bucket midpoint */
      gLog2Tbl[i] = log2(u2d(iZp));                                             /* :39 */
    }
    gLog2Tbl[0] *= 9.9999999999999927e-01;                                      /* :31,:42 so that log2(1) <= 0 */
    gLog2TblInit = 1;
  }
  return gLog2Tbl;
}

double orc_log2hot(double x) {
  const double *tbl = orc_log2hot_table();
  const uint64_t ux = d2u(x);
  const double z = u2d((ux & ~EXP_MASK_UP) | EXP0_UP);                          /* SRVectMath.h:88-89 */
  const int32_t high32 = (int32_t)(ux >> 32);                                   /* :92-94 */
  const int32_t exps32 = high32 >> (EXP_OFFS - 32);                             /* :97 arithmetic shift, sign not cleared */
  const int32_t normExps = exps32 - 1023;                                       /* :98 */
  const int32_t idx = (high32 >> (52 - 32 - 10)) & 1023;                        /* :101-102 */
  const double y = tbl[idx];                                                    /* :105-106 */
  const uint64_t uz = d2u(z);
  const double exp2Y = u2d((1ULL << (EXP_OFFS - 10 - 1)) | (uz & ~((1ULL << (EXP_OFFS - 10)) - 1))); /* :108 */
  const double tNum = z - exp2Y;                                                /* :111 */
  const double tDen = z + exp2Y;                                                /* :112 */
  const double t = tNum / tDen;                                                 /* :114 */
  const double t2 = t * t;                                                      /* :115 */
  const double t3 = t * t2;                                                     /* :117 */
  const double terms01 = fma(1.0 / 3, t3, t);                                   /* :118 */
  /* :119-120 compute terms012 but :122 uses terms01 */
  const double log2_z = fma(terms01, 2.8853900817779268147198493620038, y);    /* :122 */
  const double leading = (double)normExps;                                      /* :131 */
  return log2_z + leading;                                                      /* :133 */
}

/* ------------------------------------------------------------------------------------------------------------------
 * a3: Kahan accumulators.  SRPlatform/Interface/SRAccumVectDbl256.h:40-133, SRAccumulator.h:15-39.
 * ---------------------------------------------------------------------------------------------------------------- */
void orc_k4_reset(OrcKahan4 *a) { memset(a, 0, sizeof(*a)); }

void orc_k4_add(OrcKahan4 *a, const double v[4]) {                              /* SRAccumVectDbl256.h:40-46 */
  for (int c = 0; c < 4; c++) {
    const double y = v[c] - a->corr[c];
    const double t = a->sum[c] + y;
    a->corr[c] = (t - a->sum[c]) - y;
    a->sum[c] = t;
  }
}

void orc_k4_add_at(OrcKahan4 *a, int at, double v) {                            /* :48-54 */
  const double y = v - a->corr[at];
  const double t = a->sum[at] + y;
  a->corr[at] = (t - a->sum[at]) - y;
  a->sum[at] = t;
}

void orc_k1_init(OrcKahan1 *a, double v) { a->sum = v; a->corr = 0; }           /* SRAccumulator.h:21 */
void orc_k1_add(OrcKahan1 *a, double v) {                                       /* SRAccumulator.h:28-34 */
  const double y = v - a->corr;
  const double t = a->sum + y;
  a->corr = (t - a->sum) - y;
  a->sum = t;
}
double orc_k1_get(const OrcKahan1 *a) { return a->sum - a->corr; }              /* SRAccumulator.h:37-39 */

double orc_k4_precise_sum(const OrcKahan4 *a) {                                 /* SRAccumVectDbl256.h:83-91 */
  OrcKahan1 ans;
  orc_k1_init(&ans, a->corr[3]);
  for (int i = 2; i >= 0; i--) orc_k1_add(&ans, a->corr[i]);
  ans.sum = -ans.sum; ans.corr = -ans.corr;                                     /* Neg(), SRAccumulator.h:24 */
  for (int i = 3; i >= 0; i--) orc_k1_add(&ans, a->sum[i]);
  return orc_k1_get(&ans);
}

double orc_k4_pair_sum(const OrcKahan4 *a, const OrcKahan4 *fellow, double *fellowSum) { /* :115-132 */
  /* two independent SSE lanes running the PreciseSum sequence; the arithmetic per lane is identical */
  double s[2] = { a->corr[3], fellow->corr[3] }, c[2] = { 0, 0 };
  const OrcKahan4 *src[2] = { a, fellow };
  for (int l = 0; l < 2; l++) {
    for (int i = 2; i >= 0; i--) {
      const double y = src[l]->corr[i] - c[l];
      const double t = s[l] + y;
      c[l] = (t - s[l]) - y;
      s[l] = t;
    }
    s[l] = -s[l]; c[l] = -c[l];                                                 /* xor with sign mask :123-124 */
    for (int i = 3; i >= 0; i--) {
      const double y = src[l]->sum[i] - c[l];
      const double t = s[l] + y;
      c[l] = (t - s[l]) - y;
      s[l] = t;
    }
  }
  *fellowSum = s[1] - c[1];
  return s[0] - c[0];
}

double orc_k4_full_sum(const OrcKahan4 *a) {                                    /* :56-60 (unused on the path) */
  /* hadd(corr,sum) = {c0+c1, s0+s1, c2+c3, s2+s3}; add upper and lower 128-bit halves */
  const double cs = (a->corr[2] + a->corr[3]) + (a->corr[0] + a->corr[1]);
  const double ss = (a->sum[2] + a->sum[3]) + (a->sum[0] + a->sum[1]);
  return ss - cs;
}

/* ------------------------------------------------------------------------------------------------------------------
 * SRPoolRunner::CalcSplit.  SRPlatform/Interface/SRPoolRunner.h:96-110.
 * ---------------------------------------------------------------------------------------------------------------- */
int64_t orc_calc_split(int64_t nItems, int64_t nWorkers, int64_t *bounds) {
  int64_t nSubtasks = 0, nextStart = 0;
  const int64_t quot = nItems / nWorkers, rem = nItems % nWorkers;
  while (nSubtasks < nWorkers && nextStart < nItems) {
    nextStart += quot + ((nSubtasks < rem) ? 1 : 0);
    bounds[nSubtasks] = nextStart;
    nSubtasks++;
  }
  return nSubtasks;
}

/* ------------------------------------------------------------------------------------------------------------------
 * KB / quiz containers.  PqaCore/CpuEngine.cpp:44-84 (fresh KB: A = init^2, D = init^2 * K, B = init).
 * ---------------------------------------------------------------------------------------------------------------- */
static size_t tgap_bytes(int64_t ldT) { return (size_t)((ldT + 7) / 8) + 8; }
static size_t qbits_bytes(int64_t nQ) { return (size_t)(((nQ + 63) / 64) * 8) + 8; }

OrcKB *orc_kb_create(int64_t nAnswers, int64_t nQuestions, int64_t nTargets, double initAmount) {
  OrcKB *kb = (OrcKB *)calloc(1, sizeof(OrcKB));
  kb->nAnswers = nAnswers; kb->nQuestions = nQuestions; kb->nTargets = nTargets;
  kb->ldT = ((nTargets + 3) / 4) * 4;
  const size_t nA = (size_t)nQuestions * nAnswers * kb->ldT, nD = (size_t)nQuestions * kb->ldT;
  kb->A = (double *)aligned_alloc(64, ((nA * 8 + 63) / 64) * 64);
  kb->D = (double *)aligned_alloc(64, ((nD * 8 + 63) / 64) * 64);
  kb->B = (double *)aligned_alloc(64, (((size_t)kb->ldT * 8 + 63) / 64) * 64);
  const double init1 = initAmount, initSqr = init1 * init1;                     /* CpuEngine.cpp:45-47 */
  const double initMD = initSqr * (double)nAnswers;
  for (size_t i = 0; i < nA; i++) kb->A[i] = initSqr;
  for (size_t i = 0; i < nD; i++) kb->D[i] = initMD;
  for (int64_t i = 0; i < kb->ldT; i++) kb->B[i] = init1;
  kb->targetGaps = (uint8_t *)calloc(1, tgap_bytes(kb->ldT));
  kb->questionGaps = (uint8_t *)calloc(1, qbits_bytes(nQuestions));
  /* bits past the size read as gaps (GapTracker.h:9-10) */
  for (int64_t t = nTargets; t < (int64_t)tgap_bytes(kb->ldT) * 8; t++) bit_set(kb->targetGaps, t, 1);
  for (int64_t q = nQuestions; q < (int64_t)qbits_bytes(nQuestions) * 8; q++) bit_set(kb->questionGaps, q, 1);
  kb->nTargetGaps = 0;
  return kb;
}

void orc_kb_destroy(OrcKB *kb) {
  if (!kb) return;
  free(kb->A); free(kb->D); free(kb->B); free(kb->targetGaps); free(kb->questionGaps); free(kb);
}

void orc_kb_set_target_gap(OrcKB *kb, int64_t t, int isGap) {
  const int was = bit_test(kb->targetGaps, t);
  bit_set(kb->targetGaps, t, isGap);
  kb->nTargetGaps += (isGap ? 1 : 0) - (was ? 1 : 0);
}
void orc_kb_set_question_gap(OrcKB *kb, int64_t q, int isGap) { bit_set(kb->questionGaps, q, isGap); }

/* ---- training: PqaCore/CETrainOperation.cpp:15-83, PqaCore/CETrainTaskNumSpec.h:24-32 ------------------------------
 * The cube stores squares: one training step takes a = sqrt(A) to a + b, i.e. A += 2ab + b^2, D += the same. */
static double *kb_a(OrcKB *kb, OrcAQ aq, int64_t t) {
  return &kb->A[((size_t)aq.iQuestion * kb->nAnswers + aq.iAnswer) * kb->ldT + t];
}
static double *kb_d(OrcKB *kb, OrcAQ aq, int64_t t) { return &kb->D[(size_t)aq.iQuestion * kb->ldT + t]; }

/* CETrainOperation::ProcessOne (:15-25) */
static void train_process_one(OrcKB *kb, OrcAQ aq, int64_t t, double twoB, double bSquare) {
  double *pA = kb_a(kb, aq, t), *pD = kb_d(kb, aq, t);
  const double a = sqrt(*pA);                 /* :18 */
  const double addend = a * twoB + bSquare;   /* :19 */
  *pA = *pA + addend;                         /* :23-24 */
  *pD = *pD + addend;                         /* :25 */
}
/* CETrainOperation::Perform1 (:28-30) */
static void train_perform1(OrcKB *kb, OrcAQ aq, int64_t t, double b) { train_process_one(kb, aq, t, 2 * b, b * b); }
/* CETrainOperation::Perform2 (:32-83): two answered questions at once.  Three cases:
 *   same question, same answer (:34-35): ONE step of 2b -- ProcessOne with _inc4B = 4b and _incSquare2B = 4 b^2;
 *   same question, different answers (:37-54): each answer's cell gets its own addend, but mD gets TWICE THE FIRST answer's
 *     addend (:46-47 "twice the amount in element #2": sseAddend[0] + sseAddend[0]), not the sum of the two;
 *   different questions (:56-82): two independent Perform1 steps, vectorised. */
static void train_perform2(OrcKB *kb, OrcAQ first, OrcAQ second, int64_t t, double b) {
  const double twoB = 2 * b, bSquare = b * b;
  if (first.iQuestion == second.iQuestion) {
    if (first.iAnswer == second.iAnswer) {
      train_process_one(kb, first, t, 4 * b, 4 * bSquare);   /* :35, CETrainTaskNumSpec.h:29-30 */
    } else {
      double *pA1 = kb_a(kb, first, t), *pA2 = kb_a(kb, second, t), *pD = kb_d(kb, first, t);
      const double add1 = sqrt(*pA1) * twoB + bSquare, add2 = sqrt(*pA2) * twoB + bSquare;   /* :42-44 */
      const double addD = add1 + add1;                       /* :45-46 */
      *pA1 = *pA1 + add1;                                    /* :47-53 */
      *pA2 = *pA2 + add2;
      *pD = *pD + addD;
    }
  } else {
    train_perform1(kb, first, t, b);                         /* :62-81: lane-wise the same operations as two ProcessOne */
    train_perform1(kb, second, t, b);
  }
}

/* CpuEngine::TrainSpec (PqaCore/CpuEngine.cpp:102-183): the answered questions are distributed into nWorkers buckets by
 * iQuestion % nWorkers (CETrainSubtaskDistrib.h:46-52; each bucket is a LIFO list), then every bucket is consumed from its
 * newest entry backwards, two entries at a time through Perform2, a last odd one through Perform1 (CETrainSubtaskAdd.cpp:17-38).
 * The reference distributes with several threads racing on an atomic sequence counter; restated here for the order a
 * single distributing thread produces (sequence number = position in pAQs).  Buckets hold disjoint questions, so the order
 * between buckets does not matter.  Then _vB[iTarget] += amount (:172). */
void orc_kb_train_workers(OrcKB *kb, int64_t nAQs, const OrcAQ *aqs, int64_t iTarget, double amount, int64_t nWorkers) {
  int64_t *last = (int64_t *)malloc(sizeof(int64_t) * (size_t)nWorkers);
  int64_t *prev = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nAQs > 0 ? nAQs : 1));
  for (int64_t w = 0; w < nWorkers; w++) last[w] = -1;
  for (int64_t i = 0; i < nAQs; i++) {                       /* CETrainSubtaskDistrib::Run */
    const int64_t bucket = aqs[i].iQuestion % nWorkers;
    prev[i] = last[bucket];
    last[bucket] = i;
  }
  for (int64_t w = 0; w < nWorkers; w++) {                   /* CETrainSubtaskAdd::Run */
    int64_t iLast = last[w];
    while (iLast != -1) {
      const OrcAQ first = aqs[iLast];
      iLast = prev[iLast];
      if (iLast == -1) { train_perform1(kb, first, iTarget, amount); break; }
      const OrcAQ second = aqs[iLast];
      train_perform2(kb, first, second, iTarget, amount);
      iLast = prev[iLast];
    }
  }
  free(last);
  free(prev);
  kb->B[iTarget] += amount;
}
/* Distinct questions: every step is a Perform1 whatever the pairing (kept for the callers that train that way). */
void orc_kb_train(OrcKB *kb, int64_t nAQs, const OrcAQ *aqs, int64_t iTarget, double amount) {
  orc_kb_train_workers(kb, nAQs, aqs, iTarget, amount, 1);
}
/* CpuEngine::RecordQuizTargetSpec (PqaCore/CpuEngine.cpp:442-466): the quiz's answers in order, pairs (0,1), (2,3), ...
 * through Perform2, a last odd one through Perform1; _vB[iTarget] += amount. */
void orc_kb_record_quiz_target(OrcKB *kb, int64_t nAQs, const OrcAQ *aqs, int64_t iTarget, double amount) {
  int64_t i = 0;
  for (; i < nAQs - 1; i += 2) train_perform2(kb, aqs[i], aqs[i + 1], iTarget, amount);
  if (i == nAQs - 1) train_perform1(kb, aqs[i], iTarget, amount);
  kb->B[iTarget] += amount;
}

OrcQuiz *orc_quiz_create(const OrcKB *kb) {
  OrcQuiz *q = (OrcQuiz *)calloc(1, sizeof(OrcQuiz));
  q->mants = (double *)aligned_alloc(64, (((size_t)kb->ldT * 8 + 63) / 64) * 64);
  q->exps = (int64_t *)aligned_alloc(64, (((size_t)kb->ldT * 8 + 63) / 64) * 64);
  memset(q->mants, 0, (size_t)kb->ldT * 8);
  memset(q->exps, 0, (size_t)kb->ldT * 8);
  q->asked = (uint8_t *)calloc(1, qbits_bytes(kb->nQuestions));
  return q;
}
void orc_quiz_destroy(OrcQuiz *q) { if (!q) return; free(q->mants); free(q->exps); free(q->asked); free(q); }

/* ------------------------------------------------------------------------------------------------------------------
 * a1: CEEvalQsSubtaskConsider<SRDoubleNumber>::Run.  PqaCore/CEEvalQsSubtaskConsider.cpp:41-217.
 * ---------------------------------------------------------------------------------------------------------------- */
static double calc_velocity_component(double V, int64_t nTargets) {             /* :24-34 */
  const double cLn0Stab = -746;                                                 /* CEEvalQsSubtaskConsider.h:21 */
  const double cLnMaxV = 0.34657359027997265470861606072909;                    /* SRMath.h:23 _cLnSqrt2 */
  const double lnV = ((V == 0) ? cLn0Stab : log(V));                            /* :29 */
  const double powT = (double)nTargets * (double)nTargets;                      /* :30 */
  return 1 / (cLnMaxV - lnV + cLnMaxV / powT);                                  /* :32 */
}

void orc_eval_subtask(const OrcKB *kb, const OrcQuiz *quiz, int64_t nValidTargets, int64_t iFirst, int64_t iLimit,
                      double *runLength, double *priority) {
  const int64_t K = kb->nAnswers, ldT = kb->ldT;
  const int64_t nTargVects = (kb->nTargets + 3) >> 2;                           /* :46 */
  const double *pPriors = quiz->mants;                                          /* :47 */
  double *invDi = (double *)malloc((size_t)nTargVects * 4 * 8);                 /* :49 stack scratch */
  double *post = (double *)malloc((size_t)nTargVects * 4 * 8);                  /* :50 */
  double *mW = (double *)malloc((size_t)K * 8), *mH = (double *)malloc((size_t)K * 8),
         *mV = (double *)malloc((size_t)K * 8);                                 /* :48 AnswerMetrics */

  OrcKahan1 accRunLength; orc_k1_init(&accRunLength, 0.0);                      /* :52 */
  for (int64_t i = iFirst; i < iLimit; i++) {
    if (bit_test(kb->questionGaps, i) || bit_test(quiz->asked, i)) {            /* :54 */
      runLength[i] = orc_k1_get(&accRunLength);                                 /* :56 */
      if (priority) priority[i] = 0;
      continue;
    }
    const double *pmDi = kb->D + (size_t)i * ldT;                               /* :59 */
    OrcKahan1 accTotW; orc_k1_init(&accTotW, 0.0);                              /* :60 */
    OrcKahan4 accL; orc_k4_reset(&accL);                                        /* :61 */
    for (int64_t k = 0; k < K; k++) {                                           /* :62 */
      OrcKahan4 accLhEnt; orc_k4_reset(&accLhEnt);                              /* :63 */
      const double *psAik = kb->A + ((size_t)i * K + k) * ldT;                  /* :64 */
      for (int64_t j = 0; j < nTargVects; j++) {                                /* :66 pass 1 */
        double lh[4];
        for (int c = 0; c < 4; c++) {
          const int64_t t = j * 4 + c;
          const int gap = bit_test(kb->targetGaps, t);                          /* :67-68 */
          if (k == 0) invDi[t] = gap ? 0.0 : 1.0 / pmDi[t];                     /* :72-76 andnot(gap, 1/D) */
          const double prQk = psAik[t] * invDi[t];                              /* :81 */
          lh[c] = gap ? 0.0 : prQk * pPriors[t];                                /* :82 */
          post[t] = lh[c];                                                      /* :84 */
        }
        orc_k4_add(&accLhEnt, lh);                                              /* :86 */
      }
      const double Wk = orc_k4_precise_sum(&accLhEnt);                          /* :88 */
      orc_k1_add(&accTotW, Wk);                                                 /* :89 */
      mW[k] = Wk;                                                               /* :90 */
      const double invWk = 1.0 / Wk;                                            /* :91 */

      orc_k4_reset(&accLhEnt);                                                  /* :93 */
      OrcKahan4 accV; orc_k4_reset(&accV);                                      /* :94 */
      for (int64_t j = 0; j < nTargVects; j++) {                                /* :95 pass 2 */
        double hv[4], lv[4], vv[4];
        for (int c = 0; c < 4; c++) {
          const int64_t t = j * 4 + c;
          const double posterior = post[t] * invWk;                             /* :97 */
          const int gap = bit_test(kb->targetGaps, t);                          /* :99-100 */
          const double prior = gap ? 0.0 : pPriors[t];                          /* :103 */
          const double l2post = gap ? 0.0 : orc_log2hot(posterior);             /* :106 */
          hv[c] = posterior * l2post;                                           /* :113 */
          const double invDij = invDi[t];                                       /* :116 */
          lv[c] = gap ? 0.0 : (invDij * invDij) / l2post;                       /* :117 */
          const double diff = posterior - prior;                                /* :119 */
          vv[c] = diff * diff;                                                  /* :126 */
        }
        orc_k4_add(&accLhEnt, hv);                                              /* :114 */
        orc_k4_add(&accL, lv);                                                  /* :117 */
        orc_k4_add(&accV, vv);                                                  /* :127 */
      }
      double velocity;
      const double entropyHik = -orc_k4_pair_sum(&accLhEnt, &accV, &velocity);  /* :130 */
      mH[k] = entropyHik;                                                       /* :131 */
      mV[k] = velocity;                                                         /* :132 */
    }
    const double totW = orc_k1_get(&accTotW);                                   /* :134 */

    OrcKahan4 accAvgH, accAvgV; orc_k4_reset(&accAvgH); orc_k4_reset(&accAvgV); /* :139-140 */
    const int64_t nVectorized = (K >> 2) << 2;                                  /* :141-142 */
    for (int64_t k = 0; k < nVectorized; k += 4) {                              /* :148-159 */
      double wh[4], wv[4];
      for (int c = 0; c < 4; c++) {
        wh[c] = mW[k + c] * mH[k + c];                                          /* :152 */
        wv[c] = mW[k + c] * sqrt(mV[k + c]);                                    /* :156-157 */
      }
      orc_k4_add(&accAvgH, wh);
      orc_k4_add(&accAvgV, wv);
    }
    for (int64_t k = nVectorized; k < K; k++) {                                 /* :163-172 */
      const double velocity = sqrt(mV[k]);                                      /* :165 */
      const int at = (int)(k - nVectorized);                                    /* :168 */
      orc_k4_add_at(&accAvgH, at, mW[k] * mH[k]);                               /* :167,:170 */
      orc_k4_add_at(&accAvgV, at, mW[k] * velocity);                            /* :171 */
    }
    double avgV;
    double avgH = orc_k4_pair_sum(&accAvgH, &accAvgV, &avgV);                   /* :175 */
    avgH = avgH / totW;                                                         /* :176-177 */
    avgV = avgV / totW;

    const double nExpectedTargets = exp2(avgH);                                 /* :181 */
    const double vComp = calc_velocity_component(avgV, nValidTargets + 1);      /* :191 */
    const double lack = -orc_k4_precise_sum(&accL);                             /* :201 */
    /* :207 std::pow(lack,1)*std::pow(vComp,9)*std::pow(nExpectedTargets,-2); integer powers per the author's TODO :206 */
    const double v2 = vComp * vComp, v4 = v2 * v2, v8 = v4 * v4, v9 = v8 * vComp;
    const double nExpM2 = 1.0 / (nExpectedTargets * nExpectedTargets);
    const double prio = lack * v9 * nExpM2;
    orc_k1_add(&accRunLength, prio);                                            /* :212 */
    runLength[i] = orc_k1_get(&accRunLength);                                   /* :214 */
    if (priority) priority[i] = prio;
  }
  free(invDi); free(post); free(mW); free(mH); free(mV);
}

void orc_eval_all(const OrcKB *kb, const OrcQuiz *quiz, int64_t nSubtasks, double *runLength, double *priority) {
  int64_t *bounds = (int64_t *)malloc((size_t)nSubtasks * 8);
  const int64_t n = orc_calc_split(kb->nQuestions, nSubtasks, bounds);          /* CpuEngine.cpp:355 */
  const int64_t nValid = kb->nTargets - kb->nTargetGaps;                        /* CpuEngine.cpp:352 */
  for (int64_t s = 0; s < n; s++)
    orc_eval_subtask(kb, quiz, nValid, s == 0 ? 0 : bounds[s - 1], bounds[s], runLength, priority);
  free(bounds);
}

/* ------------------------------------------------------------------------------------------------------------------
 * a4: selection.  PqaCore/CpuEngine.cpp:362-406, PqaCore/BaseEngine.cpp:60-124.
 * ---------------------------------------------------------------------------------------------------------------- */
static inline uint64_t pack64(const uint8_t *bits, int64_t iPack) { uint64_t u; memcpy(&u, bits + iPack * 8, 8); return u; }

int64_t orc_find_nearest_question(const OrcKB *kb, const OrcQuiz *quiz, int64_t iMiddle) {
  const uint32_t dInf = 200;                                                    /* BaseEngine.cpp:61 */
  const int64_t iPack64 = iMiddle >> 6;
  const uint32_t iWithin = (uint32_t)(iMiddle & 63);
  const uint64_t available = ~(pack64(kb->questionGaps, iPack64) | pack64(quiz->asked, iPack64)); /* :64-65 */
  if (available != 0) {
    const uint64_t baseMask = (1ULL << iWithin) - 1;                            /* :67 */
    const uint64_t higher = available & ~baseMask;                              /* :68 */
    const uint64_t lower = baseMask & available;                                /* :69 */
    const uint32_t dHigher = higher ? ((uint32_t)__builtin_ctzll(higher) - iWithin) : dInf;       /* :71 */
    const uint32_t dLower = lower ? (iWithin - (uint32_t)(63 - __builtin_clzll(lower))) : dInf;   /* :72 */
    return (dHigher < dLower) ? iMiddle + dHigher : iMiddle - dLower;           /* :73-78 */
  }
  const int64_t limPack64 = (kb->nQuestions + 63) >> 6;                         /* :80 */
  int64_t i = 1;
  while ((iPack64 >= i) && (iPack64 + i < limPack64)) {                         /* :82 */
    const uint64_t availLeft = ~(pack64(kb->questionGaps, iPack64 - i) | pack64(quiz->asked, iPack64 - i));
    const uint64_t availRight = ~(pack64(kb->questionGaps, iPack64 + i) | pack64(quiz->asked, iPack64 + i));
    if ((availLeft | availRight) == 0) { i++; continue; }
    const uint32_t dHigher = availRight ? ((uint32_t)__builtin_ctzll(availRight) + 64 - iWithin) : dInf;      /* :92 */
    const uint32_t dLower = availLeft ? (iWithin + 64 - (uint32_t)(63 - __builtin_clzll(availLeft))) : dInf;  /* :93 */
    if (dHigher < dLower) return iMiddle + dHigher + ((i - 1) << 6);
    return iMiddle - dLower - ((i - 1) << 6);
  }
  while (iPack64 >= i) {                                                        /* :101 */
    const uint64_t availLeft = ~(pack64(kb->questionGaps, iPack64 - i) | pack64(quiz->asked, iPack64 - i));
    if (!availLeft) { i++; continue; }
    const uint32_t dLower = iWithin + 64 - (uint32_t)(63 - __builtin_clzll(availLeft));
    return iMiddle - dLower - ((i - 1) << 6);
  }
  while (iPack64 + i < limPack64) {                                             /* :112 */
    const uint64_t availRight = ~(pack64(kb->questionGaps, iPack64 + i) | pack64(quiz->asked, iPack64 + i));
    if (!availRight) { i++; continue; }
    const uint32_t dHigher = (uint32_t)__builtin_ctzll(availRight) + 64 - iWithin;
    return iMiddle + dHigher + ((i - 1) << 6);
  }
  return -1;                                                                    /* :123 cInvalidPqaId */
}

static int64_t upper_bound_d(const double *a, int64_t n, double v) {            /* std::upper_bound: first a[i] > v */
  int64_t lo = 0, hi = n;
  while (lo < hi) { const int64_t mid = lo + ((hi - lo) >> 1); if (!(v < a[mid])) lo = mid + 1; else hi = mid; }
  return lo;
}

int64_t orc_select_sampled(const OrcKB *kb, const OrcQuiz *quiz, int64_t nSubtasks, const double *runLength,
                           uint64_t rnd) {
  const int64_t Q = kb->nQuestions;
  int64_t *bounds = (int64_t *)malloc((size_t)nSubtasks * 8);
  double *grandTotals = (double *)malloc((size_t)nSubtasks * 8);
  const int64_t n = orc_calc_split(Q, nSubtasks, bounds);
  OrcKahan1 accTotG; orc_k1_init(&accTotG, 0.0);                                /* CpuEngine.cpp:362 */
  for (int64_t i = 0; i < n; i++) {
    orc_k1_add(&accTotG, runLength[bounds[i] - 1]);                             /* :366-367 */
    grandTotals[i] = orc_k1_get(&accTotG);                                      /* :368 */
  }
  const double totG = grandTotals[n - 1];                                       /* :375 */
  /* SRDoubleNumber::MakeRandom, SRPlatform/Interface/SRDoubleNumber.h:35-39 */
  const double selRunLen = totG * (double)rnd / (double)UINT64_MAX;             /* :379 */
  int64_t sel;
  const int64_t iWorker = upper_bound_d(grandTotals, n, selRunLen);             /* :380-381 */
  if (iWorker >= n) {
    sel = Q - 1;                                                                /* :384 */
  } else {
    const double inWorkerRunLen = selRunLen - ((iWorker == 0) ? 0.0 : grandTotals[iWorker - 1]); /* :388 */
    const int64_t iFirst = (iWorker == 0) ? 0 : bounds[iWorker - 1];            /* :389 */
    const int64_t iLimit = bounds[iWorker];                                     /* :390 */
    sel = iFirst + upper_bound_d(runLength + iFirst, iLimit - iFirst, inWorkerRunLen); /* :391 */
    if (sel >= iLimit) sel = iLimit - 1;                                        /* :392-400 */
  }
  free(bounds); free(grandTotals);
  if (bit_test(kb->questionGaps, sel) || bit_test(quiz->asked, sel))            /* :404 */
    sel = orc_find_nearest_question(kb, quiz, sel);                             /* :405 */
  return sel;                                                                   /* -1 => QuestionsExhausted :407-410 */
}

int64_t orc_select_argmax(const OrcKB *kb, const OrcQuiz *quiz, const double *priority) {
  int64_t best = -1; double bestP = 0;
  for (int64_t i = 0; i < kb->nQuestions; i++) {
    if (bit_test(kb->questionGaps, i) || bit_test(quiz->asked, i)) continue;
    if (best < 0 || priority[i] > bestP) { best = i; bestP = priority[i]; }
  }
  return best;
}

/* ------------------------------------------------------------------------------------------------------------------
 * prior updates
 * ---------------------------------------------------------------------------------------------------------------- */
/* CEBaseDivTargPriorsSubtask::RunInternal, PqaCore/CEDivTargPriorsSubtask.h:12-23: true division of every lane */
static void div_priors(OrcQuiz *quiz, int64_t nVects, double sumPriors) {
  for (int64_t t = 0; t < nVects * 4; t++) quiz->mants[t] = quiz->mants[t] / sumPriors;
}

/* Summator::ForPriors, PqaCore/Summator.h:11-21: serial Kahan over the subtask sums, in subtask order */
static double summator(const double *sums, int64_t n) {
  OrcKahan1 acc; orc_k1_init(&acc, 0.0);
  for (int64_t i = 0; i < n; i++) orc_k1_add(&acc, sums[i]);
  return orc_k1_get(&acc);
}

void orc_start_quiz(const OrcKB *kb, OrcQuiz *quiz, int64_t nWorkers) {
  const int64_t nVects = (kb->nTargets + 3) >> 2;                               /* CECreateQuizOperation.cpp:38 */
  int64_t *bounds = (int64_t *)malloc((size_t)nWorkers * 8);
  double *sums = (double *)malloc((size_t)nWorkers * 8);
  const int64_t n = orc_calc_split(nVects, nWorkers, bounds);                   /* :39 */
  memset(quiz->asked, 0, qbits_bytes(kb->nQuestions));                          /* CpuEngine.cpp:214 */
  for (int64_t s = 0; s < n; s++) {                                             /* CESetPriorsSubtaskSum.cpp:27-35 */
    OrcKahan4 acc; orc_k4_reset(&acc);
    for (int64_t j = (s == 0 ? 0 : bounds[s - 1]); j < bounds[s]; j++) {
      double v[4];
      for (int c = 0; c < 4; c++) {
        const int64_t t = j * 4 + c;
        v[c] = bit_test(kb->targetGaps, t) ? 0.0 : kb->B[t];                    /* :28-30 */
        quiz->mants[t] = v[c];                                                  /* :31 */
        quiz->exps[t] = 0;                                                      /* :32 */
      }
      orc_k4_add(&acc, v);                                                      /* :33 */
    }
    sums[s] = orc_k4_precise_sum(&acc);                                         /* :35 */
  }
  div_priors(quiz, nVects, summator(sums, n));                                  /* CECreateQuizOperation.cpp:48-51 */
  free(bounds); free(sums);
}

void orc_record_answer(const OrcKB *kb, OrcQuiz *quiz, int64_t iQuestion, int64_t iAnswer, int64_t nWorkers) {
  const int64_t nVects = (kb->nTargets + 3) >> 2;                               /* CEQuiz.h:109 */
  int64_t *bounds = (int64_t *)malloc((size_t)nWorkers * 8);
  double *sums = (double *)malloc((size_t)nWorkers * 8);
  const int64_t n = orc_calc_split(nVects, nWorkers, bounds);                   /* CEQuiz.h:110 */
  bit_set(quiz->asked, iQuestion, 1);                                           /* CEQuiz.h:91 */
  const double *pMul = kb->A + ((size_t)iQuestion * kb->nAnswers + iAnswer) * kb->ldT; /* CERecordAnswerSubtaskMul.cpp:25 */
  const double *pDiv = kb->D + (size_t)iQuestion * kb->ldT;                     /* :26 */
  for (int64_t s = 0; s < n; s++) {
    OrcKahan4 acc; orc_k4_reset(&acc);
    for (int64_t j = (s == 0 ? 0 : bounds[s - 1]); j < bounds[s]; j++) {        /* :27 */
      double v[4];
      for (int c = 0; c < 4; c++) {
        const int64_t t = j * 4 + c;
        const double pQaGivenT = pMul[t] / pDiv[t];                             /* :31 */
        const double product = quiz->mants[t] * pQaGivenT;                      /* :34 */
        v[c] = bit_test(kb->targetGaps, t) ? 0.0 : product;                     /* :35-36 */
        quiz->mants[t] = v[c];                                                  /* :37 */
      }
      orc_k4_add(&acc, v);                                                      /* :39 */
    }
    sums[s] = orc_k4_precise_sum(&acc);                                         /* :41 */
  }
  div_priors(quiz, nVects, summator(sums, n));                                  /* CEQuiz.h:117-120 */
  free(bounds); free(sums);
}

static int ceil_log2_u64(uint64_t val) {                                        /* SRPlatform/Interface/SRMath.h:46-51 */
  if (!val) return 0;
  const int index = 63 - __builtin_clzll(val);
  return index + ((val & (val - 1)) ? 1 : 0);
}

int orc_resume_quiz(const OrcKB *kb, OrcQuiz *quiz, int64_t nAnswered, const OrcAQ *aqs, int64_t nWorkers,
                    int bugCompat) {
  const int64_t nVects = (kb->nTargets + 3) >> 2;                               /* CECreateQuizOperation.cpp:72 */
  int64_t *bounds = (int64_t *)malloc((size_t)nWorkers * 8);
  double *sums = (double *)malloc((size_t)nWorkers * 8);
  const int64_t n = orc_calc_split(nVects, nWorkers, bounds);                   /* :73 */
  memset(quiz->asked, 0, qbits_bytes(kb->nQuestions));                          /* CpuEngine.cpp:214 */
  for (int64_t i = 0; i < nAnswered; i++) bit_set(quiz->asked, aqs[i].iQuestion, 1); /* CpuEngine.cpp:232 */

  /* a8: CEUpdatePriorsSubtaskMul::RunInternal, PqaCore/CEUpdatePriorsSubtaskMul.cpp:16-114.  The L1 blocking and
   * cache flushes (:38,:86-99) change no arithmetic: each target's chain of products is independent. */
  for (int64_t t = 0; t < nVects * 4; t++) {
    {                                                                           /* :43-62 first answered question */
      const OrcAQ aq = aqs[0];
      const double adjMul = kb->A[((size_t)aq.iQuestion * kb->nAnswers + aq.iAnswer) * kb->ldT + t];
      const double adjDiv = kb->D[(size_t)aq.iQuestion * kb->ldT + t];
      const double pQaGivenT = adjMul / adjDiv;                                 /* :51 */
      const double oldMant = bugCompat ? kb->B[t & 3] : kb->B[t];               /* :53 loads pvB, not pvB + j */
      const double product = oldMant * pQaGivenT;                               /* :54 */
      const uint64_t up = d2u(product);
      quiz->mants[t] = u2d(EXP0_UP | (up & ~EXP_MASK_UP));                      /* :56 MakeExponent0, SRSimd.h:194-198 */
      quiz->exps[t] = (int64_t)((up & EXP_MASK_UP) >> EXP_OFFS);                /* :59 ExtractExponents64<false>, SRSimd.h:130-137 */
    }
    for (int64_t i = 1; i < nAnswered; i++) {                                   /* :63-85 */
      const OrcAQ aq = aqs[i];
      const double adjMul = kb->A[((size_t)aq.iQuestion * kb->nAnswers + aq.iAnswer) * kb->ldT + t];
      const double adjDiv = kb->D[(size_t)aq.iQuestion * kb->ldT + t];
      const double pQaGivenT = adjMul / adjDiv;                                 /* :71 */
      const double product = quiz->mants[t] * pQaGivenT;                        /* :75 */
      const uint64_t up = d2u(product);
      quiz->mants[t] = u2d(EXP0_UP | (up & ~EXP_MASK_UP));                      /* :77 */
      quiz->exps[t] += (int64_t)((up & EXP_MASK_UP) >> EXP_OFFS);               /* :80-82 */
    }
  }

  /* a9: CpuEngine::NormalizePriors, PqaCore/CpuEngine.cpp:284-335 */
  int64_t fullMax = INT64_MIN;
  for (int64_t t = 0; t < nVects * 4; t++) {                                    /* CENormPriorsSubtaskMax.cpp:25-31,47-53 */
    const int64_t totExp = quiz->exps[t] + (int64_t)((d2u(quiz->mants[t]) & EXP_MASK_UP) >> EXP_OFFS);
    if (!bit_test(kb->targetGaps, t) && totExp > fullMax) fullMax = totExp;     /* MaxI64 with retention, SRSimd.h:221-225 */
  }
  const int64_t highBound = 1023 + 1023 - ceil_log2_u64((uint64_t)kb->nTargets) - 2; /* CpuEngine.cpp:316 */
  const int64_t minAllowed = INT64_MIN + highBound + 1;                         /* :317 */
  if (fullMax <= minAllowed) { free(bounds); free(sums); return ORC_I64_UNDERFLOW; } /* :318-321 */
  const int64_t corrExp = highBound - fullMax;                                  /* :322 */

  for (int64_t s = 0; s < n; s++) {                                             /* CENormPriorsSubtaskCorrSum.cpp:47-63 */
    OrcKahan4 acc; orc_k4_reset(&acc);
    for (int64_t j = (s == 0 ? 0 : bounds[s - 1]); j < bounds[s]; j++) {
      double v[4];
      for (int c = 0; c < 4; c++) {
        const int64_t t = j * 4 + c;
        const uint64_t um = d2u(quiz->mants[t]);                                /* :26 */
        const int64_t origExp = quiz->exps[t] + (int64_t)((um & EXP_MASK_UP) >> EXP_OFFS); /* :27-28 */
        const int64_t normExp = origExp + corrExp;                              /* :30 */
        const int assume0 = (1 > normExp) || bit_test(kb->targetGaps, t);       /* :32-33 */
        /* ReplaceExponents, SRSimd.h:200-204 */
        v[c] = assume0 ? 0.0 : u2d(((uint64_t)normExp << EXP_OFFS) | (um & ~EXP_MASK_UP)); /* :35-36 */
        quiz->exps[t] = 0;                                                      /* :37 */
        quiz->mants[t] = v[c];                                                  /* :38 */
      }
      orc_k4_add(&acc, v);                                                      /* :57 */
    }
    sums[s] = orc_k4_precise_sum(&acc);                                         /* :59 */
  }
  div_priors(quiz, nVects, summator(sums, n));                                  /* CpuEngine.cpp:327-332 */
  free(bounds); free(sums);
  return ORC_OK;
}

/* ------------------------------------------------------------------------------------------------------------------
 * f2: ListTopTargets.  CpuEngine::ListTopTargetsSpec (PqaCore/CpuEngine.cpp:417-440),
 * CEListTopTargetsAlgorithm::RunHeapifyBased (PqaCore/CEListTopTargetsAlgorithm.cpp:30-95),
 * CEHeapifyPriorsSubtaskMake (PqaCore/CEHeapifyPriorsSubtaskMake.cpp:42-52 Regard, :56-88 Run),
 * SRHeapHelper::Down (SRPlatform/Interface/SRHeap.h:16-39), RatedTarget / RatingsHeapItem order by probability only
 * (PqaCore/Interface/PqaCommon.h:58-60, PqaCore/RatingsHeap.h:18-20).
 *
 * std::make_heap / std::pop_heap are the C++ library's: the reference binary carries MSVC's.  Restated below is the
 * algorithm MSVC's <algorithm> (_Make_heap_unchecked / _Pop_heap_hole_by_index / _Push_heap_by_index), libstdc++'s
 * (__make_heap / __adjust_heap / __push_heap) and libc++'s pre-Floyd versions share: the hole sinks to a leaf taking the
 * right child unless it is less than the left one (a last lone left child is taken too), then the lifted value climbs
 * while its parent is less.  With distinct probabilities the listing does not depend on it; the order of EQUAL
 * probabilities does.  tests/test_oracle.py pins this restatement to libstdc++'s own std::make_heap / std::pop_heap
 * (a checker compiled there with g++); against MSVC's it is "parity unpinned" (no MSVC here) beyond that shared algorithm.
 *
 * The radix-sort branch (CEListTopTargetsAlgorithm.cpp:97-173, CERadixSortRatingsSubtaskSort.cpp:61-140), which the cost model
 * of CpuEngine.cpp:423-434 picks for long lists (orc_list_top_targets_takes_radix), is not restated: its scatter passes
 * never advance the bucket offsets (CERadixSortRatingsSubtaskSort.cpp:111,126 store to pOffsets[bucket] without ++), so all
 * but one slot per bucket keep whatever the memory pool held -- its output is not a function of its inputs.  Both branches
 * filter the same way: gaps and prob <= 0 are dropped (CEHeapifyPriorsSubtaskMake.cpp:43-49,
 * CERadixSortRatingsSubtaskSort.cpp:78-84).
 * ---------------------------------------------------------------------------------------------------------------- */
typedef struct { double prob; int64_t id; } HeapItem;   /* RatedTarget (id = iTarget) or RatingsHeapItem (id = iSource) */

static void heap_push_by_index(HeapItem *first, int64_t hole, int64_t top, HeapItem val) {
  for (int64_t idx = (hole - 1) >> 1; top < hole && first[idx].prob < val.prob; idx = (hole - 1) >> 1) {
    first[hole] = first[idx];
    hole = idx;
  }
  first[hole] = val;
}
static void heap_adjust(HeapItem *first, int64_t hole, int64_t bottom, HeapItem val) {
  const int64_t top = hole;
  int64_t idx = hole;
  const int64_t maxNonLeaf = (bottom - 1) >> 1;
  while (idx < maxNonLeaf) {                         /* the hole moves down to the larger child */
    idx = 2 * idx + 2;
    if (first[idx].prob < first[idx - 1].prob) --idx;
    first[hole] = first[idx];
    hole = idx;
  }
  if (idx == maxNonLeaf && bottom % 2 == 0) {        /* an only child at the bottom */
    first[hole] = first[bottom - 1];
    hole = bottom - 1;
  }
  heap_push_by_index(first, hole, top, val);
}
static void heap_make(HeapItem *first, int64_t n) {                             /* std::make_heap */
  for (int64_t hole = n >> 1; hole > 0;) {
    --hole;
    heap_adjust(first, hole, n, first[hole]);
  }
}
static void heap_pop(HeapItem *first, int64_t n) {                              /* std::pop_heap: the top goes to first[n-1] */
  if (n < 2) return;
  const HeapItem val = first[n - 1];
  first[n - 1] = first[0];
  heap_adjust(first, 0, n - 1, val);
}
static void heap_down(HeapItem *first, int64_t n) {                             /* SRHeapHelper::Down, SRHeap.h:16-39 */
  int64_t cur = 0;
  for (;;) {
    const int64_t child1 = 2 * cur + 1;
    if (child1 >= n) return;                                                    /* :21-23 */
    const int64_t child2 = child1 + 1;
    if (child2 >= n) {                                                          /* :25-30 */
      if (first[cur].prob < first[child1].prob) { const HeapItem t = first[cur]; first[cur] = first[child1]; first[child1] = t; }
      return;
    }
    const int64_t higher = (first[child2].prob < first[child1].prob) ? child1 : child2;   /* :31 */
    if (!(first[cur].prob < first[higher].prob)) return;                        /* :32-34 */
    { const HeapItem t = first[cur]; first[cur] = first[higher]; first[higher] = t; }   /* :35 */
    cur = higher;                                                               /* :36 */
  }
}

/* test access to the three heap steps above (tests/test_oracle.py holds them to libstdc++'s) */
void orc_heap_make(double *prob, int64_t *id, int64_t n) {
  HeapItem *h = (HeapItem *)malloc((size_t)(n > 0 ? n : 1) * sizeof(HeapItem));
  for (int64_t i = 0; i < n; i++) { h[i].prob = prob[i]; h[i].id = id[i]; }
  heap_make(h, n);
  for (int64_t i = 0; i < n; i++) { prob[i] = h[i].prob; id[i] = h[i].id; }
  free(h);
}
void orc_heap_pop(double *prob, int64_t *id, int64_t n) {
  HeapItem *h = (HeapItem *)malloc((size_t)(n > 0 ? n : 1) * sizeof(HeapItem));
  for (int64_t i = 0; i < n; i++) { h[i].prob = prob[i]; h[i].id = id[i]; }
  heap_pop(h, n);
  for (int64_t i = 0; i < n; i++) { prob[i] = h[i].prob; id[i] = h[i].id; }
  free(h);
}

int orc_list_top_targets_takes_radix(int64_t nTargets, int64_t nWorkers, int64_t maxCount) {   /* CpuEngine.cpp:423-434 */
  const uint64_t nTargPerThread = ((uint64_t)nTargets + (uint64_t)nWorkers - 1) / (uint64_t)nWorkers;   /* :423 PosDivideRoundUp */
  const uint64_t logW = (uint64_t)ceil_log2_u64((uint64_t)nWorkers), logT = (uint64_t)ceil_log2_u64((uint64_t)nTargets);
  const uint64_t nRadixSortOps = 9 * (nTargPerThread > 256 ? nTargPerThread : 256) + (uint64_t)maxCount * (logW > 1 ? logW : 1);   /* :426-427 */
  const uint64_t nHeapifyOps = 3 * nTargPerThread + (uint64_t)maxCount * logT;                          /* :428 */
  return nRadixSortOps < nHeapifyOps;                                                                   /* :431 */
}

int64_t orc_list_top_targets(const OrcKB *kb, const OrcQuiz *quiz, int64_t maxCount, int64_t nWorkers, OrcRatedTarget *dest) {
  const int64_t T = kb->nTargets;
  int64_t *bounds = (int64_t *)malloc((size_t)nWorkers * sizeof(int64_t));
  int64_t *pieceLimits = (int64_t *)malloc((size_t)nWorkers * sizeof(int64_t));
  HeapItem *headHeap = (HeapItem *)malloc((size_t)nWorkers * sizeof(HeapItem));
  HeapItem *ratings = (HeapItem *)malloc((size_t)(T > 0 ? T : 1) * sizeof(HeapItem));
  const int64_t nSub = orc_calc_split(T, nWorkers, bounds);                     /* CEListTopTargetsAlgorithm.cpp:46 */
  for (int64_t s = 0; s < nSub; s++) {                                          /* CEHeapifyPriorsSubtaskMake::Run, one per piece */
    const int64_t iFirst = s == 0 ? 0 : bounds[s - 1], iLimit = bounds[s];
    int64_t iSelLim = iFirst;                                                   /* CEHeapifyPriorsSubtaskMake.cpp:31 */
    for (int64_t t = iFirst; t < iLimit; t++) {                                 /* :66-83 (the unrolling changes no order) */
      if (bit_test(kb->targetGaps, t)) continue;                                /* :43-45 */
      const double prob = quiz->mants[t];                                       /* :46 */
      if (prob <= 0) continue;                                                  /* :47-49 */
      ratings[iSelLim].prob = prob;                                             /* :50-52 */
      ratings[iSelLim].id = t;
      iSelLim++;
    }
    pieceLimits[s] = iSelLim;                                                   /* :86 */
    heap_make(ratings + iFirst, iSelLim - iFirst);                              /* :87 */
  }
  /* RecalcToStarts (SRPoolRunner.h:71-77): pStarts[i] = start of piece i */
  int64_t nHh = 0;
  for (int64_t i = 0; i < nSub; i++) {                                          /* CEListTopTargetsAlgorithm.cpp:58-66 */
    const int64_t curFirst = i == 0 ? 0 : bounds[i - 1];
    if (pieceLimits[i] == curFirst) continue;
    headHeap[nHh].id = i;
    headHeap[nHh].prob = ratings[curFirst].prob;
    nHh++;
  }
  heap_make(headHeap, nHh);                                                     /* :67 */
  int64_t listed = maxCount;
  for (int64_t i = 0; i < maxCount; i++) {                                      /* :69-92 */
    if (nHh == 0) { listed = i; break; }                                        /* :71-73 */
    dest[i].prob = headHeap[0].prob;                                            /* :74 */
    const int64_t curPiece = headHeap[0].id;
    const int64_t pieceStart = curPiece == 0 ? 0 : bounds[curPiece - 1];
    dest[i].iTarget = ratings[pieceStart].id;                                   /* :77 */
    const int64_t pieceLim = pieceLimits[curPiece];
    if (pieceStart + 1 == pieceLim) {                                           /* :81-88 the piece is exhausted */
      heap_pop(headHeap, nHh);
      nHh--;
      continue;
    }
    heap_pop(ratings + pieceStart, pieceLim - pieceStart);                      /* :90 */
    pieceLimits[curPiece]--;                                                    /* :91 */
    headHeap[0].prob = ratings[pieceStart].prob;                                /* :93 */
    heap_down(headHeap, nHh);                                                   /* :94 */
  }
  free(bounds); free(pieceLimits); free(headHeap); free(ratings);
  return listed;                                                                /* :73 / :97 */
}
