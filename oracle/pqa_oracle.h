/*
 * pqa_oracle.h -- CPU restatement of ProbQA's CpuEngine<SRDoubleNumber> hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product library (probqa_amd/csrc, libPqaCore.so) includes,
 * links or calls this code.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it,
 * and only as the checker / the timed CPU baseline.
 *
 * Every function cites the reference file:line it follows (paths relative to /root/reference/ProbQA).
 * The reference itself is MSVC/Win32-only (SRPlatform/Interface/SRPlatform.h:43-45 is an unconditional #error for
 * other compilers) and cannot be built or imported here, so this restatement *is* the oracle.
 *
 * PINNING STATUS
 *   pinned   : Log2Hot (SRPlatformTests/SRVectMathTest.cpp:45-103) and the 4-lane Kahan accumulator
 *              (SRPlatformTests/SRAccumulatorTest.cpp:21-35) reproduce the reference's own known-answer tests;
 *              fresh-KB values (PqaCoreTests/Dimensions.cpp:60-77); DichotomyTest's >=98 % criterion
 *              (PqaCoreTests/DichotomyTest.cpp:99, shortened) is run end-to-end.
 *   unpinned : the reference holds NO golden priorities / posteriors / selections, so the per-question priority
 *              vector and the posterior vectors are "parity unpinned" beyond the pieces above: they are pinned
 *              only by source-order restatement + an independent high-precision definition check (tests/).
 *
 * Arithmetic: IEEE fp64, source order, compile with -ffp-contract=off; fma() appears exactly where the reference
 * writes _mm256_fmadd_pd.  AVX lanes are emulated as 4 scalar lanes (lane c handles targets j == c mod 4).
 */
#ifndef PQA_ORACLE_H
#define PQA_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Knowledge base.  Rows are padded to a multiple of 4 doubles (SRFastArray, SRPlatform/Interface/SRFastArray.h:34-36);
 * bit arrays are LSB-first and bits past the size are 1 == "gap" (PqaCore/GapTracker.h:9-15). */
typedef struct {
  int64_t nAnswers, nQuestions, nTargets;
  int64_t ldT;            /* row stride in doubles, multiple of 4, >= nTargets */
  double *A;              /* [nQuestions][nAnswers][ldT]  squares of counts  (PqaCore/CpuEngine.decl.h:31-37) */
  double *D;              /* [nQuestions][ldT]            sum over answers of A */
  double *B;              /* [ldT]                        target counts */
  uint8_t *targetGaps;    /* ceil(ldT/8)+8 bytes */
  uint8_t *questionGaps;  /* ceil(nQuestions/64)*8+8 bytes */
  int64_t nTargetGaps;
} OrcKB;

typedef struct {
  double *mants;          /* [ldT] prior mantissas == prior probabilities once normalised (PqaCore/CEQuiz.decl.h:13-44) */
  int64_t *exps;          /* [ldT] */
  uint8_t *asked;         /* question bits, same sizing as questionGaps, bits past size 0 */
} OrcQuiz;

typedef struct { int64_t iQuestion, iAnswer; } OrcAQ;

/* error codes of the path (PqaCore/Interface/PqaErrors.h:12-40) */
enum { ORC_OK = 0, ORC_I64_UNDERFLOW = 16, ORC_QUESTIONS_EXHAUSTED = 17 };

/* ---- a2: SRVectMath::Log2Hot  (SRPlatform/Interface/SRVectMath.h:87-135, table SRPlatform/SRVectMath.cpp:30-44) */
double orc_log2hot(double x);
const double *orc_log2hot_table(void); /* 1024 entries */

/* ---- a3: SRAccumVectDbl256 / SRAccumulator (SRPlatform/Interface/SRAccumVectDbl256.h, SRAccumulator.h) */
typedef struct { double sum[4], corr[4]; } OrcKahan4;
typedef struct { double sum, corr; } OrcKahan1;
void   orc_k4_reset(OrcKahan4 *a);
void   orc_k4_add(OrcKahan4 *a, const double v[4]);
void   orc_k4_add_at(OrcKahan4 *a, int at, double v);
double orc_k4_precise_sum(const OrcKahan4 *a);
double orc_k4_pair_sum(const OrcKahan4 *a, const OrcKahan4 *fellow, double *fellowSum);
double orc_k4_full_sum(const OrcKahan4 *a);
void   orc_k1_init(OrcKahan1 *a, double v);
void   orc_k1_add(OrcKahan1 *a, double v);
double orc_k1_get(const OrcKahan1 *a);

/* ---- SRPoolRunner::CalcSplit (SRPlatform/Interface/SRPoolRunner.h:96-110); returns nSubtasks, fills bounds[] */
int64_t orc_calc_split(int64_t nItems, int64_t nWorkers, int64_t *bounds);

/* ---- KB helpers (PqaCore/CpuEngine.cpp:44-84 fresh KB; PqaCore/CETrainOperation.cpp:15-25 training) */
OrcKB  *orc_kb_create(int64_t nAnswers, int64_t nQuestions, int64_t nTargets, double initAmount);
void    orc_kb_destroy(OrcKB *kb);
void    orc_kb_set_target_gap(OrcKB *kb, int64_t t, int isGap);
void    orc_kb_set_question_gap(OrcKB *kb, int64_t q, int isGap);
/* Train: A += 2*sqrt(A)*b + b^2, D += same, B[t] += b; repeated questions by the rules of CETrainOperation::Perform2
 * (PqaCore/CETrainOperation.cpp:32-83) in the bucket order of CpuEngine::TrainSpec (nWorkers buckets by iQuestion % nWorkers,
 * each consumed newest first, two at a time; CETrainSubtaskDistrib.h:46-52, CETrainSubtaskAdd.cpp:17-38). */
void    orc_kb_train_workers(OrcKB *kb, int64_t nAQs, const OrcAQ *aqs, int64_t iTarget, double amount, int64_t nWorkers);
void    orc_kb_train(OrcKB *kb, int64_t nAQs, const OrcAQ *aqs, int64_t iTarget, double amount);   /* nWorkers = 1 */
/* RecordQuizTarget (PqaCore/CpuEngine.cpp:442-466): the answers in order, pairwise through Perform2 */
void    orc_kb_record_quiz_target(OrcKB *kb, int64_t nAQs, const OrcAQ *aqs, int64_t iTarget, double amount);
OrcQuiz *orc_quiz_create(const OrcKB *kb);
void    orc_quiz_destroy(OrcQuiz *q);

/* ---- a1: CEEvalQsSubtaskConsider<SRDoubleNumber>::Run (PqaCore/CEEvalQsSubtaskConsider.cpp:41-217)
 * One subtask over questions [iFirst,iLimit).  runLength (required) receives the per-subtask inclusive Kahan running sum;
 * priority (optional) receives the raw priority of each evaluated question, 0 for gap/asked questions. */
void orc_eval_subtask(const OrcKB *kb, const OrcQuiz *quiz, int64_t nValidTargets, int64_t iFirst, int64_t iLimit,
                      double *runLength, double *priority);
/* Convenience: whole question range as nSubtasks subtasks (PqaCore/CpuEngine.cpp:355-360); single-threaded. */
void orc_eval_all(const OrcKB *kb, const OrcQuiz *quiz, int64_t nSubtasks, double *runLength, double *priority);

/* ---- a4: CpuEngine::NextQuestionSpec selection (PqaCore/CpuEngine.cpp:362-406) given runLength from orc_eval_all and
 * the 64-bit random number the reference would draw (SRPlatform/Interface/SRDoubleNumber.h:35-39).
 * Returns the selected question or -1 (QuestionsExhausted). */
int64_t orc_select_sampled(const OrcKB *kb, const OrcQuiz *quiz, int64_t nSubtasks, const double *runLength,
                           uint64_t rnd);
/* BaseEngine::FindNearestQuestion (PqaCore/BaseEngine.cpp:60-124) */
int64_t orc_find_nearest_question(const OrcKB *kb, const OrcQuiz *quiz, int64_t iMiddle);
/* north-star selector: index of the maximum priority, lowest index on ties; -1 if none is eligible */
int64_t orc_select_argmax(const OrcKB *kb, const OrcQuiz *quiz, const double *priority);

/* ---- a7+a6: StartQuiz priors (PqaCore/CESetPriorsSubtaskSum.cpp:17-40, CEDivTargPriorsSubtask.h:12-30,
 *      Summator.h:11-21, CECreateQuizOperation.cpp:22-53).  nWorkers = thread-pool size of the emulated machine. */
void orc_start_quiz(const OrcKB *kb, OrcQuiz *quiz, int64_t nWorkers);
/* ---- a5+a6: RecordAnswer (PqaCore/CERecordAnswerSubtaskMul.cpp:15-42, CEQuiz.h:77-122) */
void orc_record_answer(const OrcKB *kb, OrcQuiz *quiz, int64_t iQuestion, int64_t iAnswer, int64_t nWorkers);
/* ---- a8+a9+a6: ResumeQuiz (PqaCore/CEUpdatePriorsSubtaskMul.cpp:16-114, CpuEngine.cpp:284-335,
 *      CENormPriorsSubtaskMax.cpp:37-54, CENormPriorsSubtaskCorrSum.cpp:47-63).
 *      bugCompat!=0 reproduces CEUpdatePriorsSubtaskMul.cpp:53 (vector 0 of vB used for every target vector). */
int orc_resume_quiz(const OrcKB *kb, OrcQuiz *quiz, int64_t nAnswered, const OrcAQ *aqs, int64_t nWorkers,
                    int bugCompat);

/* ---- f2: ListTopTargets -- CEListTopTargetsAlgorithm::RunHeapifyBased (PqaCore/CEListTopTargetsAlgorithm.cpp:30-95) over the
 * pieces of CEHeapifyPriorsSubtaskMake (PqaCore/CEHeapifyPriorsSubtaskMake.cpp:42-88): gaps and prob <= 0 dropped, descending;
 * equal probabilities in the order the per-piece heaps and the head heap of an nWorkers-thread pool give.  Returns the number
 * listed (<= maxCount).  orc_list_top_targets_takes_radix: 1 where the reference's cost model (PqaCore/CpuEngine.cpp:423-434)
 * would take its radix-sort branch instead, which is not restated (pqa_oracle.c says why). */
typedef struct { int64_t iTarget; double prob; } OrcRatedTarget;               /* RatedTarget, PqaCore/Interface/PqaCommon.h:54-61 */
int64_t orc_list_top_targets(const OrcKB *kb, const OrcQuiz *quiz, int64_t maxCount, int64_t nWorkers, OrcRatedTarget *dest);
int     orc_list_top_targets_takes_radix(int64_t nTargets, int64_t nWorkers, int64_t maxCount);
/* the restated std::make_heap / std::pop_heap over (prob, id) records ordered by prob, for the test that holds them to libstdc++'s */
void    orc_heap_make(double *prob, int64_t *id, int64_t n);
void    orc_heap_pop(double *prob, int64_t *id, int64_t n);

/* ---- AVX2 + pthreads restatement of a1 for the timed CPU baseline (pqa_oracle_avx2.c).  Bit-identical to
 * orc_eval_all (checked in tests).  nThreads worker threads, nSubtasks = 8*nThreads as PqaCore/CpuEngine.cpp:339. */
void orc_eval_all_avx2_mt(const OrcKB *kb, const OrcQuiz *quiz, int64_t nThreads, int64_t nSubtasks,
                          double *runLength, double *priority);
int  orc_have_avx2(void);

#ifdef __cplusplus
}
#endif
#endif
