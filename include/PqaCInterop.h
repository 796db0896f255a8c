/*
 * PqaCInterop.h -- C ABI of libPqaCore.so, the MI355X-native drop-in for ProbQA's PqaCore.dll.
 *
 * The 40 functions and 5 PODs below are exactly the surface the reference exports and that its Python (ctypes) and
 * .NET (P/Invoke) wrappers bind:
 *     reference: ProbQA/PqaCore/Interface/PqaCInterop.h:9-42 (PODs, #pragma pack(8)), :48-108 (functions);
 *     bound by  Interop/Python/ProbQAInterop/ProbQA.py:72-296 and the .cs files of ProbQA/ProbQANetCore.
 * Names, argument order, types and the error convention are unchanged:
 *   - functions returning void* give NULL on success or an opaque PqaError* the caller frees with CiReleasePqaError;
 *   - value-returning functions take `void **ppError` (set to NULL / PqaError*) and return -1 (or 0) on failure;
 *   - PqaError_ToString returns a heap string freed with CiReleaseString.
 *
 * Behind the ABI there is ONE engine: the HIP engine (gfx950 kernels).  PqaEngineFactory_CreateCpuEngine keeps its
 * name for binary compatibility and creates that engine; there is no CPU fallback -- without a usable GPU the factory
 * returns an error.  MI355X-specific additions are in PqaHipExt.h.
 */
#ifndef PQA_C_INTEROP_H
#define PQA_C_INTEROP_H

#include <stdint.h>
#ifndef __cplusplus
#include <stdbool.h>
#endif

#if defined(__GNUC__)
#define PQACORE_API __attribute__((visibility("default")))
#else
#define PQACORE_API
#endif

#pragma pack(push, 8)
typedef struct {            /* reference PqaCInterop.h:10-19 */
  int64_t _nAnswers;
  int64_t _nQuestions;
  int64_t _nTargets;
  uint8_t _precType;        /* TPqaPrecisionType: Float=1, Double=3 (reference PqaCommon.h:17-24) */
  uint16_t _precExponent;
  uint32_t _precMantissa;
  double _initAmount;
  uint64_t _memPoolMaxBytes;
} CiEngineDefinition;

typedef struct {            /* :21-24 */
  int64_t _iQuestion;
  int64_t _iAnswer;
} CiAnsweredQuestion;

typedef struct {            /* :26-30 */
  int64_t _nAnswers;
  int64_t _nQuestions;
  int64_t _nTargets;
} CiEngineDimensions;

typedef struct {            /* :32-35 */
  int64_t _iTarget;
  double _prob;
} CiRatedTarget;

typedef struct {            /* :37-40 */
  int64_t _index;
  double _initAmount;
} CiAddQorTParam;
#pragma pack(pop)

#ifdef __cplusplus
extern "C" {
#define PQA_DEFAULT(x) = x
#else
#define PQA_DEFAULT(x)
#endif

PQACORE_API void CiDebugBreak(void);                                                             /* :48 */

PQACORE_API uint8_t Logger_Init(void **ppStrErr, const char *baseName);                          /* :50 */
PQACORE_API void CiReleaseString(void *pvString);                                                /* :51 */

PQACORE_API void *CiGetPqaEngineFactory(void);                                                   /* :53 */
PQACORE_API void *PqaEngineFactory_CreateCpuEngine(void *pvFactory, void **ppError, const CiEngineDefinition *pEngDef); /* :54 */
PQACORE_API void *PqaEngineFactory_LoadCpuEngine(void *pvFactory, void **ppError, const char *filePath,
                                                 uint64_t memPoolMaxBytes);                      /* :55-56 */

PQACORE_API void CiReleasePqaError(void *pvErr);                                                 /* :58 */
PQACORE_API void *PqaError_ToString(void *pvError, const uint8_t withParams);                    /* :59 */

PQACORE_API void CiReleasePqaEngine(void *pvEngine);                                             /* :61 */
PQACORE_API void *PqaEngine_Train(void *pvEngine, int64_t nQuestions, const CiAnsweredQuestion *const pAQs,
                                  const int64_t iTarget, const double amount PQA_DEFAULT(1.0));  /* :62-63 */

PQACORE_API uint8_t PqaEngine_QuestionPermFromComp(void *pvEngine, const int64_t count, int64_t *pIds);  /* :65 */
PQACORE_API uint8_t PqaEngine_QuestionCompFromPerm(void *pvEngine, const int64_t count, int64_t *pIds);  /* :66 */
PQACORE_API uint8_t PqaEngine_TargetPermFromComp(void *pvEngine, const int64_t count, int64_t *pIds);    /* :68 */
PQACORE_API uint8_t PqaEngine_TargetCompFromPerm(void *pvEngine, const int64_t count, int64_t *pIds);    /* :69 */
PQACORE_API uint8_t PqaEngine_QuizPermFromComp(void *pvEngine, const int64_t count, int64_t *pIds);      /* :71 */
PQACORE_API uint8_t PqaEngine_QuizCompFromPerm(void *pvEngine, const int64_t count, int64_t *pIds);      /* :72 */
PQACORE_API uint8_t PqaEngine_EnsurePermQuizGreater(void *pvEngine, const int64_t bound);                /* :74 */
PQACORE_API uint8_t PqaEngine_RemapQuizPermId(void *pvEngine, const int64_t srcPermId, const int64_t destPermId); /* :75 */

PQACORE_API uint64_t PqaEngine_GetTotalQuestionsAsked(void *pvEngine, void **ppError);           /* :77 */
PQACORE_API uint8_t PqaEngine_CopyDims(void *pvEngine, CiEngineDimensions *pDims);               /* :78 */
PQACORE_API int64_t PqaEngine_StartQuiz(void *pvEngine, void **ppError);                         /* :79 */
PQACORE_API int64_t PqaEngine_ResumeQuiz(void *pvEngine, void **ppError, const int64_t nAnswered,
                                         const CiAnsweredQuestion *const pAQs);                  /* :80-81 */
PQACORE_API int64_t PqaEngine_NextQuestion(void *pvEngine, void **ppError, const int64_t iQuiz); /* :82 */
PQACORE_API void *PqaEngine_RecordAnswer(void *pvEngine, const int64_t iQuiz, const int64_t iAnswer); /* :83 */

PQACORE_API void *PqaEngine_ClearOldQuizzes(void *pvEngine, const int64_t maxCount, const double maxAgeSec); /* :85 */

PQACORE_API int64_t PqaEngine_GetActiveQuestionId(void *pvEngine, void **ppError, const int64_t iQuiz); /* :87 */
PQACORE_API void *PqaEngine_SetActiveQuestion(void *pvEngine, const int64_t iQuiz, const int64_t iQuestion); /* :88 */

PQACORE_API int64_t PqaEngine_ListTopTargets(void *pvEngine, void **ppError, const int64_t iQuiz,
                                             const int64_t maxCount, CiRatedTarget *pDest);      /* :90-91 */
PQACORE_API void *PqaEngine_RecordQuizTarget(void *pvEngine, const int64_t iQuiz, const int64_t iTarget,
                                             const double amount PQA_DEFAULT(1.0));              /* :92-93 */
PQACORE_API void *PqaEngine_ReleaseQuiz(void *pvEngine, const int64_t iQuiz);                    /* :94 */
PQACORE_API void *PqaEngine_SaveKB(void *pvEngine, const char *const filePath, const uint8_t bDoubleBuffer); /* :95 */

PQACORE_API void *PqaEngine_StartMaintenance(void *pvEngine, const bool forceQuizzes);           /* :98 */
PQACORE_API void *PqaEngine_FinishMaintenance(void *pvEngine);                                   /* :99 */
PQACORE_API void *PqaEngine_AddQsTs(void *pvEngine, const int64_t nQuestions, CiAddQorTParam *pAddQuestionParams,
                                    const int64_t nTargets, CiAddQorTParam *pAddTargetParams);   /* :100-101 */
PQACORE_API void *PqaEngine_RemoveQuestions(void *pvEngine, const int64_t nQuestions, const int64_t *pQIds); /* :102 */
PQACORE_API void *PqaEngine_RemoveTargets(void *pvEngine, const int64_t nTargets, const int64_t *pTIds);     /* :103 */
PQACORE_API void *PqaEngine_Compact(void *pvEngine, int64_t *pnQuestions, int64_t const **const ppOldQuestions,
                                    int64_t *pnTargets, int64_t const **const ppOldTargets);     /* :104-105 */
PQACORE_API void CiReleaseCompaction(const int64_t *p);                                          /* :106 */
PQACORE_API void *PqaEngine_Shutdown(void *pvEngine, const char *const saveFilePath PQA_DEFAULT(0)); /* :107 */
PQACORE_API void *PqaEngine_SetLogger(void *pvEngine, void *pSRLogger);                          /* :108 */

#ifdef __cplusplus
} /* extern "C" */
#endif
#endif
