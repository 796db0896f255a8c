// c_abi.cpp -- the extern "C" surface of libPqaCore.so.
// Shims follow reference ProbQA/PqaCore/PqaCInterop.cpp:45-408 (AssignPqaError / ReturnPqaError and its three null-handle
// conventions: return an error object, set *ppError, or log and return 0); declarations are in include/PqaCInterop.h and
// include/PqaHipExt.h.
#include <dlfcn.h>
#include <sched.h>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "hip_engine.h"

using pqa::AQ;
using pqa::ErrCode;
using pqa::Error;
using pqa::HipEngine;

static_assert(sizeof(CiEngineDefinition) == 48, "POD layout must match reference PqaCInterop.h:10-19");
static_assert(offsetof(CiEngineDefinition, _precType) == 24 && offsetof(CiEngineDefinition, _precExponent) == 26 &&
                  offsetof(CiEngineDefinition, _precMantissa) == 28 && offsetof(CiEngineDefinition, _initAmount) == 32 &&
                  offsetof(CiEngineDefinition, _memPoolMaxBytes) == 40, "POD layout");
static_assert(sizeof(CiAnsweredQuestion) == sizeof(AQ) && sizeof(CiRatedTarget) == 16 && sizeof(CiAddQorTParam) == 16, "POD layout");
static_assert(sizeof(pqa::RatedTargetDev) == sizeof(CiRatedTarget) && offsetof(pqa::RatedTargetDev, prob) == offsetof(CiRatedTarget, _prob), "POD layout");
static_assert(sizeof(CiHipSelection) == sizeof(pqa::SelectResult), "selection record");

namespace {

struct Factory { int unused; };
Factory gFactory;  // process-global singleton, never freed (reference PqaCore/PqaEngineFactorySelector.cpp:11-15)

void AssignErr(void **ppError, Error &err) {  // PqaCInterop.cpp:45-54
  if (!ppError) return;
  *ppError = err.ok() ? nullptr : new Error(std::move(err));
}
void *ReturnErr(Error &&err) {  // PqaCInterop.cpp:56-61
  if (err.ok()) return nullptr;
  return new Error(std::move(err));
}
Error NullEngine() { return Error::Make(ErrCode::NullArgument, "Nullptr is passed in place of IPqaEngine."); }
Error NotImpl(const char *feature) {
  return Error::MakeP(ErrCode::NotImplemented, std::string("Feature=") + feature,
                      std::string(feature) + " is not built in the MI355X engine yet.");
}
char *DupString(const std::string &s) {
  char *p = new char[s.size() + 1];
  std::memcpy(p, s.c_str(), s.size() + 1);
  return p;
}

#define ENGINE_OR_RETURN_ERROR                                   \
  pqa::IEngine *pEng = static_cast<pqa::IEngine *>(pvEngine);         \
  if (pEng == nullptr) return new Error(NullEngine());
#define ENGINE_OR_SET_ERROR(retVal)                        \
  pqa::IEngine *pEng = static_cast<pqa::IEngine *>(pvEngine);         \
  if (pEng == nullptr) {                                        \
    if (ppError) *ppError = new Error(NullEngine());            \
    return retVal;                                              \
  }
#define ENGINE_OR_LOG(retVal)                                               \
  pqa::IEngine *pEng = static_cast<pqa::IEngine *>(pvEngine);                             \
  if (pEng == nullptr) {                                                            \
    std::fprintf(stderr, "PqaCore: Nullptr is passed in place of IPqaEngine.\n");   \
    return retVal;                                                                  \
  }

// PQA_DEVICES=i[,j,...]: empty when unset or malformed (a malformed value is reported and ignored)
std::vector<int> DevicesFromEnvironment() {
  std::vector<int> devices;
  const char *v = std::getenv("PQA_DEVICES");
  if (!v) return devices;
  const char *p = v;
  bool ok = *p != 0;
  while (ok && *p) {
    char *end = nullptr;
    const long d = std::strtol(p, &end, 10);
    if (end == p || d < 0 || d > 1023) { ok = false; break; }
    devices.push_back((int)d);
    p = end;
    if (*p == ',') p++; else if (*p != 0) ok = false;
  }
  if (!ok) {
    if (*v) std::fprintf(stderr, "PqaCore: ignoring PQA_DEVICES=%s (expected a comma-separated list of device ordinals)\n", v);
    devices.clear();
  }
  return devices;
}

void *CreateEngine(void *pvFactory, void **ppError, const CiEngineDefinition *pEngDef, const CiHipShard *pShard) {
  if (pvFactory == nullptr) {  // PqaCInterop.cpp:93-98
    if (ppError) *ppError = new Error(Error::Make(ErrCode::NullArgument, "Nullptr is passed in place of IPqaEngineFactory."));
    return nullptr;
  }
  if (pEngDef == nullptr) {
    if (ppError) *ppError = new Error(Error::Make(ErrCode::NullArgument, "Nullptr is passed in place of the engine definition."));
    return nullptr;
  }
  Error err;
  // PQA_DEVICES=i[,j,...] (unchanged wrappers cannot name a device): one ordinal = that device; several = one shard of
  // the question axis per listed device (an ordinal may repeat: several shards on one device), behind this one engine handle
  std::vector<int> devices;
  if (pShard == nullptr) devices = DevicesFromEnvironment();
  pqa::IEngine *eng = nullptr;
  if (devices.size() >= 2) {
    eng = pqa::CreateShardedEngine(err, *pEngDef, devices);
  } else {
    CiHipShard whole;
    if (devices.size() == 1) {
      whole._qFirst = 0; whole._qTotal = pEngDef->_nQuestions; whole._device = devices[0]; whole._reserved = 0;
      pShard = &whole;
    }
    eng = HipEngine::Create(err, *pEngDef, pShard);
  }
  AssignErr(ppError, err);
  return eng;
}

}  // namespace

extern "C" {

PQACORE_API void CiDebugBreak(void) { /* reference requests a debugger; nothing to do here */ }

// The process-wide default logger (reference SRPlatform/SRDefaultLogger.cpp:47-83): a file logger once Logger_Init has named
// it, the debug stream (here: stderr) until then.  A second initialisation is an error, reported as an owned C string.
PQACORE_API uint8_t Logger_Init(void **ppStrErr, const char *baseName) {
  const std::string err = pqa::DefaultLogger::Init(baseName);
  if (ppStrErr) *ppStrErr = err.empty() ? nullptr : DupString(err);
  return err.empty() ? 1 : 0;
}

PQACORE_API void CiReleaseString(void *pvString) { delete[] static_cast<char *>(pvString); }

PQACORE_API void *CiGetPqaEngineFactory(void) { return &gFactory; }

PQACORE_API void *PqaEngineFactory_CreateCpuEngine(void *pvFactory, void **ppError, const CiEngineDefinition *pEngDef) {
  return CreateEngine(pvFactory, ppError, pEngDef, nullptr);
}
PQACORE_API void *PqaEngineFactory_CreateHipEngine(void *pvFactory, void **ppError, const CiEngineDefinition *pEngDef) {
  return CreateEngine(pvFactory, ppError, pEngDef, nullptr);
}
PQACORE_API void *PqaEngineFactory_CreateHipEngineSharded(void *pvFactory, void **ppError,
                                                          const CiEngineDefinition *pEngDef, const CiHipShard *pShard) {
  return CreateEngine(pvFactory, ppError, pEngDef, pShard);
}

PQACORE_API void *PqaEngineFactory_LoadCpuEngine(void *pvFactory, void **ppError, const char *filePath,
                                                 uint64_t memPoolMaxBytes) {
  (void)memPoolMaxBytes;  // the device engine has no host memory pool to size
  if (pvFactory == nullptr) {
    if (ppError) *ppError = new Error(Error::Make(ErrCode::NullArgument, "Nullptr is passed in place of IPqaEngineFactory."));
    return nullptr;
  }
  Error err;
  const std::vector<int> devices = DevicesFromEnvironment();
  if (devices.size() >= 2) {
    pqa::IEngine *sharded = pqa::LoadShardedEngine(err, filePath, devices);
    AssignErr(ppError, err);
    return sharded;
  }
  if (devices.size() == 1 && hipSetDevice(devices[0]) != hipSuccess) {
    (void)hipGetLastError();
    err = Error::MakeP(ErrCode::IndexOutOfRange, "device=" + std::to_string(devices[0]), "No such HIP device (PQA_DEVICES).");
    AssignErr(ppError, err);
    return nullptr;
  }
  pqa::IEngine *eng = HipEngine::Load(err, filePath);
  AssignErr(ppError, err);
  return eng;
}

PQACORE_API void *PqaEngineFactory_LoadHipEngine(void *pvFactory, void **ppError, const char *filePath, uint64_t memPoolMaxBytes) {
  return PqaEngineFactory_LoadCpuEngine(pvFactory, ppError, filePath, memPoolMaxBytes);
}

PQACORE_API void CiReleasePqaError(void *pvErr) { delete static_cast<Error *>(pvErr); }

PQACORE_API void *PqaError_ToString(void *pvError, const uint8_t withParams) {
  Error *pErr = static_cast<Error *>(pvError);
  if (!pErr) return DupString("[Success] message=[]");
  return DupString(pErr->ToString(withParams != 0));
}

static void ReleaseEngineSideTables(void *pvEngine);   // (what c_abi.cpp keeps per engine handle: the RCCL exchange buffers)
PQACORE_API void CiReleasePqaEngine(void *pvEngine) {
  if (pvEngine) ReleaseEngineSideTables(pvEngine);
  delete static_cast<pqa::IEngine *>(pvEngine);
}

PQACORE_API void *PqaEngine_Train(void *pvEngine, int64_t nQuestions, const CiAnsweredQuestion *const pAQs,
                                  const int64_t iTarget, const double amount) {
  ENGINE_OR_RETURN_ERROR;
  return ReturnErr(pEng->Train(nQuestions, reinterpret_cast<const AQ *>(pAQs), iTarget, amount));
}

PQACORE_API uint8_t PqaEngine_QuestionPermFromComp(void *pvEngine, const int64_t count, int64_t *pIds) {
  ENGINE_OR_LOG(0);
  return pEng->MapIds(0, true, count, pIds) ? 1 : 0;
}
PQACORE_API uint8_t PqaEngine_QuestionCompFromPerm(void *pvEngine, const int64_t count, int64_t *pIds) {
  ENGINE_OR_LOG(0);
  return pEng->MapIds(0, false, count, pIds) ? 1 : 0;
}
PQACORE_API uint8_t PqaEngine_TargetPermFromComp(void *pvEngine, const int64_t count, int64_t *pIds) {
  ENGINE_OR_LOG(0);
  return pEng->MapIds(1, true, count, pIds) ? 1 : 0;
}
PQACORE_API uint8_t PqaEngine_TargetCompFromPerm(void *pvEngine, const int64_t count, int64_t *pIds) {
  ENGINE_OR_LOG(0);
  return pEng->MapIds(1, false, count, pIds) ? 1 : 0;
}
PQACORE_API uint8_t PqaEngine_QuizPermFromComp(void *pvEngine, const int64_t count, int64_t *pIds) {
  ENGINE_OR_LOG(0);
  return pEng->MapIds(2, true, count, pIds) ? 1 : 0;
}
PQACORE_API uint8_t PqaEngine_QuizCompFromPerm(void *pvEngine, const int64_t count, int64_t *pIds) {
  ENGINE_OR_LOG(0);
  return pEng->MapIds(2, false, count, pIds) ? 1 : 0;
}
PQACORE_API uint8_t PqaEngine_EnsurePermQuizGreater(void *pvEngine, const int64_t bound) {
  ENGINE_OR_LOG(0);
  return pEng->EnsurePermQuizGreater(bound) ? 1 : 0;
}
PQACORE_API uint8_t PqaEngine_RemapQuizPermId(void *pvEngine, const int64_t srcPermId, const int64_t destPermId) {
  ENGINE_OR_LOG(0);
  return pEng->RemapQuizPermId(srcPermId, destPermId) ? 1 : 0;
}

PQACORE_API uint64_t PqaEngine_GetTotalQuestionsAsked(void *pvEngine, void **ppError) {
  ENGINE_OR_SET_ERROR(0);
  Error err;
  const uint64_t n = pEng->GetTotalQuestionsAsked(err);
  AssignErr(ppError, err);
  return n;
}

PQACORE_API uint8_t PqaEngine_CopyDims(void *pvEngine, CiEngineDimensions *pDims) {
  ENGINE_OR_LOG(0);
  pEng->CopyDims(pDims);
  return 1;
}

PQACORE_API int64_t PqaEngine_StartQuiz(void *pvEngine, void **ppError) {
  ENGINE_OR_SET_ERROR(-1);
  Error err;
  const int64_t id = pEng->StartQuiz(err);
  AssignErr(ppError, err);
  return id;
}

PQACORE_API int64_t PqaEngine_ResumeQuiz(void *pvEngine, void **ppError, const int64_t nAnswered,
                                         const CiAnsweredQuestion *const pAQs) {
  ENGINE_OR_SET_ERROR(-1);
  Error err;
  const int64_t id = pEng->ResumeQuiz(err, nAnswered, reinterpret_cast<const AQ *>(pAQs));
  AssignErr(ppError, err);
  return id;
}

PQACORE_API int64_t PqaEngine_NextQuestion(void *pvEngine, void **ppError, const int64_t iQuiz) {
  ENGINE_OR_SET_ERROR(-1);
  Error err;
  const int64_t q = pEng->NextQuestion(err, iQuiz);
  AssignErr(ppError, err);
  return q;
}

PQACORE_API void *PqaEngine_RecordAnswer(void *pvEngine, const int64_t iQuiz, const int64_t iAnswer) {
  ENGINE_OR_RETURN_ERROR;
  return ReturnErr(pEng->RecordAnswer(iQuiz, iAnswer));
}

PQACORE_API void *PqaEngine_ClearOldQuizzes(void *pvEngine, const int64_t maxCount, const double maxAgeSec) {
  ENGINE_OR_RETURN_ERROR;
  return ReturnErr(pEng->ClearOldQuizzes(maxCount, maxAgeSec));
}

PQACORE_API int64_t PqaEngine_GetActiveQuestionId(void *pvEngine, void **ppError, const int64_t iQuiz) {
  ENGINE_OR_SET_ERROR(-1);
  Error err;
  const int64_t q = pEng->GetActiveQuestionId(err, iQuiz);
  AssignErr(ppError, err);
  return q;
}

PQACORE_API void *PqaEngine_SetActiveQuestion(void *pvEngine, const int64_t iQuiz, const int64_t iQuestion) {
  ENGINE_OR_RETURN_ERROR;
  return ReturnErr(pEng->SetActiveQuestion(iQuiz, iQuestion));
}

PQACORE_API int64_t PqaEngine_ListTopTargets(void *pvEngine, void **ppError, const int64_t iQuiz,
                                             const int64_t maxCount, CiRatedTarget *pDest) {
  ENGINE_OR_SET_ERROR(-1);
  Error err;
  const int64_t n = pEng->ListTopTargets(err, iQuiz, maxCount, pDest);
  AssignErr(ppError, err);
  return n;
}

PQACORE_API void *PqaEngine_RecordQuizTarget(void *pvEngine, const int64_t iQuiz, const int64_t iTarget,
                                             const double amount) {
  ENGINE_OR_RETURN_ERROR;
  return ReturnErr(pEng->RecordQuizTarget(iQuiz, iTarget, amount));
}

PQACORE_API void *PqaEngine_ReleaseQuiz(void *pvEngine, const int64_t iQuiz) {
  ENGINE_OR_RETURN_ERROR;
  return ReturnErr(pEng->ReleaseQuiz(iQuiz));
}

PQACORE_API void *PqaEngine_SaveKB(void *pvEngine, const char *const filePath, const uint8_t bDoubleBuffer) {
  ENGINE_OR_RETURN_ERROR;
  return ReturnErr(pEng->SaveKB(filePath, bDoubleBuffer != 0));
}

PQACORE_API void *PqaEngine_StartMaintenance(void *pvEngine, const bool forceQuizzes) {
  ENGINE_OR_RETURN_ERROR;
  return ReturnErr(pEng->StartMaintenance(forceQuizzes));
}
PQACORE_API void *PqaEngine_FinishMaintenance(void *pvEngine) {
  ENGINE_OR_RETURN_ERROR;
  return ReturnErr(pEng->FinishMaintenance());
}
PQACORE_API void *PqaEngine_AddQsTs(void *pvEngine, const int64_t nQuestions, CiAddQorTParam *pAddQuestionParams,
                                    const int64_t nTargets, CiAddQorTParam *pAddTargetParams) {
  ENGINE_OR_RETURN_ERROR;
  return ReturnErr(pEng->AddQsTs(nQuestions, pAddQuestionParams, nTargets, pAddTargetParams));
}
PQACORE_API void *PqaEngine_RemoveQuestions(void *pvEngine, const int64_t nQuestions, const int64_t *pQIds) {
  ENGINE_OR_RETURN_ERROR;
  return ReturnErr(pEng->RemoveQuestions(nQuestions, pQIds));
}
PQACORE_API void *PqaEngine_RemoveTargets(void *pvEngine, const int64_t nTargets, const int64_t *pTIds) {
  ENGINE_OR_RETURN_ERROR;
  return ReturnErr(pEng->RemoveTargets(nTargets, pTIds));
}
PQACORE_API void *PqaEngine_Compact(void *pvEngine, int64_t *pnQuestions, int64_t const **const ppOldQuestions,
                                    int64_t *pnTargets, int64_t const **const ppOldTargets) {
  ENGINE_OR_RETURN_ERROR;
  return ReturnErr(pEng->Compact(pnQuestions, ppOldQuestions, pnTargets, ppOldTargets));
}
PQACORE_API void CiReleaseCompaction(const int64_t *p) { std::free(const_cast<int64_t *>(p)); }

PQACORE_API void *PqaEngine_Shutdown(void *pvEngine, const char *const saveFilePath) {
  ENGINE_OR_RETURN_ERROR;
  return ReturnErr(pEng->Shutdown(saveFilePath));
}
// BaseEngine::SetLogger (reference PqaCore/BaseEngine.cpp:252-258): nullptr selects the default logger -- that case is served.
// Any other value is a pointer to an SRPlat::ISRLogger, a C++ object of the MSVC ABI (SRPlatform/Interface/ISRLogger.h:11-25,
// virtual Log(Severity, const SRString&)): none of the C-ABI wrappers can make one (ProbQA.py and the .NET layer only call
// Logger_Init), and this library cannot call through a foreign vtable.
PQACORE_API void *PqaEngine_SetLogger(void *pvEngine, void *pSRLogger) {
  ENGINE_OR_RETURN_ERROR;
  if (pSRLogger == nullptr) return nullptr;
  return ReturnErr(NotImpl("SetLogger with a caller-supplied ISRLogger object (MSVC C++ ABI)"));
}

// ---------------------------------------------------------------------------------------------------- PqaHipExt.h
PQACORE_API void *PqaHip_SetOption(void *pvEngine, const char *name, int64_t value) {
  ENGINE_OR_RETURN_ERROR;
  return ReturnErr(pEng->SetOption(name, value));
}
PQACORE_API int64_t PqaHip_GetOption(void *pvEngine, const char *name) {
  ENGINE_OR_LOG(-1);
  return pEng->GetOption(name);
}
PQACORE_API const char *PqaHip_EvalKernelName(void *pvEngine) {
  ENGINE_OR_LOG("");
  return pEng->EvalKernelName();
}
PQACORE_API void *PqaHip_SetKB(void *pvEngine, const double *pA, const double *pD, const double *pB) {
  ENGINE_OR_RETURN_ERROR;
  return ReturnErr(pEng->SetKB(pA, pD, pB));
}
PQACORE_API void *PqaHip_GetKB(void *pvEngine, double *pA, double *pD, double *pB) {
  ENGINE_OR_RETURN_ERROR;
  return ReturnErr(pEng->GetKB(pA, pD, pB));
}
PQACORE_API void *PqaHip_FillSynthetic(void *pvEngine, double nTrain, double noiseAmp, uint64_t seed) {
  ENGINE_OR_RETURN_ERROR;
  return ReturnErr(pEng->FillSynthetic(nTrain, noiseAmp, seed));
}
PQACORE_API void *PqaHip_SetTargetGaps(void *pvEngine, int64_t n, const int64_t *pTargets) {
  ENGINE_OR_RETURN_ERROR;
  return ReturnErr(pEng->SetTargetGaps(n, pTargets));
}
PQACORE_API void *PqaHip_SetQuestionGaps(void *pvEngine, int64_t n, const int64_t *pQuestions) {
  ENGINE_OR_RETURN_ERROR;
  return ReturnErr(pEng->SetQuestionGaps(n, pQuestions));
}
PQACORE_API void *PqaEngine_EvalPriorities(void *pvEngine, const int64_t iQuiz, double *pOut, const int64_t n) {
  ENGINE_OR_RETURN_ERROR;
  return ReturnErr(pEng->EvalPriorities(iQuiz, pOut, n));
}
PQACORE_API int64_t PqaEngine_NextQuestionArgmax(void *pvEngine, void **ppError, const int64_t iQuiz) {
  ENGINE_OR_SET_ERROR(-1);
  Error err;
  const int64_t q = pEng->NextQuestionArgmax(err, iQuiz);
  AssignErr(ppError, err);
  return q;
}
PQACORE_API int64_t PqaEngine_NextQuestionSampled(void *pvEngine, void **ppError, const int64_t iQuiz,
                                                  const uint64_t rnd) {
  ENGINE_OR_SET_ERROR(-1);
  Error err;
  const int64_t q = pEng->NextQuestionSampled(err, iQuiz, rnd);
  AssignErr(ppError, err);
  return q;
}
PQACORE_API void *PqaHip_GetPriors(void *pvEngine, const int64_t iQuiz, double *pOut, const int64_t n) {
  ENGINE_OR_RETURN_ERROR;
  return ReturnErr(pEng->GetPriors(iQuiz, pOut, n));
}
PQACORE_API void *PqaEngine_NextQuestionArgmaxBatch(void *pvEngine, const int64_t nQuizzes, const int64_t *pQuizzes,
                                                    int64_t *pQuestions) {
  ENGINE_OR_RETURN_ERROR;
  return ReturnErr(pEng->NextQuestionArgmaxBatch(nQuizzes, pQuizzes, pQuestions));
}
PQACORE_API void *PqaEngine_RecordAnswerBatch(void *pvEngine, const int64_t nQuizzes, const int64_t *pQuizzes, const int64_t *pAnswers) {
  ENGINE_OR_RETURN_ERROR;
  return ReturnErr(pEng->RecordAnswerBatch(nQuizzes, pQuizzes, pAnswers));
}
PQACORE_API void *PqaEngine_StartQuizBatch(void *pvEngine, const int64_t nQuizzes, int64_t *pQuizzes) {
  ENGINE_OR_RETURN_ERROR;
  return ReturnErr(pEng->StartQuizBatch(nQuizzes, pQuizzes));
}
PQACORE_API void *PqaEngine_ListTopTargetsBatch(void *pvEngine, const int64_t nQuizzes, const int64_t *pQuizzes, const int64_t maxCount,
                                                CiRatedTarget *pDest, int64_t *pCounts) {
  ENGINE_OR_RETURN_ERROR;
  return ReturnErr(pEng->ListTopTargetsBatch(nQuizzes, pQuizzes, maxCount, pDest, pCounts));
}
PQACORE_API void *PqaHip_SelectArgmaxBatch(void *pvEngine, const int64_t nQuizzes, const int64_t *pQuizzes, CiHipSelection *pOut) {
  ENGINE_OR_RETURN_ERROR;
  return ReturnErr(pEng->SelectArgmaxBatch(nQuizzes, pQuizzes, pOut));
}
PQACORE_API void *PqaEngine_EvalPrioritiesBatch(void *pvEngine, const int64_t nQuizzes, const int64_t *pQuizzes, double *pOut) {
  ENGINE_OR_RETURN_ERROR;
  return ReturnErr(pEng->EvalPrioritiesBatch(nQuizzes, pQuizzes, pOut));
}
PQACORE_API void *PqaHip_Log2Hot(void *pvEngine, const double *pIn, double *pOut, const int64_t n) {
  ENGINE_OR_RETURN_ERROR;
  return ReturnErr(pEng->Log2HotArray(pIn, pOut, n));
}
PQACORE_API void *PqaHip_GetStream(void *pvEngine) {
  ENGINE_OR_LOG(nullptr);
  return pEng->GetStream();
}
PQACORE_API void *PqaHip_SetStream(void *pvEngine, void *hipStream) {
  ENGINE_OR_RETURN_ERROR;
  return ReturnErr(pEng->SetStream(static_cast<hipStream_t>(hipStream)));
}
PQACORE_API void *PqaHip_Synchronize(void *pvEngine) {
  ENGINE_OR_RETURN_ERROR;
  return ReturnErr(pEng->Synchronize());
}
PQACORE_API void *PqaHip_Quiesce(void *pvEngine) {
  ENGINE_OR_RETURN_ERROR;
  return ReturnErr(pEng->Quiesce());
}
PQACORE_API void *PqaHip_EnqueueSelectArgmaxFlag(void *pvEngine, const int64_t iQuiz, void *pOut, void *pFlag,
                                                 const uint64_t flagValue) {
  ENGINE_OR_RETURN_ERROR;
  return ReturnErr(pEng->EnqueueSelectArgmaxFlag(iQuiz, pOut, pFlag, flagValue));
}
// Host memory (e.g. a shared-memory segment mapped by every rank) made writable by this process's GPU.
PQACORE_API void *PqaHip_HostRegister(void *pHost, const int64_t nBytes, void **ppDevice) {
  if (!pHost || !ppDevice || nBytes <= 0) return ReturnErr(Error::Make(ErrCode::NullArgument, "Bad arguments to PqaHip_HostRegister."));
  hipError_t he = hipHostRegister(pHost, (size_t)nBytes, hipHostRegisterMapped | hipHostRegisterPortable);
  if (he == hipSuccess) he = hipHostGetDevicePointer(ppDevice, pHost, 0);
  if (he != hipSuccess) return ReturnErr(Error::MakeP(ErrCode::StdException, std::string("hip=") + hipGetErrorString(he), "hipHostRegister failed."));
  return nullptr;
}
PQACORE_API void *PqaHip_HostUnregister(void *pHost) {
  if (pHost) hipHostUnregister(pHost);
  return nullptr;
}
// Host-side half of the shared-memory exchange: spin until the flags of all `world` slots equal flagValue, then pick the
// winner (maximum priority, lowest index on ties, NaN never wins, -1 if no slot has an eligible question).  A slot is
// strideBytes long and starts with {double priority; int64 index; uint64 flag}.  Returns an error after timeoutSec.
PQACORE_API void *PqaHip_PickWhenAll(const void *pSlots, const int64_t world, const int64_t strideBytes,
                                     const uint64_t flagValue, const double timeoutSec, double *pPriority, int64_t *pIndex) {
  if (!pSlots || !pPriority || !pIndex || world <= 0 || strideBytes < 24)
    return ReturnErr(Error::Make(ErrCode::NullArgument, "Bad arguments to PqaHip_PickWhenAll."));
  const char *base = (const char *)pSlots;
  const auto t0 = std::chrono::steady_clock::now();
  for (int64_t r = 0; r < world; r++) {
    const volatile uint64_t *flag = (const volatile uint64_t *)(base + r * strideBytes + 16);
    uint64_t spins = 0;
    bool yielding = false;
    while (*flag != flagValue) {   // (a pure spin for the first 200 us -- the answer is a kernel's time away --, then the core is offered between looks)
      ++spins;
      if (!yielding) {
        __builtin_ia32_pause();
        if ((spins & 63) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(200)) yielding = true;
        continue;
      }
      if ((spins & 0xFF) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeoutSec)
        return ReturnErr(Error::MakeP(ErrCode::StdException, "rank=" + std::to_string(r), "Timed out waiting for a shard's selection."));
      sched_yield();
    }
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  double bestP = 0;
  int64_t bestI = -1;
  for (int64_t r = 0; r < world; r++) {
    double p;
    int64_t i;
    std::memcpy(&p, base + r * strideBytes, 8);
    std::memcpy(&i, base + r * strideBytes + 8, 8);
    if (i < 0) continue;
    if (p != p) p = -HUGE_VAL;
    if (bestI < 0 || p > bestP || (p == bestP && i < bestI)) { bestP = p; bestI = i; }
  }
  *pPriority = bestP;
  *pIndex = bestI;
  return nullptr;
}
// One step of the shared-memory exchange in one call: enqueue this shard's selection with its record and flag in slot `rank`
// of the host's slot array (pSlotsDev: the device-visible address of the same array), then wait for every rank and pick.
PQACORE_API void *PqaHip_SelectThroughSlots(void *pvEngine, const int64_t iQuiz, const void *pSlots, void *pSlotsDev,
                                            const int64_t rank, const int64_t world, const int64_t strideBytes,
                                            const uint64_t flagValue, const double timeoutSec, double *pPriority,
                                            int64_t *pIndex) {
  ENGINE_OR_RETURN_ERROR;
  if (!pSlots || !pSlotsDev || rank < 0 || rank >= world || strideBytes < 24)
    return ReturnErr(Error::Make(ErrCode::NullArgument, "Bad arguments to PqaHip_SelectThroughSlots."));
  char *mine = (char *)pSlotsDev + rank * strideBytes;
  Error e = pEng->EnqueueSelectArgmaxFlag(iQuiz, mine, mine + 16, flagValue);
  if (!e.ok()) return ReturnErr(std::move(e));
  return PqaHip_PickWhenAll(pSlots, world, strideBytes, flagValue, timeoutSec, pPriority, pIndex);
}
// The shards' 16-byte winners gathered by ONE RCCL collective on the engine's stream, for a process-per-GPU host that owns an RCCL
// communicator and is not Python (probqa_amd/dist.py does the same through torch.distributed; north_star: "a single RCCL
// all-reduce" -- an all-gather of {priority, GLOBAL index} and the exact pick on every rank, so that ties break by the lowest
// index as in the reference's argmax).  RCCL is looked up at the first call (dlopen: libPqaCore.so itself does not link it).
namespace {
// The exchange buffers of one engine: 16 (world + 1) bytes on the ENGINE's device and their pinned mirror.  They live as long as the
// engine (ReleaseRcclBufs at CiReleasePqaEngine: an engine at a recycled address never meets an earlier engine's buffers), and `mu`
// is held over enqueue, all-gather, copy and pick, so that concurrent calls on one engine do not share them mid-flight.
struct RcclBufs {
  std::mutex mu;
  void *dSend = nullptr, *dRecv = nullptr, *hRecv = nullptr;
  int64_t world = 0;
  int device = -1;
  void Free() {
    if (dSend) { hipSetDevice(device); hipFree(dSend); }
    if (hRecv) hipHostFree(hRecv);
    dSend = dRecv = hRecv = nullptr;
    world = 0;
  }
};
std::mutex gRcclMu;
std::unordered_map<void *, std::shared_ptr<RcclBufs>> gRcclBufs;
void ReleaseRcclBufs(void *pvEngine) {
  std::shared_ptr<RcclBufs> b;
  {
    std::lock_guard<std::mutex> lk(gRcclMu);
    auto it = gRcclBufs.find(pvEngine);
    if (it == gRcclBufs.end()) return;
    b = std::move(it->second);
    gRcclBufs.erase(it);
  }
  std::lock_guard<std::mutex> lk(b->mu);   // (a call still inside finishes first)
  b->Free();
}
typedef int (*NcclAllGatherFn)(const void *, void *, size_t, int, void *, hipStream_t);
NcclAllGatherFn RcclAllGather() {
  static NcclAllGatherFn fn = [] {
    void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    return h ? reinterpret_cast<NcclAllGatherFn>(dlsym(h, "ncclAllGather")) : nullptr;
  }();
  return fn;
}
}  // namespace
PQACORE_API void *PqaHip_SelectArgmaxRccl(void *pvEngine, const int64_t iQuiz, void *pNcclComm, const int64_t world, double *pPriority,
                                          int64_t *pIndex) {
  ENGINE_OR_RETURN_ERROR;
  if (!pNcclComm || !pPriority || !pIndex || world < 1 || world > 4096)
    return ReturnErr(Error::Make(ErrCode::NullArgument, "Bad arguments to PqaHip_SelectArgmaxRccl."));
  const NcclAllGatherFn allGather = RcclAllGather();
  if (allGather == nullptr) return ReturnErr(Error::Make(ErrCode::StdException, "librccl.so (ncclAllGather) could not be loaded."));
  std::shared_ptr<RcclBufs> bufs;
  {
    std::lock_guard<std::mutex> lk(gRcclMu);
    std::shared_ptr<RcclBufs> &slot = gRcclBufs[pvEngine];
    if (!slot) slot = std::make_shared<RcclBufs>();
    bufs = slot;
  }
  RcclBufs &b = *bufs;
  std::lock_guard<std::mutex> held(b.mu);
  const int device = (int)pEng->GetOption("device");
  if (device < 0 || pEng->GetOption("shards") > 0) return ReturnErr(Error::Make(ErrCode::StdException, "PqaHip_SelectArgmaxRccl is for an engine on ONE device (a shard of a process-per-GPU host)."));
  if (hipSetDevice(device) != hipSuccess) return ReturnErr(Error::Make(ErrCode::StdException, "PqaHip_SelectArgmaxRccl: the engine's device cannot be selected."));
  if (b.world != world || b.device != device) {   // (first call, or another communicator size: allocated on the engine's device, whatever the calling thread's was)
    b.Free();
    b.device = device;
    void *d = nullptr, *h = nullptr;
    if (hipMalloc(&d, (size_t)(world + 1) * 16) != hipSuccess || hipHostMalloc(&h, (size_t)world * 16, hipHostMallocDefault) != hipSuccess) {
      if (d) hipFree(d);
      (void)hipGetLastError();
      return ReturnErr(Error::Make(ErrCode::StdException, "PqaHip_SelectArgmaxRccl: no memory for the exchange buffers."));
    }
    b.dSend = d; b.dRecv = static_cast<char *>(d) + 16; b.hRecv = h; b.world = world;
  }
  Error e = pEng->EnqueueSelectArgmax(iQuiz, b.dSend);   // {priority, GLOBAL index} of this shard's winner, in stream order
  if (!e.ok()) return ReturnErr(std::move(e));
  const hipStream_t stream = pEng->GetStream();
  const int rc = allGather(b.dSend, b.dRecv, 16, /* ncclUint8 */ 1, pNcclComm, stream);
  if (rc != 0) return ReturnErr(Error::MakeP(ErrCode::StdException, "ncclResult=" + std::to_string(rc), "ncclAllGather failed."));
  hipError_t he = hipMemcpyAsync(b.hRecv, b.dRecv, (size_t)world * 16, hipMemcpyDeviceToHost, stream);
  if (he == hipSuccess) he = hipStreamSynchronize(stream);
  if (he != hipSuccess) return ReturnErr(Error::MakeP(ErrCode::StdException, hipGetErrorString(he), "PqaHip_SelectArgmaxRccl: the gathered winners did not arrive."));
  double bestP = 0;
  int64_t bestI = -1;
  for (int64_t r = 0; r < world; r++) {   // (as PqaHip_PickWhenAll: max priority, lowest index on ties, NaN never wins, -1 if none)
    double p;
    int64_t i;
    std::memcpy(&p, static_cast<const char *>(b.hRecv) + r * 16, 8);
    std::memcpy(&i, static_cast<const char *>(b.hRecv) + r * 16 + 8, 8);
    if (i < 0) continue;
    if (p != p) p = -HUGE_VAL;
    if (bestI < 0 || p > bestP || (p == bestP && i < bestI)) { bestP = p; bestI = i; }
  }
  *pPriority = bestP;
  *pIndex = bestI;
  return nullptr;
}
static void ReleaseEngineSideTables(void *pvEngine) { ReleaseRcclBufs(pvEngine); }
PQACORE_API void *PqaHip_EnqueueSelectArgmax(void *pvEngine, const int64_t iQuiz, void *pOut) {
  ENGINE_OR_RETURN_ERROR;
  return ReturnErr(pEng->EnqueueSelectArgmax(iQuiz, pOut));
}
PQACORE_API void *PqaHip_EnqueueEval(void *pvEngine, const int64_t iQuiz) {
  ENGINE_OR_RETURN_ERROR;
  return ReturnErr(pEng->EnqueueEval(iQuiz));
}
PQACORE_API void *PqaHip_GetPriorDevicePtr(void *pvEngine, const int64_t iQuiz, void **ppDev, int64_t *pLdT) {
  ENGINE_OR_RETURN_ERROR;
  return ReturnErr(pEng->GetPriorDevicePtr(iQuiz, ppDev, pLdT));
}
PQACORE_API void *PqaHip_RecordAnswerRemote(void *pvEngine, const int64_t iQuiz, const int64_t iAnswer) {
  ENGINE_OR_RETURN_ERROR;
  return ReturnErr(pEng->RecordAnswerRemote(iQuiz, iAnswer));
}

PQACORE_API int64_t PqaHip_HostLogicProbe(const char *what, const int64_t *pIn, const int64_t nIn, int64_t *pOut, const int64_t nOut) {
  const std::string w(what ? what : "");
  if (nIn < 0 || nOut < 0 || (nIn > 0 && !pIn) || (nOut > 0 && !pOut)) return -1;
  if (w == "id_ledger") {
    pqa::IdLedger ledger;
    int64_t i = 0, nRes = 0;
    while (i < nIn) {
      if (i + 3 > nIn || nRes >= nOut) return -1;
      const int64_t op = pIn[i], a = pIn[i + 1], b = pIn[i + 2];
      i += 3;
      int64_t r;
      switch (op) {
        case 0: r = ledger.PermanentOf(a); break;
        case 1: r = ledger.SlotOf(a); break;
        case 2: r = ledger.RaiseFloor(a); break;
        case 3: r = ledger.Vacate(a); break;
        case 4: r = ledger.Reissue(a); break;
        case 5: r = ledger.Extend(a); break;
        case 6: r = ledger.Rename(a, b); break;
        case 7:
          if (a < 0 || b != a || i + a > nIn) return -1;
          r = ledger.Repack(a, pIn + i);
          i += a;
          break;
        case 8: {
          FILE *f = std::tmpfile();
          if (!f) return -1;
          pqa::IdLedger back;
          r = ledger.Write(f) && std::fseek(f, 0, SEEK_SET) == 0 && back.Read(f);
          std::fclose(f);
          if (r) ledger = back;
          break;
        }
        case 9: r = ledger.LiveSlots(); break;
        default: return -1;
      }
      pOut[nRes++] = r;
    }
    return nRes;
  }
  if (w == "let_go") {
    if (nIn < 4 || pIn[3] < 0 || nIn != 4 + 2 * pIn[3]) return -1;
    std::vector<pqa::QuizUsage> inUse;
    for (int64_t k = 0; k < pIn[3]; k++) inUse.push_back(pqa::QuizUsage{pIn[4 + 2 * k], (time_t)pIn[5 + 2 * k]});
    const std::vector<int64_t> ids = pqa::QuizzesToLetGo(inUse, (time_t)pIn[0], pIn[1], (double)pIn[2]);
    if ((int64_t)ids.size() + 1 > nOut) return -1;
    pOut[0] = (int64_t)ids.size();
    for (size_t k = 0; k < ids.size(); k++) pOut[1 + k] = ids[k];
    return (int64_t)ids.size() + 1;
  }
  return -1;
}

}  // extern "C"
