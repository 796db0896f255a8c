// hip_engine.h -- host side of the MI355X engine behind the PqaCore C ABI.
//
// Mirrors the reference's engine interface for the hot path (reference: ProbQA/PqaCore/Interface/IPqaEngine.h:14-112;
// behaviour of PqaCore/BaseEngine.cpp and PqaCore/CpuEngine.cpp for StartQuiz / ResumeQuiz / NextQuestion /
// RecordAnswer / active-question bookkeeping / quiz registry / error objects).  All arithmetic runs in the gfx950
// kernels of pqa_kernels.h; this file only owns device memory, the quiz table and the reference's error behaviour.
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <cstdint>
#include <memory>
#include <mutex>
#include <cstdio>
#include <ctime>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/PqaHipExt.h"
#include "pqa_kernels.h"

namespace pqa {

// Error codes of reference PqaCore/Interface/PqaErrors.h:12-40
enum class ErrCode : int64_t {
  None = 0, NotImplemented = 1, SRException = 2, StdException = 3, InsufficientEngineDimensions = 4,
  MaintenanceModeChangeInProgress = 5, MaintenanceModeAlreadyThis = 6, ObjectShutDown = 7, IndexOutOfRange = 8,
  Internal = 9, Aggregate = 10, NegativeCount = 11, NonPositiveAmount = 12, AbsentId = 13, WrongMode = 14,
  UnhandledCase = 15, I64Underflow = 16, QuestionsExhausted = 17, NoQuizActiveQuestion = 18, CantOpenFile = 19,
  FileOp = 20, QuizzesActive = 21, NullArgument = 22, WrongRuntimeType = 23, NotInitialized = 24
};

// PqaError (reference PqaCore/Interface/PqaErrors.h:56-91): code + message + stringified params
struct Error {
  ErrCode code = ErrCode::None;
  std::string message;
  std::string params;   // what IPqaErrorParams::ToString() would give; empty = nullptr params
  bool hasParams = false;
  bool ok() const { return code == ErrCode::None; }
  std::string ToString(bool withParams) const;  // reference PqaCore/PqaErrors.cpp:128-143
  static Error Make(ErrCode c, std::string msg) { Error e; e.code = c; e.message = std::move(msg); return e; }
  static Error MakeP(ErrCode c, std::string params, std::string msg) {
    Error e; e.code = c; e.message = std::move(msg); e.params = std::move(params); e.hasParams = true; return e;
  }
};
const char *ErrCodeName(ErrCode c);  // reference PqaCore/PqaErrors.cpp:13-62

struct AQ { int64_t iQuestion, iAnswer; };

// SRDefaultLogger (reference SRPlatform/SRDefaultLogger.cpp): stderr until Init names a file, that file afterwards.
struct DefaultLogger {
  enum class Severity : uint8_t { None = 0, Info, Warning, Error, Critical };   // ISRLogger::Severity
  static std::string Init(const char *baseName);   // "" on success, the error text otherwise
  static void Log(Severity sev, const std::string &message);
};

// The two-way map between the dense ("compact") ids the arrays are indexed by and the ids a caller may keep across edits of the
// knowledge base ("permanent": never reissued).  Behaviour of the reference's PermanentIdManager (PqaCore/PermanentIdManager.h)
// as its callers and the .kb format see it; the structure is this file's own: the forward table is the array the file stores,
// the reverse direction is a vector ordered by permanent id -- ids are issued in increasing order, so it grows at its end -- whose
// entries are only trusted when the forward table agrees (a vacated slot leaves its entry behind; it is swept out when half of
// the vector is such leftovers).
class IdLedger {
 public:
  static constexpr int64_t kNone = -1;                  // cInvalidPqaId
  int64_t PermanentOf(int64_t slot) const { return InRange(slot) ? _permOf[(size_t)slot] : kNone; }
  int64_t SlotOf(int64_t permanent) const;
  bool Write(FILE *f, bool withoutSlots = false) const; // {next id to issue, slot count, the forward table}: the file's layout
  bool Read(FILE *f);
  bool RaiseFloor(int64_t bound);                       // ids issued from now on exceed `bound`; false if they did already
  bool Vacate(int64_t slot);                            // the slot's permanent id is retired
  bool Reissue(int64_t slot);                           // a vacated slot gets the next permanent id
  bool Extend(int64_t nSlots);                          // new slots at the end, each under the next permanent id
  bool Repack(int64_t nSlots, const int64_t *from);     // slot i takes over what slot from[i] held; nothing else survives
  bool Rename(int64_t permanent, int64_t toPermanent);  // a live id is re-labelled with an unused id from the past
  int64_t LiveSlots() const { return _live; }
  void Clear() { _permOf.clear(); _byPerm.clear(); _live = 0; }

 private:
  struct Back { int64_t permanent, slot; };
  bool InRange(int64_t slot) const { return slot >= 0 && slot < (int64_t)_permOf.size(); }
  bool Live(const Back &b) const { return _permOf[(size_t)b.slot] == b.permanent; }
  size_t LowerBound(int64_t permanent) const;
  void Enter(int64_t permanent, int64_t slot);
  void Rebuild();
  int64_t _issueNext = 0;
  int64_t _live = 0;                 // slots that hold a permanent id
  std::vector<int64_t> _permOf;      // slot -> permanent id, kNone for a vacated slot
  std::vector<Back> _byPerm;         // ascending by permanent id; at most one entry per id
};

// Which quizzes a ClearOldQuizzes(maxCount, maxAgeSec) call lets go (behaviour of BaseEngine::ClearOldQuizzes): every quiz
// unused for longer than maxAgeSec, then the longest-unused of the rest until maxCount remain.  `quizzes` in registry order;
// the result in the order they are to be released (it decides which registry slots the next quizzes reuse first): the aged-out
// ones in registry order, then by age, the longest-unused first, equal ages by registry order.
struct QuizUsage { int64_t id; time_t lastUsage; };
std::vector<int64_t> QuizzesToLetGo(const std::vector<QuizUsage> &quizzes, time_t now, int64_t maxCount, double maxAgeSec);

// A quiz's own lines of host-coherent pinned memory: what the kernels working for ONE quiz hand to the host without a copy --
// the posterior's best targets (listed by RecordAnswer's kernel ahead of the ListTopTargets that follows it) and their flag.
// Per quiz, so that the quizzes of different client threads never wait for each other's results.
constexpr int kQuizTop = 32;
struct QuizPinned {
  RatedTargetDev top[kQuizTop];
  int64_t nOut;
  uint64_t topFlag;
  uint64_t pad[6];
};

struct Quiz {
  time_t lastUsage = 0;                // reference BaseQuiz::OnUsage
  double *dPrior = nullptr;            // ldT doubles, device
  uint32_t *dAsked = nullptr;          // device bitmap over LOCAL questions
  std::vector<uint32_t> hAsked;        // host mirror
  std::vector<AQ> answers;             // global question ids
  int64_t activeQuestion = -1;         // global id (reference CEQuiz::_activeQuestion)
  uint64_t priorVersion = 0;           // bumped whenever the posterior changes
  uint64_t serial = 0;                 // unique per created quiz: a registry slot reused by a later quiz is not this quiz
  int lateStreak = 0;                  // selections in a row whose (lazily fixed) sweep had listed rows at the pole: from the third on, RecordAnswer's
                                       // speculative sweep gets its fix-up launched right behind it again (it will be needed, and runs while the client is elsewhere)
  bool noServer = false;               // the resident sweep has answered a step of this quiz with -4 (a row at the pole of the lack term:
                                       // pole_kernels.hip) -- from here on its selections are launched, with the fix behind them
  QuizPinned *pin = nullptr;           // this quiz's host-coherent result lines (pooled by the engine)
  void *dRowStage = nullptr;           // sharded engine without peer access: the two rows of an answered question another shard holds, copied here
  // the listing in pin->top: made by the kernel that published `topOp` to pin->topFlag, of the posterior `topVersion`
  uint64_t topOp = 0, topVersion = 0;
  int64_t topCount = 0;
  std::atomic<bool> updatePending{false};   // a RecordAnswer of this quiz waits in the engine's list of deferred updates
  // A client of a combined sweep is selecting for this quiz outside the engine's lock (SelectFromPriorities): the quiz object
  // stays until it is done (a ReleaseQuiz from another thread -- a client's error, IPqaEngine.h:44 -- waits).
  std::atomic<bool> inSelection{false};
};

// What the C ABI (c_abi.cpp) drives: one engine on one device (HipEngine), or the question axis of one knowledge base split
// over several devices of the process (ShardedEngine, sharded_engine.cpp) -- the reference's IPqaEngine surface
// (PqaCore/Interface/IPqaEngine.h:14-113) plus the additive calls of include/PqaHipExt.h.
class IEngine {
 public:
  virtual ~IEngine() {}
  virtual Error Train(int64_t nQuestions, const AQ *pAQs, int64_t iTarget, double amount) = 0;
  virtual uint64_t GetTotalQuestionsAsked(Error &err) = 0;
  virtual void CopyDims(CiEngineDimensions *pDims) const = 0;
  virtual int64_t StartQuiz(Error &err) = 0;
  virtual int64_t ResumeQuiz(Error &err, int64_t nAnswered, const AQ *pAQs) = 0;
  virtual int64_t NextQuestion(Error &err, int64_t iQuiz) = 0;
  virtual Error RecordAnswer(int64_t iQuiz, int64_t iAnswer) = 0;
  virtual int64_t GetActiveQuestionId(Error &err, int64_t iQuiz) = 0;
  virtual Error SetActiveQuestion(int64_t iQuiz, int64_t iQuestion) = 0;
  virtual int64_t ListTopTargets(Error &err, int64_t iQuiz, int64_t maxCount, CiRatedTarget *pDest) = 0;
  virtual Error RecordQuizTarget(int64_t iQuiz, int64_t iTarget, double amount) = 0;
  virtual Error ReleaseQuiz(int64_t iQuiz) = 0;
  virtual Error StartMaintenance(bool forceQuizzes) = 0;
  virtual Error FinishMaintenance() = 0;
  virtual Error Shutdown(const char *saveFilePath) = 0;
  virtual bool MapIds(int which, bool toPerm, int64_t count, int64_t *pIds) = 0;
  virtual bool EnsurePermQuizGreater(int64_t bound) = 0;
  virtual bool RemapQuizPermId(int64_t srcPermId, int64_t destPermId) = 0;
  virtual Error SaveKB(const char *filePath, bool doubleBuffer) = 0;
  virtual Error AddQsTs(int64_t nQuestions, CiAddQorTParam *pAqps, int64_t nTargets, CiAddQorTParam *pAtps) = 0;
  virtual Error RemoveQuestions(int64_t n, const int64_t *pQIds) = 0;
  virtual Error RemoveTargets(int64_t n, const int64_t *pTIds) = 0;
  virtual Error Compact(int64_t *pnQuestions, const int64_t **ppOldQuestions, int64_t *pnTargets, const int64_t **ppOldTargets) = 0;
  virtual Error ClearOldQuizzes(int64_t maxCount, double maxAgeSec) = 0;
  // ---- additive (PqaHipExt.h)
  virtual Error SetOption(const char *name, int64_t value) = 0;
  virtual int64_t GetOption(const char *name) const = 0;
  virtual const char *EvalKernelName() const = 0;
  virtual Error SetKB(const double *pA, const double *pD, const double *pB) = 0;
  virtual Error GetKB(double *pA, double *pD, double *pB) = 0;
  virtual Error FillSynthetic(double nTrain, double noiseAmp, uint64_t seed) = 0;
  virtual Error SetTargetGaps(int64_t n, const int64_t *ids) = 0;
  virtual Error SetQuestionGaps(int64_t n, const int64_t *ids) = 0;
  virtual Error EvalPriorities(int64_t iQuiz, double *pOut, int64_t n) = 0;
  virtual int64_t NextQuestionArgmax(Error &err, int64_t iQuiz) = 0;
  virtual int64_t NextQuestionSampled(Error &err, int64_t iQuiz, uint64_t rnd) = 0;
  virtual Error GetPriors(int64_t iQuiz, double *pOut, int64_t n) = 0;
  virtual Error NextQuestionArgmaxBatch(int64_t n, const int64_t *pQuizzes, int64_t *pOut) = 0;
  virtual Error EvalPrioritiesBatch(int64_t n, const int64_t *pQuizzes, double *pOut) = 0;
  virtual Error SelectArgmaxBatch(int64_t n, const int64_t *pQuizzes, CiHipSelection *pOut) = 0;
  virtual Error Log2HotArray(const double *pIn, double *pOut, int64_t n) = 0;
  virtual hipStream_t GetStream() const = 0;
  virtual Error SetStream(hipStream_t s) = 0;
  virtual Error Synchronize() = 0;
  virtual Error Quiesce() = 0;
  virtual Error EnqueueSelectArgmax(int64_t iQuiz, void *pOut) = 0;
  virtual Error EnqueueSelectArgmaxFlag(int64_t iQuiz, void *pOut, void *pFlag, uint64_t flagValue) = 0;
  virtual Error EnqueueEval(int64_t iQuiz) = 0;
  virtual Error GetPriorDevicePtr(int64_t iQuiz, void **ppDev, int64_t *pLdT) = 0;
  virtual Error RecordAnswerRemote(int64_t iQuiz, int64_t iAnswer) = 0;
  virtual Error RecordAnswerBatch(int64_t n, const int64_t *pQuizzes, const int64_t *pAnswers) = 0;
  virtual Error StartQuizBatch(int64_t n, int64_t *pQuizzes) = 0;
  virtual Error ListTopTargetsBatch(int64_t n, const int64_t *pQuizzes, int64_t maxCount, CiRatedTarget *pDest, int64_t *pCounts) = 0;
};

class HipEngine : public IEngine {
 public:
  static HipEngine *Create(Error &err, const CiEngineDefinition &def, const CiHipShard *shard);
  ~HipEngine() override;

  // ---- reference IPqaEngine surface (names as in IPqaEngine.h)
  Error Train(int64_t nQuestions, const AQ *pAQs, int64_t iTarget, double amount) override;
  uint64_t GetTotalQuestionsAsked(Error &err) override;
  void CopyDims(CiEngineDimensions *pDims) const override;
  int64_t StartQuiz(Error &err) override;
  int64_t ResumeQuiz(Error &err, int64_t nAnswered, const AQ *pAQs) override;
  int64_t NextQuestion(Error &err, int64_t iQuiz) override;
  Error RecordAnswer(int64_t iQuiz, int64_t iAnswer) override;
  int64_t GetActiveQuestionId(Error &err, int64_t iQuiz) override;
  Error SetActiveQuestion(int64_t iQuiz, int64_t iQuestion) override;
  int64_t ListTopTargets(Error &err, int64_t iQuiz, int64_t maxCount, CiRatedTarget *pDest) override;
  Error RecordQuizTarget(int64_t iQuiz, int64_t iTarget, double amount) override;
  Error ReleaseQuiz(int64_t iQuiz) override;
  Error StartMaintenance(bool forceQuizzes) override;
  Error FinishMaintenance() override;
  Error Shutdown(const char *saveFilePath) override;
  // which: 0 questions, 1 targets, 2 quizzes; toPerm: compact -> permanent (reference BaseEngine.cpp:150-215)
  bool MapIds(int which, bool toPerm, int64_t count, int64_t *pIds) override;
  bool EnsurePermQuizGreater(int64_t bound) override;
  bool RemapQuizPermId(int64_t srcPermId, int64_t destPermId) override;
  Error SaveKB(const char *filePath, bool doubleBuffer) override;
  static HipEngine *Load(Error &err, const char *filePath);
  Error AddQsTs(int64_t nQuestions, CiAddQorTParam *pAqps, int64_t nTargets, CiAddQorTParam *pAtps) override;
  Error RemoveQuestions(int64_t n, const int64_t *pQIds) override;
  Error RemoveTargets(int64_t n, const int64_t *pTIds) override;
  Error Compact(int64_t *pnQuestions, const int64_t **ppOldQuestions, int64_t *pnTargets, const int64_t **ppOldTargets) override;
  Error ClearOldQuizzes(int64_t maxCount, double maxAgeSec) override;

  // ---- additive (PqaHipExt.h)
  Error SetOption(const char *name, int64_t value) override;
  int64_t GetOption(const char *name) const override;
  const char *EvalKernelName() const override;
  Error SetKB(const double *pA, const double *pD, const double *pB) override;
  Error GetKB(double *pA, double *pD, double *pB) override;
  Error FillSynthetic(double nTrain, double noiseAmp, uint64_t seed) override;
  Error SetTargetGaps(int64_t n, const int64_t *ids) override;
  Error SetQuestionGaps(int64_t n, const int64_t *ids) override;
  Error EvalPriorities(int64_t iQuiz, double *pOut, int64_t n) override;
  int64_t NextQuestionArgmax(Error &err, int64_t iQuiz) override;
  int64_t NextQuestionSampled(Error &err, int64_t iQuiz, uint64_t rnd) override;
  Error GetPriors(int64_t iQuiz, double *pOut, int64_t n) override;
  int64_t NextQuestionArgmaxGraph(Error &err, Quiz *q);
  Error NextQuestionArgmaxBatch(int64_t n, const int64_t *pQuizzes, int64_t *pOut) override;
  Error EvalPrioritiesBatch(int64_t n, const int64_t *pQuizzes, double *pOut) override;
  Error SelectArgmaxBatch(int64_t n, const int64_t *pQuizzes, CiHipSelection *pOut) override;   // pOut[i][q], q < local question count
  Error Log2HotArray(const double *pIn, double *pOut, int64_t n) override;  // device log2hot over an array (tests)
  hipStream_t GetStream() const override { return _stream; }
  Error SetStream(hipStream_t s) override;
  Error Synchronize() override;
  Error Quiesce() override;
  Error EnqueueSelectArgmax(int64_t iQuiz, void *pOut) override;
  Error EnqueueSelectArgmaxFlag(int64_t iQuiz, void *pOut, void *pFlag, uint64_t flagValue) override;
  Error EnqueueEval(int64_t iQuiz) override;
  Error GetPriorDevicePtr(int64_t iQuiz, void **ppDev, int64_t *pLdT) override;
  Error RecordAnswerRemote(int64_t iQuiz, int64_t iAnswer) override;
  Error RecordAnswerBatch(int64_t n, const int64_t *pQuizzes, const int64_t *pAnswers) override;
  Error StartQuizBatch(int64_t n, int64_t *pQuizzes) override;
  Error ListTopTargetsBatch(int64_t n, const int64_t *pQuizzes, int64_t maxCount, CiRatedTarget *pDest, int64_t *pCounts) override;

  // ---- what a sharded engine needs from its shards (sharded_engine.cpp; implemented in hip_engine_shard.cpp)
  // An answer of a quiz as the sharded engine hands it to EVERY shard: the answered question in global numbering; for the shards
  // that do not hold it, its two rows where they are (the owner's cube: read in place over peer access, or -- `stage` -- copied
  // into the quiz's staging rows first).  `list`: this shard's kernel also lists the new posterior's best targets.
  struct ShardAnswer { int64_t iQuiz, qGlobal, iAnswer; const void *rowA, *rowD; int srcDevice; bool stage, list; };
  // Bookkeeping of RecordAnswer for each (CEQuiz::RecordAnswer, PqaCore/CEQuiz.h:77-122) and ONE launch for all the posteriors.
  Error ApplyAnswers(int64_t n, const ShardAnswer *answers);
  // A combined sweep for n distinct quizzes in two halves (the combining of concurrent NextQuestion calls, done by the sharded
  // engine over all its shards): EnqueueCombined validates, flushes, launches into batch context `ctx` (0 / 1) and snapshots the
  // quizzes' unavailable questions; CollectCombined waits for that launch -- without the engine's lock -- and gives per quiz the
  // shard's winner (argmax batches) or a view of its priority vector on the host (hostPriorities).
  struct CombinedFlight { uint64_t tag = 0; bool hostPriorities = false, quizMinor = false, tagged = false; int64_t Bp = 0, nQ = 0, n = 0; hipError_t he = hipSuccess; };
  struct PriorityView { const double *pri = nullptr; int64_t stride = 0; uint64_t tag = 0; };   // local question k: pri[k * stride]; tag != 0: the word behind it must carry the tag first
  Error EnqueueCombined(int ctx, int64_t n, const int64_t *pQuizzes, bool hostPriorities, CombinedFlight *flight,
                        std::vector<uint32_t> *unavailable /* n x UnavailableWordCount() words, local bit order */);
  Error CollectCombined(int ctx, const CombinedFlight &flight, CiHipSelection *winners, PriorityView *views);
  size_t UnavailableWordCount() const { return _hQGap.size(); }
  void SetExternalCallers(const std::atomic<int> *n) { _extCallers = n; }   // the sharded engine's count of client threads inside it
  Error FlushDeferred();   // launch whatever posterior updates are deferred (the sharded engine's training barrier)
  int Device() const { return _device; }
  int64_t FirstQuestion() const { return _qFirst; }
  int64_t LocalQuestions() const { return _Q; }
  int64_t RowLength() const { return _ldT; }
  bool OwnsQuestion(int64_t qGlobal) const { return qGlobal >= _qFirst && qGlobal < _qFirst + _Q; }
  const double *PriorityDevicePtr() const { return _dPriority; }   // filled by EnqueueEval, local question order
  void BumpQuestionsAsked(uint64_t n) { _nQuestionsAsked.fetch_add(n, std::memory_order_relaxed); }
  Error GetRowPointers(int64_t qGlobal, int64_t iAnswer, const void **ppA, const void **ppD);
  // rowDevices / stageRow (optional): the device each row lives on, and whether it is copied to this device first (no peer access)
  int64_t ResumeQuizRows(Error &err, int64_t nAnswered, const AQ *pAQs, const void *const *rows, const int *rowDevices = nullptr,
                         const char *stageRow = nullptr);
  // .kb arrays of this engine's questions at the file's current position (hip_engine_kb.cpp); the engine's lock is not taken
  Error IoRows(FILE *f, const char *filePath, bool mD, bool write);
  Error IoVB(FILE *f, const char *filePath, bool write);
  Error SetVBFromHost(const double *vb);
  void GetGapLists(std::vector<int64_t> &questionsGlobal, std::vector<int64_t> &targets) const {
    for (int64_t q : _questionGapList) questionsGlobal.push_back(q + _qFirst);
    targets = _targetGapList;
  }
  const IdLedger &TargetIds() const { return _targetIds; }
  void SetTargetIds(const IdLedger &l) { _targetIds = l; }
  void SetQuizIds(const IdLedger &l) { _quizIds = l; }
  void SetQuestionsAsked(uint64_t n) { _nQuestionsAsked.store(n); }
  uint8_t PrecisionType() const { return _precType; }
  uint32_t PrecMantissa() const { return _precMantissa; }
  uint16_t PrecExponent() const { return _precExponent; }
  Error UnavailableWords(int64_t iQuiz, std::vector<uint32_t> &words);
  // a batched selection in two halves: every shard's sweep is in flight before the first one is waited for
  Error EnqueueBatch(int64_t n, const int64_t *pQuizzes, bool wantPriorities, uint64_t *pTag);
  Error CollectBatchSelections(int64_t n, uint64_t tag, CiHipSelection *pOut);
  Error CollectBatchPriorities(int64_t n, double *pOut);
  Error ValidateTrain(int64_t nQuestions, const AQ *pAQs, int64_t iTarget, int64_t iQuiz);
  bool IsRegularMode() const { return _mode == Mode::Regular; }
  bool IsMaintenanceMode() const { return _mode == Mode::Maintenance; }
  double InitAmount() const { return _initAmount; }
  const void *QuestionBlock(int64_t qLocal) const { return CubeAt(qLocal); }
  const void *RowPointer(int64_t qLocal, int64_t row) const { return CubeAt(qLocal, row); }   // row < K: sA[q][row][.]; row == K: mD[q][.]
  int64_t Answers() const { return _K; }
  int64_t CombinedBatchFor(int64_t waiting) const { return PreferredCombinedBatch(waiting); }
  const double *VBDevicePtr() const { return _dVB; }
  // a rebuilt shard takes its questions' rows from the old shards' cubes (in place, over peer access) ...
  Error AdoptRows(const std::vector<const void *> &srcBlocks, int64_t ldTsrc, const std::vector<int64_t> &colMap, const double *srcVB);
  // ... and initialises added target columns and questions as CpuEngine::AddQsTsSpec does (CpuEngine.cpp:497-567)
  Error ApplyFills(const std::vector<int64_t> &tIds, const std::vector<double> &tInit, const std::vector<int64_t> &qLocalIds,
                   const std::vector<double> &qInit);
  const IdLedger &QuizIds() const { return _quizIds; }

 private:
  HipEngine() = default;
  Error Init(const CiEngineDefinition &def, const CiHipShard *shard);
  KbView View() const;
  Quiz *UseQuiz(Error &err, int64_t iQuiz);                 // reference BaseEngine::UseQuiz, BaseEngine.cpp:399-419
  Error CheckRegular(const char *what) const;               // MaintenanceSwitch gate of BaseEngine.cpp:423-427 etc.
  int64_t CreateQuiz(Error &err, int64_t nAnswered, const AQ *pAQs, const void *const *rows, const double *srcPrior, int srcDevice,
                     hipEvent_t ready);  // CpuEngine::CreateQuizInternal
  void DestroyQuiz(Quiz *q);
  int64_t FinishSelection(Error &err, Quiz *q, int64_t sel);  // gap/asked fallback + bookkeeping (CpuEngine.cpp:404-413)
  int64_t FindNearestQuestion(int64_t iMiddleGlobal, const Quiz *q) const;  // BaseEngine.cpp:60-124
  bool QuestionUnavailable(const Quiz *q, int64_t qGlobal) const;
  Error RecordAnswerImpl(int64_t iQuiz, int64_t iAnswer, bool remote);
  Error RecordAnswerLocked(int64_t iQuiz, int64_t iAnswer, bool remote, bool flushNow);
  StartBatchInline *_startBatch = nullptr;   // StartQuizBatch: CreateQuiz queues the new quizzes' buffers here instead of launching
  void BuildTrainSteps(int64_t n, const AQ *pAQs, bool fromQuiz, std::vector<TrainStep> &steps, std::vector<int64_t> &chainStart) const;
  Error TrainLocked(int64_t nQuestions, const AQ *pAQs, int64_t iTarget, double amount, bool fromQuiz);
  Error ValidateTrainLocked(int64_t nQuestions, const AQ *pAQs, int64_t iTarget) const;
  uint64_t NextRandom();
  Error UploadGaps();
  Error ReallocKB(int64_t newQ, int64_t newT);               // grow the device cube / vB / per-question buffers
  int64_t AssignQuiz(Quiz *q);                               // reference BaseEngine::AssignQuiz, BaseEngine.cpp:780-793
  void UnassignQuiz(int64_t iQuiz);

  enum class Mode { Regular, Maintenance, Shutdown };

  int64_t _K = 0, _Q = 0, _T = 0, _ldT = 0, _qFirst = 0, _qTotal = 0;
  double _initAmount = 1;
  int _device = 0;
  hipStream_t _stream = nullptr, _ownStream = nullptr;
  // the cube: [Q][K+1][ldT] elements of _elem bytes -- double for Double engines, float for Float engines (reference
  // PqaCore/Interface/PqaCommon.h:17-24; the reference instantiates Float for its GPU engine, PqaEngineBaseFactory.cpp:124-142).
  // Everything else (vB, the quizzes' posteriors, priorities) is fp64 in both.
  char *_dCube = nullptr;
  int _elem = 8;
  uint8_t _precType = 3;
  char *CubeAt(int64_t q, int64_t row = 0) const { return _dCube + ((size_t)(q * (_K + 1) + row) * (size_t)_ldT) * (size_t)_elem; }
  static int64_t RoundLdT(int64_t T, int elem) { const int64_t m = 128 / elem; return ((T + m - 1) / m) * m; }   // rows start on 128-byte lines
  double *_dVB = nullptr, *_dPriority = nullptr, *_dRunLength = nullptr, *_dPoleScratch = nullptr;
  size_t PoleScratchBytes(int64_t capQ) const { return (size_t)capQ * (size_t)(2 * _K + 2) * sizeof(double) + PoleListBytes(capQ); }
  uint32_t *_dTGap = nullptr, *_dQGap = nullptr;
  int64_t *_dExps = nullptr, *_dAqs = nullptr, *_dStatus = nullptr, *_dNOut = nullptr;
  int64_t _aqCapacity = 0;
  // ListTopTargets over rows of more than 16384 targets / for many quizzes at once (kb_kernels.hip: LaunchTopTargetsBatch): the two
  // buffers of candidate lists on the device, and the host-coherent lines a batch's results arrive in (records, then the counts)
  RatedTargetDev *_dTopScratch[2] = {nullptr, nullptr};
  int64_t _topScratchRecords = 0;
  RatedTargetDev *_hTopBatch = nullptr;
  int64_t _hTopBatchRecords = 0;
  Error EnsureTopScratch(int64_t nQuizzes, int64_t want);
  int64_t ListTopTargetsOnHost(Error &err, Quiz *q, int64_t want, CiRatedTarget *pDest, bool referenceOrder = false);
  // ListTopTargets = the fast listing with one entry more than asked for; where it shows equal probabilities, the reference's own
  // order among them (its per-worker heaps, reproduced on the device: kb_kernels.hip LaunchTopTargetsExact) -- option "top_exact"
  int64_t ListTopTargetsFast(Error &err, int64_t iQuiz, int64_t maxCount, CiRatedTarget *pDest);
  int64_t ListTopTargetsExact(Error &err, int64_t iQuiz, int64_t maxCount, CiRatedTarget *pDest);
  Error EnsureTopExactScratch(int64_t nQuizzes, int64_t want);
  void *_dTopExact = nullptr;
  size_t _topExactBytes = 0;
  int64_t _optTopExact = 1;
  int64_t _topExactListings = 0;   // read-only option "top_exact_listings": listings that took the heaps' path
  // Released quizzes' device buffers, reused by the next StartQuiz / ResumeQuiz of the same dimensions: hipMalloc / hipFree
  // cost tens of microseconds and hipFree synchronises the device.  Reuse is ordered by the engine's stream.
  struct QuizBuffers { double *dPrior; uint32_t *dAsked; int64_t ldT; size_t askedWords; };
  std::vector<QuizBuffers> _quizBufferPool;
  struct GraphEntry { hipGraphExec_t exec; int64_t variant; hipStream_t stream; uint64_t kbVersion; };
  uint64_t _kbVersion = 0;
  std::unordered_map<Quiz *, GraphEntry> _graphs;  // option "use_graph"
  SelectResult *_dGraphScratch = nullptr;
  uint64_t *_dTagCell = nullptr;
  // The launched, graph-replayed and resident selections all report through _hPinned->sel / ->seq, each waiting for "its" value of
  // the flag: the three sequences live in disjoint ranges, or a selection would find the flag already holding its value -- left by
  // another path's earlier selection -- and return that one's question (graph tag 1, then the resident sweep's first request, also
  // 1: found by the soak of tools/stress_more.py).  Launch tags (NextLaunchTag) count from 1 and stay below 2^40.
  static constexpr uint64_t kGraphFlagBase = 1ull << 40, kServerFlagBase = 2ull << 40;
  uint64_t _graphTag = kGraphFlagBase + 1;
  void DropQuizBufferPool();
  SelectResult *_dSel = nullptr;
  struct Pinned {  // host-coherent: written by kernels, polled / read by the host without copies
    SelectResult sel; uint64_t seq; int64_t status[2]; int64_t nOut;
    uint64_t opFlag;               // completion flag of the sampled selection kernel
    uint64_t topFlag;              // completion flag of whatever listed into top[] last
    RatedTargetDev top[256];
  };
  // (Pinned::top: listings of more than kQuizTop targets; up to kQuizTop the quiz's own lines are used, where RecordAnswer's
  //  kernel lists the new posterior's best targets ahead of the ListTopTargets for the same quiz and posterior)
  uint64_t _opSeq = 0;
  // ---- the quizzes' pinned lines: slabs of host-coherent memory, handed out per quiz, returned at its release
  std::vector<QuizPinned *> _pinSlabs, _pinFree;
  static constexpr int kPinSlab = 256;
  QuizPinned *TakePin();
  // ---- deferred posterior updates.  While several client threads are inside the engine, RecordAnswer only does its
  // bookkeeping and leaves the kernel to whoever next needs a posterior (ListTopTargets, NextQuestion, ...): that caller launches
  // ONE kernel for all the updates that have gathered (grid.x = update; prior_kernels.hip: record_answer_batch_kernel) instead
  // of one launch each.  Alone in the engine, RecordAnswer launches at once, as before.
  struct BatchCtx;
  struct PendingUpdate { Quiz *q; int64_t qLocal, iAnswer; const void *rowA = nullptr, *rowD = nullptr; bool list = true; };   // rowA: another shard's question
  std::vector<PendingUpdate> _pendingUpdates;
  Error FlushUpdates();                       // the caller holds _mu
  void MarkStreamBusy();
  uint64_t _flushes = 0, _flushedUpdates = 0, _maxFlush = 0;
  std::atomic<int> _activeCallers{0};         // client threads inside quiz-level calls right now
  struct CallScope {
    std::atomic<int> &n;
    explicit CallScope(std::atomic<int> &c) : n(c) { n.fetch_add(1, std::memory_order_relaxed); }
    ~CallScope() { n.fetch_sub(1, std::memory_order_relaxed); }
  };
  const std::atomic<int> *_extCallers = nullptr;   // a shard: the client threads inside the sharded engine that drives it
  int Callers() const {
    const int own = _activeCallers.load(std::memory_order_relaxed);
    return _extCallers ? std::max(own, _extCallers->load(std::memory_order_relaxed)) : own;
  }
  bool Concurrent() const { return _optCombine && Callers() > 1; }
  static int AllowedCpus();                   // the CPUs this process may use: the cgroup's quota (cpu.max) or the affinity mask
  int64_t _optCombineSpin = 1;                // option "combine_spin": 1 = waiting clients spin while they are fewer than the allowed CPUs, 0 = they always sleep
  int64_t _optPostAlways = 0;                 // option "post_always" (test hook): the posted form of RecordAnswer / ListTopTargets even when the engine is free
  bool ClientsFitCpus() const { return _optCombineSpin && Callers() + 2 <= AllowedCpus(); }
  int64_t _optCombine = 1;                    // option "combine" / PQA_COMBINE: 0 = every call by itself, as before
  // ---- combining of concurrent NextQuestion calls (reference: every client's NextQuestion runs under a SHARED lock,
  // PqaCore/CpuEngine.cpp:357-361, Interface/IPqaEngine.h:44).  A caller posts its request; if a leader is at work it waits
  // for its result, otherwise it becomes the leader: it takes everything posted so far -- distinct quizzes -- and serves it
  // with ONE row-sharing / grid.y sweep (BatchSweep), selects per quiz, hands the results out and, if requests are waiting
  // again, passes the lead to the oldest of them.  One request: the single-quiz path, as before.
  struct SelRequest {
    int64_t iQuiz = -1;
    int kind = 0;                      // 0: argmax, 1: sampled (rnd)
    uint64_t rnd = 0;
    int64_t result = -1;
    Error err;
    std::atomic<int> state{0};         // 0 waiting, 1 served, 2 lead handed over: serve the queue yourself,
                                       // 3 the sweep is done: select for yourself from pri[] (the leader does not do it for everybody)
    const double *pri = nullptr;       // state 3: priority of local question k at pri[k * priStride]
    int64_t priStride = 0;
    uint64_t priTag = 0;               // != 0: pri[k * priStride + 1] is the entry's launch tag -- the entry is taken once it carries this one
    BatchCtx *ctx = nullptr;           // state 3: whose reader count is this request's to release
    uint64_t serial = 0;               // state 3: the quiz the sweep ran for
    Quiz *quiz = nullptr;              // state 3: held by inSelection
    std::vector<uint32_t> unavailable; // state 3: the quiz's asked questions and the gaps as they were when the sweep was launched
    int64_t nQ = 0, nSub = 0;
  };
  int64_t SelectFromPriorities(SelRequest *r);     // the host-side selector + FinishSelection for one request of a combined sweep
  std::atomic<size_t> _pendingCount{0};            // == _pendingUpdates.size(), readable without the lock
  std::atomic<int64_t> _flushedSinceSweep{0};      // RecordAnswers launched since the newest combined sweep: their clients' NextQuestions are on their way
  std::atomic<int64_t> _sweepNsEwma{0};            // how long a follower of a combined sweep waits for its result (moving average): it sleeps most of that, then spins
  std::atomic<int64_t> _lastCombined{0};           // requests of the newest combined sweep: as many RecordAnswers are about to arrive
  int64_t _optLingerUs = 20;                       // option "combine_linger_us": how long a ListTopTargets waits for them before it launches the updates
  std::mutex _combMu;
  std::vector<SelRequest *> _combQueue;
  bool _leaderActive = false;
  int64_t Combine(Error &err, int64_t iQuiz, int kind, uint64_t rnd);
  // ---- posted operations.  With dozens of client threads the engine's lock is not held long but changes hands through the
  // kernel every time: each RecordAnswer and ListTopTargets slept on it and was woken by the thread before it, one wake-up
  // latency per call, serially (64 threads: 200 us inside a RecordAnswer that works for 1).  So a call that finds the lock
  // taken does not queue on it: it posts its operation and sleeps on the operation's own word; whoever holds the lock runs
  // everything posted so far right before it lets go (EngineMutex::unlock) -- the RecordAnswers of a drain into the list of
  // deferred updates, ONE launch for all the posteriors its ListTopTargets ask for -- and wakes the posters, all at once.
  struct Flight;
  struct PostedOp {
    int kind = 0;                      // 1: RecordAnswer(iQuiz, arg = iAnswer, remote); 2: ListTopTargets' launch (arg = maxCount);
                                       // 3: a leader's LaunchBatch(ctx, batch, flight); 4: StartQuiz (result = the quiz);
                                       // 5: ReleaseQuiz(iQuiz); 6: RecordQuizTarget(iQuiz, arg = iTarget, amount)
    double amount = 0;
    int64_t iQuiz = -1, arg = 0;
    bool remote = false;
    Error err;
    int64_t result = 0;                // kind 2: the number of targets to take from `pin` once it carries `flagOp`; -2: take the lock yourself
    QuizPinned *pin = nullptr;
    uint64_t flagOp = 0;
    Quiz *quiz = nullptr;              // (the drain's own, between its two passes)
    uint64_t serial = 0;
    BatchCtx *ctx = nullptr;
    std::vector<SelRequest *> *batch = nullptr;
    Flight *flight = nullptr;
    std::atomic<int> state{0};         // 0 posted, 2 posted and its thread asleep on this word, 1 done
    PostedOp *next = nullptr;
  };
  std::atomic<PostedOp *> _posted{nullptr};
  std::vector<std::atomic<int> *> _postedWake;     // the drain's sleepers, woken once the lock is released
  uint64_t _postedOps = 0, _postedDrains = 0;
  void DrainPosted();                              // (the engine's lock held)
  void TrainPosted(PostedOp *ordered);             // the drain's RecordQuizTarget calls
  uint64_t _trainBatches = 0, _trainBatchCalls = 0;
  Error ReleaseQuizLocked(int64_t iQuiz, bool mayWait);
  Error RecordQuizTargetLocked(int64_t iQuiz, int64_t iTarget, double amount);
  void RunPosted(PostedOp &op);                    // post, and return when somebody has run it
  void ServeQueue(SelRequest *own);
  int64_t PreferredCombinedBatch(int64_t m) const;
  struct Flight {                      // a combined sweep between its launch and its collection
    std::vector<SelRequest *> live;
    uint64_t tag = 0;
    bool anySampled = false, quizMinor = false, tagged = false;
    int64_t Bp = 0, nQ = 0;
    hipError_t he = hipSuccess;
    std::chrono::steady_clock::time_point tA, tB, tC;
  };
  void LaunchBatch(BatchCtx &c, std::vector<SelRequest *> &batch, Flight &f);
  void LaunchBatchLocked(BatchCtx &c, std::vector<SelRequest *> &batch, Flight &f);
  bool CollectBatch(BatchCtx &c, std::vector<SelRequest *> &batch, Flight &f, SelRequest *own);   // true: `own` is to select for itself
  int64_t NextQuestionArgmaxLocked(Error &err, int64_t iQuiz);
  int64_t NextQuestionSampledLocked(Error &err, int64_t iQuiz, uint64_t rnd);
  std::mutex _rngMu;
  uint64_t _quizSerial = 0;
  uint64_t _combBatches = 0, _combRequests = 0, _combMaxBatch = 0;
  std::atomic<uint64_t> _combNs[7] = {{0}, {0}, {0}, {0}, {0}, {0}, {0}};
  volatile uint64_t *_pendingRecordFlag = nullptr;   // where _pendingRecordOp will appear
  // spin on a host-coherent flag until it holds `value` (the kernel's last store); falls back to the stream's status
  Error WaitFlag(volatile uint64_t *flag, uint64_t value, const char *what);
  Error WaitFlagNapping(volatile uint64_t *flag, uint64_t value, const char *what);   // for many waiters at once: a short spin, then naps
  SelectResult *_dSelScratch = nullptr;  // its per-workgroup winner records
  double *_dPriorScratch = nullptr;      // the long-row posterior kernels' subtask sums (KbView::priorScratch)
  int64_t _optPoleFix = 1;               // option "pole_fix"
  int64_t _optPoleFollow = 1;            // option "pole_follow" (measurement hook)
  int64_t _optLateEager = 3;             // option "late_eager": see Quiz::lateStreak (0: speculative sweeps always with the fix-up behind them)
  int64_t _optTimeSweeps = 0;            // option "time_sweeps" (measurement hook): events around every launched fp64 sweep; read-only "last_sweep_ns"
  hipEvent_t _evSweep[2] = {nullptr, nullptr};
  mutable bool _sweepTimed = false;
  int64_t _optPoleGate = 1;              // option "pole_gate": a fused single-quiz ARGMAX has only the listed questions redone that can still win (pole_kernels.hip: pole_bounds_kernel)
  int64_t _optPoleLazy = 1;              // option "pole_lazy": synchronous single-quiz selections launch the fix only when the sweep listed something (FusedSelect::lazyFix)
  int64_t _optLongRowForm = 1;           // option "long_row_form": StartQuiz / RecordAnswer over rows beyond 16384 targets as one workgroup per subtask of the sum
  // batched selections (NextQuestionArgmaxBatch); allocated on first use
  static constexpr int64_t kMaxBatch = 256, kBatchGrid = 1024;
  struct BatchPinned { QuizSlot slots[kMaxBatch]; SelectResult out[kMaxBatch]; uint64_t seq[kMaxBatch]; };
  // Everything ONE batched sweep in flight needs of its own: the staged slots and winner records, the scratch of the
  // row-sharing sweep (batch_kernels.hip; sized by its plan, grown on demand), the host copy of the priority vectors.  Two of
  // them: the batch calls of the ABI and the shards' halves use the first; the leaders of combined sweeps alternate, so that
  // the next sweep is launched while the previous one runs.
  struct BatchCtx {
    std::mutex mu;                       // one batch at a time in this context (taken before the engine's lock)
    BatchPinned *h = nullptr;
    QuizSlot *dSlots = nullptr;
    SelectResult *dScratch = nullptr;
    double *dPriority = nullptr;
    int64_t priorityQ = -1;
    void *dPT = nullptr; double *dAcc = nullptr; BatchRecord *dRecs = nullptr; double *dPriT = nullptr;
    size_t ptBytes = 0, accBytes = 0, recBytes = 0, priTBytes = 0, rerankBytes = 0;
    void *dRerank = nullptr;             // Float engines: the candidates of the fp64 re-rank and their priorities
    void *dPole = nullptr;               // Double engines: the batched sweeps' pole scratch (BatchPlan::pole)
    size_t poleBytes = 0;
    int lastBp = 0;
    double *hPri = nullptr;              // pinned: the batch's priority vectors for the host-side selector -- copied there behind the
    size_t hPriDoubles = 0;              // row-sharing sweep, or written there by the grid.y = quiz sweep itself as {priority, launch tag} records
    bool hPriCoherent = false;           // hPri is host-coherent mapped memory (the kernels write it) rather than a copy's destination
    std::atomic<int> readers{0};         // clients still selecting out of hPri
    hipEvent_t event = nullptr;
    std::atomic<bool> inFlight{false};   // a leader's sweep launched and not yet collected
  };
  BatchCtx _ctx[2];
  int _ctxNext = 0;                      // (the leader's)
  // wantPriorities: the row-sharing sweep with its priority matrix kept (EvalPrioritiesBatch).  hostPriorities: whichever form
  // suits the batch, and the quizzes' priority vectors copied into _hBatchPri (layout: *pQuizMinor) for the host's selector.
  Error BatchSweep(BatchCtx &c, int64_t n, const int64_t *pQuizzes, std::vector<Quiz *> &quizzes, bool wantPriorities, uint64_t tag,
                   bool hostPriorities = false, bool *pQuizMinor = nullptr, bool *pTagged = nullptr);
  Error WaitBatchFlags(BatchCtx &c, int64_t n, uint64_t tag);
  Error EnqueueBatchLocked(int64_t n, const int64_t *pQuizzes, bool wantPriorities, uint64_t *pTag);
  Error CollectBatchSelectionsLocked(int64_t n, uint64_t tag, CiHipSelection *pOut);
  Error CollectBatchPrioritiesLocked(int64_t n, double *pOut);
  std::vector<Quiz *> _batchQuizzes;   // the quizzes of the batch between its two halves
  void *_dClusterScratch = nullptr;   // exchange buffers of the long-row sweep (cluster_kernels.hip), grown on demand
  size_t _clusterScratchBytes = 0;
  bool UseClusterSweep() const;       // rows beyond the register shapes, automatic variant, shape supported
  Error LaunchSingleSweep(Quiz *q, const FusedSelect *fused);
  bool LazyFix() const;
  Error RunLazyFix(Quiz *q, const FusedSelect &swept, const char *what);   // the single-quiz sweep of this engine's precision, on _stream
  uint64_t _selSeq = 0;
  // Tag of the next fused launch: consecutive launches differ in the low 32 bits, and those are never 0 (the state of
  // freshly cleared records)
  uint64_t NextLaunchTag() {
    if ((uint32_t)++_selSeq == 0) ++_selSeq;
    return _selSeq;
  }
  Pinned *_hPinned = nullptr;
  std::vector<uint32_t> _hTGap, _hQGap;   // host mirrors; qgap over local questions, bits past size set
  int64_t _nTargetGaps = 0;
  int64_t _capQ = 0;                        // questions the device buffers are allocated for (>= _Q)
  std::vector<int64_t> _questionGapList, _targetGapList;  // LIFO, like reference PqaCore/GapTracker.h
  IdLedger _questionIds, _targetIds, _quizIds;
  uint32_t _precMantissa = 0;
  uint16_t _precExponent = 0;
  std::vector<Quiz *> _quizzes;
  std::vector<int64_t> _quizGaps;
  friend struct KbIo;
  // The engine's lock.  Taking it also marks the engine's stream as possibly busy: the resident sweep runs on its own
  // stream, so a request is posted only after whatever the other operations enqueued on `_stream` has finished
  // (ServerPost); the selection paths, which leave nothing running, restore the mark they found.
  struct EngineMutex {
    // (a sleeping lock on purpose: the client threads of a server may outnumber the cores it is allowed -- the GPU boxes of
    //  this project give a container 16 of 256 hardware threads -- and spinning waiters, tried in round 3 as a test-and-set and
    //  as a ticket lock, burn that allowance: 64 client threads fell from 45 k to 13 k questions/s, 256 to 0.3 k)
    std::mutex m;
    bool busy = false, wasBusy = false;
    // (spinFirst -- set while the client threads are fewer than the CPUs the process may use: a bounded spin before sleeping; the
    //  holder is usually a microsecond of bookkeeping or one kernel launch away from releasing, and being woken through the kernel
    //  costs tens of microseconds.  With more clients than CPUs every spinning waiter takes time from a thread that has work.)
    std::atomic<bool> spinFirst{false};
    HipEngine *owner = nullptr;
    bool try_lock() {
      if (!m.try_lock()) return false;
      wasBusy = busy; busy = true;
      return true;
    }
    void lock() {
      if (spinFirst.load(std::memory_order_relaxed))
        for (int i = 0; i < 400; i++) {
          if (m.try_lock()) { wasBusy = busy; busy = true; return; }
          for (int j = 0; j < 4; j++) __builtin_ia32_pause();
        }
      m.lock(); wasBusy = busy; busy = true;
    }
    void lock_urgent() { lock(); }
    void unlock();   // (runs the posted operations first: hip_engine_combine.cpp)
  };
  struct UrgentLock {   // (RAII for lock_urgent, re-lockable like std::unique_lock)
    EngineMutex &mu;
    bool held = false;
    explicit UrgentLock(EngineMutex &m) : mu(m) { lock(); }
    ~UrgentLock() { if (held) unlock(); }
    void lock() { mu.lock_urgent(); held = true; }
    void unlock() { mu.unlock(); held = false; }
  };
  mutable EngineMutex _mu;
  std::atomic<uint64_t> _nQuestionsAsked{0};
  Mode _mode = Mode::Regular;
  // options
  int64_t _optSelect = 0, _optWorkers = 16, _optEvalSubtasks = 0, _optEvalVariant = 0;
  // ResumeQuiz seeds the first answered question's product from vector 0 of vB for every target vector, as the reference binary
  // does (PqaCore/CEUpdatePriorsSubtaskMul.cpp:53 loads pvB, not pvB + j): the drop-in default.  0 = the evident intent.
  int64_t _optBugCompat = 1;
  void ApplyEnvironment();   // PQA_SELECT / PQA_SERVER / PQA_BUG_COMPAT / PQA_WORKERS / PQA_SEED: defaults for unchanged wrappers
  int64_t _optHostSampled = 1;    // the sampled NextQuestion as ONE launch + the selector on the host (the finisher workgroup hands over the priority vector)
  TaggedPriority *_hHostPriority = nullptr;   // host-coherent, _hostPriorityCap records {priority, launch tag}
  std::vector<double> _hostRun;               // the vector the host-side selector works on
  Error CollectHostPriority(uint64_t tag, const Quiz *q);   // the launch's entries out of _hHostPriority into _hostRun
  int64_t _hostPriorityCap = 0;
  hipError_t EnsureHostPriority();
  int64_t _optFusedSampled = 0;   // the sampled NextQuestion as ONE launch (the sweep's finisher workgroup runs the selector): correct,
                                  // but 38.3 vs 36.4 us at 1000 x 5 x 1000 -- one workgroup's serial selection costs more than a launch
  int64_t _optEvalMaxGrid = 0;    // test hook: KbView::maxGrid
  int64_t _optBatchMin = 0;       // batches of at least this many quizzes take the row-sharing sweep (lane = quiz), smaller ones grid.y = quiz; 0 = by the number of waves the batch gives the row-sharing sweep
  int64_t _optBatchForm = 0;      // 0: the batch's form by its size and the cube's shape; 1 grid.y = quiz, 2 row-sharing, 3 (quiz, chunk) lanes
  int64_t _optRerank = 1;         // Float engines' batched argmax: the fp32 sweep's best 8 questions per quiz re-ranked in fp64
  int64_t _optBatchQb = 0;        // questions per block of that sweep (0 = default)
  int64_t _optBatchTile = 0;      // targets per LDS tile of that sweep (0 = default)
  int64_t _optClusterShape = 0;   // ... the shape of the form that runs ahead (cluster_kernels.hip: kAheadVariants), 0 = default
  // Rows longer than this many elements take the cluster sweep (option cluster_from, 1024..16384).  10240: what the register shapes hold
  // without spilling -- the 16-wave shapes behind them (128 registers a lane) ran 10500^2 at 2552 us against the cluster's 1597, 12000^2
  // at 3004 against 1974, 16000^2 at 4230 against 3430 (round 6, one box); they stay selectable (eval_variant 6, 7, 11).
  int64_t _optClusterFrom = 10240;
  int64_t ClusterFrom() const { return _elem == 8 ? _optClusterFrom : 16384; }   // (Float engines: their register shapes hold 16384 elements; not re-measured)
  int64_t _optClusterForm = 0;    // long rows, one quiz (cluster_kernels.hip): 0 = default, 1 = question by question, 2 = pass 1 a question ahead
  int64_t _optBatchTail = 1;      // that sweep's last, partial round as a launch of its own with fewer questions per group (LaunchEvalBatch)
  int64_t _optBatchGroups = 0;    // question groups per workgroup of that sweep for batches under 129 quizzes (0 = automatic)
  // ---- the next sweep ahead of its request (option "speculate"): RecordAnswer enqueues, right behind its posterior kernel, the
  // sweep the NextQuestion that normally follows would launch -- the client's time between the two calls (the wrapper's own
  // overhead, ListTopTargets, a person reading the question) overlaps with it, and that NextQuestion only waits for the flag.
  // The result is used only if nothing has touched the quiz, the cube, the gaps or the hand-over buffers since (every such
  // operation drops it); the random number of the sampled selector is drawn when NextQuestion is called, as before.
  struct Speculation {
    Quiz *quiz = nullptr;
    uint64_t priorVersion = 0, tag = 0;
    int kind = 0;            // 1: argmax record in _hPinned->sel; 2: priority vector in _hHostPriority; 3: priorities in _dPriority
    int64_t variant = 0;
    hipStream_t stream = nullptr;
    FusedSelect fs{};        // the launch's own arguments: a speculative sweep is launched WITHOUT the pole fix-up behind it (lazyFix) --
                             // most of a quiz's last speculations are never used, and their fix-ups were the longest kernels of the loop
  } _spec;
  // A sweep launched with lazyFix whose answer nobody has looked at yet may have left entries in the engine's suspect list: the next
  // launch that uses the list empties it first (and the speculation that left them is dropped: its fix-up would find nothing)
  bool _poleListPending = false;
  Error SettlePoleList();
  int64_t _optSpeculate = 1;
  int _specScore = 0;        // +1 per speculation used, -1 per speculation dropped: below -4 only every 32nd RecordAnswer speculates
  uint64_t _specProbe = 0, _specHits = 0, _specDropped = 0;
  bool Speculate(Quiz *q, int64_t updQuestion = -1, int64_t updAnswer = -1);
  int64_t _optFuseUpdate = 1;   // option "fuse_update"
  uint64_t _fusedUpdates = 0;
  int64_t SpeculateFor(int64_t iQuiz);
  int TakeSpeculation(Quiz *q, int kindMask, uint64_t *pTag);
  void DropSpeculation() {
    if (_spec.quiz != nullptr) { _spec.quiz = nullptr; _specDropped++; if (_specScore > -8) _specScore--; }
  }
  int64_t _optUseGraph = 0;   // NextQuestion (argmax) replays a per-quiz HIP graph instead of launching
  int64_t _topWantRecent = 10;   // what ListTopTargets has been asked for lately
  int64_t _optTopCache = 10;  // targets RecordAnswer's kernel lists ahead of the ListTopTargets that follows it (0: none)
  // ---- resident sweep (option "server"; pqa_kernels.h: ServerMailbox)
  int64_t _optServer = 0, _optServerIdleUs = 500, _optServerVramMailbox = 1;
  hipStream_t _serverStream = nullptr;
  ServerMailbox *_hMailbox = nullptr;     // pinned
  ServerCtl *_dServerCtl = nullptr;
  volatile uint64_t *_serverRequest = nullptr;   // the request line: host-visible device memory if the platform maps it, else the mailbox's
  bool _serverRequestInVram = false;
  uint64_t _serverReqSeq = 0;                    // the sequence number last written there (the line is never read by the host)
  bool _serverLaunched = false;           // a kernel instance has been launched and not yet seen to have left
  uint64_t _serverKb = 0, _serverPosted = 0;
  int64_t _serverVariant = 0;
  uint64_t _pendingRecordOp = 0;          // RecordAnswer's kernel is the newest work on _stream and publishes this op number
  bool ServerUsable() const;
  Error ServerPost(Quiz *q, SelectResult *out, uint64_t *flag, uint64_t flagValue, int64_t outBase);
  Error ServerWait(volatile uint64_t *flag, uint64_t value, const char *what);
  void StopServer();
  void ServerQuiesce();   // returns once the posted step (if any) has finished: before anything that writes what it reads
  uint64_t _rng[2] = {0, 0};
};

void LogAnomaly(DefaultLogger::Severity sev, const char *what, double value);   // the reference's numeric-anomaly log entries (rate-limited)
void CheckPriority(double priority, int64_t index);
int64_t SelectSampledHost(double *run, int64_t n, int64_t nWorkers, uint64_t rnd, const std::function<bool(int64_t)> &skipped);
int64_t SelectSampledHostBits(double *run, int64_t n, int64_t nWorkers, uint64_t rnd, const uint32_t *a, const uint32_t *b);
int64_t FindNearestInPacks(int64_t iMiddle, int64_t nQuestions, const std::function<uint64_t(int64_t)> &avail);
// One knowledge base over several devices of this process (sharded_engine.cpp); devices.size() >= 2.
IEngine *CreateShardedEngine(Error &err, const CiEngineDefinition &def, const std::vector<int> &devices);
IEngine *LoadShardedEngine(Error &err, const char *filePath, const std::vector<int> &devices);

}  // namespace pqa
