// sharded_engine.cpp -- one knowledge base, its question axis split over several devices of ONE process, behind the same C ABI.
//
// every question's priority depends only on its own sA / mD rows and on the (small, replicated) posterior, so
// the sweep shards by question with no data-path collective; what the shards exchange per selection is 16 bytes each.
// PQA_DEVICES=0,1,...,7 makes PqaEngineFactory_CreateCpuEngine build this engine instead of a single-device one, so that the
// reference's unchanged wrappers -- which can only call PqaEngine_NextQuestion (PqaCInterop.cpp:261-267) -- reach all the GPUs
// of a node.  Shard s is a complete HipEngine over the contiguous question range SRPoolRunner::CalcSplit gives it
// (SRPlatform/Interface/SRPoolRunner.h:96-110), with replicas of vB, the gap bitmaps and every quiz's posterior.
//
//   NextQuestion           one client: every shard's sweep is enqueued on its own device and stream; each finisher writes {priority,
//                          GLOBAL index} and then the step number into this engine's pinned slots; the host picks when all have
//                          landed (maximum priority, lowest index on ties) -- no collective launch, no copies, no stream sync; the
//                          reference's sampled selector (CpuEngine.cpp:362-400) runs on the host over the shards' priority vectors,
//                          with the subtask split over the GLOBAL question range.
//                          Many clients (the reference serves them under a SHARED lock, CpuEngine.cpp:357-361; its published rate
//                          is the sum over its learner threads, PqaClient.cpp:238-245): the calls that arrive together are
//                          combined -- ONE batched sweep per shard for all their quizzes, all shards in flight at once, the
//                          per-quiz winners (or priority vectors) merged on the host by the clients themselves.
//   RecordAnswer           is recorded here and returns; before anything reads a posterior, ALL gathered answers go to EVERY shard
//                          in one call and one launch each: a shard computes the posterior itself -- for a question another shard
//                          holds it reads that question's two rows where they are, over peer access (xGMI), or from a staged copy
//                          when the devices cannot map each other -- so all replicas hold the same bits and nothing is copied
//                          or waited for between the shards.
//   ResumeQuiz             every shard computes the posterior from row POINTERS (the other shards' rows in place, or staged).
//   Train / RecordQuizTarget   every shard applies the steps that fall on its questions (and its vB replica); other shards' reads
//                          of those rows are ordered around it by events (no host synchronisation).
//   SaveKB / LoadCpuEngine the file orders its rows by question: every shard streams its own block (same byte layout as a whole-cube
//                          engine's file: a KB saved sharded loads unsharded and vice versa).
// One lock (_opMu) orders what must happen in the same order on every shard -- quiz registry changes, trainings, the launch of a
// combined sweep; it is never held while the GPU is waited for, and calls that find it taken post their operation and are served
// by its holder on the way out (as hip_engine_combine.cpp's posted operations).  Maintenance-mode edits of the dimensions rebuild the
// shards (Rebuild below).  Not sharded (NotImplemented on this engine): SetStream and the stream-ordered single-shard entry points.
#include "hip_engine_internal.h"

namespace pqa {

namespace {

Error NotSharded(const char *what) {
  return Error::MakeP(ErrCode::NotImplemented, std::string("Feature=") + what + " on a sharded engine",
                      std::string(what) + " is not available when the question axis is split over several devices (PQA_DEVICES).");
}

struct alignas(64) Slot {          // one per shard, host-coherent pinned memory
  double priority;
  int64_t index;
  uint64_t flag;
};

}  // namespace

class ShardedEngine final : public IEngine {
 public:
  static ShardedEngine *Create(Error &err, const CiEngineDefinition &def, const std::vector<int> &devices);
  ~ShardedEngine() override;

  Error Train(int64_t n, const AQ *pAQs, int64_t iTarget, double amount) override;
  uint64_t GetTotalQuestionsAsked(Error &err) override { return _sh[0]->GetTotalQuestionsAsked(err); }
  void CopyDims(CiEngineDimensions *pDims) const override { _sh[0]->CopyDims(pDims); }
  int64_t StartQuiz(Error &err) override;
  int64_t ResumeQuiz(Error &err, int64_t nAnswered, const AQ *pAQs) override;
  int64_t NextQuestion(Error &err, int64_t iQuiz) override;
  Error RecordAnswer(int64_t iQuiz, int64_t iAnswer) override;
  int64_t GetActiveQuestionId(Error &err, int64_t iQuiz) override;
  Error SetActiveQuestion(int64_t iQuiz, int64_t iQuestion) override;
  int64_t ListTopTargets(Error &err, int64_t iQuiz, int64_t maxCount, CiRatedTarget *pDest) override;
  Error RecordQuizTarget(int64_t iQuiz, int64_t iTarget, double amount) override;
  Error ReleaseQuiz(int64_t iQuiz) override;
  // A forced switch destroys the shards' quizzes (BaseEngine.cpp:650-668), a shutdown likewise: what this engine keeps per quiz goes too
  Error StartMaintenance(bool force) override {
    std::lock_guard<OpMutex> lk(_opMu);
    Error e = FlushAnswers();
    if (!e.ok()) return e;
    e = AllLocked([&](HipEngine &sh) { return sh.StartMaintenance(force); });
    if (e.ok() && force) ForgetQuizzes();
    return e;
  }
  Error FinishMaintenance() override { return All([&](HipEngine &e) { return e.FinishMaintenance(); }); }
  Error Shutdown(const char *saveFilePath) override {
    if (saveFilePath && *saveFilePath) { Error e = SaveKB(saveFilePath, false); if (!e.ok()) return e; }   // BaseEngine.cpp:270-300
    std::lock_guard<OpMutex> lk(_opMu);
    (void)FlushAnswers();
    Error e = AllLocked([&](HipEngine &sh) { return sh.Shutdown(nullptr); });
    if (e.ok()) ForgetQuizzes();
    return e;
  }
  bool MapIds(int which, bool toPerm, int64_t count, int64_t *pIds) override {
    if (which == 0) {   // questions: the global compact <-> permanent map lives here (the shards' own maps are over local ids)
      std::lock_guard<OpMutex> lk(_opMu);
      bool ok = true;
      for (int64_t i = 0; i < count; i++) {
        pIds[i] = toPerm ? _questionIds.PermanentOf(pIds[i]) : _questionIds.SlotOf(pIds[i]);
        ok = ok && pIds[i] != -1;
      }
      return ok;
    }
    return _sh[0]->MapIds(which, toPerm, count, pIds);
  }
  bool EnsurePermQuizGreater(int64_t bound) override { std::lock_guard<OpMutex> lk(_opMu); bool ok = true; for (auto &s : _sh) ok = s->EnsurePermQuizGreater(bound) && ok; return ok; }
  bool RemapQuizPermId(int64_t a, int64_t b) override { std::lock_guard<OpMutex> lk(_opMu); bool ok = true; for (auto &s : _sh) ok = s->RemapQuizPermId(a, b) && ok; return ok; }
  Error SaveKB(const char *filePath, bool doubleBuffer) override;
  static ShardedEngine *Load(Error &err, const char *filePath, const std::vector<int> &devices);
  Error AddQsTs(int64_t nQuestions, CiAddQorTParam *pAqps, int64_t nTargets, CiAddQorTParam *pAtps) override;
  Error RemoveQuestions(int64_t n, const int64_t *pQIds) override;
  Error RemoveTargets(int64_t n, const int64_t *pTIds) override;
  Error Compact(int64_t *pnQuestions, const int64_t **ppOldQuestions, int64_t *pnTargets, const int64_t **ppOldTargets) override;
  Error ClearOldQuizzes(int64_t maxCount, double maxAgeSec) override;

  Error SetOption(const char *name, int64_t value) override {
    const std::string n(name ? name : "");
    std::lock_guard<OpMutex> lk(_opMu);
    Error e = FlushAnswers();
    if (!e.ok()) return e;
    if (n == "combine") { _optCombine = value ? 1 : 0; return Error(); }          // (this engine's own combining; the shards are driven one call at a time)
    if (n == "combine_linger_us") { if (value < 0 || value > 10000) return Error::Make(ErrCode::UnhandledCase, "Unknown option or value out of range: " + n); _optLingerUs = value; return Error(); }
    e = AllLocked([&](HipEngine &sh) { return sh.SetOption(name, value); });
    if (!e.ok()) return e;   // (a value the shards refuse changes nothing here either)
    if (n == "select") _select = value;   // (kept here too: NextQuestion dispatches on it)
    if (n == "seed") Seed((uint64_t)value);
    return Error();
  }
  int64_t GetOption(const char *name) const override {
    const std::string n(name ? name : "");
    if (n == "shards") return (int64_t)_sh.size();
    if (n == "shards_in_flight_max") return _shardsInFlightMax;   // the most shards whose sweeps were enqueued before the first was waited for (newest call)
    if (n == "combine") return _optCombine;
    if (n == "combine_linger_us") return _optLingerUs;
    if (n == "combined_batches") return (int64_t)_combBatches.load();       // sweeps that served more than one NextQuestion call ...
    if (n == "combined_requests") return (int64_t)_combRequests.load();     // ... the calls they served ...
    if (n == "combined_max_batch") return (int64_t)_combMaxBatch.load();    // ... and the largest of them
    if (n == "posted_ops") return (int64_t)_postedOps.load();               // calls that found the engine taken and were run by its holder
    if (n == "answer_flushes") return (int64_t)_answerFlushes.load();       // hand-overs of gathered answers to the shards ...
    if (n == "answers_flushed") return (int64_t)_answersFlushed.load();     // ... the answers they carried ...
    if (n == "answer_max_flush") return (int64_t)_answerMaxFlush.load();    // ... and the most in one
    if (n == "update_max_flush") return (int64_t)_answerMaxFlush.load();    // (the one-device engine's name for it)
    if (n == "start_batches") return (int64_t)_startBatches.load();         // launches that started several quizzes
    if (n == "peer_access") return _peerAll ? 1 : 0;                        // every pair of this engine's devices maps each other's memory
    if (n == "staged_rows") return (int64_t)_stagedRows.load();             // rows of other shards' questions that were copied instead of read in place
    if (n == "train_barriers") return (int64_t)_trainBarriers.load();
    return _sh[0]->GetOption(name);
  }
  const char *EvalKernelName() const override { return _sh[0]->EvalKernelName(); }
  Error SetKB(const double *pA, const double *pD, const double *pB) override {
    std::lock_guard<OpMutex> lk(_opMu);
    { Error fe = FlushAnswers(); if (!fe.ok()) return fe; }   // (the gathered answers read the cube as it was when they were given)
    for (auto &s : _sh) {
      const size_t q0 = (size_t)s->FirstQuestion();
      Error e = s->SetKB(pA + q0 * (size_t)_K * (size_t)_T, pD + q0 * (size_t)_T, pB);
      if (!e.ok()) return e;
    }
    return Error();
  }
  Error GetKB(double *pA, double *pD, double *pB) override {
    std::lock_guard<OpMutex> lk(_opMu);
    { Error fe = FlushAnswers(); if (!fe.ok()) return fe; }
    for (auto &s : _sh) {
      const size_t q0 = (size_t)s->FirstQuestion();
      Error e = s->GetKB(pA ? pA + q0 * (size_t)_K * (size_t)_T : nullptr, pD ? pD + q0 * (size_t)_T : nullptr, s == _sh[0] ? pB : nullptr);
      if (!e.ok()) return e;
    }
    return Error();
  }
  Error FillSynthetic(double nTrain, double noiseAmp, uint64_t seed) override { return All([&](HipEngine &e) { return e.FillSynthetic(nTrain, noiseAmp, seed); }); }
  Error SetTargetGaps(int64_t n, const int64_t *ids) override { return All([&](HipEngine &e) { return e.SetTargetGaps(n, ids); }); }
  Error SetQuestionGaps(int64_t n, const int64_t *ids) override {
    std::lock_guard<OpMutex> lk(_opMu);
    { Error fe = FlushAnswers(); if (!fe.ok()) return fe; }
    Error e = AllLocked([&](HipEngine &eng) { return eng.SetQuestionGaps(n, ids); });
    if (e.ok())
      for (int64_t i = 0; i < n; i++)
        if (std::find(_qGapList.begin(), _qGapList.end(), ids[i]) == _qGapList.end()) { _qGapList.push_back(ids[i]); _questionIds.Vacate(ids[i]); }
    RefreshGapBits();
    return e;
  }
  Error EvalPriorities(int64_t iQuiz, double *pOut, int64_t n) override {
    if (n != _Q) return Error::MakeP(ErrCode::IndexOutOfRange, "n=" + std::to_string(n), "Priority buffer length must equal the question count.");
    std::lock_guard<OpMutex> lk(_opMu);
    Error e = FlushAnswers();
    if (!e.ok()) return e;
    for (auto &s : _sh) { e = s->EvalPriorities(iQuiz, pOut + s->FirstQuestion(), s->LocalQuestions()); if (!e.ok()) return e; }
    return Error();
  }
  int64_t NextQuestionArgmax(Error &err, int64_t iQuiz) override { return Combine(err, iQuiz, 0, 0); }
  int64_t NextQuestionSampled(Error &err, int64_t iQuiz, uint64_t rnd) override { return Combine(err, iQuiz, 1, rnd); }
  Error GetPriors(int64_t iQuiz, double *pOut, int64_t n) override {
    Error e = EnsureApplied(iQuiz);
    if (!e.ok()) return e;
    Touch(iQuiz);
    return _sh[0]->GetPriors(iQuiz, pOut, n);
  }
  Error NextQuestionArgmaxBatch(int64_t n, const int64_t *pQuizzes, int64_t *pOut) override;
  Error EvalPrioritiesBatch(int64_t n, const int64_t *pQuizzes, double *pOut) override;
  Error SelectArgmaxBatch(int64_t n, const int64_t *pQuizzes, CiHipSelection *pOut) override;
  Error Log2HotArray(const double *pIn, double *pOut, int64_t n) override { return _sh[0]->Log2HotArray(pIn, pOut, n); }
  hipStream_t GetStream() const override { return _sh[0]->GetStream(); }
  Error SetStream(hipStream_t) override { return NotSharded("SetStream"); }
  Error Synchronize() override {
    std::lock_guard<OpMutex> lk(_opMu);
    Error e = FlushAnswers();
    if (!e.ok()) return e;
    return AllLocked([&](HipEngine &sh) { return sh.Synchronize(); });
  }
  Error Quiesce() override {
    std::lock_guard<OpMutex> lk(_opMu);
    Error e = FlushAnswers();
    if (!e.ok()) return e;
    return AllLocked([&](HipEngine &sh) { return sh.Quiesce(); });
  }
  Error EnqueueSelectArgmax(int64_t, void *) override { return NotSharded("EnqueueSelectArgmax"); }
  Error EnqueueSelectArgmaxFlag(int64_t, void *, void *, uint64_t) override { return NotSharded("EnqueueSelectArgmaxFlag"); }
  Error EnqueueEval(int64_t iQuiz) override {
    std::lock_guard<OpMutex> lk(_opMu);
    Error e = FlushAnswers();
    if (!e.ok()) return e;
    return AllLocked([&](HipEngine &sh) { return sh.EnqueueEval(iQuiz); });
  }
  Error GetPriorDevicePtr(int64_t iQuiz, void **ppDev, int64_t *pLdT) override {
    Error e = EnsureApplied(iQuiz);
    if (!e.ok()) return e;
    return _sh[0]->GetPriorDevicePtr(iQuiz, ppDev, pLdT);
  }
  Error RecordAnswerRemote(int64_t, int64_t) override { return NotSharded("RecordAnswerRemote"); }
  Error RecordAnswerBatch(int64_t n, const int64_t *pQuizzes, const int64_t *pAnswers) override {   // (gathered like any other answers: one hand-over to the shards)
    if (n > 0 && (!pQuizzes || !pAnswers)) return Error::Make(ErrCode::NullArgument, "Nullptr is passed in place of a batch buffer.");
    for (int64_t i = 0; i < n; i++) { Error e = RecordAnswerDeferred(pQuizzes[i], pAnswers[i]); if (!e.ok()) return e; }
    return FlushNow();
  }
  Error ListTopTargetsBatch(int64_t n, const int64_t *pQuizzes, int64_t maxCount, CiRatedTarget *pDest, int64_t *pCounts) override {
    // every shard holds every quiz's whole posterior: the gathered answers reach the shards, then one of them lists
    Error e = FlushNow();
    if (!e.ok()) return e;
    return _sh[0]->ListTopTargetsBatch(n, pQuizzes, maxCount, pDest, pCounts);
  }
  Error StartQuizBatch(int64_t n, int64_t *pQuizzes) override {
    if (n > 0 && !pQuizzes) return Error::Make(ErrCode::NullArgument, "Nullptr is passed in place of a batch buffer.");
    for (int64_t i = 0; i < n; i++) {
      Error e;
      pQuizzes[i] = StartQuiz(e);
      if (pQuizzes[i] < 0) { for (int64_t j = 0; j < i; j++) (void)ReleaseQuiz(pQuizzes[j]); return e; }
    }
    return Error();
  }

 private:
  ShardedEngine() { _opMu.owner = this; }

  // ---- the engine's lock and its posted operations ---------------------------------------------------------------------------
  struct SelRequest;
  struct Flight;
  struct BatchCtx;
  struct Op {
    int kind = 0;                      // 1 StartQuiz (result = the quiz), 2 ReleaseQuiz(iQuiz), 3 RecordQuizTarget(iQuiz, iTarget, amount),
                                       // 4 hand the gathered answers to the shards, 5 a leader's LaunchBatch(ctx, batch, flight)
    int64_t iQuiz = -1, iTarget = -1;
    double amount = 0;
    Error err;
    int64_t result = -1;
    int ctx = 0;
    std::vector<SelRequest *> *batch = nullptr;
    Flight *flight = nullptr;
    std::atomic<int> state{0};         // 0 posted, 2 posted and its thread asleep on this word, 1 done
    Op *next = nullptr;
  };
  struct OpMutex {                     // (a sleeping lock; releasing it runs whatever was posted meanwhile)
    std::mutex m;
    ShardedEngine *owner = nullptr;
    bool try_lock() { return m.try_lock(); }
    void lock() { m.lock(); }
    void unlock();
  };
  mutable OpMutex _opMu;
  std::atomic<Op *> _posted{nullptr};
  std::vector<std::atomic<int> *> _wake;   // the drain's sleepers, woken once the lock is released
  void RunOp(Op &op);                  // runs it under the lock -- here, or by the lock's holder on its way out
  void Drain();                        // (the lock held)
  void Execute(Op *ordered);

  template <typename F>
  Error All(F &&f) {   // (the gathered answers first: what follows reads or changes what they read)
    std::lock_guard<OpMutex> lk(_opMu);
    Error fe = FlushAnswers();
    return fe.ok() ? AllLocked(f) : fe;
  }
  template <typename F>
  Error AllLocked(F &&f) { for (auto &s : _sh) { Error e = f(*s); if (!e.ok()) return e; } return Error(); }
  int OwnerOf(int64_t qGlobal) const {
    for (size_t s = 0; s < _sh.size(); s++) if (_sh[s]->OwnsQuestion(qGlobal)) return (int)s;
    return -1;
  }

  // ---- per-quiz state kept HERE (the shards keep the posterior, the asked bits and the answers): last use (BaseQuiz::OnUsage,
  // BaseEngine.cpp:417 -- ClearOldQuizzes is decided once for all shards), the active question (CEQuiz::_activeQuestion) and
  // whether an answer of the quiz is still among the gathered ones.  Read and written by the quiz's own client without a lock
  // (no concurrent calls on one quiz, IPqaEngine.h:44): a table of chunks that only grows, under _opMu.
  // failed: a hand-over of this quiz's answer did not reach every shard (FlushAnswers): the posterior replicas may differ, and every
  // later call on the quiz says so instead of selecting from one of them; released like any other quiz
  struct QuizRow { std::atomic<int64_t> lastUse{0}, active{-1}; std::atomic<int> pending{0}, failed{0}; };
  static constexpr int64_t kRowsPerChunk = 4096, kRowChunks = 8192;
  std::atomic<QuizRow *> _rows[kRowChunks] = {};
  std::atomic<int64_t> _registrySize{0};       // ids below this have been handed out at some time
  QuizRow *Row(int64_t id) const {
    if (id < 0 || id >= kRowsPerChunk * kRowChunks) return nullptr;
    QuizRow *c = _rows[id / kRowsPerChunk].load(std::memory_order_acquire);
    return c ? c + id % kRowsPerChunk : nullptr;
  }
  QuizRow *LiveRow(int64_t id) const { QuizRow *r = Row(id); return r && r->lastUse.load(std::memory_order_relaxed) != 0 ? r : nullptr; }
  QuizRow *EnsureRow(int64_t id) {             // (_opMu held)
    if (id < 0 || id >= kRowsPerChunk * kRowChunks) return nullptr;
    QuizRow *c = _rows[id / kRowsPerChunk].load(std::memory_order_acquire);
    if (!c) { c = new QuizRow[kRowsPerChunk]; _rows[id / kRowsPerChunk].store(c, std::memory_order_release); }
    if (id >= _registrySize.load(std::memory_order_relaxed)) _registrySize.store(id + 1, std::memory_order_relaxed);
    return c + id % kRowsPerChunk;
  }
  void Touch(int64_t iQuiz) { if (QuizRow *r = LiveRow(iQuiz)) r->lastUse.store((int64_t)NowStamp(), std::memory_order_relaxed); }
  static int64_t NowStamp() { const time_t t = time(nullptr); return t == 0 ? 1 : (int64_t)t; }
  void NewQuiz(int64_t id) { QuizRow *r = EnsureRow(id); if (r) { r->active.store(-1); r->pending.store(0); r->failed.store(0); r->lastUse.store(NowStamp()); } }
  void ForgetQuizzes() {
    { std::lock_guard<std::mutex> lk(_pendMu); _pending.clear(); _pendingCount.store(0); }
    for (int64_t id = 0; id < _registrySize.load(); id++) if (QuizRow *r = Row(id)) { r->lastUse.store(0); r->active.store(-1); r->pending.store(0); r->failed.store(0); }
  }
  // The error a call on quiz `iQuiz` gets when the quiz is not there or the mode is wrong: the shards' own (BaseEngine::UseQuiz,
  // BaseEngine.cpp:399-419; the MaintenanceSwitch gate), asked of shard 0 by a call that changes nothing.
  Error QuizError(int64_t iQuiz) { Error e; (void)_sh[0]->GetActiveQuestionId(e, iQuiz); return e; }

  // ---- the gathered answers ---------------------------------------------------------------------------------------------------
  struct PendingAnswer { int64_t iQuiz, qGlobal, iAnswer; };
  std::mutex _pendMu;
  std::vector<PendingAnswer> _pending;
  std::atomic<size_t> _pendingCount{0};
  std::atomic<int64_t> _answersSinceSweep{0}, _lastCombined{0};
  Error RecordAnswerDeferred(int64_t iQuiz, int64_t iAnswer);
  Error FlushAnswers();                 // (_opMu held) everything gathered so far: one ApplyAnswers per shard
  Error FlushNow() {                    // through the lock: when it returns, every hand-over begun before it has reached all shards
    Op op;
    op.kind = 4;
    RunOp(op);
    return op.err;
  }
  // Before anything reads quiz `iQuiz`'s posterior on a shard: its answer -- if one is among the gathered ones, or in a hand-over
  // another thread is making right now (the quiz's flag falls only after the last shard has it) -- has reached the shards.
  Error FailedQuiz(int64_t iQuiz) const {
    QuizRow *r = LiveRow(iQuiz);
    if (r && r->failed.load(std::memory_order_acquire))
      return Error::MakeP(ErrCode::Internal, "quizId=" + std::to_string(iQuiz), "An answer of this quiz did not reach every shard: its posteriors may differ. Release the quiz.");
    return Error();
  }
  Error EnsureApplied(int64_t iQuiz) {
    QuizRow *r = LiveRow(iQuiz);
    Error e = r && r->pending.load(std::memory_order_acquire) ? FlushNow() : Error();
    return e.ok() ? FailedQuiz(iQuiz) : e;
  }
  std::vector<char> _qGapBit;           // per global question: a gap (mirror of _qGapList for RecordAnswer's check)
  void RefreshGapBits() { _qGapBit.assign((size_t)_Q, 0); for (int64_t g : _qGapList) if (g >= 0 && g < _Q) _qGapBit[(size_t)g] = 1; }

  // ---- other shards' rows: peer access, or staged copies; trainings are ordered around the reads by events -------------------
  bool _peerAll = true, _forceNoPeer = false;
  std::vector<char> _peer;              // [s * N + o]: shard s's device reads shard o's memory in place
  bool InPlace(size_t s, size_t o) const { return !_forceNoPeer && _peer[s * _sh.size() + o]; }
  std::vector<hipEvent_t> _readsDone, _trainDone;   // per shard: behind its reads of other shards' rows / behind its training kernels
  std::vector<char> _remoteReads;       // per shard: it has read other shards' rows since its last _readsDone
  std::vector<uint64_t> _trainEpoch;    // per shard: trainings it has run
  std::vector<uint64_t> _seenTrain;     // [s * N + o]: the training epoch of shard o that shard s's stream is ordered behind
  Error WaitForTraining(size_t s, size_t o);     // before shard s reads shard o's rows
  Error BeforeTraining();               // every shard's reads of other shards' rows so far are ordered before every shard's training
  Error AfterTraining();
  std::atomic<uint64_t> _stagedRows{0}, _trainBarriers{0};

  // ---- one NextQuestion by itself ------------------------------------------------------------------------------------------------
  int64_t SelectArgmaxLocked(Error &err, int64_t iQuiz, double *pPriority);
  int64_t SelectSampledLocked(Error &err, int64_t iQuiz, uint64_t rnd);
  int64_t Commit(Error &err, int64_t iQuiz, int64_t qGlobal);
  uint64_t NextRandom() {   // xorshift128+, the generator family of SRPlatform/Interface/SRFastRandom.h:60-72
    uint64_t s1 = _rng[0];
    const uint64_t s0 = _rng[1];
    _rng[0] = s0;
    s1 ^= s1 << 23;
    _rng[1] = s1 ^ s0 ^ (s1 >> 18) ^ (s0 >> 5);
    return _rng[1] + s0;
  }

  // ---- concurrent NextQuestion calls: combined (as hip_engine_combine.cpp's Combine / ServeQueue / LaunchBatch / CollectBatch) --------
  struct SelRequest {
    int64_t iQuiz = -1;
    int kind = 0;                      // 0 argmax, 1 sampled (rnd)
    uint64_t rnd = 0;
    int64_t result = -1;
    Error err;
    std::atomic<int> state{0};         // 0 waiting, 1 served, 2 lead handed over: serve the queue yourself, 3 select for yourself from `views`
    std::vector<HipEngine::PriorityView> views;   // state 3: per shard, this quiz's priority vector on the host
    std::vector<uint64_t> skip;        // state 3: asked questions and gaps in GLOBAL numbering as the sweeps saw them (64-bit packs)
    BatchCtx *ctx = nullptr;
  };
  struct BatchCtx {
    std::mutex mu;                     // one combined sweep at a time in this context (and the batch calls of the ABI in context 0)
    std::atomic<int> readers{0};       // clients still selecting out of the shards' host buffers of this context
    std::atomic<bool> inFlight{false};
  };
  struct Flight {
    std::vector<SelRequest *> live;    // the requests whose sweep is in flight
    std::vector<HipEngine::CombinedFlight> shard;
    std::vector<std::vector<uint32_t>> unavailable;   // per shard: live.size() x words
    bool anySampled = false;
  };
  BatchCtx _bctx[2];
  int _ctxNext = 0;
  std::mutex _combMu;
  std::vector<SelRequest *> _combQueue;
  bool _leaderActive = false;
  std::mutex _rngMu;
  int64_t _optCombine = 1, _optLingerUs = 20;
  std::atomic<int> _activeCallers{0};
  struct CallScope {
    std::atomic<int> &n;
    explicit CallScope(std::atomic<int> &c) : n(c) { n.fetch_add(1, std::memory_order_relaxed); }
    ~CallScope() { n.fetch_sub(1, std::memory_order_relaxed); }
  };
  bool Concurrent() const { return _optCombine && _activeCallers.load(std::memory_order_relaxed) > 1; }
  int64_t Combine(Error &err, int64_t iQuiz, int kind, uint64_t rnd);
  void ServeQueue(SelRequest *own);
  void LaunchBatch(int ctx, std::vector<SelRequest *> &batch, Flight &f);
  void LaunchBatchLocked(int ctx, std::vector<SelRequest *> &batch, Flight &f);
  bool CollectBatch(int ctx, std::vector<SelRequest *> &batch, Flight &f, SelRequest *own);
  int64_t SelectFromViews(SelRequest *r);
  void ServeAlone(SelRequest *r) {     // (_opMu held)
    r->err = FailedQuiz(r->iQuiz);
    if (!r->err.ok()) { r->result = -1; return; }
    r->result = r->kind == 0 ? SelectArgmaxAlone(r->err, r->iQuiz) : SelectSampledLocked(r->err, r->iQuiz, r->rnd);
  }
  int64_t SelectArgmaxAlone(Error &err, int64_t iQuiz) {
    const int64_t q = SelectArgmaxLocked(err, iQuiz, nullptr);
    if (!err.ok()) return -1;
    return Commit(err, iQuiz, q);
  }
  std::atomic<uint64_t> _combBatches{0}, _combRequests{0}, _combMaxBatch{0}, _postedOps{0}, _answerFlushes{0}, _answersFlushed{0},
      _answerMaxFlush{0}, _startBatches{0};

  std::vector<std::unique_ptr<HipEngine>> _sh;
  int64_t _K = 0, _Q = 0, _T = 0;
  int64_t _select = 0;
  Slot *_slots = nullptr;               // [shards], pinned + mapped: written by the sweeps' finishers, polled here
  uint64_t _step = 0;
  std::vector<double> _hostPriority;
  IdLedger _questionIds;                     // global question ids
  uint64_t _rng[2] = {0x9E3779B97F4A7C15ULL, 0xBF58476D1CE4E5B9ULL};
  void Seed(uint64_t x) {   // SplitMix64 into the two words of the generator, as HipEngine does
    _rng[0] = SplitMix64(x);
    _rng[1] = SplitMix64(x);
  }
  int64_t _shardsInFlightMax = 0;
  // ---- maintenance-mode edits of the dimensions (CpuEngine.cpp:468-658, BaseEngine.cpp:721-873): the ids are worked out HERE,
  // over the global question axis, exactly as the unsharded engine works them out; the data moves by REBUILDING the shards --
  // new shards over SRPoolRunner::CalcSplit of the new question count, every new question taking its rows from wherever the old
  // shards hold them (in place over peer access, columns picked by the target map), all-or-nothing (the old shards stay until
  // the new ones are complete).
  std::vector<int64_t> _qGapList;           // global question gaps, LIFO like PqaCore/GapTracker.h (the shards keep bitmaps of their own)
  std::vector<int> _devices;
  Error Rebuild(int64_t newQ, int64_t newT, const std::vector<int64_t> &srcQ, const std::vector<int64_t> &srcT,
                const std::vector<int64_t> &qGaps, const std::vector<int64_t> &tGaps, const IdLedger &targetIds,
                const std::vector<int64_t> &fillT, const std::vector<double> &fillTInit, const std::vector<int64_t> &fillQ,
                const std::vector<double> &fillQInit);
  Error MaintenanceOnly(const char *what) const {
    if (_sh[0]->IsMaintenanceMode()) return Error();
    return Error::Make(ErrCode::WrongMode, std::string("Can't perform maintenance-only mode operation - ") + what +
                                               " - because current mode is not maintenance (but regular/shutdown?).");
  }
  Error InitCrossShard();               // peer access between the devices, the ordering events
  void AdoptShards();                   // what every (re)built set of shards is told
  void ReleaseEverywhere(int64_t iQuiz, size_t nShards) {   // roll a partly created quiz back
    for (size_t s = 0; s < nShards; s++) (void)_sh[s]->ReleaseQuiz(iQuiz);
  }
};

ShardedEngine::~ShardedEngine() {
  for (size_t s = 0; s < _sh.size(); s++) {
    hipSetDevice(_sh[s]->Device());
    if (s < _readsDone.size() && _readsDone[s]) hipEventDestroy(_readsDone[s]);
    if (s < _trainDone.size() && _trainDone[s]) hipEventDestroy(_trainDone[s]);
  }
  _sh.clear();
  if (_slots) hipHostFree(_slots);
  for (auto &c : _rows) delete[] c.load();
}

// ---- the lock ---------------------------------------------------------------------------------------------------------------------
void ShardedEngine::OpMutex::unlock() {
  for (;;) {
    std::vector<std::atomic<int> *> wake;
    if (owner != nullptr && owner->_posted.load(std::memory_order_acquire) != nullptr) {
      owner->Drain();
      wake.swap(owner->_wake);
    }
    m.unlock();
    for (std::atomic<int> *w : wake) FutexWakeOne(w);
    // posted between the drain and the release: its thread saw the lock taken and waits.  (Post then try_lock there, release then
    // this load here: one of the two sees the other.)  If somebody else has the lock by now, the operation is theirs to run.
    if (owner == nullptr || owner->_posted.load(std::memory_order_seq_cst) == nullptr || !m.try_lock()) return;
  }
}

void ShardedEngine::RunOp(Op &op) {
  if (_opMu.try_lock()) {            // free: run it here (and whatever else has been posted, on the way out)
    op.next = nullptr;
    Execute(&op);
    _opMu.unlock();
    return;
  }
  _postedOps.fetch_add(1, std::memory_order_relaxed);
  Op *head = _posted.load(std::memory_order_relaxed);
  do op.next = head; while (!_posted.compare_exchange_weak(head, &op, std::memory_order_seq_cst, std::memory_order_relaxed));
  for (;;) {
    if (_opMu.try_lock()) _opMu.unlock();   // (free after all: the release runs it)
    for (int spins = 0; spins < 300; spins++) {
      if (op.state.load(std::memory_order_acquire) == 1) return;
      _mm_pause();
    }
    int expected = 0;
    if (op.state.compare_exchange_strong(expected, 2, std::memory_order_seq_cst) || expected == 2) {
      struct timespec ts{0, 1000000};   // (a millisecond, then the lock is tried again: a belt to the braces above)
      syscall(SYS_futex, reinterpret_cast<int *>(&op.state), FUTEX_WAIT_PRIVATE, 2, &ts, nullptr, 0);
    }
    if (op.state.load(std::memory_order_acquire) == 1) return;
  }
}

void ShardedEngine::Drain() {
  Op *list = _posted.exchange(nullptr, std::memory_order_acq_rel);
  if (list == nullptr) return;
  Op *ordered = nullptr;
  while (list != nullptr) { Op *n = list->next; list->next = ordered; ordered = list; list = n; }   // the order they were posted in
  Execute(ordered);
  for (Op *op = ordered; op != nullptr;) {
    Op *const next = op->next;   // (the operation is its thread's again the moment its state says so)
    std::atomic<int> *word = &op->state;
    if (word->exchange(1, std::memory_order_acq_rel) == 2) _wake.push_back(word);
    op = next;
  }
}

// A list of operations under the lock (one, from a caller that found the lock free; or everything posted so far).  All of them are
// concurrent calls, so any order among them is a valid one: first the gathered answers go to the shards (whatever follows reads
// posteriors or the quizzes' answer lists), then the registry changes in their order, the quiz starts in ONE launch per shard, the
// trainings between their two barriers, and last the combined sweeps -- they are what the most clients wait for.
void ShardedEngine::Execute(Op *ordered) {
  Error flushErr = FlushAnswers();
  int64_t nStarts = 0, nTrains = 0;
  for (Op *op = ordered; op != nullptr; op = op->next) {
    if (op->kind == 4) { op->err = flushErr; continue; }
    if (op->kind == 1) { nStarts++; continue; }
    if (op->kind == 3) { nTrains++; continue; }
    if (op->kind == 2) {   // ReleaseQuiz: every shard releases (a shard that has not got the quiz says so): the registries stay in step
      Error first;
      size_t released = 0;
      for (auto &s : _sh) { Error e = s->ReleaseQuiz(op->iQuiz); if (e.ok()) released++; else if (first.ok()) first = e; }
      // (some shards released and one refused -- a NextQuestion of the quiz selecting on another thread, the client's own error: the
      //  refusing shards are asked again until they agree, so that every shard's registry holds the same quizzes)
      for (int tries = 0; !first.ok() && released > 0 && released < _sh.size() && tries < 2000; tries++) {
        first = Error();
        released = 0;
        for (auto &s : _sh) {
          Error qe;
          (void)s->GetActiveQuestionId(qe, op->iQuiz);
          if (!qe.ok()) { released++; continue; }              // (this shard has let it go already)
          Error e = s->ReleaseQuiz(op->iQuiz);
          if (e.ok()) released++; else if (first.ok()) first = e;
        }
        if (!first.ok()) { struct timespec ts{0, 50000}; nanosleep(&ts, nullptr); }
      }
      if (first.ok()) if (QuizRow *r = Row(op->iQuiz)) { r->lastUse.store(0); r->active.store(-1); r->pending.store(0); r->failed.store(0); }
      op->err = first;
    }
  }
  if (nStarts > 0) {
    // the StartQuiz calls that arrived together: ONE launch per shard sets all their priors (prior_kernels.hip: start_quiz_batch_kernel)
    std::vector<Op *> starts;
    for (Op *op = ordered; op != nullptr; op = op->next) if (op->kind == 1) starts.push_back(op);
    const int64_t k = (int64_t)starts.size();
    std::vector<int64_t> ids((size_t)k), got((size_t)k);
    Error err;
    size_t done = 0;
    for (; done < _sh.size() && err.ok(); done++) {
      std::vector<int64_t> &dst = done == 0 ? ids : got;
      err = _sh[done]->StartQuizBatch(k, dst.data());
      if (err.ok() && done > 0 && got != ids) {
        for (int64_t id : got) (void)_sh[done]->ReleaseQuiz(id);
        err = Error::Make(ErrCode::Internal, "The shards' quiz registries have diverged.");
      }
      if (!err.ok()) break;
    }
    if (!err.ok()) {
      for (size_t s = 0; s < done; s++) for (int64_t id : ids) (void)_sh[s]->ReleaseQuiz(id);   // all or nothing
      for (Op *op : starts) { op->err = err; op->result = -1; }
    } else {
      for (int64_t i = 0; i < k; i++) { NewQuiz(ids[(size_t)i]); starts[(size_t)i]->result = ids[(size_t)i]; }
      if (k > 1) _startBatches.fetch_add(1, std::memory_order_relaxed);
    }
  }
  if (nTrains > 0) {
    // RecordQuizTarget (BaseEngine.cpp:529-566, CpuEngine.cpp:442-466): every shard validates before any trains -- a gap question
    // owned by shard k must not leave shards 0..k-1 trained and their vB replicas ahead -- then every shard applies the steps on
    // its own questions; the other shards' reads of those rows are ordered around the trainings by events
    std::vector<Op *> valid;
    for (Op *op = ordered; op != nullptr; op = op->next) {
      if (op->kind != 3) continue;
      op->err = flushErr;
      if (op->err.ok() && op->amount > 0)
        for (auto &s : _sh) { op->err = s->ValidateTrain(0, nullptr, op->iTarget, op->iQuiz); if (!op->err.ok()) break; }
      if (op->err.ok()) valid.push_back(op);
    }
    if (!valid.empty()) {
      Error be = BeforeTraining();
      for (Op *op : valid) {
        op->err = be;
        if (!op->err.ok()) continue;
        Touch(op->iQuiz);
        for (auto &s : _sh) { op->err = s->RecordQuizTarget(op->iQuiz, op->iTarget, op->amount); if (!op->err.ok()) break; }
      }
      Error ae = AfterTraining();
      if (!ae.ok()) for (Op *op : valid) if (op->err.ok()) op->err = ae;
    }
  }
  for (Op *op = ordered; op != nullptr; op = op->next)
    if (op->kind == 5) LaunchBatchLocked(op->ctx, *op->batch, *op->flight);
}

// ---- creation -------------------------------------------------------------------------------------------------------------------
ShardedEngine *ShardedEngine::Create(Error &err, const CiEngineDefinition &def, const std::vector<int> &devices) {
  std::unique_ptr<ShardedEngine> eng(new ShardedEngine());
  const int64_t N = (int64_t)devices.size();
  if (def._nQuestions < N) {
    err = Error::MakeP(ErrCode::InsufficientEngineDimensions, "[nQuestions=" + std::to_string(def._nQuestions) + " of " + std::to_string(N) + "]",
                       "Fewer questions than devices in PQA_DEVICES.");
    return nullptr;
  }
  eng->_K = def._nAnswers; eng->_Q = def._nQuestions; eng->_T = def._nTargets;
  eng->_devices = devices;
  // SRPoolRunner::CalcSplit (SRPlatform/Interface/SRPoolRunner.h:96-110): the first Q % N shards hold one question more
  const int64_t quot = def._nQuestions / N, rem = def._nQuestions % N;
  int64_t first = 0;
  for (int64_t s = 0; s < N; s++) {
    CiEngineDefinition d = def;
    d._nQuestions = quot + (s < rem ? 1 : 0);
    CiHipShard sh;
    sh._qFirst = first; sh._qTotal = def._nQuestions; sh._device = devices[(size_t)s]; sh._reserved = 0;
    HipEngine *e = HipEngine::Create(err, d, &sh);
    if (!e) return nullptr;
    eng->_sh.emplace_back(e);
    first += d._nQuestions;
  }
  err = eng->InitCrossShard();
  if (!err.ok()) return nullptr;
  if (hipHostMalloc((void **)&eng->_slots, sizeof(Slot) * (size_t)N, hipHostMallocMapped | hipHostMallocCoherent | hipHostMallocPortable) != hipSuccess) {
    err = Error::Make(ErrCode::Internal, "Can't allocate the shards' selection slots.");
    return nullptr;
  }
  std::memset(eng->_slots, 0, sizeof(Slot) * (size_t)N);
  eng->_select = eng->_sh[0]->GetOption("select");
  {   // the selector's generator: from the system's entropy like the reference's (SRFastRandom.h:31-40), or PQA_SEED
    std::random_device rd;
    uint64_t seed = ((uint64_t)rd() << 32) ^ rd();
    if (const char *v = std::getenv("PQA_SEED")) {
      char *end = nullptr;
      const long long x = std::strtoll(v, &end, 10);
      if (end != v && *end == 0) seed = (uint64_t)x;
    }
    eng->Seed(seed);
  }
  eng->_hostPriority.resize((size_t)def._nQuestions);
  eng->_questionIds.Extend(def._nQuestions);
  eng->RefreshGapBits();
  eng->AdoptShards();
  err = Error();
  return eng.release();
}

void ShardedEngine::AdoptShards() {
  for (auto &s : _sh) s->SetExternalCallers(&_activeCallers);
}

// Where the rows of another shard's question are read from.  Peer access is enabled between every pair of distinct devices; a pair
// that cannot map each other's memory (or PQA_FORCE_NO_PEER=1: every pair, a test hook that runs on one GPU) gets staged copies --
// hipMemcpyPeerAsync needs no peer access -- of the two rows an answer needs and of the rows ResumeQuiz reads; what has no staged
// form says so when it is called (the maintenance-mode rebuild of the shards reads whole question blocks in place).  The pair's
// state is decided here, once, and logged: nothing dereferences a pointer the device cannot reach.
Error ShardedEngine::InitCrossShard() {
  const size_t N = _sh.size();
  _peer.assign(N * N, 1);
  _peerAll = true;
  if (const char *v = std::getenv("PQA_FORCE_NO_PEER")) _forceNoPeer = *v && std::strcmp(v, "0") != 0;
  for (size_t a = 0; a < N; a++)
    for (size_t b = 0; b < N; b++) {
      const int da = _sh[a]->Device(), db = _sh[b]->Device();
      if (da == db) continue;
      int can = 0;
      bool ok = hipDeviceCanAccessPeer(&can, da, db) == hipSuccess && can;
      if (ok) {
        hipSetDevice(da);
        const hipError_t e = hipDeviceEnablePeerAccess(db, 0);
        ok = e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled;
        (void)hipGetLastError();
      }
      if (!ok) {
        _peer[a * N + b] = 0;
        _peerAll = false;
        DefaultLogger::Log(DefaultLogger::Severity::Warning, "Device " + std::to_string(da) + " cannot map the memory of device " + std::to_string(db) +
                                                                 ": rows of its questions are copied instead of read in place (PQA_DEVICES).");
      }
    }
  if (_forceNoPeer) _peerAll = false;
  _readsDone.assign(N, nullptr);
  _trainDone.assign(N, nullptr);
  for (size_t s = 0; s < N; s++) {
    hipSetDevice(_sh[s]->Device());
    if (hipEventCreateWithFlags(&_readsDone[s], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&_trainDone[s], hipEventDisableTiming) != hipSuccess)
      return Error::Make(ErrCode::Internal, "Can't create the shards' ordering events.");
  }
  _remoteReads.assign(N, 0);
  _trainEpoch.assign(N, 0);
  _seenTrain.assign(N * N, 0);
  return Error();
}

// Shard s is about to read rows of shard o: not before o's training kernels so far have run.
Error ShardedEngine::WaitForTraining(size_t s, size_t o) {
  const size_t N = _sh.size();
  if (s == o || _seenTrain[s * N + o] == _trainEpoch[o]) return Error();
  hipSetDevice(_sh[s]->Device());
  if (hipStreamWaitEvent(_sh[s]->GetStream(), _trainDone[o], 0) != hipSuccess) return Error::Make(ErrCode::Internal, "hipStreamWaitEvent failed.");
  _seenTrain[s * N + o] = _trainEpoch[o];
  return Error();
}

// A training writes rows that other shards' posterior kernels may still be reading (they were launched before the training call
// arrived, so they must see the rows as they were): every shard that has read other shards' rows since its last event records
// one behind those reads, and every other shard's stream waits for it before its training kernel.
Error ShardedEngine::BeforeTraining() {
  const size_t N = _sh.size();
  bool any = false;
  for (size_t t = 0; t < N; t++) {
    if (!_remoteReads[t]) continue;
    any = true;
    hipSetDevice(_sh[t]->Device());
    if (hipEventRecord(_readsDone[t], _sh[t]->GetStream()) != hipSuccess) return Error::Make(ErrCode::Internal, "hipEventRecord failed.");
  }
  if (!any) return Error();
  _trainBarriers.fetch_add(1, std::memory_order_relaxed);
  for (size_t s = 0; s < N; s++) {
    hipSetDevice(_sh[s]->Device());
    for (size_t t = 0; t < N; t++)
      if (t != s && _remoteReads[t] && hipStreamWaitEvent(_sh[s]->GetStream(), _readsDone[t], 0) != hipSuccess)
        return Error::Make(ErrCode::Internal, "hipStreamWaitEvent failed.");
  }
  std::fill(_remoteReads.begin(), _remoteReads.end(), 0);
  return Error();
}

Error ShardedEngine::AfterTraining() {
  for (size_t s = 0; s < _sh.size(); s++) {
    hipSetDevice(_sh[s]->Device());
    if (hipEventRecord(_trainDone[s], _sh[s]->GetStream()) != hipSuccess) return Error::Make(ErrCode::Internal, "hipEventRecord failed.");
    _trainEpoch[s]++;
  }
  return Error();
}

// ---- quiz registry ----------------------------------------------------------------------------------------------------------------
int64_t ShardedEngine::StartQuiz(Error &err) {
  CallScope scope(_activeCallers);
  Op op;
  op.kind = 1;
  RunOp(op);
  err = op.err;
  return op.result;
}

Error ShardedEngine::ReleaseQuiz(int64_t iQuiz) {
  CallScope scope(_activeCallers);
  Op op;
  op.kind = 2; op.iQuiz = iQuiz;
  RunOp(op);
  return op.err;
}

int64_t ShardedEngine::ResumeQuiz(Error &err, int64_t nAnswered, const AQ *pAQs) {
  if (nAnswered < 0) { err = Error::MakeP(ErrCode::NegativeCount, "count=" + std::to_string(nAnswered), "|nAnswered| must be non-negative."); return -1; }
  if (nAnswered == 0) return StartQuiz(err);   // BaseEngine.cpp:393-395
  if (pAQs == nullptr) { err = Error::Make(ErrCode::NullArgument, "Nullptr is passed in place of answered questions."); return -1; }
  CallScope scope(_activeCallers);
  std::lock_guard<OpMutex> lk(_opMu);
  err = FlushAnswers();
  if (!err.ok()) return -1;
  // every shard computes the posterior itself from row POINTERS: the rows of other shards' questions in place, or staged
  std::vector<const void *> rows(2 * (size_t)nAnswered);
  std::vector<int> rowDev(2 * (size_t)nAnswered), owners((size_t)nAnswered);
  for (int64_t i = 0; i < nAnswered; i++) {
    const int owner = OwnerOf(pAQs[i].iQuestion);
    if (owner < 0) { err = Error::MakeP(ErrCode::IndexOutOfRange, "subjIndex=" + std::to_string(pAQs[i].iQuestion), "Question index is not in KB range."); return -1; }
    err = _sh[(size_t)owner]->GetRowPointers(pAQs[i].iQuestion, pAQs[i].iAnswer, &rows[2 * (size_t)i], &rows[2 * (size_t)i + 1]);
    if (!err.ok()) return -1;
    owners[(size_t)i] = owner;
    rowDev[2 * (size_t)i] = rowDev[2 * (size_t)i + 1] = _sh[(size_t)owner]->Device();
  }
  int64_t id = -1;
  for (size_t s = 0; s < _sh.size(); s++) {
    bool allInPlace = true;
    std::vector<char> stage(2 * (size_t)nAnswered, 0);
    for (int64_t i = 0; i < nAnswered; i++) {
      const size_t o = (size_t)owners[(size_t)i];
      if (o == s) continue;
      err = WaitForTraining(s, o);
      if (!err.ok()) { ReleaseEverywhere(id, s); return -1; }
      if (!InPlace(s, o)) {
        allInPlace = false;
        stage[2 * (size_t)i] = stage[2 * (size_t)i + 1] = 1;
        _stagedRows.fetch_add(2, std::memory_order_relaxed);
      }
    }
    const int64_t got = _sh[s]->ResumeQuizRows(err, nAnswered, pAQs, rows.data(), allInPlace ? nullptr : rowDev.data(), allInPlace ? nullptr : stage.data());
    if (got < 0) { ReleaseEverywhere(id, s); return -1; }   // (e.g. out of memory on shard s: the earlier shards' quiz goes again)
    if (s == 0) id = got;
    else if (got != id) {
      (void)_sh[s]->ReleaseQuiz(got);
      ReleaseEverywhere(id, s);
      err = Error::Make(ErrCode::Internal, "The shards' quiz registries have diverged.");
      return -1;
    }
    // (ResumeQuiz synchronises the shard's stream: its reads of the other shards' rows are done when it returns)
  }
  NewQuiz(id);
  return id;
}

// ClearOldQuizzes (behaviour: BaseEngine.cpp:814-873), decided once for all shards by the rule the one-device engine uses
// (QuizzesToLetGo, hip_engine_kb.cpp) over the usage times kept here.
Error ShardedEngine::ClearOldQuizzes(int64_t maxCount, double maxAgeSec) {
  if (maxCount < 0)
    return Error::MakeP(ErrCode::NegativeCount, "count=" + std::to_string(maxCount), "The number of quizzes to keep cannot be less than 0.");
  std::lock_guard<OpMutex> lk(_opMu);
  if (!_sh[0]->IsRegularMode()) return Error();   // quizzes are not expected to exist in maintenance / shutdown mode
  Error first = FlushAnswers();
  std::vector<QuizUsage> inUse;
  for (int64_t id = 0; id < _registrySize.load(); id++)
    if (QuizRow *r = LiveRow(id)) inUse.push_back(QuizUsage{id, (time_t)r->lastUse.load()});   // (registry order)
  for (int64_t id : QuizzesToLetGo(inUse, time(nullptr), maxCount, maxAgeSec)) {
    if (QuizRow *r = Row(id)) { r->lastUse.store(0); r->active.store(-1); r->pending.store(0); }
    for (auto &s : _sh) { Error e = s->ReleaseQuiz(id); if (!e.ok() && first.ok()) first = e; }
  }
  return first;
}

// ---- active question, answers, listings -----------------------------------------------------------------------------------------
int64_t ShardedEngine::GetActiveQuestionId(Error &err, int64_t iQuiz) {
  QuizRow *r = _sh[0]->IsRegularMode() ? LiveRow(iQuiz) : nullptr;
  if (!r) { err = QuizError(iQuiz); if (!err.ok()) return -1; r = LiveRow(iQuiz); if (!r) { err = Error::Make(ErrCode::Internal, "The quiz tables have diverged."); return -1; } }
  err = Error();
  r->lastUse.store(NowStamp(), std::memory_order_relaxed);
  return r->active.load(std::memory_order_relaxed);
}

Error ShardedEngine::SetActiveQuestion(int64_t iQuiz, int64_t iQuestion) {
  QuizRow *r = _sh[0]->IsRegularMode() ? LiveRow(iQuiz) : nullptr;
  if (!r) { Error e = QuizError(iQuiz); if (!e.ok()) return e; r = LiveRow(iQuiz); if (!r) return Error::Make(ErrCode::Internal, "The quiz tables have diverged."); }
  r->lastUse.store(NowStamp(), std::memory_order_relaxed);
  r->active.store(iQuestion, std::memory_order_relaxed);   // unchecked, as reference PqaCore/BaseEngine.cpp:507-508
  return Error();
}

// Everything NextQuestion does after the pick (CpuEngine.cpp:403-413): the question becomes the quiz's active question, the
// asked-questions counter moves once.
int64_t ShardedEngine::Commit(Error &err, int64_t iQuiz, int64_t qGlobal) {
  if (qGlobal < 0) { err = Error::Make(ErrCode::QuestionsExhausted, "Found no unasked question that is not in a gap."); return -1; }
  QuizRow *r = LiveRow(iQuiz);
  if (!r) { err = QuizError(iQuiz); return -1; }
  r->active.store(qGlobal, std::memory_order_relaxed);
  _sh[0]->BumpQuestionsAsked(1);
  err = Error();
  return qGlobal;
}

// RecordAnswer (BaseEngine.cpp:441-466, CEQuiz::RecordAnswer PqaCore/CEQuiz.h:77-122): validated and recorded here; the shards
// get it -- with everything else that has gathered -- before the next thing that reads a posterior.
Error ShardedEngine::RecordAnswerDeferred(int64_t iQuiz, int64_t iAnswer) {
  if (!_sh[0]->IsRegularMode()) return QuizError(iQuiz);
  if (iAnswer < 0 || iAnswer >= _K)  // reference PqaCore/BaseEngine.cpp:447-451
    return Error::MakeP(ErrCode::IndexOutOfRange, RangeParams(iAnswer, 0, _K - 1), "Answer index is not in the answer range.");
  QuizRow *r = LiveRow(iQuiz);
  if (!r) { Error e = QuizError(iQuiz); return e.ok() ? Error::Make(ErrCode::Internal, "The quiz tables have diverged.") : e; }
  { Error fe = FailedQuiz(iQuiz); if (!fe.ok()) return fe; }
  r->lastUse.store(NowStamp(), std::memory_order_relaxed);
  const int64_t aq = r->active.load(std::memory_order_relaxed);
  if (aq == -1)
    return Error::MakeP(ErrCode::NoQuizActiveQuestion, "answerId=" + std::to_string(iAnswer),
                        "An attempt to record an answer in a quiz that doesn't have an active question");
  if (aq < 0 || aq >= _Q || _qGapBit[(size_t)aq])
    return Error::MakeP(ErrCode::NoQuizActiveQuestion, "answerId=" + std::to_string(iAnswer),
                        "An attempt to record an answer in a quiz that has invalid active question");
  if (r->pending.load(std::memory_order_acquire)) {   // (a second answer of a quiz whose first is still among the gathered ones)
    Error e = FlushNow();
    if (!e.ok()) return e;
  }
  r->active.store(-1, std::memory_order_relaxed);
  r->pending.store(1, std::memory_order_release);
  {
    std::lock_guard<std::mutex> lk(_pendMu);
    _pending.push_back(PendingAnswer{iQuiz, aq, iAnswer});
    _pendingCount.store(_pending.size(), std::memory_order_release);
  }
  return Error();
}

Error ShardedEngine::RecordAnswer(int64_t iQuiz, int64_t iAnswer) {
  CallScope scope(_activeCallers);
  Error e = RecordAnswerDeferred(iQuiz, iAnswer);
  if (!e.ok()) return e;
  // Alone in the engine: the shards' kernels start now, under whatever the client does next.  Other clients inside: the answer
  // waits for the next call that needs a posterior -- which hands over all that have gathered by then, in one launch per shard.
  return Concurrent() ? Error() : FlushNow();
}

Error ShardedEngine::FlushAnswers() {
  std::vector<PendingAnswer> taken;
  {
    std::lock_guard<std::mutex> lk(_pendMu);
    taken.swap(_pending);
    _pendingCount.store(0, std::memory_order_release);
  }
  if (taken.empty()) return Error();
  const size_t N = _sh.size();
  _answerFlushes.fetch_add(1, std::memory_order_relaxed);
  _answersFlushed.fetch_add(taken.size(), std::memory_order_relaxed);
  _answersSinceSweep.fetch_add((int64_t)taken.size(), std::memory_order_relaxed);
  if (taken.size() > _answerMaxFlush.load(std::memory_order_relaxed)) _answerMaxFlush.store(taken.size(), std::memory_order_relaxed);
  std::vector<HipEngine::ShardAnswer> forShard(taken.size());
  Error first;
  for (size_t s = 0; s < N; s++) {
    for (size_t i = 0; i < taken.size(); i++) {
      const PendingAnswer &p = taken[i];
      const int owner = OwnerOf(p.qGlobal);
      HipEngine::ShardAnswer &a = forShard[i];
      a = HipEngine::ShardAnswer{p.iQuiz, p.qGlobal, p.iAnswer, nullptr, nullptr, -1, false, (size_t)(p.iQuiz % (int64_t)N) == s};
      if ((size_t)owner == s) continue;
      HipEngine &o = *_sh[(size_t)owner];
      a.rowA = o.RowPointer(p.qGlobal - o.FirstQuestion(), p.iAnswer);
      a.rowD = o.RowPointer(p.qGlobal - o.FirstQuestion(), _K);
      a.srcDevice = o.Device();
      a.stage = !InPlace(s, (size_t)owner);
      if (a.stage) _stagedRows.fetch_add(2, std::memory_order_relaxed);
      Error we = WaitForTraining(s, (size_t)owner);
      if (!we.ok() && first.ok()) first = we;
      _remoteReads[s] = 1;
    }
    Error e = _sh[s]->ApplyAnswers((int64_t)forShard.size(), forShard.data());
    if (!e.ok() && first.ok()) first = e;
  }
  // (a failure on any shard: which of the batch's answers that shard still applied is not known here -- every quiz of the batch is
  //  marked, and says so from now on)
  for (const PendingAnswer &p : taken)
    if (QuizRow *r = Row(p.iQuiz)) {
      if (!first.ok()) r->failed.store(1, std::memory_order_release);
      r->pending.store(0, std::memory_order_release);
    }
  return first;
}

int64_t ShardedEngine::ListTopTargets(Error &err, int64_t iQuiz, int64_t maxCount, CiRatedTarget *pDest) {
  CallScope scope(_activeCallers);
  QuizRow *r = LiveRow(iQuiz);
  if (r && r->pending.load(std::memory_order_acquire)) {
    // Group commit.  This quiz's answer is among the gathered ones, and the clients that got their questions from the same combined
    // sweep are recording theirs right now: a moment for them, so that ONE hand-over carries all of them.
    if (_optLingerUs > 0 && Concurrent()) {
      const size_t target = (size_t)std::max<int64_t>(2, std::min<int64_t>(_lastCombined.load(std::memory_order_relaxed), _activeCallers.load(std::memory_order_relaxed) - 1));
      const auto t0 = std::chrono::steady_clock::now();
      const auto limit = std::chrono::microseconds(_optLingerUs);
      for (;;) {
        const size_t have = _pendingCount.load(std::memory_order_relaxed);
        if (have == 0 || have >= target) break;   // (0: somebody has handed them over)
        for (int i = 0; i < 32; i++) _mm_pause();
        if (std::chrono::steady_clock::now() - t0 > limit) break;
      }
    }
    err = FlushNow();
    if (!err.ok()) return -1;
  }
  if (r) r->lastUse.store(NowStamp(), std::memory_order_relaxed);
  // (the shard whose kernel listed the new posterior's best targets with the update: FlushAnswers)
  return _sh[(size_t)(iQuiz >= 0 ? iQuiz % (int64_t)_sh.size() : 0)]->ListTopTargets(err, iQuiz, maxCount, pDest);
}

// ---- training -------------------------------------------------------------------------------------------------------------------
// The reference validates every answered question before any Add subtask runs (CETrainSubtaskDistrib.h:26-45): a gap question
// owned by shard k must not leave shards 0..k-1 trained and their vB replicas ahead -- every shard validates, then every shard trains.
Error ShardedEngine::Train(int64_t n, const AQ *pAQs, int64_t iTarget, double amount) {
  std::lock_guard<OpMutex> lk(_opMu);
  Error e = FlushAnswers();
  if (!e.ok()) return e;
  if (n >= 0 && amount > 0 && (n == 0 || pAQs != nullptr))   // (else: shard 0 produces the reference's argument error, before any kernel)
    for (auto &s : _sh) { e = s->ValidateTrain(n, pAQs, iTarget, -1); if (!e.ok()) return e; }
  e = BeforeTraining();
  if (!e.ok()) return e;
  for (auto &s : _sh) { e = s->Train(n, pAQs, iTarget, amount); if (!e.ok()) break; }
  Error ae = AfterTraining();
  return e.ok() ? ae : e;
}

Error ShardedEngine::RecordQuizTarget(int64_t iQuiz, int64_t iTarget, double amount) {
  CallScope scope(_activeCallers);
  Op op;
  op.kind = 3; op.iQuiz = iQuiz; op.iTarget = iTarget; op.amount = amount;
  RunOp(op);
  return op.err;
}

// ---- one NextQuestion by itself (_opMu held through the wait: nobody else is asking) ------------------------------------------------
int64_t ShardedEngine::SelectArgmaxLocked(Error &err, int64_t iQuiz, double *pPriority) {
  if (++_step == 0) ++_step;
  const uint64_t step = _step;
  for (size_t s = 0; s < _sh.size(); s++) {
    // the slots are mapped + portable: their host address is what every device sees
    err = _sh[s]->EnqueueSelectArgmaxFlag(iQuiz, &_slots[s].priority, &_slots[s].flag, step);
    if (!err.ok()) return -1;
  }
  Touch(iQuiz);
  SpinWait w;
  for (size_t s = 0; s < _sh.size(); s++) {
    volatile uint64_t *flag = &_slots[s].flag;
    while (*flag != step)
      if (!w.Tick(std::chrono::seconds(60))) {
        err = Error::MakeP(ErrCode::Internal, "shard=" + std::to_string(s), "Timed out waiting for a shard's selection.");
        return -1;
      }
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  double bestP = 0;
  int64_t bestI = -1;
  for (size_t s = 0; s < _sh.size(); s++) {
    double p = _slots[s].priority;
    const int64_t i = _slots[s].index;
    if (i == -3) { err = Error::Make(ErrCode::Internal, "A shard's sweep did not complete."); return -1; }
    if (i < 0) continue;
    if (p != p) p = -HUGE_VAL;
    if (bestI < 0 || p > bestP || (p == bestP && i < bestI)) { bestP = p; bestI = i; }
  }
  CheckPriority(bestP, bestI);   // (the reference's "Got priority=" warning, for the question that was selected)
  if (pPriority) *pPriority = bestP;
  err = Error();
  return bestI;
}

// The reference's selector (PqaCore/CpuEngine.cpp:362-400) over the GLOBAL question range: the same per-subtask Kahan run
// lengths, grand totals and upper_bounds as select_sampled_wg_impl (pqa_device.h) runs on one device, here on the host over the
// shards' priority vectors -- with the same priorities, subtask count and random number it picks the question an unsharded
// engine picks.
int64_t ShardedEngine::SelectSampledLocked(Error &err, int64_t iQuiz, uint64_t rnd) {
  Touch(iQuiz);
  for (auto &s : _sh) { err = s->EnqueueEval(iQuiz); if (!err.ok()) return -1; }
  for (auto &s : _sh) {
    hipSetDevice(s->Device());
    if (hipMemcpyAsync(_hostPriority.data() + s->FirstQuestion(), s->PriorityDevicePtr(), (size_t)s->LocalQuestions() * sizeof(double),
                       hipMemcpyDeviceToHost, s->GetStream()) != hipSuccess) { err = Error::Make(ErrCode::Internal, "Copy of a shard's priorities failed."); return -1; }
  }
  for (auto &s : _sh) {
    hipSetDevice(s->Device());
    if (hipStreamSynchronize(s->GetStream()) != hipSuccess) { err = Error::Make(ErrCode::Internal, "A shard's sweep failed."); return -1; }
  }
  // which questions are asked or gaps, in global numbering (the shards' ranges are not multiples of 32)
  std::vector<uint64_t> skip((size_t)((_Q + 63) / 64) + 1, 0);
  {
    std::vector<uint32_t> words;
    for (auto &s : _sh) {
      err = s->UnavailableWords(iQuiz, words);
      if (!err.ok()) return -1;
      for (int64_t i = 0; i < s->LocalQuestions(); i++)
        if ((words[(size_t)(i >> 5)] >> (i & 31)) & 1u) skip[(size_t)((s->FirstQuestion() + i) >> 6)] |= 1ULL << ((s->FirstQuestion() + i) & 63);
    }
    for (int64_t q = _Q; q < (int64_t)skip.size() * 64; q++) skip[(size_t)(q >> 6)] |= 1ULL << (q & 63);
  }
  auto skipped = [&](int64_t q) { return (skip[(size_t)(q >> 6)] >> (q & 63)) & 1ULL; };
  const int64_t n = _Q;
  int64_t sel = SelectSampledHost(_hostPriority.data(), n, _sh[0]->GetOption("eval_subtasks"), rnd, [&](int64_t q) { return skipped(q) != 0; });
  // :403-407 a gap / asked pick falls to BaseEngine::FindNearestQuestion, over the global bitmap
  if (skipped(sel)) sel = FindNearestInPacks(sel, n, [&](int64_t p) { return ~skip[(size_t)p]; });
  return Commit(err, iQuiz, sel);
}

int64_t ShardedEngine::NextQuestion(Error &err, int64_t iQuiz) {
  if (_select == 1) return Combine(err, iQuiz, 0, 0);
  uint64_t rnd;
  { std::lock_guard<std::mutex> lk(_rngMu); rnd = NextRandom(); }   // (drawn when the call arrives, whatever sweep serves it)
  return Combine(err, iQuiz, 1, rnd);
}

// ---- concurrent NextQuestion calls ------------------------------------------------------------------------------------------------
// A caller posts its request; if a leader is at work it waits for its result, otherwise it becomes the leader: it takes everything
// posted so far -- distinct quizzes -- launches ONE batched sweep per shard for it (every shard's on its own device and stream, all
// in flight before the first is waited for), hands the lead to the oldest request still waiting as soon as the sweeps are
// launched, and then waits for its own.  Two batch contexts alternate, so that the next leader launches while this one's sweeps
// run.  One request alone takes the single-quiz path (the shards' fused argmax through the pinned slots).
int64_t ShardedEngine::Combine(Error &err, int64_t iQuiz, int kind, uint64_t rnd) {
  CallScope scope(_activeCallers);
  SelRequest r;
  r.iQuiz = iQuiz; r.kind = kind; r.rnd = rnd;
  if (!_optCombine) {
    std::lock_guard<OpMutex> lk(_opMu);
    err = FlushAnswers();
    if (!err.ok()) return -1;
    ServeAlone(&r);
    err = r.err;
    return r.result;
  }
  bool lead;
  {
    std::lock_guard<std::mutex> lk(_combMu);
    _combQueue.push_back(&r);
    lead = !_leaderActive;
    if (lead) _leaderActive = true;
  }
  if (!lead) {
    int st = 0;
    for (int spins = 0; spins < 1500 && (st = r.state.load(std::memory_order_acquire)) == 0; spins++) _mm_pause();
    while (st == 0) {
      FutexWait(&r.state, 0);   // (returns at once if the state is no longer 0)
      st = r.state.load(std::memory_order_acquire);
    }
    if (st == 1) { err = r.err; return r.result; }
    if (st == 3) {   // the sweeps have run: this quiz's priorities are on the host, the selection is this thread's own work
      const int64_t sel = SelectFromViews(&r);
      r.ctx->readers.fetch_sub(1, std::memory_order_release);
      err = r.err;
      return sel;
    }
    // (2: the leader before has launched its batch and handed the lead to this, the oldest waiting request)
  }
  ServeQueue(&r);
  err = r.err;
  return r.result;
}

void ShardedEngine::ServeQueue(SelRequest *own) {
  // The clients whose answers were handed over since the last combined sweep are on their way here: a leader that starts at once
  // sweeps for the two or three that were quickest and makes the rest wait for a second sweep.  So it waits -- microseconds --
  // until most of them have posted, or nobody new comes; while the previous leader's sweeps still run there is no hurry at all.
  if (_optLingerUs > 0 && Concurrent()) {
    const int64_t expect = std::min<int64_t>(_answersSinceSweep.load(std::memory_order_relaxed), _activeCallers.load(std::memory_order_relaxed) - 1);
    const BatchCtx &other = _bctx[_ctxNext ^ 1];
    const auto t0 = std::chrono::steady_clock::now();
    const auto limit = std::chrono::microseconds(_optLingerUs), limitBusy = std::chrono::microseconds(8 * _optLingerUs);
    for (;;) {
      size_t have;
      { std::lock_guard<std::mutex> lk(_combMu); have = _combQueue.size(); }
      const bool busy = other.inFlight.load(std::memory_order_relaxed);
      if (!busy && (expect <= 1 || (int64_t)have * 5 >= expect * 4)) break;
      if (busy && (int64_t)have >= _activeCallers.load(std::memory_order_relaxed) - 1) break;   // (everybody is here)
      for (int i = 0; i < 32; i++) _mm_pause();
      if (std::chrono::steady_clock::now() - t0 > (busy ? limitBusy : limit)) break;
    }
  }
  const int ctx = _ctxNext;
  _ctxNext ^= 1;
  BatchCtx &c = _bctx[ctx];
  std::unique_lock<std::mutex> ctxLock(c.mu);
  while (c.readers.load(std::memory_order_acquire) != 0) _mm_pause();
  std::vector<SelRequest *> batch;
  {
    std::lock_guard<std::mutex> lk(_combMu);
    std::vector<SelRequest *> rest;
    for (SelRequest *r : _combQueue) {
      bool take = (int64_t)batch.size() < 256;
      for (size_t i = 0; take && i < batch.size(); i++) take = batch[i]->iQuiz != r->iQuiz;   // a quiz once per sweep
      (take ? batch : rest).push_back(r);
    }
    // (the sweeps' lanes come in groups: the newest requests beyond the last well-filled group wait for the next sweep)
    const size_t keep = (size_t)_sh[0]->CombinedBatchFor((int64_t)batch.size());
    if (keep < batch.size()) {
      rest.insert(rest.begin(), batch.begin() + (std::ptrdiff_t)keep, batch.end());
      batch.resize(keep);
    }
    _combQueue.swap(rest);
  }
  Flight f;
  LaunchBatch(ctx, batch, f);   // (under _opMu; what could not be launched has its error -- or its result, for a batch of one)
  {
    std::lock_guard<std::mutex> lk(_combMu);
    if (_combQueue.empty()) _leaderActive = false;
    else PublishState(&_combQueue.front()->state, 2);
  }
  const bool ownSelects = f.live.empty() ? false : CollectBatch(ctx, batch, f, own);
  ctxLock.unlock();
  for (SelRequest *r : batch)
    if (r != nullptr && r != own) PublishState(&r->state, 1);   // (r is its caller's again from here on)
  if (ownSelects) {
    SelectFromViews(own);
    c.readers.fetch_sub(1, std::memory_order_release);
  }
}

void ShardedEngine::LaunchBatch(int ctx, std::vector<SelRequest *> &batch, Flight &f) {
  Op op;
  op.kind = 5; op.ctx = ctx; op.batch = &batch; op.flight = &f;
  RunOp(op);
}

void ShardedEngine::LaunchBatchLocked(int ctx, std::vector<SelRequest *> &batch, Flight &f) {
  _answersSinceSweep.store(0, std::memory_order_relaxed);
  std::vector<SelRequest *> live;
  std::vector<int64_t> ids;
  const bool regular = _sh[0]->IsRegularMode();
  for (SelRequest *r : batch) {
    if (!regular || LiveRow(r->iQuiz) == nullptr) { r->err = QuizError(r->iQuiz); r->result = -1; if (r->err.ok()) r->err = Error::Make(ErrCode::Internal, "The quiz tables have diverged."); continue; }
    { Error fe = FailedQuiz(r->iQuiz); if (!fe.ok()) { r->err = fe; r->result = -1; continue; } }
    live.push_back(r);
    ids.push_back(r->iQuiz);
    f.anySampled = f.anySampled || r->kind == 1;
  }
  if (live.empty()) return;
  if (live.size() == 1) { ServeAlone(live[0]); return; }
  const int64_t n = (int64_t)live.size();
  const size_t N = _sh.size();
  f.shard.assign(N, HipEngine::CombinedFlight());
  f.unavailable.assign(N, std::vector<uint32_t>());
  Error err;
  _shardsInFlightMax = 0;
  for (size_t s = 0; s < N && err.ok(); s++) {
    err = _sh[s]->EnqueueCombined(ctx, n, ids.data(), f.anySampled, &f.shard[s], f.anySampled ? &f.unavailable[s] : nullptr);
    if (err.ok()) _shardsInFlightMax++;
  }
  if (!err.ok()) {
    // (a shard refused the batch -- e.g. a quiz released under its own NextQuestion, the caller's error: every request by itself;
    //  the sweeps already launched run to their end unread)
    for (size_t s = 0; s < N; s++) if (f.shard[s].n > 0) (void)_sh[s]->CollectCombined(ctx, f.shard[s], nullptr, nullptr);
    for (SelRequest *r : live) ServeAlone(r);
    return;
  }
  for (SelRequest *r : live) Touch(r->iQuiz);
  _combBatches.fetch_add(1, std::memory_order_relaxed);
  _combRequests.fetch_add((uint64_t)n, std::memory_order_relaxed);
  if ((uint64_t)n > _combMaxBatch.load(std::memory_order_relaxed)) _combMaxBatch.store((uint64_t)n, std::memory_order_relaxed);
  _lastCombined.store(n, std::memory_order_relaxed);
  _bctx[ctx].inFlight.store(true, std::memory_order_relaxed);
  f.live.swap(live);
}

// Wait for every shard's sweep and hand the results out -- the engine open to the other clients' calls meanwhile.  Returns true if
// `own` is to select for itself.
bool ShardedEngine::CollectBatch(int ctx, std::vector<SelRequest *> &batch, Flight &f, SelRequest *own) {
  const int64_t n = (int64_t)f.live.size();
  const size_t N = _sh.size();
  BatchCtx &c = _bctx[ctx];
  std::vector<std::vector<CiHipSelection>> winners(N);
  std::vector<std::vector<HipEngine::PriorityView>> views(N);
  Error err;
  for (size_t s = 0; s < N; s++) {   // (every shard is waited for, whatever the others reported)
    winners[s].resize((size_t)n);
    views[s].resize((size_t)n);
    Error e = _sh[s]->CollectCombined(ctx, f.shard[s], winners[s].data(), views[s].data());
    if (!e.ok() && err.ok()) err = e;
  }
  c.inFlight.store(false, std::memory_order_relaxed);
  if (!err.ok()) {
    for (SelRequest *r : f.live) { r->err = err; r->result = -1; }
    return false;
  }
  if (f.anySampled) {
    // The priority vectors are on the host: every client selects for ITSELF (the O(Q) scalar Kahan steps of the reference's
    // selector run on as many cores as there are clients), the leader only for its own request.
    c.readers.fetch_add((int)n, std::memory_order_acq_rel);
    const int64_t packs = (_Q + 63) / 64 + 1;
    bool ownLive = false;
    for (int64_t i = 0; i < n; i++) {
      SelRequest *r = f.live[(size_t)i];
      r->views.resize(N);
      r->skip.assign((size_t)packs, 0);
      for (size_t s = 0; s < N; s++) {
        r->views[s] = views[s][(size_t)i];
        const size_t words = _sh[s]->UnavailableWordCount();
        const uint32_t *w = f.unavailable[s].data() + (size_t)i * words;
        const int64_t q0 = _sh[s]->FirstQuestion();
        for (int64_t k = 0; k < _sh[s]->LocalQuestions(); k++)
          if ((w[(size_t)(k >> 5)] >> (k & 31)) & 1u) r->skip[(size_t)((q0 + k) >> 6)] |= 1ULL << ((q0 + k) & 63);
      }
      for (int64_t q = _Q; q < packs * 64; q++) r->skip[(size_t)(q >> 6)] |= 1ULL << (q & 63);
      r->ctx = &c;
      if (r == own) { ownLive = true; continue; }
      for (SelRequest *&slot : batch) if (slot == r) slot = nullptr;   // (published here: not the caller's to publish again)
      PublishState(&r->state, 3);
    }
    return ownLive;
  }
  // the kernels' choices: per quiz the best of the shards' winners (maximum priority, lowest index on ties, NaN never wins)
  for (int64_t i = 0; i < n; i++) {
    SelRequest *r = f.live[(size_t)i];
    double bestP = 0;
    int64_t bestI = -1;
    for (size_t s = 0; s < N; s++) {
      double p = winners[s][(size_t)i]._priority;
      const int64_t q = winners[s][(size_t)i]._iQuestion;
      if (q < 0) continue;
      if (p != p) p = -HUGE_VAL;
      if (bestI < 0 || p > bestP || (p == bestP && q < bestI)) { bestP = p; bestI = q; }
    }
    r->result = Commit(r->err, r->iQuiz, bestI);
  }
  return false;
}

// One request of a combined sweep, after the sweeps: this quiz's priority vector out of the shards' host buffers into global order,
// then the selector (the reference's: SelectSampledHost; the argmax by the device's rule) and NextQuestion's bookkeeping -- on the
// client's own thread, no lock.
int64_t ShardedEngine::SelectFromViews(SelRequest *r) {
  const int64_t nQ = _Q;
  auto skipped = [&](int64_t q) { return ((r->skip[(size_t)(q >> 6)] >> (q & 63)) & 1ULL) != 0; };
  std::vector<double> run((size_t)nQ);
  SpinWait w;
  for (size_t s = 0; s < _sh.size(); s++) {
    const HipEngine::PriorityView &v = r->views[s];
    const int64_t q0 = _sh[s]->FirstQuestion(), nLocal = _sh[s]->LocalQuestions();
    for (int64_t k = 0; k < nLocal; k++) {
      if (skipped(q0 + k)) { run[(size_t)(q0 + k)] = 0.0; continue; }
      const volatile double *rec = v.pri + (size_t)k * (size_t)v.stride;
      if (v.tag != 0) {
        // (the quiz's flag said that every workgroup had reported, not that every one of its stores had landed: an entry is taken
        //  once it carries the launch's tag -- it almost always does by now)
        const volatile uint64_t *tagWord = reinterpret_cast<const volatile uint64_t *>(rec + 1);
        while (*tagWord != v.tag)
          if (!w.Tick(std::chrono::seconds(30))) {
            r->err = Error::Make(ErrCode::Internal, "Timed out waiting for a shard's priority vector.");
            return r->result = -1;
          }
        std::atomic_thread_fence(std::memory_order_acquire);
      }
      run[(size_t)(q0 + k)] = *rec;
    }
  }
  int64_t pick = -1;
  if (r->kind == 1) {
    pick = SelectSampledHost(run.data(), nQ, _sh[0]->GetOption("eval_subtasks"), r->rnd, [&](int64_t q) { return skipped(q); });
  } else {
    double best = 0;
    for (int64_t q = 0; q < nQ; q++) {
      if (skipped(q)) continue;
      double p = run[(size_t)q];
      if (p != p) p = -HUGE_VAL;
      if (pick < 0 || p > best) { best = p; pick = q; }
    }
    if (pick >= 0) CheckPriority(run[(size_t)pick], pick);
  }
  // :403-407 a gap / asked pick falls to BaseEngine::FindNearestQuestion, over the global bitmap
  if (pick >= 0 && skipped(pick)) pick = FindNearestInPacks(pick, nQ, [&](int64_t p) { return ~r->skip[(size_t)p]; });
  return r->result = Commit(r->err, r->iQuiz, pick);
}

// ---- the batch calls of the ABI -------------------------------------------------------------------------------------------------
// Every shard's batched sweep is enqueued -- on its own device and stream -- before the first one is waited for: on N devices a
// batch takes one shard's time, not N shards' (SRPoolRunner's subtasks run side by side too, SRPlatform/Interface/SRPoolRunner.h:96-110).
Error ShardedEngine::SelectArgmaxBatch(int64_t n, const int64_t *pQuizzes, CiHipSelection *pOut) {
  if (n > 0 && !pOut) return Error::Make(ErrCode::NullArgument, "Nullptr is passed in place of a batch buffer.");
  std::lock_guard<std::mutex> ctxLock(_bctx[0].mu);   // (the shards' batch context 0: not while a combined sweep uses it)
  while (_bctx[0].readers.load(std::memory_order_acquire) != 0) _mm_pause();
  std::vector<uint64_t> tags(_sh.size(), 0);
  {
    std::lock_guard<OpMutex> lk(_opMu);
    Error e = FlushAnswers();
    if (!e.ok()) return e;
    _shardsInFlightMax = 0;
    for (size_t s = 0; s < _sh.size(); s++) {
      e = _sh[s]->EnqueueBatch(n, pQuizzes, false, &tags[s]);
      if (!e.ok()) return e;   // (validation fails on shard 0, before anything was launched)
      _shardsInFlightMax++;
    }
    for (int64_t i = 0; i < n; i++) Touch(pQuizzes[i]);
  }
  std::vector<CiHipSelection> part((size_t)n);
  Error first;
  for (size_t s = 0; s < _sh.size(); s++) {
    Error e = _sh[s]->CollectBatchSelections(n, tags[s], part.data());
    if (!e.ok()) { if (first.ok()) first = e; continue; }   // (the other shards' launches are still waited for)
    for (int64_t i = 0; i < n; i++) {
      const CiHipSelection &c = part[(size_t)i];
      CiHipSelection &b = pOut[i];
      if (s == 0) { b = c; continue; }
      if (c._iQuestion >= 0 && (b._iQuestion < 0 || c._priority > b._priority || (c._priority == b._priority && c._iQuestion < b._iQuestion))) b = c;
    }
  }
  return first;
}

Error ShardedEngine::NextQuestionArgmaxBatch(int64_t n, const int64_t *pQuizzes, int64_t *pOut) {
  if (n < 0 || n > 256) return Error::MakeP(ErrCode::IndexOutOfRange, "n=" + std::to_string(n), "Batch size is out of range.");
  if (n == 0) return Error();
  if (!pQuizzes || !pOut) return Error::Make(ErrCode::NullArgument, "Nullptr is passed in place of a batch buffer.");
  std::vector<CiHipSelection> best((size_t)n);
  Error e = SelectArgmaxBatch(n, pQuizzes, best.data());
  if (!e.ok()) return e;
  for (int64_t i = 0; i < n; i++) {
    Error ce;
    pOut[i] = Commit(ce, pQuizzes[i], best[(size_t)i]._iQuestion);   // -1 + QuestionsExhausted: reported as -1 only
  }
  return Error();
}

Error ShardedEngine::EvalPrioritiesBatch(int64_t n, const int64_t *pQuizzes, double *pOut) {
  std::lock_guard<std::mutex> ctxLock(_bctx[0].mu);
  while (_bctx[0].readers.load(std::memory_order_acquire) != 0) _mm_pause();
  {
    std::lock_guard<OpMutex> lk(_opMu);
    Error e = FlushAnswers();
    if (!e.ok()) return e;
    // every shard's sweep is in flight (its own device and stream) before the first one is waited for
    _shardsInFlightMax = 0;
    for (auto &s : _sh) {
      uint64_t tag = 0;
      e = s->EnqueueBatch(n, pQuizzes, true, &tag);
      if (!e.ok()) return e;
      _shardsInFlightMax++;
    }
    for (int64_t i = 0; i < n && pQuizzes; i++) Touch(pQuizzes[i]);
  }
  std::vector<double> part;
  for (auto &s : _sh) {
    part.resize((size_t)n * (size_t)s->LocalQuestions());
    Error e = s->CollectBatchPriorities(n, part.data());
    if (!e.ok()) return e;
    for (int64_t i = 0; i < n; i++)
      std::memcpy(pOut + (size_t)i * (size_t)_Q + (size_t)s->FirstQuestion(), part.data() + (size_t)i * (size_t)s->LocalQuestions(),
                  (size_t)s->LocalQuestions() * sizeof(double));
  }
  return Error();
}

// ---- maintenance ------------------------------------------------------------------------------------------------------------
Error ShardedEngine::RemoveQuestions(int64_t n, const int64_t *pQIds) {   // BaseEngine.cpp:722-743; all ids validated before the first is removed
  std::lock_guard<OpMutex> lk(_opMu);
  Error e = MaintenanceOnly("remove questions");
  if (!e.ok()) return e;
  if (n < 0) return Error::MakeP(ErrCode::NegativeCount, "count=" + std::to_string(n), "Counts must be non-negative.");
  if (n > 0 && !pQIds) return Error::Make(ErrCode::NullArgument, "Nullptr ids array.");
  for (int64_t i = 0; i < n; i++) {
    const int64_t iq = pQIds[i];
    bool bad = iq < 0 || iq >= _Q || std::find(_qGapList.begin(), _qGapList.end(), iq) != _qGapList.end();
    for (int64_t j = 0; j < i && !bad; j++) bad = pQIds[j] == iq;
    if (bad) return Error::MakeP(ErrCode::AbsentId, "id=" + std::to_string(iq), "Question index is not in KB.");
  }
  for (auto &s : _sh) { e = s->SetQuestionGaps(n, pQIds); if (!e.ok()) return e; }
  for (int64_t i = 0; i < n; i++) { _qGapList.push_back(pQIds[i]); _questionIds.Vacate(pQIds[i]); }
  RefreshGapBits();
  return Error();
}

Error ShardedEngine::RemoveTargets(int64_t n, const int64_t *pTIds) {   // BaseEngine.cpp:745-765
  std::lock_guard<OpMutex> lk(_opMu);
  Error e = MaintenanceOnly("remove targets");
  if (!e.ok()) return e;
  // (the target axis is replicated: every shard validates and removes the same ids; shard 0 refuses a bad call before any other is asked)
  for (auto &s : _sh) { e = s->RemoveTargets(n, pTIds); if (!e.ok()) return e; }
  return Error();
}

// New shards for new dimensions.  srcQ[q'] / srcT[t']: the old global question / target whose data new q' / t' takes, -1 for none
// (a gap, or an id that fillQ / fillT initialise).  The registries that are not data -- gaps, id maps, options, mode -- are
// carried over.  On failure nothing has changed.
Error ShardedEngine::Rebuild(int64_t newQ, int64_t newT, const std::vector<int64_t> &srcQ, const std::vector<int64_t> &srcT,
                             const std::vector<int64_t> &qGaps, const std::vector<int64_t> &tGaps, const IdLedger &targetIds,
                             const std::vector<int64_t> &fillT, const std::vector<double> &fillTInit, const std::vector<int64_t> &fillQ,
                             const std::vector<double> &fillQInit) {
  const int64_t N = (int64_t)_sh.size();
  if (!_peerAll)   // (adopt_rows_kernel reads the old shards' question blocks in place)
    return Error::MakeP(ErrCode::NotImplemented, "Feature=maintenance-mode edits of the dimensions without peer access between the devices",
                        "Rebuilding the shards reads the old shards' rows in place: every listed device must map the others' memory (PQA_DEVICES).");
  if (newQ < N) return Error::MakeP(ErrCode::InsufficientEngineDimensions, "[nQuestions=" + std::to_string(newQ) + " of " + std::to_string(N) + "]",
                                    "Fewer questions than devices in PQA_DEVICES.");
  HipEngine &s0 = *_sh[0];
  CiEngineDefinition def;
  std::memset(&def, 0, sizeof(def));
  def._nAnswers = _K; def._nTargets = newT;
  def._precType = s0.PrecisionType(); def._precMantissa = s0.PrecMantissa(); def._precExponent = s0.PrecExponent();
  def._initAmount = s0.InitAmount();
  Error err;
  for (auto &s : _sh) { err = s->Synchronize(); if (!err.ok()) return err; }
  std::vector<std::unique_ptr<HipEngine>> fresh;
  const int64_t quot = newQ / N, rem = newQ % N;   // SRPoolRunner::CalcSplit
  int64_t first = 0;
  for (int64_t s = 0; s < N; s++) {
    CiEngineDefinition d = def;
    d._nQuestions = quot + (s < rem ? 1 : 0);
    CiHipShard sh;
    sh._qFirst = first; sh._qTotal = newQ; sh._device = _devices[(size_t)s]; sh._reserved = 0;
    HipEngine *e = HipEngine::Create(err, d, &sh);
    if (!e) return err;
    fresh.emplace_back(e);
    // its questions' rows, from wherever the old shards hold them
    std::vector<const void *> blocks((size_t)d._nQuestions, nullptr);
    for (int64_t q = 0; q < d._nQuestions; q++) {
      const int64_t old = srcQ[(size_t)(first + q)];
      if (old < 0) continue;
      const int owner = OwnerOf(old);
      blocks[(size_t)q] = _sh[(size_t)owner]->QuestionBlock(old - _sh[(size_t)owner]->FirstQuestion());
    }
    err = e->AdoptRows(blocks, s0.RowLength(), srcT, _sh[(size_t)s]->VBDevicePtr());   // (vB: the replica on the same device)
    if (!err.ok()) return err;
    std::vector<int64_t> localQ;
    std::vector<double> localInit;
    for (size_t i = 0; i < fillQ.size(); i++)
      if (fillQ[i] >= first && fillQ[i] < first + d._nQuestions) { localQ.push_back(fillQ[i] - first); localInit.push_back(fillQInit[i]); }
    err = e->ApplyFills(fillT, fillTInit, localQ, localInit);
    if (!err.ok()) return err;
    err = e->SetQuestionGaps((int64_t)qGaps.size(), qGaps.data());
    if (err.ok()) err = e->SetTargetGaps((int64_t)tGaps.size(), tGaps.data());
    if (!err.ok()) return err;
    e->SetTargetIds(targetIds);
    e->SetQuizIds(_sh[(size_t)s]->QuizIds());
    Error ae;
    e->SetQuestionsAsked(s == 0 ? s0.GetTotalQuestionsAsked(ae) : 0);
    for (const char *opt : {"select", "workers", "eval_subtasks", "eval_variant", "bug_compat", "top_cache", "speculate", "host_sampled",
                            "fused_sampled", "batch_min", "batch_qb", "batch_tile", "batch_groups", "batch_tail", "batch_form", "cluster_form", "cluster_shape", "rerank", "combine", "server", "use_graph",
                            "pole_fix", "pole_lazy", "late_eager", "long_row_form", "fuse_update", "combine_spin", "combine_linger_us", "post_always", "server_idle_us"}) {
      const int64_t v = _sh[(size_t)s]->GetOption(opt);
      if (v >= 0) (void)e->SetOption(opt, std::string(opt) == "eval_subtasks" && v == 8 * _sh[(size_t)s]->GetOption("workers") ? 0 : v);
    }
    err = e->StartMaintenance(false);
    if (!err.ok()) return err;
    first += d._nQuestions;
  }
  // ---- commit
  _sh.swap(fresh);
  _Q = newQ;
  _T = newT;
  _qGapList = qGaps;
  _hostPriority.assign((size_t)newQ, 0.0);
  ForgetQuizzes();
  RefreshGapBits();
  AdoptShards();
  std::fill(_remoteReads.begin(), _remoteReads.end(), 0);   // (every old shard was synchronised above; the new ones start clean)
  std::fill(_trainEpoch.begin(), _trainEpoch.end(), 0);
  std::fill(_seenTrain.begin(), _seenTrain.end(), 0);
  return Error();
}

Error ShardedEngine::AddQsTs(int64_t nQuestions, CiAddQorTParam *pAqps, int64_t nTargets, CiAddQorTParam *pAtps) {
  std::lock_guard<OpMutex> lk(_opMu);
  Error e = MaintenanceOnly("add questions/targets");
  if (!e.ok()) return e;
  if (nQuestions < 0 || nTargets < 0)
    return Error::MakeP(ErrCode::NegativeCount, "count=" + std::to_string(std::min(nQuestions, nTargets)), "Counts must be non-negative.");
  if ((nQuestions > 0 && !pAqps) || (nTargets > 0 && !pAtps)) return Error::Make(ErrCode::NullArgument, "Nullptr parameters array.");
  // CpuEngine::AddQsTsSpec, reference PqaCore/CpuEngine.cpp:468-575, over the GLOBAL ids (as HipEngine::AddQsTs over its own):
  // gaps are reused LIFO, the rest is appended
  std::vector<int64_t> tGapList, tmp;
  _sh[0]->GetGapLists(tmp, tGapList);
  const int64_t nQReuse = std::min<int64_t>(nQuestions, (int64_t)_qGapList.size()), nQNew = nQuestions - nQReuse;
  const int64_t nTReuse = std::min<int64_t>(nTargets, (int64_t)tGapList.size()), nTNew = nTargets - nTReuse;
  std::vector<int64_t> qIds, tIds;
  std::vector<double> qInit, tInit;
  for (int64_t i = 0; i < nQReuse; i++) qIds.push_back(_qGapList[_qGapList.size() - 1 - (size_t)i]);   // :476-482
  for (int64_t i = 0; i < nTReuse; i++) tIds.push_back(tGapList[tGapList.size() - 1 - (size_t)i]);      // :488-493
  for (int64_t i = 0; i < nQNew; i++) qIds.push_back(_Q + i);                                           // :500
  for (int64_t j = 0; j < nTNew; j++) tIds.push_back(_T + j);                                           // :531
  for (int64_t i = 0; i < nQuestions; i++) qInit.push_back(pAqps[i]._initAmount);
  for (int64_t j = 0; j < nTargets; j++) tInit.push_back(pAtps[j]._initAmount);
  const int64_t newQ = _Q + nQNew, newT = _T + nTNew;
  std::vector<int64_t> srcQ((size_t)newQ, -1), srcT((size_t)newT, -1);
  for (int64_t q = 0; q < _Q; q++) srcQ[(size_t)q] = q;      // (a gap's rows travel too: they are nobody's)
  for (int64_t t = 0; t < _T; t++) srcT[(size_t)t] = t;
  for (int64_t id : qIds) if (id < _Q) srcQ[(size_t)id] = -1;   // re-initialised below
  std::vector<int64_t> qGaps(_qGapList.begin(), _qGapList.end() - nQReuse), tGaps(tGapList.begin(), tGapList.end() - nTReuse);
  IdLedger targetIds = _sh[0]->TargetIds();
  for (int64_t i = 0; i < nTReuse; i++) targetIds.Reissue(tIds[(size_t)i]);
  targetIds.Extend(newT);                                           // :541-542
  IdLedger questionIds = _questionIds;
  for (int64_t i = 0; i < nQReuse; i++) questionIds.Reissue(qIds[(size_t)i]);
  questionIds.Extend(newQ);
  // (the rebuilt shards mark their remaining gaps themselves: their own id maps are over local ids and are not consulted)
  e = Rebuild(newQ, newT, srcQ, srcT, qGaps, tGaps, targetIds, tIds, tInit, qIds, qInit);
  if (!e.ok()) return e;
  _questionIds = questionIds;
  for (int64_t i = 0; i < nQuestions; i++) pAqps[i]._index = qIds[(size_t)i];
  for (int64_t j = 0; j < nTargets; j++) pAtps[j]._index = tIds[(size_t)j];
  return Error();
}

Error ShardedEngine::Compact(int64_t *pnQuestions, const int64_t **ppOldQuestions, int64_t *pnTargets, const int64_t **ppOldTargets) {
  std::lock_guard<OpMutex> lk(_opMu);
  Error e = MaintenanceOnly("compact the KB");
  if (!e.ok()) return e;
  if (!pnQuestions || !ppOldQuestions || !pnTargets || !ppOldTargets) return Error::Make(ErrCode::NullArgument, "Nullptr output.");
  // CpuEngine::CompactSpec, CpuEngine.cpp:577-658, over the global axes (the pairing of HipEngine::Compact)
  std::vector<int64_t> tGapList, tmp;
  _sh[0]->GetGapLists(tmp, tGapList);
  std::vector<char> qGap((size_t)_Q, 0), tGap((size_t)_T, 0);
  for (int64_t g : _qGapList) qGap[(size_t)g] = 1;
  for (int64_t g : tGapList) tGap[(size_t)g] = 1;
  const int64_t nQ = _Q - (int64_t)_qGapList.size(), nT = _T - (int64_t)tGapList.size();
  if (nQ < (int64_t)_sh.size())
    return Error::MakeP(ErrCode::InsufficientEngineDimensions, "[nQuestions=" + std::to_string(nQ) + " of " + std::to_string(_sh.size()) + "]",
                        "Fewer questions than devices in PQA_DEVICES would remain.");
  std::vector<int64_t> oldQ((size_t)std::max<int64_t>(nQ, 1)), oldT((size_t)std::max<int64_t>(nT, 1));
  {   // questions: a gap in the kept prefix takes the LAST surviving question (:586-601)
    int64_t iFirst = 0, iLast = _Q - 1;
    for (; iFirst <= iLast; iFirst++) {
      if (!qGap[(size_t)iFirst]) { oldQ[(size_t)iFirst] = iFirst; continue; }
      while (qGap[(size_t)iLast] && iLast > iFirst) iLast--;
      if (iFirst == iLast) break;
      oldQ[(size_t)iFirst] = iLast;
      iLast--;
    }
  }
  {   // targets: gaps of the kept prefix (ascending) take the survivors of the dropped tail (ascending) (:604-618)
    std::vector<int64_t> dst, src;
    for (int64_t t = 0; t < nT; t++) if (tGap[(size_t)t]) dst.push_back(t); else oldT[(size_t)t] = t;
    for (int64_t t = nT; t < _T; t++) if (!tGap[(size_t)t]) src.push_back(t);
    for (size_t i = 0; i < dst.size(); i++) oldT[dst[i]] = src[i];
  }
  oldQ.resize((size_t)nQ);
  oldT.resize((size_t)nT);
  IdLedger targetIds = _sh[0]->TargetIds(), questionIds = _questionIds;
  targetIds.Repack(nT, oldT.data());
  questionIds.Repack(nQ, oldQ.data());
  e = Rebuild(nQ, nT, oldQ, oldT, {}, {}, targetIds, {}, {}, {}, {});
  if (!e.ok()) return e;
  _questionIds = questionIds;
  int64_t *outQ = (int64_t *)std::malloc(sizeof(int64_t) * (size_t)std::max<int64_t>(nQ, 1));
  int64_t *outT = (int64_t *)std::malloc(sizeof(int64_t) * (size_t)std::max<int64_t>(nT, 1));
  std::copy(oldQ.begin(), oldQ.end(), outQ);
  std::copy(oldT.begin(), oldT.end(), outT);
  *pnQuestions = nQ; *pnTargets = nT;
  *ppOldQuestions = outQ; *ppOldTargets = outT;
  return Error();
}

// ---- .kb file (layout of reference PqaCore/BaseEngine.cpp:323-385 + PqaCore/CpuEngine.cpp:664-688, see hip_engine_kb.cpp): the
// file orders its rows by question, so the shards' blocks follow each other -- every shard streams its own rows through its
// own staging buffer; vB, the target gaps and the target / quiz id maps are replicas (shard 0's are written).
namespace {
Error KbFileErr(const char *path, const char *msg) {
  return Error::MakeP(ErrCode::FileOp, std::string("filePath=[") + (path ? path : "") + "]", msg);
}
struct FileGuard {
  FILE *f;
  ~FileGuard() { if (f) std::fclose(f); }
};
}  // namespace

Error ShardedEngine::SaveKB(const char *filePath, bool doubleBuffer) {
  (void)doubleBuffer;
  if (!filePath) return Error::Make(ErrCode::NullArgument, "Nullptr is passed in place of KB file name.");
  std::lock_guard<OpMutex> lk(_opMu);
  { Error e = FlushAnswers(); if (!e.ok()) return e; }
  for (auto &s : _sh) { Error e = s->Synchronize(); if (!e.ok()) return e; }   // (also parks resident sweeps)
  FileGuard fg{std::fopen(filePath, "wb")};
  if (!fg.f) return Error::MakeP(ErrCode::CantOpenFile, std::string("filePath=[") + filePath + "]", "Can't open the file to write KB to.");
  HipEngine &s0 = *_sh[0];
  const uint64_t prec = (uint64_t)(s0.PrecisionType() & 0xF) | ((uint64_t)(s0.PrecMantissa() & 0xFFFFFFF) << 4) | ((uint64_t)s0.PrecExponent() << 32);
  const int64_t dims[3] = {_K, _Q, _T};
  Error ae;
  const uint64_t nAsked = s0.GetTotalQuestionsAsked(ae);
  if (std::fwrite(&prec, 8, 1, fg.f) != 1 || std::fwrite(dims, sizeof(dims), 1, fg.f) != 1 || std::fwrite(&nAsked, 8, 1, fg.f) != 1)
    return KbFileErr(filePath, "Can't write the KB file header.");
  for (auto &s : _sh) { Error e = s->IoRows(fg.f, filePath, false, true); if (!e.ok()) return e; }
  for (auto &s : _sh) { Error e = s->IoRows(fg.f, filePath, true, true); if (!e.ok()) return e; }
  { Error e = s0.IoVB(fg.f, filePath, true); if (!e.ok()) return e; }
  std::vector<int64_t> qGaps = _qGapList, tGaps, tmp;   // (LIFO order, as the reference's GapTracker saves it)
  s0.GetGapLists(tmp, tGaps);
  auto writeGaps = [&](const std::vector<int64_t> &gaps) {
    const int64_t n = (int64_t)gaps.size();
    return std::fwrite(&n, 8, 1, fg.f) == 1 && std::fwrite(gaps.data(), 8, (size_t)n, fg.f) == (size_t)n;
  };
  if (!writeGaps(qGaps) || !writeGaps(tGaps)) return KbFileErr(filePath, "Can't write the gaps.");
  // (the live quiz map with empty = true keeps its next permanent id, as BaseEngine.cpp:379 and HipEngine::SaveKB write it)
  if (!_questionIds.Write(fg.f) || !s0.TargetIds().Write(fg.f) || !s0.QuizIds().Write(fg.f, true))
    return KbFileErr(filePath, "Can't write the permanent-compact ID mappings.");
  if (std::fflush(fg.f) != 0) return KbFileErr(filePath, "Failed in hard flushing the KB.");
  FILE *f = fg.f;
  fg.f = nullptr;
  if (std::fclose(f) != 0) return KbFileErr(filePath, "Failed in closing the file.");
  return Error();
}

ShardedEngine *ShardedEngine::Load(Error &err, const char *filePath, const std::vector<int> &devices) {
  if (!filePath) { err = Error::Make(ErrCode::NullArgument, "Nullptr is passed in place of KB file name."); return nullptr; }
  FileGuard fg{std::fopen(filePath, "rb")};
  if (!fg.f) { err = Error::MakeP(ErrCode::CantOpenFile, std::string("filePath=[") + filePath + "]", "Can't open the KB file to read."); return nullptr; }
  uint64_t prec = 0, nAsked = 0;
  int64_t dims[3];
  if (std::fread(&prec, 8, 1, fg.f) != 1 || std::fread(dims, sizeof(dims), 1, fg.f) != 1 || std::fread(&nAsked, 8, 1, fg.f) != 1) {
    err = KbFileErr(filePath, "Can't read the KB file header.");
    return nullptr;
  }
  CiEngineDefinition def;
  std::memset(&def, 0, sizeof(def));
  def._nAnswers = dims[0]; def._nQuestions = dims[1]; def._nTargets = dims[2];
  def._precType = (uint8_t)(prec & 0xF);
  def._precMantissa = (uint32_t)((prec >> 4) & 0xFFFFFFF);
  def._precExponent = (uint16_t)((prec >> 32) & 0xFFFF);
  def._initAmount = 1.0;   // not stored in the file; every count is overwritten below
  std::unique_ptr<ShardedEngine> eng(Create(err, def, devices));
  if (!eng) return nullptr;
  auto fail = [&](Error e) { err = std::move(e); return (ShardedEngine *)nullptr; };
  for (auto &s : eng->_sh) { Error e = s->IoRows(fg.f, filePath, false, false); if (!e.ok()) return fail(std::move(e)); }
  for (auto &s : eng->_sh) { Error e = s->IoRows(fg.f, filePath, true, false); if (!e.ok()) return fail(std::move(e)); }
  {
    Error e = eng->_sh[0]->IoVB(fg.f, filePath, false);
    if (!e.ok()) return fail(std::move(e));
    std::vector<double> vb((size_t)dims[2]);   // shard 0's vB (already in the engine's number type) to the other replicas
    e = eng->_sh[0]->GetKB(nullptr, nullptr, vb.data());
    for (size_t i = 1; e.ok() && i < eng->_sh.size(); i++) e = eng->_sh[i]->SetVBFromHost(vb.data());
    if (!e.ok()) return fail(std::move(e));
  }
  eng->_sh[0]->SetQuestionsAsked(nAsked);
  auto readGaps = [&](std::vector<int64_t> &gaps, int64_t limit) {
    int64_t n;
    if (std::fread(&n, 8, 1, fg.f) != 1 || n < 0 || n > limit) return false;
    gaps.resize((size_t)n);
    if (std::fread(gaps.data(), 8, (size_t)n, fg.f) != (size_t)n) return false;
    for (int64_t g : gaps) if (g < 0 || g >= limit) return false;
    return true;
  };
  std::vector<int64_t> qGaps, tGaps;
  if (!readGaps(qGaps, dims[1]) || !readGaps(tGaps, dims[2])) return fail(KbFileErr(filePath, "Can't read the gaps."));
  eng->_qGapList = qGaps;
  eng->RefreshGapBits();
  for (auto &s : eng->_sh) {
    Error e = s->SetQuestionGaps((int64_t)qGaps.size(), qGaps.data());
    if (e.ok()) e = s->SetTargetGaps((int64_t)tGaps.size(), tGaps.data());
    if (!e.ok()) return fail(std::move(e));
  }
  IdLedger targetIds, quizIds;
  if (!eng->_questionIds.Read(fg.f) || !targetIds.Read(fg.f) || !quizIds.Read(fg.f)) return fail(KbFileErr(filePath, "Can't read the permanent-compact ID mappings."));
  for (auto &s : eng->_sh) { s->SetTargetIds(targetIds); s->SetQuizIds(quizIds); }
  err = Error();
  return eng.release();
}

IEngine *LoadShardedEngine(Error &err, const char *filePath, const std::vector<int> &devices) {
  return ShardedEngine::Load(err, filePath, devices);
}

IEngine *CreateShardedEngine(Error &err, const CiEngineDefinition &def, const std::vector<int> &devices) {
  return ShardedEngine::Create(err, def, devices);
}

}  // namespace pqa
