// sharded_engine.cpp -- one knowledge base, its question axis split over several devices of ONE process, behind the same C ABI.
//
// every question's priority depends only on its own sA / mD rows and on the (small, replicated) posterior, so
// the sweep shards by question with no data-path collective; what the shards exchange per selection is 16 bytes each.
// PQA_DEVICES=0,1,...,7 makes PqaEngineFactory_CreateCpuEngine build this engine instead of a single-device one, so that the
// reference's unchanged wrappers -- which can only call PqaEngine_NextQuestion (PqaCInterop.cpp:261-267) -- reach all the GPUs
// of a node.  Shard s is a complete HipEngine over the contiguous question range SRPoolRunner::CalcSplit gives it
// (SRPlatform/Interface/SRPoolRunner.h:96-110), with replicas of vB, the gap bitmaps and every quiz's posterior.
//
//   NextQuestion (argmax)  every shard's sweep is enqueued on its own device and stream; each finisher writes {priority, GLOBAL
//                          index} and then the step number into this engine's pinned slots; the host picks when all have
//                          landed (maximum priority, lowest index on ties).  No collective launch, no copies, no stream sync.
//   NextQuestion (sampled) the reference's selector needs every priority in global order: each shard's priority vector
//                          (8 bytes per question) is copied to the host and the selection -- per-subtask Kahan run lengths,
//                          grand totals, two upper_bounds (CpuEngine.cpp:362-400) -- runs there, with the same subtask split
//                          over the GLOBAL question range as an unsharded engine.
//   RecordAnswer           runs on the owner of the active question; the new posterior goes to the other shards by peer copies
//                          (hipMemcpyPeerAsync, ordered by events: no host synchronisation).
//   ResumeQuiz             shard 0 computes the posterior from row POINTERS, reading other shards' rows in place over peer
//                          access (xGMI); the other shards adopt it.
//   Train / RecordQuizTarget   every shard applies the steps that fall on its questions (and its vB replica).
//   SaveKB / LoadCpuEngine the file orders its rows by question: every shard streams its own block (same byte layout as a whole-cube
//                          engine's file: a KB saved sharded loads unsharded and vice versa).
// Where the rows are: peer access is enabled between all listed devices at creation; several shards on one device (tests on a
// single GPU) need none.  Maintenance-mode edits of the dimensions rebuild the shards (Rebuild below).  Not sharded (NotImplemented on
// this engine): SetStream and the stream-ordered single-shard entry points.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>

#include "hip_engine.h"

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

struct HostKahan {                 // SRAccumulator<SRDoubleNumber> (SRPlatform/Interface/SRAccumulator.h:15-39)
  double sum = 0, corr = 0;
  void add(double v) {
    const double y = v - corr;
    const double t = sum + y;
    corr = (t - sum) - y;
    sum = t;
  }
  double get() const { return sum - corr; }
};

}  // namespace

class ShardedEngine final : public IEngine {
 public:
  static ShardedEngine *Create(Error &err, const CiEngineDefinition &def, const std::vector<int> &devices);
  ~ShardedEngine() override;

  // The reference validates every answered question before any Add subtask runs (CETrainSubtaskDistrib.h:26-45): a gap question
  // owned by shard k must not leave shards 0..k-1 trained and their vB replicas ahead -- every shard validates, then every shard trains.
  Error Train(int64_t n, const AQ *pAQs, int64_t iTarget, double amount) override {
    std::lock_guard<std::mutex> lk(_mu);
    if (n >= 0 && amount > 0 && (n == 0 || pAQs != nullptr))   // (else: shard 0 produces the reference's argument error, before any kernel)
      for (auto &s : _sh) { Error e = s->ValidateTrain(n, pAQs, iTarget, -1); if (!e.ok()) return e; }
    for (auto &s : _sh) { Error e = s->Train(n, pAQs, iTarget, amount); if (!e.ok()) return e; }
    return SyncShards();
  }
  // A shard's training kernel is ordered before that shard's later work only (its own stream); another shard's ResumeQuiz reads
  // this shard's rows in place over peer access: the cube is settled before the call returns.
  Error SyncShards() {
    for (auto &s : _sh) { Error e = s->Synchronize(); if (!e.ok()) return e; }
    return Error();
  }
  uint64_t GetTotalQuestionsAsked(Error &err) override { return _sh[0]->GetTotalQuestionsAsked(err); }
  void CopyDims(CiEngineDimensions *pDims) const override { _sh[0]->CopyDims(pDims); }
  int64_t StartQuiz(Error &err) override;
  int64_t ResumeQuiz(Error &err, int64_t nAnswered, const AQ *pAQs) override;
  int64_t NextQuestion(Error &err, int64_t iQuiz) override;
  Error RecordAnswer(int64_t iQuiz, int64_t iAnswer) override;
  int64_t GetActiveQuestionId(Error &err, int64_t iQuiz) override {
    std::lock_guard<std::mutex> lk(_mu);
    Touch(iQuiz);
    return _sh[0]->GetActiveQuestionId(err, iQuiz);
  }
  Error SetActiveQuestion(int64_t iQuiz, int64_t iQuestion) override {
    std::lock_guard<std::mutex> lk(_mu);
    Touch(iQuiz);
    for (auto &s : _sh) { Error e = s->SetActiveQuestion(iQuiz, iQuestion); if (!e.ok()) return e; }
    return Error();
  }
  int64_t ListTopTargets(Error &err, int64_t iQuiz, int64_t maxCount, CiRatedTarget *pDest) override {
    std::lock_guard<std::mutex> lk(_mu);
    Touch(iQuiz);
    return _sh[_lastOwner.count(iQuiz) ? _lastOwner[iQuiz] : 0]->ListTopTargets(err, iQuiz, maxCount, pDest);   // (its kernel listed them already)
  }
  Error RecordQuizTarget(int64_t iQuiz, int64_t iTarget, double amount) override {
    std::lock_guard<std::mutex> lk(_mu);
    Touch(iQuiz);
    if (amount > 0)
      for (auto &s : _sh) { Error e = s->ValidateTrain(0, nullptr, iTarget, iQuiz); if (!e.ok()) return e; }
    for (auto &s : _sh) { Error e = s->RecordQuizTarget(iQuiz, iTarget, amount); if (!e.ok()) return e; }
    return SyncShards();
  }
  Error ReleaseQuiz(int64_t iQuiz) override {
    std::lock_guard<std::mutex> lk(_mu);
    _lastOwner.erase(iQuiz);
    _usage.erase(iQuiz);
    Error first;   // every shard releases (a shard that has not got the quiz says so): the registries stay in step
    for (auto &s : _sh) { Error e = s->ReleaseQuiz(iQuiz); if (!e.ok() && first.ok()) first = e; }
    return first;
  }
  // A forced switch destroys the shards' quizzes (BaseEngine.cpp:650-668), a shutdown likewise: what this engine keeps per quiz goes too
  // -- a stale usage time would make ClearOldQuizzes release an id no shard has, a stale owner would serve a later quiz of that id.
  Error StartMaintenance(bool force) override {
    Error e = All([&](HipEngine &sh) { return sh.StartMaintenance(force); });
    if (e.ok() && force) ForgetQuizzes();
    return e;
  }
  Error FinishMaintenance() override { return All([&](HipEngine &e) { return e.FinishMaintenance(); }); }
  Error Shutdown(const char *saveFilePath) override {
    if (saveFilePath && *saveFilePath) { Error e = SaveKB(saveFilePath, false); if (!e.ok()) return e; }   // BaseEngine.cpp:270-300
    Error e = All([&](HipEngine &sh) { return sh.Shutdown(nullptr); });
    if (e.ok()) ForgetQuizzes();
    return e;
  }
  bool MapIds(int which, bool toPerm, int64_t count, int64_t *pIds) override {
    if (which == 0) {   // questions: the global compact <-> permanent map lives here (the shards' own maps are over local ids)
      bool ok = true;
      for (int64_t i = 0; i < count; i++) {
        pIds[i] = toPerm ? _questionIds.PermanentOf(pIds[i]) : _questionIds.SlotOf(pIds[i]);
        ok = ok && pIds[i] != -1;
      }
      return ok;
    }
    return _sh[0]->MapIds(which, toPerm, count, pIds);
  }
  bool EnsurePermQuizGreater(int64_t bound) override { bool ok = true; for (auto &s : _sh) ok = s->EnsurePermQuizGreater(bound) && ok; return ok; }
  bool RemapQuizPermId(int64_t a, int64_t b) override { bool ok = true; for (auto &s : _sh) ok = s->RemapQuizPermId(a, b) && ok; return ok; }
  Error SaveKB(const char *filePath, bool doubleBuffer) override;
  static ShardedEngine *Load(Error &err, const char *filePath, const std::vector<int> &devices);
  Error AddQsTs(int64_t nQuestions, CiAddQorTParam *pAqps, int64_t nTargets, CiAddQorTParam *pAtps) override;
  Error RemoveQuestions(int64_t n, const int64_t *pQIds) override;
  Error RemoveTargets(int64_t n, const int64_t *pTIds) override;
  Error Compact(int64_t *pnQuestions, const int64_t **ppOldQuestions, int64_t *pnTargets, const int64_t **ppOldTargets) override;
  Error ClearOldQuizzes(int64_t maxCount, double maxAgeSec) override;

  Error SetOption(const char *name, int64_t value) override {
    const std::string n(name ? name : "");
    Error e = All([&](HipEngine &sh) { return sh.SetOption(name, value); });
    if (!e.ok()) return e;   // (a value the shards refuse changes nothing here either)
    std::lock_guard<std::mutex> lk(_mu);
    if (n == "select") _select = value;   // (kept here too: NextQuestion dispatches on it)
    if (n == "seed") Seed((uint64_t)value);
    return Error();
  }
  int64_t GetOption(const char *name) const override {
    const std::string n(name ? name : "");
    if (n == "shards") return (int64_t)_sh.size();
    if (n == "shards_in_flight_max") return _shardsInFlightMax;   // the most shards whose sweeps were enqueued before the first was waited for (newest call)
    return _sh[0]->GetOption(name);
  }
  const char *EvalKernelName() const override { return _sh[0]->EvalKernelName(); }
  Error SetKB(const double *pA, const double *pD, const double *pB) override {
    std::lock_guard<std::mutex> lk(_mu);
    for (auto &s : _sh) {
      const size_t q0 = (size_t)s->FirstQuestion();
      Error e = s->SetKB(pA + q0 * (size_t)_K * (size_t)_T, pD + q0 * (size_t)_T, pB);
      if (!e.ok()) return e;
    }
    return Error();
  }
  Error GetKB(double *pA, double *pD, double *pB) override {
    std::lock_guard<std::mutex> lk(_mu);
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
    Error e = All([&](HipEngine &eng) { return eng.SetQuestionGaps(n, ids); });
    if (e.ok())
      for (int64_t i = 0; i < n; i++)
        if (std::find(_qGapList.begin(), _qGapList.end(), ids[i]) == _qGapList.end()) { _qGapList.push_back(ids[i]); _questionIds.Vacate(ids[i]); }
    return e;
  }
  Error EvalPriorities(int64_t iQuiz, double *pOut, int64_t n) override {
    if (n != _Q) return Error::MakeP(ErrCode::IndexOutOfRange, "n=" + std::to_string(n), "Priority buffer length must equal the question count.");
    std::lock_guard<std::mutex> lk(_mu);
    for (auto &s : _sh) { Error e = s->EvalPriorities(iQuiz, pOut + s->FirstQuestion(), s->LocalQuestions()); if (!e.ok()) return e; }
    return Error();
  }
  int64_t NextQuestionArgmax(Error &err, int64_t iQuiz) override;
  int64_t NextQuestionSampled(Error &err, int64_t iQuiz, uint64_t rnd) override;
  Error GetPriors(int64_t iQuiz, double *pOut, int64_t n) override {
    std::lock_guard<std::mutex> lk(_mu);
    Touch(iQuiz);
    return _sh[0]->GetPriors(iQuiz, pOut, n);
  }
  Error NextQuestionArgmaxBatch(int64_t n, const int64_t *pQuizzes, int64_t *pOut) override;
  Error EvalPrioritiesBatch(int64_t n, const int64_t *pQuizzes, double *pOut) override {
    std::lock_guard<std::mutex> lk(_mu);
    // every shard's sweep is in flight (its own device and stream) before the first one is waited for
    _shardsInFlightMax = 0;
    for (auto &s : _sh) {
      uint64_t tag = 0;
      Error e = s->EnqueueBatch(n, pQuizzes, true, &tag);
      if (!e.ok()) return e;
      _shardsInFlightMax++;
    }
    for (int64_t i = 0; i < n && pQuizzes; i++) Touch(pQuizzes[i]);
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
  Error SelectArgmaxBatch(int64_t n, const int64_t *pQuizzes, CiHipSelection *pOut) override;
  Error Log2HotArray(const double *pIn, double *pOut, int64_t n) override { return _sh[0]->Log2HotArray(pIn, pOut, n); }
  hipStream_t GetStream() const override { return _sh[0]->GetStream(); }
  Error SetStream(hipStream_t) override { return NotSharded("SetStream"); }
  Error Synchronize() override { return All([&](HipEngine &e) { return e.Synchronize(); }); }
  Error EnqueueSelectArgmax(int64_t, void *) override { return NotSharded("EnqueueSelectArgmax"); }
  Error EnqueueSelectArgmaxFlag(int64_t, void *, void *, uint64_t) override { return NotSharded("EnqueueSelectArgmaxFlag"); }
  Error EnqueueEval(int64_t iQuiz) override { return All([&](HipEngine &e) { return e.EnqueueEval(iQuiz); }); }
  Error GetPriorDevicePtr(int64_t iQuiz, void **ppDev, int64_t *pLdT) override { return _sh[0]->GetPriorDevicePtr(iQuiz, ppDev, pLdT); }
  Error RecordAnswerRemote(int64_t, int64_t) override { return NotSharded("RecordAnswerRemote"); }
  Error RecordAnswerBatch(int64_t n, const int64_t *pQuizzes, const int64_t *pAnswers) override {   // (each answer on the shard that owns its question)
    if (n > 0 && (!pQuizzes || !pAnswers)) return Error::Make(ErrCode::NullArgument, "Nullptr is passed in place of a batch buffer.");
    for (int64_t i = 0; i < n; i++) { Error e = RecordAnswer(pQuizzes[i], pAnswers[i]); if (!e.ok()) return e; }
    return Error();
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
  ShardedEngine() = default;
  template <typename F>
  Error All(F &&f) {
    std::lock_guard<std::mutex> lk(_mu);
    for (auto &s : _sh) { Error e = f(*s); if (!e.ok()) return e; }
    return Error();
  }
  int OwnerOf(int64_t qGlobal) const {
    for (size_t s = 0; s < _sh.size(); s++) if (_sh[s]->OwnsQuestion(qGlobal)) return (int)s;
    return -1;
  }
  int64_t SelectArgmaxLocked(Error &err, int64_t iQuiz, double *pPriority);
  int64_t Commit(Error &err, int64_t iQuiz, int64_t qGlobal);
  uint64_t NextRandom() {   // xorshift128+, the generator family of SRPlatform/Interface/SRFastRandom.h:60-72
    uint64_t s1 = _rng[0];
    const uint64_t s0 = _rng[1];
    _rng[0] = s0;
    s1 ^= s1 << 23;
    _rng[1] = s1 ^ s0 ^ (s1 >> 18) ^ (s0 >> 5);
    return _rng[1] + s0;
  }

  std::vector<std::unique_ptr<HipEngine>> _sh;
  std::mutex _mu;
  int64_t _K = 0, _Q = 0, _T = 0;
  int64_t _select = 0;
  Slot *_slots = nullptr;               // [shards], pinned + mapped: written by the sweeps' finishers, polled here
  uint64_t _step = 0;
  std::vector<hipEvent_t> _posteriorReady;   // per shard: recorded behind the RecordAnswer kernel whose posterior the others copy
  std::vector<hipEvent_t> _copyDone;         // per shard: recorded behind its copy of another shard's posterior
  std::unordered_map<int64_t, int> _lastOwner;   // quiz -> the shard whose RecordAnswer kernel ran last (it listed the top targets)
  std::vector<double> _hostPriority;
  IdLedger _questionIds;                   // global question ids
  uint64_t _rng[2] = {0x9E3779B97F4A7C15ULL, 0xBF58476D1CE4E5B9ULL};
  void Seed(uint64_t x) {   // SplitMix64 into the two words of the generator, as HipEngine does
    auto next = [&x] {
      uint64_t z = (x += 0x9E3779B97F4A7C15ULL);
      z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
      z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
      return z ^ (z >> 31);
    };
    _rng[0] = next();
    _rng[1] = next();
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
  // BaseQuiz::OnUsage (BaseEngine.cpp:417) for the engine as a whole: the shards are touched at different times (ListTopTargets
  // reaches one of them, GetPriors shard 0), so which quizzes ClearOldQuizzes evicts is decided HERE, once, and every shard
  // releases the same ids in the same order -- the shards' registries (ids, gaps) never diverge.
  std::unordered_map<int64_t, time_t> _usage;
  void ForgetQuizzes() { std::lock_guard<std::mutex> lk(_mu); _usage.clear(); _lastOwner.clear(); }
  void Touch(int64_t iQuiz) { auto it = _usage.find(iQuiz); if (it != _usage.end()) it->second = time(nullptr); }
  void ReleaseEverywhere(int64_t iQuiz, size_t nShards) {   // roll a partly created quiz back
    for (size_t s = 0; s < nShards; s++) (void)_sh[s]->ReleaseQuiz(iQuiz);
  }
};

ShardedEngine::~ShardedEngine() {
  for (size_t s = 0; s < _sh.size(); s++) {
    hipSetDevice(_sh[s]->Device());
    if (s < _posteriorReady.size() && _posteriorReady[s]) hipEventDestroy(_posteriorReady[s]);
    if (s < _copyDone.size() && _copyDone[s]) hipEventDestroy(_copyDone[s]);
  }
  _sh.clear();
  if (_slots) hipHostFree(_slots);
}

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
  // peer access between all pairs of distinct devices: the posterior copies and ResumeQuiz's in-place row reads go over xGMI
  for (int a : devices)
    for (int b : devices)
      if (a != b) {
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, a, b) == hipSuccess && can) {
          hipSetDevice(a);
          const hipError_t e = hipDeviceEnablePeerAccess(b, 0);
          if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
          (void)hipGetLastError();
        }
      }
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
  if (hipHostMalloc((void **)&eng->_slots, sizeof(Slot) * (size_t)N, hipHostMallocMapped | hipHostMallocCoherent | hipHostMallocPortable) != hipSuccess) {
    err = Error::Make(ErrCode::Internal, "Can't allocate the shards' selection slots.");
    return nullptr;
  }
  std::memset(eng->_slots, 0, sizeof(Slot) * (size_t)N);
  eng->_posteriorReady.assign((size_t)N, nullptr);
  eng->_copyDone.assign((size_t)N, nullptr);
  for (int64_t s = 0; s < N; s++) {
    hipSetDevice(devices[(size_t)s]);
    if (hipEventCreateWithFlags(&eng->_posteriorReady[(size_t)s], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&eng->_copyDone[(size_t)s], hipEventDisableTiming) != hipSuccess) {
      err = Error::Make(ErrCode::Internal, "Can't create the shards' ordering events.");
      return nullptr;
    }
  }
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
  err = Error();
  return eng.release();
}

int64_t ShardedEngine::StartQuiz(Error &err) {
  std::lock_guard<std::mutex> lk(_mu);
  int64_t id = -1;
  for (size_t s = 0; s < _sh.size(); s++) {
    const int64_t got = _sh[s]->StartQuiz(err);
    if (got < 0) { if (s > 0) ReleaseEverywhere(id, s); return -1; }   // (e.g. out of memory on shard s: the earlier shards' quiz goes again)
    if (s == 0) id = got;
    else if (got != id) {
      (void)_sh[s]->ReleaseQuiz(got);
      ReleaseEverywhere(id, s);
      err = Error::Make(ErrCode::Internal, "The shards' quiz registries have diverged.");
      return -1;
    }
  }
  _usage[id] = time(nullptr);
  return id;
}

int64_t ShardedEngine::ResumeQuiz(Error &err, int64_t nAnswered, const AQ *pAQs) {
  if (nAnswered < 0) { err = Error::MakeP(ErrCode::NegativeCount, "count=" + std::to_string(nAnswered), "|nAnswered| must be non-negative."); return -1; }
  if (nAnswered == 0) return StartQuiz(err);   // BaseEngine.cpp:393-395
  if (pAQs == nullptr) { err = Error::Make(ErrCode::NullArgument, "Nullptr is passed in place of answered questions."); return -1; }
  std::lock_guard<std::mutex> lk(_mu);
  std::vector<const void *> rows(2 * (size_t)nAnswered);
  for (int64_t i = 0; i < nAnswered; i++) {
    const int owner = OwnerOf(pAQs[i].iQuestion);
    if (owner < 0) { err = Error::MakeP(ErrCode::IndexOutOfRange, "subjIndex=" + std::to_string(pAQs[i].iQuestion), "Question index is not in KB range."); return -1; }
    err = _sh[(size_t)owner]->GetRowPointers(pAQs[i].iQuestion, pAQs[i].iAnswer, &rows[2 * (size_t)i], &rows[2 * (size_t)i + 1]);
    if (!err.ok()) return -1;
  }
  // shard 0 computes (remote rows are read in place), the others adopt its posterior
  const int64_t id = _sh[0]->ResumeQuizRows(err, nAnswered, pAQs, rows.data());
  if (id < 0) return -1;
  void *src = nullptr;
  int64_t ld = 0;
  err = _sh[0]->GetPriorDevicePtr(id, &src, &ld);
  if (!err.ok()) return -1;
  hipSetDevice(_sh[0]->Device());
  if (hipEventRecord(_posteriorReady[0], _sh[0]->GetStream()) != hipSuccess) { err = Error::Make(ErrCode::Internal, "hipEventRecord failed."); return -1; }
  for (size_t s = 1; s < _sh.size(); s++) {
    const int64_t got = _sh[s]->ResumeQuizAdopt(err, nAnswered, pAQs, (const double *)src, _sh[0]->Device(), _posteriorReady[0]);
    if (got < 0) { ReleaseEverywhere(id, s); return -1; }
    if (got != id) {
      (void)_sh[s]->ReleaseQuiz(got);
      ReleaseEverywhere(id, s);
      err = Error::Make(ErrCode::Internal, "The shards' quiz registries have diverged.");
      return -1;
    }
    // shard 0's next posterior kernel of this quiz (RecordAnswer on owner 0) waits for this copy of the one it rewrites
    hipSetDevice(_sh[s]->Device());
    if (hipEventRecord(_copyDone[s], _sh[s]->GetStream()) != hipSuccess) { ReleaseEverywhere(id, _sh.size()); err = Error::Make(ErrCode::Internal, "hipEventRecord failed."); return -1; }
  }
  _lastOwner[id] = 0;
  _usage[id] = time(nullptr);
  return id;
}

// ClearOldQuizzes (behaviour: BaseEngine.cpp:814-873), decided once for all shards by the rule the one-device engine uses
// (QuizzesToLetGo, hip_engine_kb.cpp) over the usage times kept here.
Error ShardedEngine::ClearOldQuizzes(int64_t maxCount, double maxAgeSec) {
  if (maxCount < 0)
    return Error::MakeP(ErrCode::NegativeCount, "count=" + std::to_string(maxCount), "The number of quizzes to keep cannot be less than 0.");
  std::lock_guard<std::mutex> lk(_mu);
  if (!_sh[0]->IsRegularMode()) return Error();   // quizzes are not expected to exist in maintenance / shutdown mode
  std::vector<QuizUsage> inUse;
  for (const auto &kv : _usage) inUse.push_back(QuizUsage{kv.first, kv.second});
  std::sort(inUse.begin(), inUse.end(), [](const QuizUsage &x, const QuizUsage &y) { return x.id < y.id; });   // registry order
  Error first;
  for (int64_t id : QuizzesToLetGo(inUse, time(nullptr), maxCount, maxAgeSec)) {
    _lastOwner.erase(id);
    _usage.erase(id);
    for (auto &s : _sh) { Error e = s->ReleaseQuiz(id); if (!e.ok() && first.ok()) first = e; }
  }
  return first;
}

// Everything NextQuestion does after the pick (CpuEngine.cpp:403-413): the question becomes the quiz's active question on every
// shard, the asked-questions counter moves once.
int64_t ShardedEngine::Commit(Error &err, int64_t iQuiz, int64_t qGlobal) {
  if (qGlobal < 0) { err = Error::Make(ErrCode::QuestionsExhausted, "Found no unasked question that is not in a gap."); return -1; }
  for (auto &s : _sh) { err = s->SetActiveQuestion(iQuiz, qGlobal); if (!err.ok()) return -1; }
  _sh[0]->BumpQuestionsAsked(1);
  return qGlobal;
}

int64_t ShardedEngine::SelectArgmaxLocked(Error &err, int64_t iQuiz, double *pPriority) {
  if (++_step == 0) ++_step;
  const uint64_t step = _step;
  for (size_t s = 0; s < _sh.size(); s++) {
    // the slots are mapped + portable: their host address is what every device sees
    err = _sh[s]->EnqueueSelectArgmaxFlag(iQuiz, &_slots[s].priority, &_slots[s].flag, step);
    if (!err.ok()) return -1;
  }
  const auto t0 = std::chrono::steady_clock::now();
  for (size_t s = 0; s < _sh.size(); s++) {
    volatile uint64_t *flag = &_slots[s].flag;
    uint64_t spins = 0;
    while (*flag != step)
      if ((++spins & 0x3FFF) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60)) {
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

int64_t ShardedEngine::NextQuestionArgmax(Error &err, int64_t iQuiz) {
  std::lock_guard<std::mutex> lk(_mu);
  Touch(iQuiz);
  const int64_t q = SelectArgmaxLocked(err, iQuiz, nullptr);
  if (!err.ok()) return -1;
  return Commit(err, iQuiz, q);
}

// The reference's selector (PqaCore/CpuEngine.cpp:362-400) over the GLOBAL question range: the same per-subtask Kahan run
// lengths, grand totals and upper_bounds as select_sampled_wg_impl (pqa_device.h) runs on one device, here on the host over the
// shards' priority vectors -- with the same priorities, subtask count and random number it picks the question an unsharded
// engine picks.
int64_t ShardedEngine::NextQuestionSampled(Error &err, int64_t iQuiz, uint64_t rnd) {
  std::lock_guard<std::mutex> lk(_mu);
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
  if (_select == 1) return NextQuestionArgmax(err, iQuiz);
  uint64_t rnd;
  { std::lock_guard<std::mutex> lk(_mu); rnd = NextRandom(); }
  return NextQuestionSampled(err, iQuiz, rnd);
}

Error ShardedEngine::RecordAnswer(int64_t iQuiz, int64_t iAnswer) {
  std::lock_guard<std::mutex> lk(_mu);
  Touch(iQuiz);
  Error err;
  const int64_t aq = _sh[0]->GetActiveQuestionId(err, iQuiz);
  if (!err.ok()) return err;
  const int owner = aq < 0 ? 0 : OwnerOf(aq);   // (no / invalid active question: let a shard produce the reference's error)
  if (owner < 0) return _sh[0]->RecordAnswer(iQuiz, iAnswer);
  HipEngine &o = *_sh[(size_t)owner];
  // the owner's kernel rewrites its posterior: not before the other shards have taken their copies of the previous one
  hipSetDevice(o.Device());
  for (size_t s = 0; s < _sh.size(); s++)
    if ((int)s != owner && hipStreamWaitEvent(o.GetStream(), _copyDone[s], 0) != hipSuccess) return Error::Make(ErrCode::Internal, "hipStreamWaitEvent failed.");
  err = o.RecordAnswer(iQuiz, iAnswer);
  if (!err.ok()) return err;
  if (hipEventRecord(_posteriorReady[(size_t)owner], o.GetStream()) != hipSuccess) return Error::Make(ErrCode::Internal, "hipEventRecord failed.");
  void *src = nullptr;
  int64_t ld = 0;
  err = o.GetPriorDevicePtr(iQuiz, &src, &ld);
  if (!err.ok()) return err;
  for (size_t s = 0; s < _sh.size(); s++) {
    if ((int)s == owner) continue;
    err = _sh[s]->RecordAnswerRemote(iQuiz, iAnswer);
    if (!err.ok()) return err;
    err = _sh[s]->AdoptPrior(iQuiz, (const double *)src, o.Device(), _posteriorReady[(size_t)owner]);
    if (!err.ok()) return err;
    hipSetDevice(_sh[s]->Device());
    if (hipEventRecord(_copyDone[s], _sh[s]->GetStream()) != hipSuccess) return Error::Make(ErrCode::Internal, "hipEventRecord failed.");
  }
  _lastOwner[iQuiz] = owner;
  return Error();
}

// Every shard's batched sweep is enqueued -- on its own device and stream -- before the first one is waited for: on N devices a
// batch takes one shard's time, not N shards' (SRPoolRunner's subtasks run side by side too, SRPlatform/Interface/SRPoolRunner.h:96-110).
Error ShardedEngine::SelectArgmaxBatch(int64_t n, const int64_t *pQuizzes, CiHipSelection *pOut) {
  std::lock_guard<std::mutex> lk(_mu);
  if (n > 0 && !pOut) return Error::Make(ErrCode::NullArgument, "Nullptr is passed in place of a batch buffer.");
  std::vector<uint64_t> tags(_sh.size(), 0);
  _shardsInFlightMax = 0;
  for (size_t s = 0; s < _sh.size(); s++) {
    Error e = _sh[s]->EnqueueBatch(n, pQuizzes, false, &tags[s]);
    if (!e.ok()) return e;   // (validation fails on shard 0, before anything was launched)
    _shardsInFlightMax++;
  }
  for (int64_t i = 0; i < n; i++) Touch(pQuizzes[i]);
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
  std::lock_guard<std::mutex> lk(_mu);
  for (int64_t i = 0; i < n; i++) {
    Error ce;
    pOut[i] = Commit(ce, pQuizzes[i], best[(size_t)i]._iQuestion);   // -1 + QuestionsExhausted: reported as -1 only
  }
  return Error();
}

// ---- maintenance ------------------------------------------------------------------------------------------------------------
Error ShardedEngine::RemoveQuestions(int64_t n, const int64_t *pQIds) {   // BaseEngine.cpp:722-743; all ids validated before the first is removed
  std::lock_guard<std::mutex> lk(_mu);
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
  return Error();
}

Error ShardedEngine::RemoveTargets(int64_t n, const int64_t *pTIds) {   // BaseEngine.cpp:745-765
  std::lock_guard<std::mutex> lk(_mu);
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
                            "fused_sampled", "batch_min", "batch_qb", "batch_tile", "rerank", "combine", "server", "use_graph"}) {
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
  _lastOwner.clear();
  _usage.clear();
  return Error();
}

Error ShardedEngine::AddQsTs(int64_t nQuestions, CiAddQorTParam *pAqps, int64_t nTargets, CiAddQorTParam *pAtps) {
  std::lock_guard<std::mutex> lk(_mu);
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
  std::lock_guard<std::mutex> lk(_mu);
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
  std::lock_guard<std::mutex> lk(_mu);
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
