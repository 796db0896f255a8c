// hip_engine_select.cpp -- HipEngine: the selection paths of one caller (hip_engine.h): the single-quiz sweeps and their selectors
// (argmax on the device, the reference's sampled selector on the host), the batched sweeps of the batch ABI, the graph replay.
#include "hip_engine_internal.h"

namespace pqa {
// ------------------------------------------------------------------------------------------------------------------
// NextQuestion
// ------------------------------------------------------------------------------------------------------------------
bool HipEngine::QuestionUnavailable(const Quiz *q, int64_t qLocal) const {
  return BitTest(_hQGap, qLocal) || BitTest(q->hAsked, qLocal);
}

// BaseEngine::FindNearestQuestion, reference PqaCore/BaseEngine.cpp:60-124: the available question "nearest" to iMiddle as the
// reference finds it -- exact within iMiddle's own 64-bit pack, then pack by pack outwards, comparing only the two packs at the
// same pack distance.  avail(p): bit i set = question 64 p + i is neither asked nor a gap (bits past nQuestions clear).
int64_t FindNearestInPacks(int64_t iMiddle, int64_t nQuestions, const std::function<uint64_t(int64_t)> &avail) {
  const uint32_t dInf = 200;
  const int64_t iPack64 = iMiddle >> 6;
  const uint32_t iWithin = (uint32_t)(iMiddle & 63);
  const uint64_t available = avail(iPack64);
  if (available != 0) {
    const uint64_t baseMask = (1ULL << iWithin) - 1;
    const uint64_t higher = available & ~baseMask, lower = baseMask & available;
    const uint32_t dHigher = higher ? ((uint32_t)__builtin_ctzll(higher) - iWithin) : dInf;
    const uint32_t dLower = lower ? (iWithin - (uint32_t)(63 - __builtin_clzll(lower))) : dInf;
    return (dHigher < dLower) ? iMiddle + dHigher : iMiddle - dLower;
  }
  const int64_t limPack64 = (nQuestions + 63) >> 6;
  int64_t i = 1;
  while ((iPack64 >= i) && (iPack64 + i < limPack64)) {
    const uint64_t availLeft = avail(iPack64 - i), availRight = avail(iPack64 + i);
    if ((availLeft | availRight) == 0) { i++; continue; }
    const uint32_t dHigher = availRight ? ((uint32_t)__builtin_ctzll(availRight) + 64 - iWithin) : dInf;
    const uint32_t dLower = availLeft ? (iWithin + 64 - (uint32_t)(63 - __builtin_clzll(availLeft))) : dInf;
    if (dHigher < dLower) return iMiddle + dHigher + ((i - 1) << 6);
    return iMiddle - dLower - ((i - 1) << 6);
  }
  while (iPack64 >= i) {
    const uint64_t availLeft = avail(iPack64 - i);
    if (!availLeft) { i++; continue; }
    return iMiddle - (iWithin + 64 - (uint32_t)(63 - __builtin_clzll(availLeft))) - ((i - 1) << 6);
  }
  while (iPack64 + i < limPack64) {
    const uint64_t availRight = avail(iPack64 + i);
    if (!availRight) { i++; continue; }
    return iMiddle + ((uint32_t)__builtin_ctzll(availRight) + 64 - iWithin) + ((i - 1) << 6);
  }
  return -1;
}

// The reference's selector (PqaCore/CpuEngine.cpp:362-400) on the host, over a priority vector the sweep has delivered: the same
// per-subtask Kahan run lengths (CEEvalQsSubtaskConsider.cpp:52-58, :212-214), Kahan grand totals and two upper_bounds as
// select_sampled_wg_impl (pqa_device.h) -- operation for operation, so with the same priorities, subtask count and random number it
// picks the same question.  run: priorities in, run lengths out.  Returns the pick before the gap / asked fallback (:403-407).
// The reference reports numeric anomalies of a sweep in its log -- non-finite grand totals of the priorities (CpuEngine.cpp:370-373),
// a non-positive grand total (:375-377), a priority that is not a positive finite number (CEEvalQsSubtaskConsider.cpp:209-211) --
// and goes on.  So does this engine, for what reaches the host: the selected question's priority, the totals of the sampled
// selector.  (NaN never wins an argmax here, so a NaN winner means that every available question's priority is NaN.)  At most
// kAnomalyLogLimit entries per process: a broken knowledge base would otherwise write one per selection.
namespace {
std::atomic<int> gAnomaliesLogged{0};
constexpr int kAnomalyLogLimit = 200;
}  // namespace
void LogAnomaly(DefaultLogger::Severity sev, const char *what, double value) {
  if (gAnomaliesLogged.fetch_add(1, std::memory_order_relaxed) >= kAnomalyLogLimit) return;
  char buf[64];
  std::snprintf(buf, sizeof(buf), "%.17g", value);
  DefaultLogger::Log(sev, std::string(what) + buf);
}
void CheckPriority(double priority, int64_t index) {   // CEEvalQsSubtaskConsider.cpp:209-211, for the question that was selected
  if (index >= 0 && !(priority > 0 && std::isfinite(priority))) LogAnomaly(DefaultLogger::Severity::Warning, "Got priority=", priority);
}

namespace {
template <class Skip>
int64_t SelectSampledHostT(double *run, int64_t n, int64_t nWorkers, uint64_t rnd, const Skip &skipped) {
  struct Kahan {                 // SRAccumulator<SRDoubleNumber> (SRPlatform/Interface/SRAccumulator.h:15-39)
    double sum = 0, corr = 0;
    void add(double v) { const double y = v - corr; const double t = sum + y; corr = (t - sum) - y; sum = t; }
    double get() const { return sum - corr; }
  };
  const int64_t quot = n / nWorkers, rem = n % nWorkers, nSubtasks = quot == 0 ? rem : nWorkers;   // SRPoolRunner::CalcSplit
  auto bound = [&](int64_t i) { return (i + 1) * quot + std::min<int64_t>(i + 1, rem); };           // end of subtask i
  std::vector<double> grand((size_t)nSubtasks);
  for (int64_t s = 0; s < nSubtasks; s++) {
    Kahan acc;
    for (int64_t i = s == 0 ? 0 : bound(s - 1); i < bound(s); i++) {
      if (!skipped(i)) acc.add(run[i]);   // gap / asked questions only copy the running sum
      run[i] = acc.get();
    }
    grand[(size_t)s] = acc.get();
  }
  Kahan tot;                                                     // CpuEngine.cpp:362-368
  for (int64_t s = 0; s < nSubtasks; s++) {
    tot.add(grand[(size_t)s]);
    grand[(size_t)s] = tot.get();
    if (!std::isfinite(grand[(size_t)s]))                          // :370-373
      LogAnomaly(DefaultLogger::Severity::Error, "Overflow or underflow has happened in the question evaluation subtasks: ", grand[(size_t)s]);
  }
  const double totG = grand[(size_t)nSubtasks - 1];
  if (totG <= 0) LogAnomaly(DefaultLogger::Severity::Warning, "Grand-grand total is ", totG);   // :375-377
  const double selRunLen = totG * (double)rnd / 18446744073709551615.0;   // :379, SRDoubleNumber::MakeRandom
  const int64_t iWorker = std::upper_bound(grand.begin(), grand.end(), selRunLen) - grand.begin();   // :380-381
  if (iWorker >= nSubtasks) return n - 1;                         // :384
  const double inWorker = selRunLen - (iWorker == 0 ? 0.0 : grand[(size_t)iWorker - 1]);   // :388
  const int64_t first = iWorker == 0 ? 0 : bound(iWorker - 1), limit = bound(iWorker);
  int64_t sel = std::upper_bound(run + first, run + limit, inWorker) - run;   // :391
  if (sel >= limit) sel = limit - 1;                              // :392-400
  return sel;
}
}  // namespace
int64_t SelectSampledHost(double *run, int64_t n, int64_t nWorkers, uint64_t rnd, const std::function<bool(int64_t)> &skipped) {
  return SelectSampledHostT(run, n, nWorkers, rnd, skipped);
}
// (the same over bit words -- a question is skipped if its bit is set in either array; `b` may be null: the test inlined
//  instead of a call through std::function per question, 1000 of them per selection)
int64_t SelectSampledHostBits(double *run, int64_t n, int64_t nWorkers, uint64_t rnd, const uint32_t *a, const uint32_t *b) {
  if (b == nullptr) return SelectSampledHostT(run, n, nWorkers, rnd, [a](int64_t i) { return ((a[i >> 5] >> (i & 31)) & 1u) != 0; });
  return SelectSampledHostT(run, n, nWorkers, rnd, [a, b](int64_t i) { return (((a[i >> 5] | b[i >> 5]) >> (i & 31)) & 1u) != 0; });
}

int64_t HipEngine::FindNearestQuestion(int64_t iMiddle, const Quiz *q) const {   // (over the local question range)
  return FindNearestInPacks(iMiddle, _Q, [&](int64_t p) { return ~(Pack64(_hQGap, p) | Pack64(q->hAsked, p)); });
}

int64_t HipEngine::FinishSelection(Error &err, Quiz *q, int64_t selLocal) {
  // reference PqaCore/CpuEngine.cpp:403-413
  if (selLocal >= 0 && QuestionUnavailable(q, selLocal)) selLocal = FindNearestQuestion(selLocal, q);
  if (selLocal < 0) {
    err = Error::Make(ErrCode::QuestionsExhausted, "Found no unasked question that is not in a gap.");
    return -1;
  }
  q->activeQuestion = _qFirst + selLocal;
  _nQuestionsAsked.fetch_add(1, std::memory_order_relaxed);
  return q->activeQuestion;
}

Error HipEngine::EnqueueEval(int64_t iQuiz) {
  std::lock_guard<EngineMutex> lk(_mu);
  Error err = CheckRegular("compute next question");
  if (!err.ok()) return err;
  Quiz *q = UseQuiz(err, iQuiz);
  if (!q) return err;
  hipSetDevice(_device);
  StopServer();   // a launched sweep has no room beside the resident one and would wait for it to idle out
  return LaunchSingleSweep(q, nullptr);
}

// The single-quiz sweep of this engine's precision on the engine's stream: the register-resident fp64 shapes with the fused
// argmax (eval_kernels.hip) for Double engines; for Float engines the fp32 streaming sweep and, where a selection is asked
// for, the argmax kernel behind it (batch_kernels.hip, select_kernels.hip).
bool HipEngine::UseClusterSweep() const { return _optEvalVariant == 0 && _ldT > ClusterFrom() && EvalClusterSupported(View()); }

Error HipEngine::LaunchSingleSweep(Quiz *q, const FusedSelect *fused) {
  { Error se = SettlePoleList(); if (!se.ok()) return se; }
  if (UseClusterSweep()) {
    // long rows, either precision: the question split over a cluster of workgroups, then the epilogues, then (where a selection
    // is asked for) the argmax kernel
    const size_t need = EvalClusterScratchBytes(View());
    if (need > _clusterScratchBytes) {
      HIP_TRY(hipStreamSynchronize(_stream));
      hipFree(_dClusterScratch);
      _dClusterScratch = nullptr;
      _clusterScratchBytes = 0;
      HIP_TRY(hipMalloc(&_dClusterScratch, need));
      HIP_TRY(hipMemsetAsync(_dClusterScratch, 0, need, _stream));   // (no record of fresh memory may look like a launch's)
      _clusterScratchBytes = need;
    }
    HIP_TRY(LaunchEvalCluster(View(), q->dPrior, q->dAsked, _dPriority, _dClusterScratch, _stream));
    if (fused != nullptr)
      HIP_TRY(LaunchSelectArgmax(_dPriority, _dQGap, q->dAsked, 0, _Q, fused->outBase, fused->out, fused->seq, fused->flagValue, _stream));
    return Error();
  }
  if (_elem == 8) {
    // (measurement hook, option "time_sweeps": HIP events on the engine's stream right around this launch -- what the launch of a
    //  SYNCHRONOUS selection takes on the device, dispatch to retirement, as a profiler's kernel trace sees it; back-to-back launches,
    //  whose ramps overlap, are shorter)
    if (_optTimeSweeps) {
      if (!_evSweep[0]) { HIP_TRY(hipEventCreate(&_evSweep[0])); HIP_TRY(hipEventCreate(&_evSweep[1])); }
      HIP_TRY(hipEventRecord(_evSweep[0], _stream));
    }
    HIP_TRY(LaunchEvalQuestions(View(), q->dPrior, q->dAsked, 0, _Q, _dPriority, (int)_optEvalVariant, fused, _stream));
    if (_optTimeSweeps) { HIP_TRY(hipEventRecord(_evSweep[1], _stream)); _sweepTimed = true; }
    return Error();
  }
  if (_optEvalVariant != 99 && EvalF32RegisterShape(View(), (int)_optEvalVariant))   // (variant 99: the streaming form, as for Double engines)
    HIP_TRY(LaunchEvalQuestionsF32Reg(View(), q->dPrior, q->dAsked, _dPriority, (int)_optEvalVariant, _stream));
  else
    HIP_TRY(LaunchEvalQuestionsF32(View(), q->dPrior, q->dAsked, _dPriority, _stream));
  if (fused != nullptr)
    HIP_TRY(LaunchSelectArgmax(_dPriority, _dQGap, q->dAsked, 0, _Q, fused->outBase, fused->out, fused->seq, fused->flagValue, _stream));
  return Error();
}

Error HipEngine::EnqueueSelectArgmax(int64_t iQuiz, void *pOut) {
  std::lock_guard<EngineMutex> lk(_mu);
  Error err = CheckRegular("compute next question");
  if (!err.ok()) return err;
  Quiz *q = UseQuiz(err, iQuiz);
  if (!q) return err;
  // one launch: the sweep's last workgroup picks the argmax; reported index = local position + qFirst (GLOBAL id)
  const FusedSelect fs{_dSelScratch, pOut ? (SelectResult *)pOut : _dSel, nullptr, NextLaunchTag(), _qFirst, 0, 0, nullptr, 0, 0, nullptr, nullptr};
  hipSetDevice(_device);
  StopServer();   // a launched sweep has no room beside the resident one and would wait for it to idle out
  return LaunchSingleSweep(q, &fs);
}

// The same, for a multi-process host loop that exchanges the shards' winners through host memory shared by the ranks
// (probqa_amd/dist.py: ShmSelector): the record goes to pOut and then flagValue to pFlag, both device-visible addresses of
// host-coherent (registered) memory, straight from the sweep's finisher -- no copy, no stream synchronisation.
Error HipEngine::EnqueueSelectArgmaxFlag(int64_t iQuiz, void *pOut, void *pFlag, uint64_t flagValue) {
  std::lock_guard<EngineMutex> lk(_mu);
  Error err = CheckRegular("compute next question");
  if (!err.ok()) return err;
  Quiz *q = UseQuiz(err, iQuiz);
  if (!q) return err;
  if (!pOut || !pFlag) return Error::Make(ErrCode::NullArgument, "Nullptr is passed in place of the record or the flag.");
  hipSetDevice(_device);
  // The resident sweep cannot hand a quiz over to the fix of pole_kernels.hip here -- its record goes to another process, where a "-4,
  // take the launched path" means nothing -- so while that fix is on (the default) this selection is always launched, with the fix
  // behind it: a late quiz state gets the reference-order sums on the sharded path as on the single-engine one.
  if (_optServer && !_optPoleFix && ServerUsable()) return ServerPost(q, (SelectResult *)pOut, (uint64_t *)pFlag, flagValue, _qFirst | (int64_t)kServerNoWatch);
  const FusedSelect fs{_dSelScratch, (SelectResult *)pOut, (uint64_t *)pFlag, NextLaunchTag(), _qFirst, 0, flagValue, nullptr, 0, 0, nullptr, nullptr};
  StopServer();   // a launched sweep has no room beside the resident one and would wait for it to idle out
  return LaunchSingleSweep(q, &fs);
}

hipError_t HipEngine::EnsureHostPriority() {
  if (_hostPriorityCap >= _capQ && _hHostPriority != nullptr) return hipSuccess;
  StopServer();   // (its launch arguments hold the old buffer)
  if (_hHostPriority) hipHostFree(_hHostPriority);
  _hHostPriority = nullptr;
  _hostPriorityCap = 0;
  const hipError_t e = hipHostMalloc((void **)&_hHostPriority, (size_t)_capQ * sizeof(TaggedPriority), hipHostMallocMapped | hipHostMallocCoherent);
  if (e == hipSuccess) {
    std::memset(_hHostPriority, 0, (size_t)_capQ * sizeof(TaggedPriority));   // (no launch has tag 0)
    _hostPriorityCap = _capQ;
  }
  return e;
}

// After the flag: the entries of the questions the sweep evaluated, each taken once it carries the launch's tag (the flag says
// that every workgroup has reported, not that every one of its stores has landed).
Error HipEngine::CollectHostPriority(uint64_t tag, const Quiz *q) {
  _hostRun.resize((size_t)_Q);
  const volatile TaggedPriority *rec = _hHostPriority;
  SpinWait w;
  for (int64_t i = 0; i < _Q; i++) {
    if (BitTest(_hQGap, i) || BitTest(q->hAsked, i)) { _hostRun[(size_t)i] = 0.0; continue; }
    while (rec[i].tag != tag)
      if (!w.Tick(std::chrono::seconds(30))) return HipErr(hipErrorNotReady, "priority vector hand-over");
    std::atomic_thread_fence(std::memory_order_acquire);
    _hostRun[(size_t)i] = rec[i].priority;
  }
  return Error();
}

// The same wait for MANY client threads at once (ListTopTargets while other clients are inside the engine): each waits for its own
// quiz's flag, typically behind a combined sweep of a few hundred microseconds -- spinning all the while, dozens of them eat the
// cores the process is allowed.  A short spin (the kernel may be about to finish), then naps of ~20 us.
Error HipEngine::WaitFlagNapping(volatile uint64_t *flag, uint64_t value, const char *what) {
  for (int spins = 0; spins < 2000; spins++) {
    if (*flag == value) { std::atomic_thread_fence(std::memory_order_acquire); return Error(); }
    _mm_pause();
  }
  static thread_local bool slackSet = false;
  if (!slackSet) { prctl(PR_SET_TIMERSLACK, 2000UL, 0, 0, 0); slackSet = true; }   // (the default slack rounds a 20 us nap up to 70)
  const auto t0 = std::chrono::steady_clock::now();
  uint64_t naps = 0;
  while (*flag != value) {
    struct timespec ts{0, 20000};
    nanosleep(&ts, nullptr);
    if ((++naps & 0x3FF) == 0) {
      if (hipStreamQuery(_stream) == hipSuccess && *flag != value) {  // the kernel retired without publishing
        const hipError_t he = hipStreamSynchronize(_stream);
        if (he != hipSuccess || *flag != value) return HipErr(he == hipSuccess ? hipErrorUnknown : he, what);
      }
      if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(30)) return HipErr(hipErrorNotReady, what);
    }
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  return Error();
}

Error HipEngine::WaitFlag(volatile uint64_t *flag, uint64_t value, const char *what) {
  SpinWait w;
  while (*flag != value) {
    if (!w.Tick(std::chrono::seconds(30))) return HipErr(hipErrorNotReady, what);
    if (w.Due() && hipStreamQuery(_stream) == hipSuccess && *flag != value) {  // the kernel retired without publishing
      const hipError_t he = hipStreamSynchronize(_stream);
      if (he != hipSuccess || *flag != value) return HipErr(he == hipSuccess ? hipErrorUnknown : he, what);
    }
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  return Error();
}

int64_t HipEngine::NextQuestionArgmax(Error &err, int64_t iQuiz) { return Combine(err, iQuiz, 0, 0); }

// Synchronous single-quiz selections of a Double engine through a register shape (whose finisher knows whether anything was listed):
// the fix of pole_kernels.hip is launched only when the sweep says so (FusedSelect::lazyFix) -- one launch per selection of a fresh quiz.
bool HipEngine::LazyFix() const {
  return _optPoleFix && _optPoleLazy && _optPoleFollow && _elem == 8 && !UseClusterSweep() && EvalVariantHasFinisherWorkgroup(View(), (int)_optEvalVariant);
}
Error HipEngine::SettlePoleList() {
  if (!_poleListPending) return Error();
  _poleListPending = false;
  DropSpeculation();
  const KbView kb = View();
  if (kb.poleList != nullptr) HIP_TRY(hipMemsetAsync(kb.poleList, 0, sizeof(PoleHeader), _stream));   // (stream order: behind the sweep that wrote it)
  return Error();
}
Error HipEngine::RunLazyFix(Quiz *q, const FusedSelect &swept, const char *what) {
  FusedSelect fs = swept;
  fs.lazyFix = 0;
  fs.flagValue = NextLaunchTag();   // (the records of a hand-over keep the sweep's tag, seqValue)
  HIP_TRY(LaunchEvalPoleFixup(View(), q->dPrior, q->dAsked, _dPriority, fs, _stream));
  Error err = WaitFlag(fs.seq, fs.flagValue, what);
  std::atomic_thread_fence(std::memory_order_acquire);
  if (err.ok()) _poleListPending = false;   // (the fix-up has emptied the list)
  return err;
}

// One quiz, by itself (the caller holds _mu)
int64_t HipEngine::NextQuestionArgmaxLocked(Error &err, int64_t iQuiz) {
  err = CheckRegular("compute next question");
  if (!err.ok()) return -1;
  Quiz *q = UseQuiz(err, iQuiz);
  if (!q) return -1;
  hipSetDevice(_device);
  err = FlushUpdates();
  if (!err.ok()) return -1;
  if (_optUseGraph && _elem == 8) return NextQuestionArgmaxGraph(err, q);
  if (_optServer && !q->noServer && ServerUsable()) {
    // resident sweep: post the request, poll the answer -- no launch on the critical path
    const uint64_t value = kServerFlagBase | ++_opSeq;   // (its own range: see kGraphFlagBase)
    err = ServerPost(q, &_hPinned->sel, &_hPinned->seq, value, 0);
    if (err.ok()) err = ServerWait(&_hPinned->seq, value, "NextQuestionArgmax");
    if (!err.ok()) return -1;
    if (_hPinned->sel.index == -3) {
      err = HipErr(hipErrorLaunchFailure, "NextQuestionArgmax (incomplete sweep)");
      return -1;
    }
    if (_hPinned->sel.index != -4) {
      CheckPriority(_hPinned->sel.priority, _hPinned->sel.index);
      return FinishSelection(err, q, _hPinned->sel.index);
    }
    q->noServer = true;   // (a row at the pole of the lack term: this quiz's selections are launched from here on, the fix behind them)
  }
  // One launch; the last workgroup writes the winner and then a sequence number straight into host-coherent pinned
  // memory, which this thread polls: no D2H copy, no stream synchronisation on the critical path.
  uint64_t seq;
  FusedSelect fs{};
  if (TakeSpeculation(q, 1 << 1, &seq) != 0) fs = _spec.fs;   // (RecordAnswer has launched this very sweep already)
  else {
    seq = NextLaunchTag();
    fs = FusedSelect{_dSelScratch, &_hPinned->sel, &_hPinned->seq, seq, 0, 0, seq, nullptr, 0, 0, nullptr, nullptr, LazyFix() && q->lateStreak < _optLateEager ? 1 : 0};   // (a quiz whose last selections all needed the fix: launched behind the sweep again)
    StopServer();   // a launched sweep has no room beside the resident one and would wait for it to idle out
    err = LaunchSingleSweep(q, &fs);
    if (!err.ok()) return -1;
    if (fs.lazyFix) _poleListPending = true;   // (whatever it lists stays listed until the fix has run or the sweep has said "nothing": a wait that fails leaves it for SettlePoleList)
  }
  err = WaitFlag(&_hPinned->seq, seq, "NextQuestionArgmax");
  if (!err.ok()) return -1;
  std::atomic_thread_fence(std::memory_order_acquire);
  if (fs.lazyFix) {
    if (_hPinned->sel.index == -4) {   // the sweep listed rows at the pole of the lack term: the fix now, and its answer
      err = RunLazyFix(q, fs, "NextQuestionArgmax");   // (clears the mark once the fix has emptied the list)
      if (!err.ok()) return -1;
      q->lateStreak++;
    } else {
      q->lateStreak = 0;
      if (_hPinned->sel.index != -3) _poleListPending = false;   // (a complete sweep that listed nothing)
    }
  }
  if (_hPinned->sel.index == -3) {  // the sweep's finisher gave up: some workgroup of the launch never reported
    err = HipErr(hipErrorLaunchFailure, "NextQuestionArgmax (incomplete sweep)");
    return -1;
  }
  CheckPriority(_hPinned->sel.priority, _hPinned->sel.index);
  return FinishSelection(err, q, _hPinned->sel.index);
}


Error HipEngine::BatchSweep(BatchCtx &c, int64_t n, const int64_t *pQuizzes, std::vector<Quiz *> &quizzes, bool wantPriorities, uint64_t tag,
                            bool hostPriorities, bool *pQuizMinor, bool *pTagged) {
  if (!c.h) {  // first batch: staging in host-coherent pinned memory, winner records
    HIP_TRY(hipHostMalloc(&c.h, sizeof(BatchPinned), hipHostMallocMapped | hipHostMallocCoherent));
    std::memset(c.h, 0, sizeof(BatchPinned));
    HIP_TRY(hipMalloc(&c.dSlots, kMaxBatch * sizeof(QuizSlot)));
    HIP_TRY(hipMalloc(&c.dScratch, (size_t)kMaxBatch * kBatchGrid * sizeof(SelectResult)));
    HIP_TRY(hipMemset(c.dScratch, 0, (size_t)kMaxBatch * kBatchGrid * sizeof(SelectResult)));
  }
  // Which form: the row-sharing sweep has one wave per 64 quizzes and block of questions -- on a small cube a small batch
  // leaves most of the chip's 1024 SIMDs without a wave (1000 x 5 x 1000, 64 quizzes: 500 waves, 40 k selections/s against
  // 92 k for grid.y = quiz, whose 48 MB cube is re-read from the Infinity Cache), while 256 quizzes fill it (133 k vs 95 k).
  // batch_min = 0 (default) decides by the wave count; an explicit value decides by the batch size alone.
  // (the count is in waves of TWO questions per lane, what the fp64 sweep had when the bar was measured; with one per lane since
  //  round 6 the bars were measured again -- tools/batch_bench.py, row-sharing against grid.y, ms per batch: 1000 x 5 x 1000 64 quizzes
  //  0.61 / 0.41, 128: 0.77 / 0.76, 192: 1.29 / 1.15, 256: 1.21 / 1.47 -- as before; 2000 x 5 x 2000 32: 1.24 / 1.10, 48: 1.55 / 2.00,
  //  64: 1.53 / 1.99, 128: 2.35 / 3.84 -- the row-sharing sweep from 33 quizzes on where rows are longer than 1024 targets and it has a
  //  thousand such waves; 4000 x 5 x 4000 and 10000 x 5 x 10000: from 32 quizzes, as before)
  const int64_t qb = _optBatchQb > 0 ? _optBatchQb : (_elem == 4 ? 4 : 2), wavesRowSharing = ((n + 63) / 64) * ((_Q + qb - 1) / qb);
  bool rowSharing = _elem == 4 || wantPriorities ||
                    (_optBatchMin > 0 ? n >= _optBatchMin : (n >= 32 && (wavesRowSharing >= 1536 || (n > 32 && _ldT > 1024 && wavesRowSharing >= 1000))));
  // ... and between the two, for a few dozen quizzes over short rows (a server's combined sweeps): a lane is a (quiz, chunk of the
  // row) -- batch_kernels.hip: eval_midbatch_kernel.  Option batch_form: 0 = by these rules, 1 grid.y = quiz, 2 row-sharing, 3 this one.
  // By the measured costs at 1000 x 5 x 1000 (tools/midbatch_bench.py): grid.y ~11.3 us per quiz + 25; this form 87 / 138 / 229 us for up
  // to 8 / 16 / 32 quizzes (its lanes come in 8, 16 or 32 quiz slots) and 6.2 us per slot of 64 beyond: it won at 7 and 8 quizzes and from
  // 11 on, except 17 and 18.  Round 6, measured again (grid.y / this form, us per batch): 6: 91 / 91, 7: 103 / 92, 9: 145 / 133, 10: 157 / 134,
  // 12: 182 / 131, 17: 248 / 213, 18: 260 / 211, 24: 339 / 214, 32: 443 / 221 -- from seven quizzes on.
  bool mid = EvalMidBatchSupported(View()) && ((_optBatchForm == 0 && !rowSharing && n >= 7) || _optBatchForm == 3);
  if (_optBatchForm == 1 && _elem == 8) { rowSharing = false; mid = false; }   // (with priorities wanted too: the quizzes' own vectors, CollectBatchPriorities)
  if (_optBatchForm == 2) { rowSharing = true; mid = false; }
  if (mid) rowSharing = false;
  if (hostPriorities) {
    wantPriorities = rowSharing;   // (the row-sharing sweep keeps its priority matrix; grid.y = quiz writes per-quiz vectors anyway)
    if (pQuizMinor) *pQuizMinor = rowSharing;
    if (!c.event) HIP_TRY(hipEventCreateWithFlags(&c.event, hipEventDisableTiming));
  }
  auto copyToHost = [&](const double *src, size_t doubles) -> Error {
    if (c.readers.load(std::memory_order_acquire) != 0)   // (ServeQueue has waited for them before it took the lock they need)
      return Error::Make(ErrCode::Internal, "A priority buffer is still being read.");
    if (doubles > c.hPriDoubles) {
      HIP_TRY(hipStreamSynchronize(_stream));   // (nothing of an earlier batch is on its way into the old buffer)
      if (c.hPri) hipHostFree(c.hPri);
      c.hPri = nullptr;
      c.hPriDoubles = 0;
      HIP_TRY(hipHostMalloc((void **)&c.hPri, doubles * sizeof(double), hipHostMallocDefault));
      c.hPriDoubles = doubles;
      c.hPriCoherent = false;
    }
    HIP_TRY(hipMemcpyAsync(c.hPri, src, doubles * sizeof(double), hipMemcpyDeviceToHost, _stream));
    return Error();
  };
  if (!rowSharing && c.priorityQ != _Q) {  // per-quiz priority vectors of the grid.y form, (re)sized with the knowledge base
    if (c.dPriority) hipFree(c.dPriority);
    c.dPriority = nullptr;
    c.priorityQ = -1;
    HIP_TRY(hipMalloc(&c.dPriority, (size_t)kMaxBatch * (size_t)_Q * sizeof(double)));
    c.priorityQ = _Q;
  }
  Error err;
  quizzes.assign((size_t)n, nullptr);
  for (int64_t i = 0; i < n; i++) {
    quizzes[i] = UseQuiz(err, pQuizzes[i]);
    if (!quizzes[i]) return err;
    for (int64_t j = 0; j < i; j++)
      if (pQuizzes[j] == pQuizzes[i])
        return Error::MakeP(ErrCode::IndexOutOfRange, "quizId=" + std::to_string(pQuizzes[i]), "A quiz appears twice in one batch.");
    c.h->slots[i] = QuizSlot{quizzes[i]->dPrior, quizzes[i]->dAsked, rowSharing ? nullptr : c.dPriority + (size_t)i * (size_t)_Q,
                                 &c.h->out[i], &c.h->seq[i], nullptr};
  }
  // grid.y = quiz and the priorities wanted on the host: every workgroup stores the priorities of its questions there itself, one
  // {priority, launch tag} record each (as the single-quiz sweep's hand-over, FusedSelect::hostPriority) -- no copy behind the
  // sweep and no event: the quiz's flag says that every workgroup has reported, an entry is taken once it carries the tag
  const bool tagged = hostPriorities && !rowSharing && (mid || EvalVariantHasFinisherWorkgroup(View(), (int)_optEvalVariant));
  if (pTagged) *pTagged = tagged;
  if (tagged) {
    const size_t doubles = 2 * (size_t)n * (size_t)_Q;
    if (c.readers.load(std::memory_order_acquire) != 0) return Error::Make(ErrCode::Internal, "A priority buffer is still being read.");
    if (doubles > c.hPriDoubles || !c.hPriCoherent) {
      HIP_TRY(hipStreamSynchronize(_stream));
      if (c.hPri) hipHostFree(c.hPri);
      c.hPri = nullptr;
      c.hPriDoubles = 0;
      const size_t want = std::max(doubles, 2 * (size_t)64 * (size_t)_Q);
      HIP_TRY(hipHostMalloc((void **)&c.hPri, want * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent));
      std::memset(c.hPri, 0, want * sizeof(double));   // (no launch has tag 0)
      c.hPriDoubles = want;
      c.hPriCoherent = true;
    }
    for (int64_t i = 0; i < n; i++) c.h->slots[i].hostPriority = reinterpret_cast<TaggedPriority *>(c.hPri) + (size_t)i * (size_t)_Q;
  }
  HIP_TRY(hipMemcpyAsync(c.dSlots, c.h->slots, (size_t)n * sizeof(QuizSlot), hipMemcpyHostToDevice, _stream));
  StopServer();   // a launched sweep has no room beside the resident one and would wait for it to idle out
  auto grow = [&](void **p, size_t &have, size_t need) -> hipError_t {
    if (need <= have) return hipSuccess;
    hipStreamSynchronize(_stream);
    hipFree(*p);
    *p = nullptr;
    have = 0;
    const hipError_t e = hipMalloc(p, need);
    if (e == hipSuccess) have = need;
    return e;
  };
  // the batched sweeps' pole scratch (pole_kernels.hip): grown with the batch, its head cleared once
  auto growPole = [&](BatchPlan &plan) -> hipError_t {
    plan.pole = nullptr;
    if (plan.poleBytes == 0) return hipSuccess;
    if (plan.poleBytes > c.poleBytes) {
      const hipError_t e = grow(&c.dPole, c.poleBytes, plan.poleBytes);
      if (e != hipSuccess) return e;
      const hipError_t me = hipMemsetAsync(c.dPole, 0, kBatchPoleClear, _stream);
      if (me != hipSuccess) return me;
    }
    plan.pole = c.dPole;
    return hipSuccess;
  };
  if (mid) {
    const KbView kb = View();
    BatchPlan plan{};
    HIP_TRY(LaunchEvalMidBatch(kb, c.dSlots, (int)n, &plan, nullptr, nullptr, nullptr, 0, tag, true, _stream));
    HIP_TRY(grow(&c.dPT, c.ptBytes, plan.ptBytes));
    HIP_TRY(grow((void **)&c.dRecs, c.recBytes, plan.recBytes));
    HIP_TRY(growPole(plan));
    const bool matrix = wantPriorities || plan.poleBytes > 0;   // (the fix corrects the priority matrix, the pick reads it)
    if (matrix) HIP_TRY(grow((void **)&c.dPriT, c.priTBytes, (size_t)_Q * (size_t)plan.Bp * sizeof(double)));
    HIP_TRY(LaunchEvalMidBatch(kb, c.dSlots, (int)n, &plan, c.dPT, c.dRecs, matrix ? c.dPriT : nullptr, 0, tag, false, _stream));
    c.lastBp = plan.Bp;
    return Error();
  }
  c.lastBp = 0;   // (0: the quizzes' priorities are their own vectors in dPriority, not a matrix)
  if (!rowSharing) {
    const FusedSelect fs{c.dScratch, nullptr, nullptr, tag, 0, kBatchGrid, tag, nullptr, tagged ? 1 : 0, 0, nullptr,
                         tagged ? reinterpret_cast<TaggedPriority *>(c.hPri) : nullptr};
    // (the batch's suspect list and records -- pole_kernels.hip: grown with the batch, the head cleared once)
    const KbView kbv = View();
    BatchPlan pp{};
    pp.poleBytes = EvalBatchPoleBytes(kbv, (int)n);
    HIP_TRY(growPole(pp));
    HIP_TRY(LaunchEvalQuestionsBatch(kbv, c.dSlots, (int)n, 0, _Q, (int)_optEvalVariant, fs, _stream, pp.pole));
    if (hostPriorities && !tagged) return copyToHost(c.dPriority, (size_t)n * (size_t)_Q);
    return Error();
  }
  const KbView kb = View();
  BatchPlan plan{};
  plan.tileTargets = (int)_optBatchTile;
  plan.questionsPerBlock = (int)_optBatchQb;
  plan.questionGroups = (int)_optBatchGroups;
  plan.splitTail = (int)_optBatchTail;
  HIP_TRY(LaunchEvalBatch(kb, c.dSlots, (int)n, &plan, nullptr, nullptr, nullptr, nullptr, 0, tag, true, _stream));
  HIP_TRY(grow(&c.dPT, c.ptBytes, plan.ptBytes));
  HIP_TRY(grow((void **)&c.dAcc, c.accBytes, plan.accBytes));
  HIP_TRY(grow((void **)&c.dRecs, c.recBytes, plan.recBytes));
  // Float engines: the fp32 sweep nominates every quiz's best questions, fp64 decides among them (option "rerank", default on)
  const bool rerank = _elem == 4 && _optRerank != 0;
  HIP_TRY(growPole(plan));
  const bool matrix = wantPriorities || rerank || plan.poleBytes > 0;   // (the fix corrects the priority matrix, the pick reads it)
  if (matrix) HIP_TRY(grow((void **)&c.dPriT, c.priTBytes, (size_t)_Q * (size_t)plan.Bp * sizeof(double)));
  if (rerank) HIP_TRY(grow(&c.dRerank, c.rerankBytes, BatchRerankScratchBytes()));
  HIP_TRY(LaunchEvalBatch(kb, c.dSlots, (int)n, &plan, c.dPT, c.dAcc, c.dRecs, matrix ? c.dPriT : nullptr, 0, tag,
                          false, _stream, rerank));
  if (rerank) HIP_TRY(LaunchBatchRerank(kb, c.dSlots, (int)n, plan.Bp, c.dPriT, c.dRerank, 0, tag, _stream));
  c.lastBp = plan.Bp;
  if (hostPriorities) return copyToHost(c.dPriT, (size_t)_Q * (size_t)plan.Bp);
  return Error();
}

// A batched selection in two halves, so that a caller driving several engines (sharded_engine.cpp) has every engine's sweep in
// flight before it waits for the first: EnqueueBatch validates, stages the quizzes' slots and launches (nothing is waited for),
// CollectBatch* wait for that launch's flags.  The batch staging buffers are the engine's: one batch at a time between the two.
Error HipEngine::EnqueueBatchLocked(int64_t n, const int64_t *pQuizzes, bool wantPriorities, uint64_t *pTag) {
  Error err = CheckRegular("compute next questions");
  if (!err.ok()) return err;
  if (n < 0 || n > kMaxBatch)
    return Error::MakeP(ErrCode::IndexOutOfRange, RangeParams(n, 0, kMaxBatch), "Batch size is out of range.");
  *pTag = 0;
  if (n == 0) return Error();
  if (!pQuizzes) return Error::Make(ErrCode::NullArgument, "Nullptr is passed in place of a batch buffer.");
  hipSetDevice(_device);
  err = FlushUpdates();
  if (!err.ok()) return err;
  const uint64_t tag = NextLaunchTag();
  err = BatchSweep(_ctx[0], n, pQuizzes, _batchQuizzes, wantPriorities, tag);
  if (!err.ok()) return err;
  *pTag = tag;
  return Error();
}

Error HipEngine::CollectBatchSelectionsLocked(int64_t n, uint64_t tag, CiHipSelection *pOut) {
  if (n == 0) return Error();
  if (!pOut) return Error::Make(ErrCode::NullArgument, "Nullptr is passed in place of a batch buffer.");
  hipSetDevice(_device);
  BatchCtx &c = _ctx[0];
  Error err = WaitBatchFlags(c, n, tag);
  if (!err.ok()) return err;
  for (int64_t i = 0; i < n; i++) {
    if (c.h->out[i].index == -3) return HipErr(hipErrorLaunchFailure, "batched selection (incomplete sweep)");
    CheckPriority(c.h->out[i].priority, c.h->out[i].index);
    pOut[i]._priority = c.h->out[i].priority;
    pOut[i]._iQuestion = c.h->out[i].index < 0 ? -1 : c.h->out[i].index + _qFirst;
  }
  return Error();
}

Error HipEngine::EnqueueBatch(int64_t n, const int64_t *pQuizzes, bool wantPriorities, uint64_t *pTag) {
  std::lock_guard<EngineMutex> lk(_mu);
  return EnqueueBatchLocked(n, pQuizzes, wantPriorities, pTag);
}

Error HipEngine::CollectBatchSelections(int64_t n, uint64_t tag, CiHipSelection *pOut) {
  std::lock_guard<EngineMutex> lk(_mu);
  return CollectBatchSelectionsLocked(n, tag, pOut);
}

// pOut[i * Q + q] (local questions) of the batch enqueued with wantPriorities
Error HipEngine::CollectBatchPriorities(int64_t n, double *pOut) {
  std::lock_guard<EngineMutex> lk(_mu);
  return CollectBatchPrioritiesLocked(n, pOut);
}

Error HipEngine::CollectBatchPrioritiesLocked(int64_t n, double *pOut) {
  if (n == 0) return Error();
  if (!pOut) return Error::Make(ErrCode::NullArgument, "Nullptr is passed in place of a batch buffer.");
  hipSetDevice(_device);
  BatchCtx &c = _ctx[0];
  if (c.lastBp == 0) {   // grid.y = quiz: every quiz's own vector
    HIP_TRY(hipMemcpyAsync(pOut, c.dPriority, (size_t)n * (size_t)_Q * sizeof(double), hipMemcpyDeviceToHost, _stream));
    HIP_TRY(hipStreamSynchronize(_stream));
    return Error();
  }
  std::vector<double> host((size_t)_Q * (size_t)c.lastBp);
  HIP_TRY(hipMemcpyAsync(host.data(), c.dPriT, host.size() * sizeof(double), hipMemcpyDeviceToHost, _stream));
  HIP_TRY(hipStreamSynchronize(_stream));
  for (int64_t i = 0; i < n; i++)
    for (int64_t q = 0; q < _Q; q++) pOut[(size_t)i * (size_t)_Q + (size_t)q] = host[(size_t)q * (size_t)c.lastBp + (size_t)i];
  return Error();
}

Error HipEngine::NextQuestionArgmaxBatch(int64_t n, const int64_t *pQuizzes, int64_t *pOut) {
  std::lock_guard<std::mutex> selLk(_ctx[0].mu);   // (this context's staging buffers: not while a leader's combined sweep uses them)
  std::lock_guard<EngineMutex> lk(_mu);
  if (n > 0 && !pOut) return Error::Make(ErrCode::NullArgument, "Nullptr is passed in place of a batch buffer.");
  uint64_t tag = 0;
  Error err = EnqueueBatchLocked(n, pQuizzes, false, &tag);
  if (!err.ok() || n == 0) return err;
  std::vector<CiHipSelection> sel((size_t)n);
  err = CollectBatchSelectionsLocked(n, tag, sel.data());
  if (!err.ok()) return err;
  for (int64_t i = 0; i < n; i++) {
    Error e;   // -1 + QuestionsExhausted: reported as -1 only
    pOut[i] = FinishSelection(e, _batchQuizzes[(size_t)i], sel[(size_t)i]._iQuestion < 0 ? -1 : sel[(size_t)i]._iQuestion - _qFirst);
  }
  return Error();
}

// The batch's local winners without the bookkeeping of NextQuestion: pOut[i] = {priority, GLOBAL question index or -1} of
// pQuizzes[i] over this engine's questions -- what a host that shards the question axis exchanges between the shards before it
// sets the active questions (PqaEngine_SetActiveQuestion).
Error HipEngine::SelectArgmaxBatch(int64_t n, const int64_t *pQuizzes, CiHipSelection *pOut) {
  std::lock_guard<std::mutex> selLk(_ctx[0].mu);   // (this context's staging buffers: not while a leader's combined sweep uses them)
  std::lock_guard<EngineMutex> lk(_mu);
  if (n > 0 && !pOut) return Error::Make(ErrCode::NullArgument, "Nullptr is passed in place of a batch buffer.");
  uint64_t tag = 0;
  Error err = EnqueueBatchLocked(n, pQuizzes, false, &tag);
  if (!err.ok() || n == 0) return err;
  return CollectBatchSelectionsLocked(n, tag, pOut);
}

Error HipEngine::WaitBatchFlags(BatchCtx &c, int64_t n, uint64_t tag) {
  SpinWait w;
  for (int64_t i = 0; i < n; i++) {
    volatile uint64_t *flag = &c.h->seq[i];
    while (*flag != tag) {
      if (!w.Tick(std::chrono::seconds(600))) return HipErr(hipErrorNotReady, "batched selection (timeout)");
      if (w.Due() && hipStreamQuery(_stream) == hipSuccess && *flag != tag) {  // the kernel retired without publishing
        const hipError_t he = hipStreamSynchronize(_stream);
        if (he != hipSuccess || *flag != tag) return HipErr(he == hipSuccess ? hipErrorUnknown : he, "batched selection (result flag)");
      }
    }
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  return Error();
}

// The priority vectors of n quizzes from ONE row-sharing sweep: pOut[i * Q + q] = priority of local question q for quiz
// pQuizzes[i] (0 for gap / asked questions).  The deterministic output of the batched path, for parity checks.
Error HipEngine::EvalPrioritiesBatch(int64_t n, const int64_t *pQuizzes, double *pOut) {
  std::lock_guard<std::mutex> selLk(_ctx[0].mu);   // (this context's staging buffers: not while a leader's combined sweep uses them)
  std::lock_guard<EngineMutex> lk(_mu);
  if (n > 0 && !pOut) return Error::Make(ErrCode::NullArgument, "Nullptr is passed in place of a batch buffer.");
  uint64_t tag = 0;
  Error err = EnqueueBatchLocked(n, pQuizzes, true, &tag);
  if (!err.ok() || n == 0) return err;
  return CollectBatchPrioritiesLocked(n, pOut);
}

// The same selection replayed from a HIP graph (option "use_graph"; SURVEY 8(d) asks for the variant).  One graph per quiz:
// a single kernel node, the fused sweep with CONSTANT arguments -- the per-launch tag lives in a device word that the
// sweep's finisher advances (FusedSelect::tagCell), and the host mirrors the count.  Own record strip and tag cell, so
// graph replays and plain launches never share tags.
int64_t HipEngine::NextQuestionArgmaxGraph(Error &err, Quiz *q) {
  if (!_dGraphScratch) {
    hipError_t he = hipMalloc(&_dGraphScratch, kFusedMaxGrid * sizeof(SelectResult));
    if (he == hipSuccess) he = hipMemsetAsync(_dGraphScratch, 0, kFusedMaxGrid * sizeof(SelectResult), _stream);
    if (he == hipSuccess) he = hipMalloc(&_dTagCell, sizeof(uint64_t));
    const uint64_t one = kGraphFlagBase + 1;
    if (he == hipSuccess) he = hipMemcpyAsync(_dTagCell, &one, sizeof(one), hipMemcpyHostToDevice, _stream);
    if (he == hipSuccess) he = hipStreamSynchronize(_stream);
    if (he != hipSuccess) { err = HipErr(he, "graph selection buffers"); return -1; }
    _graphTag = one;
  }
  auto it = _graphs.find(q);
  if (it == _graphs.end() || it->second.variant != _optEvalVariant || it->second.stream != _stream ||
      it->second.kbVersion != _kbVersion) {
    if (it != _graphs.end()) { hipGraphExecDestroy(it->second.exec); _graphs.erase(it); }
    const FusedSelect fs{_dGraphScratch, &_hPinned->sel, &_hPinned->seq, 0, 0, 0, 0, _dTagCell, 0, 0, nullptr, nullptr};
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    hipError_t he = hipStreamBeginCapture(_stream, hipStreamCaptureModeThreadLocal);
    if (he == hipSuccess) {
      const hipError_t le = LaunchEvalQuestions(View(), q->dPrior, q->dAsked, 0, _Q, _dPriority, (int)_optEvalVariant, &fs, _stream);
      he = hipStreamEndCapture(_stream, &graph);
      if (he == hipSuccess) he = le;
    }
    if (he == hipSuccess) he = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    if (graph) hipGraphDestroy(graph);
    if (he != hipSuccess) { err = HipErr(he, "graph capture of the selection"); return -1; }
    it = _graphs.emplace(q, GraphEntry{exec, _optEvalVariant, _stream, _kbVersion}).first;
  }
  const uint64_t expect = _graphTag;
  StopServer();   // a launched sweep has no room beside the resident one and would wait for it to idle out
  err = SettlePoleList();
  if (!err.ok()) return -1;
  const hipError_t he = hipGraphLaunch(it->second.exec, _stream);
  if (he != hipSuccess) { err = HipErr(he, "hipGraphLaunch"); return -1; }
  uint64_t next = _graphTag + 1;                     // the finisher's own rule (fused_select)
  if ((uint32_t)next == 0) next++;
  _graphTag = next;
  err = WaitFlag(&_hPinned->seq, expect, "NextQuestionArgmax (graph)");
  if (!err.ok()) return -1;
  if (_hPinned->sel.index == -3) { err = HipErr(hipErrorLaunchFailure, "NextQuestionArgmax (incomplete sweep)"); return -1; }
  CheckPriority(_hPinned->sel.priority, _hPinned->sel.index);
  return FinishSelection(err, q, _hPinned->sel.index);
}

int64_t HipEngine::NextQuestionSampled(Error &err, int64_t iQuiz, uint64_t rnd) { return Combine(err, iQuiz, 1, rnd); }

int64_t HipEngine::NextQuestionSampledLocked(Error &err, int64_t iQuiz, uint64_t rnd) {
  err = CheckRegular("compute next question");
  if (!err.ok()) return -1;
  Quiz *q = UseQuiz(err, iQuiz);
  if (!q) return -1;
  hipSetDevice(_device);
  err = FlushUpdates();
  if (!err.ok()) return -1;
  const KbView kb = View();
  const int64_t nSub = _optEvalSubtasks ? _optEvalSubtasks : 8 * _optWorkers;  // reference PqaCore/CpuEngine.cpp:339
  if (_optServer && !q->noServer && _optHostSampled && !_optFusedSampled && ServerUsable()) {
    // resident sweep: post the request with the hand-over mark, poll the flag, select on the host -- no launch on the path
    const uint64_t value = kServerFlagBase | ++_opSeq;   // (its own range: see kGraphFlagBase)
    err = ServerPost(q, &_hPinned->sel, &_hPinned->seq, value, (int64_t)kServerHandOver);
    if (err.ok()) err = ServerWait(&_hPinned->seq, value, "NextQuestionSampled");
    if (!err.ok()) return -1;
    if (_hPinned->sel.index == -3) { err = HipErr(hipErrorLaunchFailure, "NextQuestionSampled (incomplete sweep)"); return -1; }
    if (_hPinned->sel.index != -4) {
      err = CollectHostPriority(_serverPosted, q);
      if (!err.ok()) return -1;
      const int64_t sel = SelectSampledHostBits(_hostRun.data(), _Q, nSub, rnd, _hQGap.data(), q->hAsked.data());
      return FinishSelection(err, q, sel);
    }
    q->noServer = true;   // (as NextQuestionArgmaxLocked)
  }
  uint64_t specTag = 0;
  const int took = TakeSpeculation(q, (1 << 2) | (1 << 3), &specTag);   // 2 / 3: RecordAnswer has launched the sweep already
  const bool speculated = took == 2;
  if (took == 0) StopServer();   // a launched sweep has no room beside the resident one and would wait for it to idle out
  if (took != 3 && _optHostSampled && !_optFusedSampled && _elem == 8 && EvalVariantHasFinisherWorkgroup(kb, (int)_optEvalVariant)) {
    // ONE launch, and the selection on the host: the sweep's finisher workgroup copies the finished priority vector (8 bytes per
    // question) into host-coherent memory and sets the flag; the selector's O(Q) scalar Kahan steps take the host a few
    // microseconds -- less than the dispatch of the selector kernel they replace.
    const hipError_t ae = EnsureHostPriority();
    if (ae != hipSuccess) { err = HipErr(ae, "host priority buffer"); return -1; }
    uint64_t seq = specTag;
    FusedSelect fs{};
    if (speculated) fs = _spec.fs;
    else {
      { Error se = SettlePoleList(); if (!se.ok()) { err = se; return -1; } }
      seq = NextLaunchTag();
      fs = FusedSelect{_dSelScratch, &_hPinned->sel, &_hPinned->seq, seq, 0, 0, seq, nullptr, 1, 0, nullptr, _hHostPriority, LazyFix() && q->lateStreak < _optLateEager ? 1 : 0};   // (a quiz whose last selections all needed the fix: launched behind the sweep again)
      const hipError_t he = LaunchEvalQuestions(kb, q->dPrior, q->dAsked, 0, _Q, _dPriority, (int)_optEvalVariant, &fs, _stream);
      if (he != hipSuccess) { err = HipErr(he, "NextQuestionSampled"); return -1; }
      if (fs.lazyFix) _poleListPending = true;   // (as NextQuestionArgmaxLocked)
    }
    err = WaitFlag(&_hPinned->seq, seq, "NextQuestionSampled");
    if (!err.ok()) return -1;
    if (fs.lazyFix) {
      if (_hPinned->sel.index == -4) {   // (as NextQuestionArgmaxLocked: the corrected entries carry the sweep's tag)
        err = RunLazyFix(q, fs, "NextQuestionSampled");
        if (!err.ok()) return -1;
        q->lateStreak++;
      } else {
        q->lateStreak = 0;
        if (_hPinned->sel.index != -3) _poleListPending = false;
      }
    }
    if (_hPinned->sel.index == -3) { err = HipErr(hipErrorLaunchFailure, "NextQuestionSampled (incomplete sweep)"); return -1; }
    err = CollectHostPriority(seq, q);
    if (!err.ok()) return -1;
    const int64_t sel = SelectSampledHostBits(_hostRun.data(), _Q, nSub, rnd, _hQGap.data(), q->hAsked.data());
    return FinishSelection(err, q, sel);
  }
  if (took != 3 && _optFusedSampled && _elem == 8 && EvalVariantFusesSampled(kb, (int)_optEvalVariant, nSub)) {
    // ONE launch: the sweep's finisher workgroup runs the reference's selector once every workgroup has reported
    { Error se = SettlePoleList(); if (!se.ok()) { err = se; return -1; } }
    const uint64_t seq = NextLaunchTag();
    const FusedSelect fs{_dSelScratch, &_hPinned->sel, &_hPinned->seq, seq, 0, 0, seq, nullptr, nSub, rnd, _dRunLength, nullptr};
    const hipError_t he = LaunchEvalQuestions(kb, q->dPrior, q->dAsked, 0, _Q, _dPriority, (int)_optEvalVariant, &fs, _stream);
    if (he != hipSuccess) { err = HipErr(he, "NextQuestionSampled"); return -1; }
    err = WaitFlag(&_hPinned->seq, seq, "NextQuestionSampled");
    if (!err.ok()) return -1;
    if (_hPinned->sel.index == -3) { err = HipErr(hipErrorLaunchFailure, "NextQuestionSampled (incomplete sweep)"); return -1; }
    CheckPriority(_hPinned->sel.priority, _hPinned->sel.index);
  return FinishSelection(err, q, _hPinned->sel.index);
  }
  if (took != 3) {   // (else: the priorities are in _dPriority already, or on their way there in stream order)
    err = LaunchSingleSweep(q, nullptr);
    if (!err.ok()) return -1;
  }
  hipError_t he = hipSuccess;
  const uint64_t op = ++_opSeq;  // the selector writes its record and then this number into host-coherent memory
  if (he == hipSuccess)
    he = LaunchSelectSampled(_dPriority, _dQGap, q->dAsked, 0, _Q, nSub, rnd, _dRunLength, &_hPinned->sel, &_hPinned->opFlag,
                             op, _stream);
  if (he != hipSuccess) { err = HipErr(he, "NextQuestionSampled"); return -1; }
  err = WaitFlag(&_hPinned->opFlag, op, "NextQuestionSampled");
  if (!err.ok()) return -1;
  CheckPriority(_hPinned->sel.priority, _hPinned->sel.index);
  return FinishSelection(err, q, _hPinned->sel.index);
}

int64_t HipEngine::NextQuestion(Error &err, int64_t iQuiz) {
  if (_optSelect == 1) return Combine(err, iQuiz, 0, 0);
  uint64_t rnd;
  { std::lock_guard<std::mutex> lk(_rngMu); rnd = NextRandom(); }   // (drawn when the call arrives, whatever sweep serves it)
  return Combine(err, iQuiz, 1, rnd);
}

}  // namespace pqa
