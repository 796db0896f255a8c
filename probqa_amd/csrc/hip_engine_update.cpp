// hip_engine_update.cpp -- HipEngine: what changes state -- RecordAnswer and its deferred posterior updates, speculation,
// ListTopTargets, training (hip_engine.h).
#include "hip_engine_internal.h"

namespace pqa {
// ------------------------------------------------------------------------------------------------------------------
// RecordAnswer and friends
// ------------------------------------------------------------------------------------------------------------------
Error HipEngine::RecordAnswerImpl(int64_t iQuiz, int64_t iAnswer, bool remote) {
  CallScope scope(_activeCallers);
  if (_optCombine && (_optPostAlways || !_mu.try_lock())) {   // somebody is inside the engine: it runs this call's bookkeeping on its way out
    PostedOp op;
    op.kind = 1; op.iQuiz = iQuiz; op.arg = iAnswer; op.remote = remote;
    RunPosted(op);
    return op.err;
  }
  std::unique_lock<EngineMutex> lk(_mu, std::defer_lock);
  if (_optCombine) lk = std::unique_lock<EngineMutex>(_mu, std::adopt_lock);
  else lk.lock();
  return RecordAnswerLocked(iQuiz, iAnswer, remote, !Concurrent());
}

// Several quizzes' answers in one call and ONE launch (grid.x = quiz: record_answer_batch_kernel; every quiz's posterior is the
// one RecordAnswer gives it, bit for bit -- the same workgroup code and summation order).  Quiz i must have an active question
// (NextQuestion / SetActiveQuestion).  An invalid entry fails the call; the entries before it stay recorded.
Error HipEngine::RecordAnswerBatch(int64_t n, const int64_t *pQuizzes, const int64_t *pAnswers) {
  if (n < 0) return Error::MakeP(ErrCode::NegativeCount, "count=" + std::to_string(n), "|nQuizzes| must be non-negative.");
  if (n > 0 && (!pQuizzes || !pAnswers)) return Error::Make(ErrCode::NullArgument, "Nullptr is passed in place of a batch buffer.");
  CallScope scope(_activeCallers);
  std::lock_guard<EngineMutex> lk(_mu);
  Error first;
  for (int64_t i = 0; i < n && first.ok(); i++) first = RecordAnswerLocked(pQuizzes[i], pAnswers[i], false, false);
  Error fe = FlushUpdates();
  return first.ok() ? fe : first;
}

// n new quizzes and ONE launch for their priors (grid.x = quiz).  All or nothing.
Error HipEngine::StartQuizBatch(int64_t n, int64_t *pQuizzes) {
  if (n < 0) return Error::MakeP(ErrCode::NegativeCount, "count=" + std::to_string(n), "|nQuizzes| must be non-negative.");
  if (n > 0 && !pQuizzes) return Error::Make(ErrCode::NullArgument, "Nullptr is passed in place of a batch buffer.");
  CallScope scope(_activeCallers);
  std::lock_guard<EngineMutex> lk(_mu);
  hipSetDevice(_device);
  StartBatchInline batch;
  batch.n = 0;
  batch.askedWords = (int64_t)BitWords(_Q);
  Error err;
  auto launch = [&]() -> Error {
    if (batch.n == 0) return Error();
    HIP_TRY(LaunchStartQuizBatch(View(), batch, _optWorkers, _stream));
    batch.n = 0;
    return Error();
  };
  int64_t made = 0;
  for (; made < n; made++) {
    _startBatch = &batch;
    pQuizzes[made] = CreateQuiz(err, 0, nullptr, nullptr, nullptr, 0, nullptr);
    _startBatch = nullptr;
    if (pQuizzes[made] < 0) break;
    if (batch.n == kStartInline) { err = launch(); if (!err.ok()) { made++; break; } }
  }
  if (err.ok()) err = launch();
  if (!err.ok()) {   // roll back: the call creates all its quizzes or none
    for (int64_t i = 0; i < made; i++)
      if (pQuizzes[i] >= 0 && (size_t)pQuizzes[i] < _quizzes.size() && _quizzes[(size_t)pQuizzes[i]]) {
        Quiz *q = _quizzes[(size_t)pQuizzes[i]];
        UnassignQuiz(pQuizzes[i]);
        DestroyQuiz(q);
      }
    return err;
  }
  return Error();
}

Error HipEngine::RecordAnswerLocked(int64_t iQuiz, int64_t iAnswer, bool remote, bool flushNow) {
  Error err = CheckRegular("record an answer");
  if (!err.ok()) return err;
  if (iAnswer < 0 || iAnswer >= _K)  // reference PqaCore/BaseEngine.cpp:447-451
    return Error::MakeP(ErrCode::IndexOutOfRange, RangeParams(iAnswer, 0, _K - 1), "Answer index is not in the answer range.");
  Quiz *q = UseQuiz(err, iQuiz);
  if (!q) return err;
  // CEQuiz::RecordAnswer, reference PqaCore/CEQuiz.h:77-122
  const int64_t aq = q->activeQuestion;
  if (aq == -1)
    return Error::MakeP(ErrCode::NoQuizActiveQuestion, "answerId=" + std::to_string(iAnswer),
                        "An attempt to record an answer in a quiz that doesn't have an active question");
  const bool local = aq >= _qFirst && aq < _qFirst + _Q;
  if (aq < 0 || aq >= _qTotal || (local && BitTest(_hQGap, aq - _qFirst)))
    return Error::MakeP(ErrCode::NoQuizActiveQuestion, "answerId=" + std::to_string(iAnswer),
                        "An attempt to record an answer in a quiz that has invalid active question");
  if (local == remote)
    return Error::MakeP(ErrCode::IndexOutOfRange, RangeParams(aq, _qFirst, _qFirst + _Q - 1),
                        remote ? "RecordAnswerRemote on the shard that owns the active question."
                               : "The active question belongs to another shard: use PqaHip_RecordAnswerRemote.");
  ServerQuiesce();
  if (q->updatePending) {   // (a second answer for a quiz whose first is still deferred: that one runs now)
    Error fe = FlushUpdates();
    if (!fe.ok()) return fe;
  }
  q->answers.push_back(AQ{aq, iAnswer});
  q->activeQuestion = -1;
  q->priorVersion++;  // (remote: the caller writes the owner's posterior into the quiz's buffer)
  if (!local) return Error();
  hipSetDevice(_device);
  const int64_t ql = aq - _qFirst;
  BitSet(q->hAsked, ql, true);
  // Alone in the engine, the client's next call but one is NextQuestion: where the sweep's shape allows it, ONE launch updates the
  // posterior and sweeps with it (Speculate with the update: eval_kernels.hip, eval_questions_f64_upd)
  if (flushNow && _pendingUpdates.empty() && Speculate(q, ql, iAnswer)) return Error();
  _pendingUpdates.push_back(PendingUpdate{q, ql, iAnswer});
  _pendingCount.store(_pendingUpdates.size(), std::memory_order_relaxed);
  q->updatePending = true;
  // Other client threads inside the engine: leave the kernel to whoever next needs a posterior -- it runs all the updates that
  // have gathered by then in one launch.  Alone: launch now, and the sweep of the NextQuestion that follows right behind it.
  if (!flushNow) return Error();
  Error fe = FlushUpdates();
  if (!fe.ok()) return fe;
  Speculate(q);
  return Error();
}

// Work has been put on the engine's stream that no completion flag covers: the next request to the resident sweep -- in this hold
// of the lock or a later one -- synchronises the stream first (ServerPost reads wasBusy of the CURRENT hold, busy becomes the next
// hold's wasBusy).
void HipEngine::MarkStreamBusy() {
  _mu.busy = _mu.wasBusy = true;
  _pendingRecordOp = 0;
  _pendingRecordFlag = nullptr;
}

// The deferred RecordAnswers, on the engine's stream: one launch, no copy, no synchronisation -- the kernel also sets the
// question's bit in the quiz's device bitmap and lists the new posterior's best targets into the quiz's own pinned lines (as
// many as ListTopTargets has been asking for lately; every listed target is a round of the kernel's selection, `top_cache` at
// most), and everything that reads a posterior or a bitmap afterwards is ordered behind it on the stream.
Error HipEngine::FlushUpdates() {
  if (_pendingUpdates.empty()) return Error();
  std::vector<PendingUpdate> ups;
  ups.swap(_pendingUpdates);
  _pendingCount.store(0, std::memory_order_relaxed);
  for (PendingUpdate &u : ups) u.q->updatePending = false;
  // A launch that fails leaves its updates (and those behind them) deferred: the host's bookkeeping has advanced and the calls
  // that recorded them have returned, so whoever next needs one of those posteriors gets the error instead of a stale posterior.
  auto requeue = [&](size_t from, hipError_t he, const char *what) {
    (void)hipGetLastError();
    for (size_t i = from; i < ups.size(); i++) ups[i].q->updatePending = true;
    _pendingUpdates.insert(_pendingUpdates.begin(), ups.begin() + (std::ptrdiff_t)from, ups.end());
    _pendingCount.store(_pendingUpdates.size(), std::memory_order_relaxed);
    MarkStreamBusy();
    return HipErr(he, what);
  };
  hipSetDevice(_device);
  ServerQuiesce();
  // NLooseWorkers = max(1, hw - 1): reference PqaCore/CEQuiz.h:98, PqaCore/BaseCpuEngine.cpp:22
  const int64_t nLoose = std::max<int64_t>(1, _optWorkers - 1);
  const int64_t topCount = _T <= 16384 ? std::min<int64_t>(std::min<int64_t>(std::min<int64_t>(_optTopCache, _topWantRecent), kQuizTop), _T) : 0;
  auto listed = [&](Quiz *q, uint64_t op) { q->topOp = op; q->topVersion = q->priorVersion; q->topCount = topCount; };
  auto counted = [&](size_t n) {
    _flushes++;
    _flushedUpdates += n;
    _flushedSinceSweep.fetch_add((int64_t)n, std::memory_order_relaxed);
    if (n > _maxFlush) _maxFlush = n;
  };
  if (ups.size() == 1) {
    const PendingUpdate &u = ups[0];
    const uint64_t op = _opSeq + 1;
    const hipError_t he = LaunchRecordAnswer(View(), u.q->dPrior, u.q->dAsked, u.qLocal, u.iAnswer, nLoose, u.list ? u.q->pin->top : nullptr, &u.q->pin->nOut,
                                             &u.q->pin->topFlag, op, topCount, _stream, u.rowA, u.rowD);
    if (he != hipSuccess) return requeue(0, he, "LaunchRecordAnswer");
    _opSeq = op;
    counted(1);
    if (u.list) listed(u.q, op);
    if (topCount > 0 && u.list) {
      // the kernel stores `op` last: whoever sees it knows that everything enqueued on the stream so far has finished
      _pendingRecordOp = op;
      _pendingRecordFlag = &u.q->pin->topFlag;
      _mu.busy = _mu.wasBusy;   // (busy only if it was before this call: `op` covers this call's launch)
    } else {
      MarkStreamBusy();         // (no flag of this launch to wait for)
    }
    return Error();
  }
  static_assert(kQuizTopDev == kQuizTop && offsetof(QuizPinned, nOut) == kQuizTop * sizeof(RatedTargetDev) &&
                offsetof(QuizPinned, topFlag) == offsetof(QuizPinned, nOut) + 8, "the batched kernel addresses the quiz's lines by layout");
  const KbView kb = View();
  static thread_local RecordBatchInline b;   // (10 KB: not on a client thread's stack for every flush)
  for (size_t first = 0; first < ups.size(); first += kRecordInline) {
    b.n = (int32_t)std::min<size_t>(kRecordInline, ups.size() - first);
    b.topCount = (int32_t)topCount;
    for (int32_t i = 0; i < b.n; i++) {
      const PendingUpdate &u = ups[first + (size_t)i];
      b.s[i] = RecordSlot{u.q->dPrior, u.q->dAsked, u.list ? (void *)u.q->pin : nullptr, (int32_t)u.qLocal, (int32_t)u.iAnswer, _opSeq + 1 + (uint64_t)i, u.rowA, u.rowD};
    }
    const hipError_t he = LaunchRecordAnswerBatch(kb, b, nLoose, _stream);
    if (he != hipSuccess) return requeue(first, he, "LaunchRecordAnswerBatch");
    for (int32_t i = 0; i < b.n; i++) if (ups[first + (size_t)i].list) listed(ups[first + (size_t)i].q, _opSeq + 1 + (uint64_t)i);
    _opSeq += (uint64_t)b.n;
    counted((size_t)b.n);
  }
  // the workgroups of a batched launch finish in any order: no one flag says that the stream is idle -- whoever needs it idle
  // (the resident sweep's request, ServerPost) synchronises the stream, in this hold of the lock as well as in the next
  MarkStreamBusy();
  return Error();
}

// The sweep NextQuestion would launch for `q` now, launched now (see Speculation in hip_engine.h).  Whole-cube engines with the
// launched selection paths only: the resident sweep and graph replay have no launch to move, and shards' selections are driven by
// the sharded engine.  Where the sweep has no finisher that hands its result over (Float engines, long rows), the sampled selector's
// kernel -- it needs the random number -- is launched by NextQuestion over the priorities the speculative sweep left.
// updQuestion >= 0: the answer RecordAnswer has just been given and has NOT launched an update for -- the sweep's launch computes
// the posterior itself (eval_questions_f64_upd: no posterior kernel for the sweep to wait for).  Returns true if that launch was
// made (the posterior, the asked bit and the listing of the best targets are on their way, as FlushUpdates would have them);
// false: nothing was launched for the update, the caller goes the usual way.
bool HipEngine::Speculate(Quiz *q, int64_t updQuestion, int64_t updAnswer) {
  const bool withUpdate = updQuestion >= 0;
  if (!withUpdate) DropSpeculation();   // (one at a time: the hand-over buffers are the engine's)
  if (!_optSpeculate || (_optServer && !q->noServer) || _optUseGraph || _qTotal != _Q || _Q <= 0) return false;   // (a quiz the resident sweep has handed over -- rows at the pole of the lack term -- is served as if there were none)
  if (_optServer && _serverLaunched) return false;   // (a launched sweep has no room beside the resident one: this quiz's last selection has sent it away, unless another quiz called it back)
  if (Concurrent()) return false;   // (several clients: their NextQuestions are served together, by a batched sweep)
  if (withUpdate && (!_optFuseUpdate || _specScore < -4)) return false;
  if (_specScore < -4 && (++_specProbe & 31) != 0) return false;   // the client does not follow RecordAnswer with NextQuestion: probe now and then
  const KbView kb = View();
  int kind = 0;
  if (_optSelect == 1) kind = 1;
  else if (_optHostSampled && !_optFusedSampled && _elem == 8 && EvalVariantHasFinisherWorkgroup(kb, (int)_optEvalVariant)) kind = 2;
  else if (!(_optFusedSampled && _elem == 8)) kind = 3;   // Float engines, long rows: the sweep now, the selector kernel at NextQuestion
  if (kind == 0) return false;
  const int64_t nLoose = std::max<int64_t>(1, _optWorkers - 1);   // reference PqaCore/CEQuiz.h:98, PqaCore/BaseCpuEngine.cpp:22
  if (withUpdate && (kind == 3 || UseClusterSweep() || !EvalFusesUpdate(kb, (int)_optEvalVariant, nLoose))) return false;
  if (kind == 2 && EnsureHostPriority() != hipSuccess) return false;
  if (withUpdate) DropSpeculation();
  const uint64_t seq = NextLaunchTag();
  if (!SettlePoleList().ok()) return false;
  const FusedSelect fs{_dSelScratch, &_hPinned->sel, &_hPinned->seq, seq, 0, 0, seq, nullptr, kind == 2 ? 1 : 0, 0, nullptr,
                       kind == 2 ? _hHostPriority : nullptr, (kind == 1 || kind == 2) && LazyFix() && q->lateStreak < _optLateEager ? 1 : 0};
  if (withUpdate) {
    const int64_t topCount = std::min<int64_t>(std::min<int64_t>(std::min<int64_t>(_optTopCache, _topWantRecent), kQuizTop), _T);
    const uint64_t op = _opSeq + 1;
    if (LaunchEvalQuestionsWithUpdate(kb, q->dPrior, q->dAsked, _dPriority, (int)_optEvalVariant, fs, updQuestion, updAnswer, nLoose, q->pin->top,
                                      &q->pin->nOut, &q->pin->topFlag, op, topCount, _stream) != hipSuccess) {
      (void)hipGetLastError();   // the usual way: posterior kernel, then the sweep
      return false;
    }
    _opSeq = op;
    q->topOp = op; q->topVersion = q->priorVersion; q->topCount = topCount;
    _flushes++; _flushedUpdates++; _fusedUpdates++;
    if (_maxFlush < 1) _maxFlush = 1;
  } else if (kind == 1   ? !LaunchSingleSweep(q, &fs).ok()
             : kind == 3 ? !LaunchSingleSweep(q, nullptr).ok()
                         : LaunchEvalQuestions(kb, q->dPrior, q->dAsked, 0, _Q, _dPriority, (int)_optEvalVariant, &fs, _stream) != hipSuccess) {
    (void)hipGetLastError();   // NextQuestion will launch for itself and report
    return false;
  }
  _spec.quiz = q; _spec.priorVersion = q->priorVersion; _spec.tag = seq; _spec.kind = kind; _spec.fs = fs;
  if (fs.lazyFix) _poleListPending = true;
  _spec.variant = _optEvalVariant; _spec.stream = _stream;
  _pendingRecordOp = 0;   // the posterior kernel's flag no longer says that the stream is idle
  _pendingRecordFlag = nullptr;
  _mu.busy = true;
  return withUpdate;
}

// The kind (and the launch tag to wait for) of the pending speculative sweep if it is exactly a launch a NextQuestion accepting the
// kinds of `kindMask` (bit k: kind k) would make for `q` now -- same quiz and posterior, no fused launch since (they share the
// records and the hand-over buffers) -- else 0, and the speculation is dropped.
int HipEngine::TakeSpeculation(Quiz *q, int kindMask, uint64_t *pTag) {
  if (_spec.quiz == nullptr) return 0;
  const bool match = _spec.quiz == q && ((kindMask >> _spec.kind) & 1) && _spec.priorVersion == q->priorVersion && _spec.tag == _selSeq &&
                     _spec.variant == _optEvalVariant && _spec.stream == _stream && (!_optServer || q->noServer) && !_optUseGraph;
  if (!match) { DropSpeculation(); return 0; }
  _spec.quiz = nullptr;
  _specHits++;
  if (_specScore < 8) _specScore++;
  *pTag = _spec.tag;
  return _spec.kind;
}

Error HipEngine::RecordAnswer(int64_t iQuiz, int64_t iAnswer) { return RecordAnswerImpl(iQuiz, iAnswer, false); }
Error HipEngine::RecordAnswerRemote(int64_t iQuiz, int64_t iAnswer) { return RecordAnswerImpl(iQuiz, iAnswer, true); }

int64_t HipEngine::GetActiveQuestionId(Error &err, int64_t iQuiz) {
  std::lock_guard<EngineMutex> lk(_mu);
  err = CheckRegular("get active question ID for a quiz");
  if (!err.ok()) return -1;
  Quiz *q = UseQuiz(err, iQuiz);
  if (!q) return -1;
  return q->activeQuestion;
}

Error HipEngine::SetActiveQuestion(int64_t iQuiz, int64_t iQuestion) {
  std::lock_guard<EngineMutex> lk(_mu);
  Error err = CheckRegular("set active question ID for a quiz");
  if (!err.ok()) return err;
  Quiz *q = UseQuiz(err, iQuiz);
  if (!q) return err;
  q->activeQuestion = iQuestion;  // unchecked, as reference PqaCore/BaseEngine.cpp:507-508
  return Error();
}

Error HipEngine::GetPriors(int64_t iQuiz, double *pOut, int64_t n) {
  std::lock_guard<EngineMutex> lk(_mu);
  Error err;
  Quiz *q = UseQuiz(err, iQuiz);
  if (!q) return err;
  if (!pOut) return Error::Make(ErrCode::NullArgument, "Nullptr is passed in place of the prior buffer.");
  if (n != _T) return Error::MakeP(ErrCode::IndexOutOfRange, RangeParams(n, _T, _T), "Prior buffer length must equal nTargets.");
  hipSetDevice(_device);
  { Error fe = FlushUpdates(); if (!fe.ok()) return fe; }
  HIP_TRY(hipMemcpyAsync(pOut, q->dPrior, (size_t)_T * sizeof(double), hipMemcpyDeviceToHost, _stream));
  HIP_TRY(hipStreamSynchronize(_stream));
  return Error();
}

Error HipEngine::Log2HotArray(const double *pIn, double *pOut, int64_t n) {
  std::lock_guard<EngineMutex> lk(_mu);
  if (n < 0 || (n > 0 && (!pIn || !pOut))) return Error::Make(ErrCode::NullArgument, "Nullptr is passed in place of a Log2Hot buffer.");
  if (n == 0) return Error();
  hipSetDevice(_device);
  double *dIn = nullptr, *dOut = nullptr;
  HIP_TRY(hipMalloc(&dIn, (size_t)n * sizeof(double)));
  hipError_t he = hipMalloc(&dOut, (size_t)n * sizeof(double));
  if (he == hipSuccess) he = hipMemcpyAsync(dIn, pIn, (size_t)n * sizeof(double), hipMemcpyHostToDevice, _stream);
  if (he == hipSuccess) he = LaunchLog2HotArray(dIn, dOut, n, _stream);
  if (he == hipSuccess) he = hipMemcpyAsync(pOut, dOut, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, _stream);
  if (he == hipSuccess) he = hipStreamSynchronize(_stream);
  hipFree(dIn);
  hipFree(dOut);
  HIP_TRY(he);
  return Error();
}

Error HipEngine::GetPriorDevicePtr(int64_t iQuiz, void **ppDev, int64_t *pLdT) {
  std::lock_guard<EngineMutex> lk(_mu);
  Error err;
  Quiz *q = UseQuiz(err, iQuiz);
  if (!q) return err;
  { Error fe = FlushUpdates(); if (!fe.ok()) return fe; }   // (the caller is about to read the buffer in stream order)
  if (ppDev) *ppDev = q->dPrior;
  if (pLdT) *pLdT = _ldT;
  return Error();
}

// Where the listed probabilities (and the next one below the list) are all different, the listing is the reference's whatever the order
// among equals: the fast listing -- by (probability descending, target ascending) -- with one entry more than asked for says so.  Where
// they tie, the reference's order is its heaps' (CEListTopTargetsAlgorithm.cpp:30-95), and ListTopTargetsExact reproduces them.
static bool ListingHasTies(const CiRatedTarget *got, int64_t nGot, int64_t maxCount, int64_t probed) {
  // (fewer than probed came back: every candidate is listed -- only ties among them matter)
  (void)probed;
  for (int64_t i = 0; i + 1 < nGot; i++)
    if (got[i]._prob == got[i + 1]._prob && i < maxCount) return true;
  return false;
}

int64_t HipEngine::ListTopTargets(Error &err, int64_t iQuiz, int64_t maxCount, CiRatedTarget *pDest) {
  if (maxCount <= 0 || pDest == nullptr || !_optTopExact) return ListTopTargetsFast(err, iQuiz, maxCount, pDest);
  const int64_t probed = std::min<int64_t>(maxCount, _T) + 1;   // (one beyond the list: the boundary)
  CiRatedTarget small[kQuizTop + 1];
  std::vector<CiRatedTarget> large;
  CiRatedTarget *tmp = small;
  if (probed > kQuizTop + 1) { large.resize((size_t)probed); tmp = large.data(); }
  const int64_t n = ListTopTargetsFast(err, iQuiz, probed, tmp);
  if (n < 0) return n;
  if (!ListingHasTies(tmp, n, maxCount, probed)) {
    const int64_t take = std::min(n, maxCount);
    std::memcpy(pDest, tmp, (size_t)take * sizeof(CiRatedTarget));
    return take;
  }
  return ListTopTargetsExact(err, iQuiz, maxCount, pDest);
}

// The reference's listing where probabilities tie: on the device for lists of up to 256 targets (kb_kernels.hip: LaunchTopTargetsExact),
// on the host beyond (the posterior copied: a bulk export, as ListTopTargetsOnHost).
int64_t HipEngine::ListTopTargetsExact(Error &err, int64_t iQuiz, int64_t maxCount, CiRatedTarget *pDest) {
  CallScope scope(_activeCallers);
  std::lock_guard<EngineMutex> lk(_mu);
  err = CheckRegular("list top targets");
  if (!err.ok()) return -1;
  Quiz *q = UseQuiz(err, iQuiz);
  if (!q) return -1;
  hipSetDevice(_device);
  err = FlushUpdates();
  if (!err.ok()) return -1;
  const int64_t want = std::min<int64_t>(maxCount, _T);
  if (want > 256) return ListTopTargetsOnHost(err, q, want, pDest, true);
  StopServer();   // (workgroups of up to 128 KB of LDS: no room beside a resident sweep, they would wait for it to idle out)
  err = EnsureTopExactScratch(1, want);
  if (!err.ok()) return -1;
  TopBatchPriors pr;
  pr.prior[0] = q->dPrior;
  const uint64_t op = ++_opSeq;
  const hipError_t he = LaunchTopTargetsExact(View(), pr, 1, _optWorkers, want, _dTopExact, _hPinned->top, &_hPinned->nOut, &_hPinned->topFlag, op, _stream);
  if (he != hipSuccess) { err = HipErr(he, "ListTopTargets"); return -1; }
  err = WaitFlag(&_hPinned->topFlag, op, "ListTopTargets");
  if (!err.ok()) return -1;
  _mu.busy = false;
  _pendingRecordOp = 0;
  _topExactListings++;
  const int64_t n = std::min<int64_t>(_hPinned->nOut, want);
  std::memcpy(pDest, _hPinned->top, (size_t)n * sizeof(RatedTargetDev));
  return n;
}

Error HipEngine::EnsureTopExactScratch(int64_t nQuizzes, int64_t want) {
  const size_t need = TopExactScratchBytes(_T, _optWorkers, want, nQuizzes);
  if (need <= _topExactBytes) return Error();
  HIP_TRY(hipStreamSynchronize(_stream));
  if (_dTopExact) hipFree(_dTopExact);
  _dTopExact = nullptr; _topExactBytes = 0;
  HIP_TRY(hipMalloc(&_dTopExact, need));
  _topExactBytes = need;
  return Error();
}

int64_t HipEngine::ListTopTargetsFast(Error &err, int64_t iQuiz, int64_t maxCount, CiRatedTarget *pDest) {
  CallScope scope(_activeCallers);
  std::unique_lock<EngineMutex> lk(_mu, std::defer_lock);
  if (!_optCombine || maxCount <= 0 || pDest == nullptr) lk.lock();
  else if (!_optPostAlways && _mu.try_lock()) lk = std::unique_lock<EngineMutex>(_mu, std::adopt_lock);
  else {
    // somebody is inside the engine: it launches what this call needs on its way out (the quiz's deferred update among all that
    // have gathered, the listing if the update kernel has not made it); the wait for the quiz's own lines is this thread's
    PostedOp op;
    op.kind = 2; op.iQuiz = iQuiz; op.arg = maxCount;
    RunPosted(op);
    if (op.result != -2) {
      err = op.err;
      if (!err.ok() || op.result < 0) return -1;
      err = WaitFlagNapping(&op.pin->topFlag, op.flagOp, "ListTopTargets");
      if (!err.ok()) return -1;
      const int64_t n = std::min<int64_t>(op.pin->nOut, op.result);
      std::memcpy(pDest, op.pin->top, (size_t)n * sizeof(RatedTargetDev));
      return n;
    }
    lk.lock();   // (a list longer than the quiz's lines hold)
  }
  err = CheckRegular("list top targets");
  if (!err.ok()) return -1;
  Quiz *q = UseQuiz(err, iQuiz);
  if (!q) return -1;
  if (maxCount <= 0) return 0;
  if (!pDest) { err = Error::Make(ErrCode::NullArgument, "Nullptr is passed in place of the destination."); return -1; }
  hipSetDevice(_device);
  if (q->updatePending && _optLingerUs > 0 && Concurrent()) {
    // Group commit.  This quiz's RecordAnswer is deferred, and the clients that got their questions from the same combined sweep
    // are recording their answers right now: give them a moment, so that ONE launch runs all of them.  Whoever comes out of the
    // wait first launches; the others find their update on its way.
    const size_t target = (size_t)std::max<int64_t>(2, std::min<int64_t>(_lastCombined.load(std::memory_order_relaxed), _activeCallers.load(std::memory_order_relaxed) - 1));
    if (_pendingUpdates.size() < target) {
      lk.unlock();
      const auto t0 = std::chrono::steady_clock::now();
      const auto limit = std::chrono::microseconds(_optLingerUs);
      for (;;) {
        const size_t have = _pendingCount.load(std::memory_order_relaxed);
        if (have == 0 || have >= target) break;   // (0: somebody has launched them)
        for (int i = 0; i < 32; i++) _mm_pause();
        if (std::chrono::steady_clock::now() - t0 > limit) break;
      }
      lk.lock();
      q = UseQuiz(err, iQuiz);
      if (!q) return -1;
    }
  }
  err = FlushUpdates();   // (this quiz's RecordAnswer, and whatever other quizzes' have gathered, in one launch)
  if (!err.ok()) return -1;
  const int64_t want = std::min<int64_t>(maxCount, _T);
  _topWantRecent = want >= _topWantRecent ? want : want + (_topWantRecent - want) * 7 / 8;   // (decays towards smaller requests)
  static_assert(sizeof(RatedTargetDev) == sizeof(CiRatedTarget), "listed straight into the caller's layout");
  if (want <= kQuizTop && _T <= 16384) {  // (the kernel keeps every target in registers: 16 per thread at most)
    // the kernel lists straight into the quiz's host-coherent lines and then stores the operation number: no copy, no synchronise
    const bool cached = q->topOp != 0 && q->topVersion == q->priorVersion && want <= q->topCount;
    if (!cached) {
      const uint64_t op = ++_opSeq;
      const hipError_t he = LaunchTopTargets(View(), q->dPrior, want, q->pin->top, &q->pin->nOut, &q->pin->topFlag, op, _stream);
      if (he != hipSuccess) { err = HipErr(he, "ListTopTargets"); return -1; }
      q->topOp = op; q->topVersion = q->priorVersion; q->topCount = want;
    }
    QuizPinned *pin = q->pin;
    const uint64_t op = q->topOp;
    if (Concurrent()) {
      // other clients are inside the engine: wait with the engine open to them (the lines are this quiz's own)
      lk.unlock();
      err = WaitFlagNapping(&pin->topFlag, op, "ListTopTargets");
      if (!err.ok()) return -1;
      const int64_t n = std::min<int64_t>(pin->nOut, want);
      std::memcpy(pDest, pin->top, (size_t)n * sizeof(RatedTargetDev));
      return n;
    }
    err = WaitFlag(&pin->topFlag, op, "ListTopTargets");
    if (!err.ok()) return -1;
    // what was waited for was the newest work on the stream (this call's own launch, or RecordAnswer's kernel with nothing
    // enqueued behind it): the stream is idle.  Otherwise this call has added nothing to it.
    if (!cached || (_pendingRecordOp == op && !_mu.wasBusy)) { _mu.busy = false; _pendingRecordOp = 0; }
    else _mu.busy = _mu.wasBusy;
    const int64_t n = std::min<int64_t>(pin->nOut, want);
    std::memcpy(pDest, pin->top, (size_t)n * sizeof(RatedTargetDev));
    return n;
  }
  if (want <= 256 && _T <= 16384) {   // longer lists: the engine's own lines, the engine held while the kernel runs
    const uint64_t op = ++_opSeq;
    const hipError_t he = LaunchTopTargets(View(), q->dPrior, want, _hPinned->top, &_hPinned->nOut, &_hPinned->topFlag, op, _stream);
    if (he != hipSuccess) { err = HipErr(he, "ListTopTargets"); return -1; }
    err = WaitFlag(&_hPinned->topFlag, op, "ListTopTargets");
    if (!err.ok()) return -1;
    _mu.busy = false;   // (this call's own launch was the newest work on the stream)
    _pendingRecordOp = 0;
    const int64_t n = std::min<int64_t>(_hPinned->nOut, want);
    std::memcpy(pDest, _hPinned->top, (size_t)n * sizeof(RatedTargetDev));
    return n;
  }
  if (want <= 256) {   // rows beyond one workgroup's registers: chunk lists and their merge (kb_kernels.hip), the engine's own lines
    err = EnsureTopScratch(1, want);
    if (!err.ok()) return -1;
    TopBatchPriors pr;
    pr.prior[0] = q->dPrior;
    const uint64_t op = ++_opSeq;
    const hipError_t he = LaunchTopTargetsBatch(View(), pr, 1, want, _dTopScratch[0], _dTopScratch[1], _hPinned->top, &_hPinned->nOut, &_hPinned->topFlag, op, _stream);
    if (he != hipSuccess) { err = HipErr(he, "ListTopTargets"); return -1; }
    err = WaitFlag(&_hPinned->topFlag, op, "ListTopTargets");
    if (!err.ok()) return -1;
    _mu.busy = false;   // (this call's own launches were the newest work on the stream)
    _pendingRecordOp = 0;
    const int64_t n = std::min<int64_t>(_hPinned->nOut, want);
    std::memcpy(pDest, _hPinned->top, (size_t)n * sizeof(RatedTargetDev));
    return n;
  }
  return ListTopTargetsOnHost(err, q, want, pDest);
}

// Lists of more than 256 targets: sort on the host (a listing of that length is the caller's bulk export, not a quiz step).
int64_t HipEngine::ListTopTargetsOnHost(Error &err, Quiz *q, int64_t want, CiRatedTarget *pDest, bool referenceOrder) {
  std::vector<double> pri((size_t)_T);
  hipError_t he = hipMemcpyAsync(pri.data(), q->dPrior, (size_t)_T * sizeof(double), hipMemcpyDeviceToHost, _stream);
  if (he == hipSuccess) he = hipStreamSynchronize(_stream);
  if (he != hipSuccess) { err = HipErr(he, "ListTopTargets"); return -1; }
  _mu.busy = false;
  _pendingRecordOp = 0;
  if (referenceOrder) {
    // CEListTopTargetsAlgorithm::RunHeapifyBased (CEListTopTargetsAlgorithm.cpp:30-95) as written, with the C++ library's own heap calls
    struct Rated { int64_t t; double p; bool operator<(const Rated &o) const { return p < o.p; } };
    struct Head { double p; int64_t piece; bool operator<(const Head &o) const { return p < o.p; } };
    const int64_t W = _optWorkers, quot = _T / W, rem = _T % W, nSub = quot == 0 ? rem : W;
    std::vector<Rated> ratings((size_t)_T);
    std::vector<int64_t> start((size_t)nSub), lim((size_t)nSub);
    for (int64_t i = 0; i < nSub; i++) {
      const int64_t first = i * quot + std::min(i, rem), limit = first + quot + (i < rem ? 1 : 0);
      int64_t sel = first;
      for (int64_t t = first; t < limit; t++)
        if (!BitTest(_hTGap, t) && pri[(size_t)t] > 0.0) ratings[(size_t)sel++] = Rated{t, pri[(size_t)t]};   // CEHeapifyPriorsSubtaskMake.cpp:42-52
      start[(size_t)i] = first; lim[(size_t)i] = sel;
      std::make_heap(ratings.begin() + first, ratings.begin() + sel);                                        // :87
    }
    std::vector<Head> head;
    for (int64_t i = 0; i < nSub; i++) if (lim[(size_t)i] != start[(size_t)i]) head.push_back(Head{ratings[(size_t)start[(size_t)i]].p, i});
    std::make_heap(head.begin(), head.end());
    int64_t listed = 0;
    for (; listed < want && !head.empty(); listed++) {
      const int64_t piece = head.front().piece, ps = start[(size_t)piece];
      pDest[listed]._iTarget = ratings[(size_t)ps].t; pDest[listed]._prob = head.front().p;
      if (ps + 1 == lim[(size_t)piece]) { std::pop_heap(head.begin(), head.end()); head.pop_back(); continue; }
      std::pop_heap(ratings.begin() + ps, ratings.begin() + lim[(size_t)piece]);
      lim[(size_t)piece]--;
      head.front().p = ratings[(size_t)ps].p;
      // SRHeapHelper::Down (SRHeap.h:16-39)
      size_t cur = 0;
      for (;;) {
        const size_t c1 = 2 * cur + 1;
        if (c1 >= head.size()) break;
        const size_t c2 = c1 + 1;
        if (c2 >= head.size()) { if (head[cur] < head[c1]) std::swap(head[cur], head[c1]); break; }
        const size_t hi = head[c2] < head[c1] ? c1 : c2;
        if (!(head[cur] < head[hi])) break;
        std::swap(head[cur], head[hi]);
        cur = hi;
      }
    }
    return listed;
  }
  std::vector<int64_t> idx;
  idx.reserve((size_t)_T);
  // gaps and probabilities <= 0 are no candidates (reference PqaCore/CEHeapifyPriorsSubtaskMake.cpp:43-49)
  for (int64_t t = 0; t < _T; t++) if (!BitTest(_hTGap, t) && pri[(size_t)t] > 0.0) idx.push_back(t);
  const int64_t n = std::min<int64_t>(want, (int64_t)idx.size());
  std::partial_sort(idx.begin(), idx.begin() + n, idx.end(),
                    [&](int64_t a, int64_t b) { return pri[a] > pri[b] || (pri[a] == pri[b] && a < b); });
  for (int64_t i = 0; i < n; i++) { pDest[i]._iTarget = idx[i]; pDest[i]._prob = pri[idx[i]]; }
  return n;
}

// The device scratch of the chunked listing (two buffers of candidate lists) for nQuizzes quizzes at once.
Error HipEngine::EnsureTopScratch(int64_t nQuizzes, int64_t want) {
  const int64_t need = nQuizzes * TopBatchScratchRecords(_T, want);
  if (need <= _topScratchRecords) return Error();
  HIP_TRY(hipStreamSynchronize(_stream));   // (nothing in flight reads the buffers about to go)
  for (int i = 0; i < 2; i++) { if (_dTopScratch[i]) hipFree(_dTopScratch[i]); _dTopScratch[i] = nullptr; }
  _topScratchRecords = 0;
  for (int i = 0; i < 2; i++) HIP_TRY(hipMalloc((void **)&_dTopScratch[i], (size_t)need * sizeof(RatedTargetDev)));
  _topScratchRecords = need;
  return Error();
}

// ListTopTargets for n quizzes with one launch sequence per 256 of them: pDest[i * maxCount + j], j < pCounts[i], is quiz
// pQuizzes[i]'s listing -- record for record what PqaEngine_ListTopTargets(pQuizzes[i], maxCount) returns.  What comes back from the
// device is n x maxCount records; no posterior is copied (the reference's GPU engine: all T of them per quiz,
// PqaCore/CudaEngine.cpp:251-289).
Error HipEngine::ListTopTargetsBatch(int64_t n, const int64_t *pQuizzes, int64_t maxCount, CiRatedTarget *pDest, int64_t *pCounts) {
  if (n < 0) return Error::MakeP(ErrCode::NegativeCount, "count=" + std::to_string(n), "|nQuizzes| must be non-negative.");
  if (maxCount < 0) return Error::MakeP(ErrCode::NegativeCount, "count=" + std::to_string(maxCount), "|maxCount| must be non-negative.");
  if (n > 0 && (!pQuizzes || !pCounts || (maxCount > 0 && !pDest))) return Error::Make(ErrCode::NullArgument, "Nullptr is passed in place of a batch buffer.");
  CallScope scope(_activeCallers);
  std::lock_guard<EngineMutex> lk(_mu);
  Error err = CheckRegular("list top targets");
  if (!err.ok()) return err;
  std::vector<Quiz *> quizzes((size_t)n);
  for (int64_t i = 0; i < n; i++) {
    quizzes[(size_t)i] = UseQuiz(err, pQuizzes[i]);
    if (!quizzes[(size_t)i]) return err;
  }
  for (int64_t i = 0; i < n; i++) pCounts[i] = 0;
  if (n == 0 || maxCount == 0) return Error();
  hipSetDevice(_device);
  err = FlushUpdates();
  if (!err.ok()) return err;
  const int64_t want = std::min<int64_t>(maxCount, _T);
  const bool exact = _optTopExact != 0;
  if (want > 256) {
    for (int64_t i = 0; i < n; i++) {
      // (bulk exports: on the host, in the reference's order right away)
      pCounts[i] = ListTopTargetsOnHost(err, quizzes[(size_t)i], want, pDest + i * maxCount, exact);
      if (pCounts[i] < 0) { pCounts[i] = 0; return err; }
    }
    return Error();
  }
  // the fast listing with one entry beyond the list where that fits (the boundary); a quiz whose listing shows equal probabilities
  // -- or whose boundary cannot be seen -- is listed again in the reference's own order among them (LaunchTopTargetsExact)
  const int64_t probe = exact && want + 1 <= 256 && want + 1 <= _T ? want + 1 : want;
  const int64_t group = std::min<int64_t>(n, kTopBatchQuizzes);
  StopServer();   // (thousands of workgroups: they would wait for a resident sweep to idle out)
  err = EnsureTopScratch(group, probe);
  if (!err.ok()) return err;
  // the results' lines: host-coherent, written by the last level's workgroups
  const int64_t needRecords = group * probe;
  if (needRecords > _hTopBatchRecords) {
    HIP_TRY(hipStreamSynchronize(_stream));
    if (_hTopBatch) hipHostFree(_hTopBatch);
    _hTopBatch = nullptr; _hTopBatchRecords = 0;
    HIP_TRY(hipHostMalloc((void **)&_hTopBatch, (size_t)needRecords * sizeof(RatedTargetDev) + (size_t)kTopBatchQuizzes * sizeof(int64_t), hipHostMallocDefault));
    _hTopBatchRecords = needRecords;
  }
  int64_t *hCounts = reinterpret_cast<int64_t *>(_hTopBatch + _hTopBatchRecords);
  std::vector<int64_t> tied;   // positions in the batch
  for (int64_t first = 0; first < n; first += group) {
    const int64_t m = std::min<int64_t>(group, n - first);
    TopBatchPriors pr;
    for (int64_t i = 0; i < m; i++) pr.prior[i] = quizzes[(size_t)(first + i)]->dPrior;
    HIP_TRY(LaunchTopTargetsBatch(View(), pr, m, probe, _dTopScratch[0], _dTopScratch[1], _hTopBatch, hCounts, nullptr, 0, _stream));
    HIP_TRY(hipStreamSynchronize(_stream));
    for (int64_t i = 0; i < m; i++) {
      const int64_t got = std::min<int64_t>(hCounts[i], probe), c = std::min(got, want);
      const CiRatedTarget *rec = reinterpret_cast<const CiRatedTarget *>(_hTopBatch + i * probe);
      pCounts[first + i] = c;
      std::memcpy(pDest + (first + i) * maxCount, rec, (size_t)c * sizeof(RatedTargetDev));
      if (exact && ((probe == want && got == want && want < _T) || ListingHasTies(rec, got, want, probe))) tied.push_back(first + i);
    }
  }
  _mu.busy = false;   // (the stream has just been synchronised)
  _pendingRecordOp = 0;
  if (tied.empty()) return Error();
  const int64_t tgroup = std::min<int64_t>((int64_t)tied.size(), kTopBatchQuizzes);
  err = EnsureTopExactScratch(tgroup, want);
  if (!err.ok()) return err;
  for (size_t first = 0; first < tied.size(); first += (size_t)tgroup) {
    const int64_t m = std::min<int64_t>(tgroup, (int64_t)(tied.size() - first));
    TopBatchPriors pr;
    for (int64_t i = 0; i < m; i++) pr.prior[i] = quizzes[(size_t)tied[first + (size_t)i]]->dPrior;
    HIP_TRY(LaunchTopTargetsExact(View(), pr, m, _optWorkers, want, _dTopExact, _hTopBatch, hCounts, nullptr, 0, _stream));   // (want <= probe: the lines hold it)
    HIP_TRY(hipStreamSynchronize(_stream));
    for (int64_t i = 0; i < m; i++) {
      const int64_t at = tied[first + (size_t)i], c = std::min<int64_t>(hCounts[i], want);
      pCounts[at] = c;
      std::memcpy(pDest + at * maxCount, _hTopBatch + i * want, (size_t)c * sizeof(RatedTargetDev));
    }
    _topExactListings += m;
  }
  return Error();
}

// ------------------------------------------------------------------------------------------------------------------
// training (reference PqaCore/CpuEngine.cpp:102-183, :442-466; PqaCore/CETrainOperation.cpp:15-25)
// ------------------------------------------------------------------------------------------------------------------
// The steps of one training call in the reference's pairing, for this engine's (shard's) questions.
//   fromQuiz = false: CpuEngine::TrainSpec (CpuEngine.cpp:102-183) -- the answered questions go into nWorkers LIFO buckets by
//     iQuestion % nWorkers (CETrainSubtaskDistrib.h:46-52; restated for one distributing thread, i.e. sequence = position in
//     pAQs: the reference's distributing threads race for the sequence numbers), every bucket is consumed newest first, two
//     entries at a time through Perform2, a last odd one through Perform1 (CETrainSubtaskAdd.cpp:17-38);
//   fromQuiz = true: CpuEngine::RecordQuizTargetSpec (CpuEngine.cpp:442-466) -- the quiz's answers in order, pairs (0,1), (2,3) ...
// Perform2 over two different questions is two independent Perform1 steps (CETrainOperation.cpp:56-82); over one question it
// is a step of kind 2 (same answer) or 3 (different answers), see kb_kernels.hip.  The steps come out grouped by question
// (chains), each chain in execution order; steps on other shards' questions are dropped.
void HipEngine::BuildTrainSteps(int64_t n, const AQ *pAQs, bool fromQuiz, std::vector<TrainStep> &steps, std::vector<int64_t> &chainStart) const {
  std::vector<std::pair<int64_t, TrainStep>> ordered;   // (execution rank, step)
  int64_t rank = 0;
  auto local = [&](int64_t q) { return q >= _qFirst && q < _qFirst + _Q; };
  auto perform1 = [&](const AQ &aq) {
    if (local(aq.iQuestion)) ordered.push_back({rank++, TrainStep{1, aq.iQuestion - _qFirst, aq.iAnswer, aq.iAnswer}});
  };
  auto perform2 = [&](const AQ &first, const AQ &second) {
    if (first.iQuestion != second.iQuestion) { perform1(first); perform1(second); return; }
    if (!local(first.iQuestion)) return;
    ordered.push_back({rank++, TrainStep{first.iAnswer == second.iAnswer ? 2 : 3, first.iQuestion - _qFirst, first.iAnswer, second.iAnswer}});
  };
  if (fromQuiz) {
    int64_t i = 0;
    for (; i < n - 1; i += 2) perform2(pAQs[i], pAQs[i + 1]);
    if (i == n - 1) perform1(pAQs[i]);
  } else {
    const int64_t nWorkers = _optWorkers;
    std::vector<int64_t> last((size_t)nWorkers, -1), prev((size_t)std::max<int64_t>(n, 1), -1);
    for (int64_t i = 0; i < n; i++) {
      const int64_t bucket = pAQs[i].iQuestion % nWorkers;
      prev[i] = last[bucket];
      last[bucket] = i;
    }
    for (int64_t w = 0; w < nWorkers; w++) {
      int64_t iLast = last[w];
      while (iLast != -1) {
        const AQ &first = pAQs[iLast];
        iLast = prev[iLast];
        if (iLast == -1) { perform1(first); break; }
        perform2(first, pAQs[iLast]);
        iLast = prev[iLast];
      }
    }
  }
  std::stable_sort(ordered.begin(), ordered.end(), [](const auto &x, const auto &y) { return x.second.q < y.second.q; });
  steps.clear();
  chainStart.clear();
  for (size_t i = 0; i < ordered.size(); i++) {
    if (i == 0 || ordered[i].second.q != ordered[i - 1].second.q) chainStart.push_back((int64_t)i);
    steps.push_back(ordered[i].second);
  }
  chainStart.push_back((int64_t)ordered.size());
}

// Validation of a training call (CETrainSubtaskDistrib.h:26-45, CpuEngine.cpp:138-155): ranges over the GLOBAL question range, gaps
// for this engine's own questions.  The reference validates every answered question before any Add subtask runs.
Error HipEngine::ValidateTrainLocked(int64_t nQuestions, const AQ *pAQs, int64_t iTarget) const {
  if (iTarget < 0 || iTarget >= _T)
    return Error::MakeP(ErrCode::IndexOutOfRange, RangeParams(iTarget, 0, _T - 1), "Target index is not in KB range.");
  if (BitTest(_hTGap, iTarget))
    return Error::MakeP(ErrCode::AbsentId, "id=" + std::to_string(iTarget), "Target index is not in KB (but rather at a gap).");
  for (int64_t i = 0; i < nQuestions; i++) {
    const int64_t iq = pAQs[i].iQuestion, ia = pAQs[i].iAnswer;
    if (iq < 0 || iq >= _qTotal)
      return Error::MakeP(ErrCode::IndexOutOfRange, RangeParams(iq, 0, _qTotal - 1), "Question index is not in KB range.");
    if (iq >= _qFirst && iq < _qFirst + _Q && BitTest(_hQGap, iq - _qFirst))
      return Error::MakeP(ErrCode::AbsentId, "id=" + std::to_string(iq), "Question index is not in KB (but rather at a gap).");
    if (ia < 0 || ia >= _K)
      return Error::MakeP(ErrCode::IndexOutOfRange, RangeParams(ia, 0, _K - 1), "Answer index is not in KB range.");
  }
  return Error();
}

// What a sharded engine asks of every shard BEFORE any shard trains (a gap question owned by shard k must not leave shards
// 0..k-1 trained): the validation of Train (iQuiz < 0) or of RecordQuizTarget (the quiz's own answers), nothing else.
Error HipEngine::ValidateTrain(int64_t nQuestions, const AQ *pAQs, int64_t iTarget, int64_t iQuiz) {
  std::lock_guard<EngineMutex> lk(_mu);
  if (iQuiz < 0) return ValidateTrainLocked(nQuestions, pAQs, iTarget);
  Error err = CheckRegular("record quiz target");
  if (!err.ok()) return err;
  Quiz *q = UseQuiz(err, iQuiz);
  if (!q) return err;
  return ValidateTrainLocked((int64_t)q->answers.size(), q->answers.data(), iTarget);
}

// Validation + the steps on the device; the caller holds the lock.
Error HipEngine::TrainLocked(int64_t nQuestions, const AQ *pAQs, int64_t iTarget, double amount, bool fromQuiz) {
  StopServer();   // the cube changes: the resident sweep's XCD-local L2s would keep stale rows
  {
    Error ve = ValidateTrainLocked(nQuestions, pAQs, iTarget);
    if (!ve.ok()) return ve;
  }
  std::vector<TrainStep> steps;
  std::vector<int64_t> chainStart;
  BuildTrainSteps(nQuestions, pAQs, fromQuiz, steps, chainStart);
  hipSetDevice(_device);
  if (steps.size() <= (size_t)kTrainInlineSteps && chainStart.size() <= (size_t)kTrainInlineSteps + 1) {
    TrainStepsInline in;
    in.nChains = (int64_t)chainStart.size() - 1;
    std::copy(chainStart.begin(), chainStart.end(), in.chainStart);
    std::copy(steps.begin(), steps.end(), in.steps);
    HIP_TRY(LaunchTrainStepsInline(_dCube, _elem, _dVB, _K, _ldT, in, iTarget, amount, _stream));
    return Error();   // (later operations of the engine are ordered behind it on the stream)
  }
  // one device buffer for both arrays: [steps | chainStart]
  const size_t stepBytes = steps.size() * sizeof(TrainStep), chainBytes = chainStart.size() * sizeof(int64_t);
  const int64_t needWords = (int64_t)((stepBytes + chainBytes) / sizeof(int64_t));
  if (needWords > 2 * _aqCapacity) {
    hipFree(_dAqs);
    _dAqs = nullptr;
    _aqCapacity = 0;
    const int64_t cap = std::max<int64_t>((needWords + 1) / 2, 64);
    HIP_TRY(hipMalloc(&_dAqs, (size_t)cap * 2 * sizeof(int64_t)));
    _aqCapacity = cap;
  }
  char *dBuf = reinterpret_cast<char *>(_dAqs);
  if (stepBytes > 0) HIP_TRY(hipMemcpyAsync(dBuf, steps.data(), stepBytes, hipMemcpyHostToDevice, _stream));
  HIP_TRY(hipMemcpyAsync(dBuf + stepBytes, chainStart.data(), chainBytes, hipMemcpyHostToDevice, _stream));
  HIP_TRY(LaunchTrainSteps(_dCube, _elem, _dVB, _K, _ldT, reinterpret_cast<const TrainStep *>(dBuf),
                           reinterpret_cast<const int64_t *>(dBuf + stepBytes), (int64_t)chainStart.size() - 1, iTarget, amount, _stream));
  HIP_TRY(hipStreamSynchronize(_stream));   // (the host vectors are the copies' sources)
  return Error();
}

Error HipEngine::Train(int64_t nQuestions, const AQ *pAQs, int64_t iTarget, double amount) {
  if (nQuestions < 0)
    return Error::MakeP(ErrCode::NegativeCount, "count=" + std::to_string(nQuestions), "|nQuestions| must be non-negative.");
  if (amount <= 0)
    return Error::MakeP(ErrCode::NonPositiveAmount, "amount=" + std::to_string(amount), "|amount| must be positive.");
  if (nQuestions > 0 && pAQs == nullptr) return Error::Make(ErrCode::NullArgument, "Nullptr is passed in place of answered questions.");
  std::lock_guard<EngineMutex> lk(_mu);
  if (_mode == Mode::Shutdown) return Error::MakeP(ErrCode::ObjectShutDown, "RejectedOperation=Train", "Engine is shut down.");
  Error e = TrainLocked(nQuestions, pAQs, iTarget, amount, false);
  if (e.ok()) _nQuestionsAsked.fetch_add((uint64_t)nQuestions, std::memory_order_relaxed);  // reference CpuEngine.cpp:176
  return e;
}

Error HipEngine::RecordQuizTarget(int64_t iQuiz, int64_t iTarget, double amount) {
  // reference PqaCore/BaseEngine.cpp:529-566, PqaCore/CpuEngine.cpp:442-466: the quiz's answers, pairwise in order, under ONE
  // hold of the lock (the quiz cannot be answered or released in between); the asked-questions counter is not touched
  if (amount <= 0)
    return Error::MakeP(ErrCode::NonPositiveAmount, "amount=" + std::to_string(amount), "|amount| must be positive.");
  CallScope scope(_activeCallers);
  if (_optCombine && (_optPostAlways || !_mu.try_lock())) {
    PostedOp op;
    op.kind = 6; op.iQuiz = iQuiz; op.arg = iTarget; op.amount = amount;
    RunPosted(op);
    return op.err;
  }
  std::unique_lock<EngineMutex> lk(_mu, std::defer_lock);
  if (_optCombine) lk = std::unique_lock<EngineMutex>(_mu, std::adopt_lock);
  else lk.lock();
  return RecordQuizTargetLocked(iQuiz, iTarget, amount);
}

Error HipEngine::RecordQuizTargetLocked(int64_t iQuiz, int64_t iTarget, double amount) {
  Error err = CheckRegular("record quiz target");
  if (!err.ok()) return err;
  if (iTarget < 0 || iTarget >= _T)
    return Error::MakeP(ErrCode::IndexOutOfRange, RangeParams(iTarget, 0, _T - 1), "Target index is not in KB range.");
  if (BitTest(_hTGap, iTarget))
    return Error::MakeP(ErrCode::AbsentId, "id=" + std::to_string(iTarget), "Target index is not in KB (but rather at a gap).");
  Quiz *q = UseQuiz(err, iQuiz);
  if (!q) return err;
  return TrainLocked((int64_t)q->answers.size(), q->answers.data(), iTarget, amount, true);
}

}  // namespace pqa
