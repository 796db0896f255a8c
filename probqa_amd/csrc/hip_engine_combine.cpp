// hip_engine_combine.cpp -- HipEngine under concurrent callers: operations posted to the holder of the engine lock, and the
// flat combining of concurrent NextQuestion calls into one sweep (hip_engine.h: Combine).
#include "hip_engine_internal.h"

namespace pqa {
// ------------------------------------------------------------------------------------------------------------------
// concurrent NextQuestion calls (see SelRequest in hip_engine.h)
// ------------------------------------------------------------------------------------------------------------------
// How many CPUs the process may keep busy: a container's CPU quota (cgroup v2 cpu.max / v1 cfs quota) or else the affinity mask.
// The GPU boxes of this project allow a container 16 of the host's 256 hardware threads: waiting policies that spin are right for
// up to that many client threads and wrong beyond (measured: 64 spinning clients 40 k questions/s against 54 k sleeping).
int HipEngine::AllowedCpus() {
  static const int n = [] {
    int cpus = (int)std::thread::hardware_concurrency();
    if (cpus <= 0) cpus = 1;
    if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
      char quota[32] = {0};
      long long period = 0;
      if (std::fscanf(f, "%31s %lld", quota, &period) == 2 && period > 0 && quota[0] != 'm') {
        const long long q = std::atoll(quota);
        if (q > 0) cpus = std::min<int>(cpus, (int)std::max<long long>(1, q / period));
      }
      std::fclose(f);
    } else if (FILE *g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
      long long q = -1, period = 100000;
      if (std::fscanf(g, "%lld", &q) != 1) q = -1;
      std::fclose(g);
      if (FILE *h = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (std::fscanf(h, "%lld", &period) != 1) period = 100000; std::fclose(h); }
      if (q > 0 && period > 0) cpus = std::min<int>(cpus, (int)std::max<long long>(1, q / period));
    }
    return cpus;
  }();
  return n;
}

// ---- posted operations (hip_engine.h)
void HipEngine::EngineMutex::unlock() {
  for (;;) {
    std::atomic<int> *wake[64];
    size_t nWake = 0;
    std::vector<std::atomic<int> *> more;
    if (owner != nullptr && owner->_posted.load(std::memory_order_acquire) != nullptr) {
      owner->DrainPosted();
      std::vector<std::atomic<int> *> &w = owner->_postedWake;
      if (w.size() <= 64) { nWake = w.size(); std::copy(w.begin(), w.end(), wake); }
      else more.swap(w);
      w.clear();
    }
    m.unlock();
    for (size_t i = 0; i < nWake; i++) FutexWakeOne(wake[i]);
    for (std::atomic<int> *word : more) FutexWakeOne(word);
    // Posted between the drain and the release: its thread saw the lock taken and sleeps.  (Both sides are a locked
    // read-modify-write followed by a load -- the post then try_lock there, the release then this load here: one of the two sees
    // the other.)  If somebody else has the lock by now, the operation is theirs to run.
    if (owner == nullptr || owner->_posted.load(std::memory_order_seq_cst) == nullptr || !m.try_lock()) return;
  }
}

void HipEngine::RunPosted(PostedOp &op) {
  PostedOp *head = _posted.load(std::memory_order_relaxed);
  do op.next = head; while (!_posted.compare_exchange_weak(head, &op, std::memory_order_seq_cst, std::memory_order_relaxed));
  for (;;) {
    if (_mu.try_lock()) { _mu.unlock(); }   // (free after all: run it -- and the others' -- here)
    for (int spins = 0; spins < 300; spins++) {
      if (op.state.load(std::memory_order_acquire) == 1) return;
      _mm_pause();
    }
    int expected = 0;
    if (op.state.compare_exchange_strong(expected, 2, std::memory_order_seq_cst) || expected == 2) {
      // (the timeout is a belt to the braces above: a millisecond, then the lock is tried again)
      struct timespec ts{0, 1000000};
      syscall(SYS_futex, reinterpret_cast<int *>(&op.state), FUTEX_WAIT_PRIVATE, 2, &ts, nullptr, 0);
    }
    if (op.state.load(std::memory_order_acquire) == 1) return;
  }
}

// Everything posted so far, in the order it was posted.  The RecordAnswers first go where RecordAnswer puts them (the list of
// deferred updates), ReleaseQuiz and RecordQuizTarget run as they come; the StartQuiz calls then share one launch; then ONE launch runs every deferred update if a ListTopTargets of this drain needs its quiz's posterior;
// then the combined sweeps leaders have posted; then the listings that the update kernel has not made already.
void HipEngine::DrainPosted() {
  PostedOp *list = _posted.exchange(nullptr, std::memory_order_acq_rel);
  if (list == nullptr) return;
  PostedOp *ordered = nullptr;
  while (list != nullptr) { PostedOp *n = list->next; list->next = ordered; ordered = list; list = n; }
  _postedDrains++;
  bool needFlush = false;
  int64_t nStarts = 0, nTrains = 0;
  for (PostedOp *op = ordered; op != nullptr; op = op->next) {
    _postedOps++;
    if (op->kind == 1) { op->err = RecordAnswerLocked(op->iQuiz, op->arg, op->remote, false); continue; }
    if (op->kind == 5) { op->err = ReleaseQuizLocked(op->iQuiz, false); continue; }
    if (op->kind == 6) { nTrains++; continue; }
    if (op->kind == 4) { nStarts++; continue; }
    if (op->kind == 3) continue;
    op->result = -1;
    op->err = CheckRegular("list top targets");
    if (!op->err.ok()) continue;
    op->quiz = UseQuiz(op->err, op->iQuiz);
    if (op->quiz != nullptr) op->serial = op->quiz->serial;
    if (op->quiz != nullptr && op->quiz->updatePending) needFlush = true;
  }
  Error flushErr;
  if (needFlush) flushErr = FlushUpdates();
  if (nTrains > 0) { TrainPosted(ordered); MarkStreamBusy(); }
  if (nStarts > 0) {
    MarkStreamBusy();
    // the StartQuiz calls of this drain: ONE launch sets all their priors (as StartQuizBatch; chunks of kStartInline)
    hipSetDevice(_device);
    static thread_local StartBatchInline batch;   // (4 KB of pointers: not on a client thread's stack)
    batch.n = 0;
    batch.askedWords = (int64_t)BitWords(_Q);
    std::vector<PostedOp *> chunk;
    auto launch = [&]() {
      if (batch.n > 0) {
        const hipError_t he = LaunchStartQuizBatch(View(), batch, _optWorkers, _stream);
        if (he != hipSuccess)
          for (PostedOp *o : chunk)
            if (o->result >= 0) {
              Quiz *q = _quizzes[(size_t)o->result];
              UnassignQuiz(o->result);
              DestroyQuiz(q);
              o->result = -1;
              o->err = HipErr(he, "StartQuiz");
            }
      }
      batch.n = 0;
      chunk.clear();
    };
    for (PostedOp *op = ordered; op != nullptr; op = op->next) {
      if (op->kind != 4) continue;
      _startBatch = &batch;
      op->result = CreateQuiz(op->err, 0, nullptr, nullptr, nullptr, 0, nullptr);
      _startBatch = nullptr;
      chunk.push_back(op);
      if (batch.n == kStartInline) launch();
    }
    launch();
  }
  for (PostedOp *op = ordered; op != nullptr; op = op->next)
    if (op->kind == 3) LaunchBatchLocked(*op->ctx, *op->batch, *op->flight);   // (behind the updates, ahead of the listings: the sweep is what the most clients wait for)
  for (PostedOp *op = ordered; op != nullptr;) {
    PostedOp *const next = op->next;   // (the operation is its thread's again the moment its state says so)
    if (op->kind == 2 && op->quiz != nullptr) {
      Quiz *q = op->quiz;
      const int64_t want = std::min<int64_t>(op->arg, _T);
      // (a ReleaseQuiz of the same quiz later in this drain -- a client's error, IPqaEngine.h:44 -- has taken it away since)
      const bool gone = (size_t)op->iQuiz >= _quizzes.size() || _quizzes[(size_t)op->iQuiz] != q || q->serial != op->serial;
      if (gone) { op->result = -1; op->err = Error::MakeP(ErrCode::AbsentId, "id=" + std::to_string(op->iQuiz), "Quiz index is not in the registry (but rather at a gap)."); }
      else if (!flushErr.ok()) op->err = flushErr;
      else if (want > kQuizTop || _T > 16384) op->result = -2;
      else {
        _topWantRecent = want >= _topWantRecent ? want : want + (_topWantRecent - want) * 7 / 8;
        const bool cached = q->topOp != 0 && q->topVersion == q->priorVersion && want <= q->topCount;
        hipError_t he = hipSuccess;
        if (!cached) {
          hipSetDevice(_device);
          const uint64_t opNo = ++_opSeq;
          he = LaunchTopTargets(View(), q->dPrior, want, q->pin->top, &q->pin->nOut, &q->pin->topFlag, opNo, _stream);
          if (he == hipSuccess) { q->topOp = opNo; q->topVersion = q->priorVersion; q->topCount = want; }
        }
        if (he != hipSuccess) op->err = HipErr(he, "ListTopTargets");
        else { op->pin = q->pin; op->flagOp = q->topOp; op->result = want; }
      }
    }
    std::atomic<int> *word = &op->state;
    if (word->exchange(1, std::memory_order_acq_rel) == 2) _postedWake.push_back(word);
    op = next;
  }
  // The drain runs on the holder's way out -- possibly after a selection path declared the stream idle -- and may have launched
  // updates, trainings, quiz starts and listings: whoever takes the lock next finds the stream marked busy.
  MarkStreamBusy();
}

// The RecordQuizTarget calls of a drain (kind 6), in the order they were posted: calls with different targets touch disjoint cells
// and go out in ONE launch (train_batch_inline_kernel: a workgroup per call); a call whose target is already in the batch, or that
// does not fit the kernel's arguments, closes the batch first (or runs alone, the usual way).
void HipEngine::TrainPosted(PostedOp *ordered) {
  static thread_local TrainBatchInline tb;   // (2.5 KB)
  tb.nCalls = 0; tb.nChainsTotal = 0; tb.nSteps = 0;
  std::vector<PostedOp *> inBatch;
  hipSetDevice(_device);
  auto launch = [&]() {
    if (tb.nCalls > 0) {
      const hipError_t he = LaunchTrainBatchInline(_dCube, _elem, _dVB, _K, _ldT, tb, _stream);
      if (he != hipSuccess) for (PostedOp *o : inBatch) o->err = HipErr(he, "RecordQuizTarget");
      _trainBatches++;
      _trainBatchCalls += (uint64_t)tb.nCalls;
    }
    tb.nCalls = 0; tb.nChainsTotal = 0; tb.nSteps = 0;
    inBatch.clear();
  };
  bool stopped = false;
  for (PostedOp *op = ordered; op != nullptr; op = op->next) {
    if (op->kind != 6) continue;
    op->err = CheckRegular("record quiz target");
    if (!op->err.ok()) continue;
    const int64_t iTarget = op->arg;
    Quiz *q = UseQuiz(op->err, op->iQuiz);
    if (q == nullptr) continue;
    op->err = ValidateTrainLocked((int64_t)q->answers.size(), q->answers.data(), iTarget);
    if (!op->err.ok()) continue;
    if (!stopped) { StopServer(); stopped = true; }   // the cube changes (and the deferred updates read it as it was: they run first)
    std::vector<TrainStep> steps;
    std::vector<int64_t> chainStart;
    BuildTrainSteps((int64_t)q->answers.size(), q->answers.data(), true, steps, chainStart);
    const int64_t nChains = (int64_t)chainStart.size() - 1;
    bool fits = (int64_t)steps.size() <= kTrainBatchSteps && nChains + 1 <= (int64_t)(sizeof(tb.chainStart) / sizeof(tb.chainStart[0]));
    for (const TrainStep &st : steps) fits = fits && st.q <= INT32_MAX && st.a1 < 256 && st.a2 < 256;
    if (!fits) {   // a long quiz: the usual way, in its place in the order
      launch();
      op->err = TrainLocked((int64_t)q->answers.size(), q->answers.data(), iTarget, op->amount, true);
      continue;
    }
    bool clash = tb.nCalls == kTrainBatchCalls || tb.nSteps + (int64_t)steps.size() > kTrainBatchSteps ||
                 tb.nChainsTotal + tb.nCalls + nChains + 1 > (int64_t)(sizeof(tb.chainStart) / sizeof(tb.chainStart[0]));
    for (int c = 0; c < tb.nCalls && !clash; c++) clash = tb.calls[c].iTarget == iTarget;
    if (clash) launch();
    TrainBatchCall &call = tb.calls[tb.nCalls];
    call.iTarget = iTarget; call.amount = op->amount; call.firstChain = tb.nChainsTotal; call.nChains = (int32_t)nChains;
    uint16_t *cs = tb.chainStart + tb.nChainsTotal + tb.nCalls;   // (every call's chain starts are followed by one end marker)
    for (int64_t c = 0; c <= nChains; c++) cs[c] = (uint16_t)(tb.nSteps + chainStart[(size_t)c]);
    for (size_t i = 0; i < steps.size(); i++)
      tb.steps[tb.nSteps + (int64_t)i] = TrainBatchStep{(int32_t)steps[i].q, (uint8_t)steps[i].kind, (uint8_t)steps[i].a1, (uint8_t)steps[i].a2, 0};
    tb.nSteps += (int32_t)steps.size();
    tb.nChainsTotal += (int32_t)nChains;
    tb.nCalls++;
    inBatch.push_back(op);
  }
  launch();
}

int64_t HipEngine::Combine(Error &err, int64_t iQuiz, int kind, uint64_t rnd) {
  CallScope scope(_activeCallers);
  _mu.spinFirst.store(ClientsFitCpus() && _activeCallers.load(std::memory_order_relaxed) > 1, std::memory_order_relaxed);
  if (!_optCombine) {
    std::lock_guard<EngineMutex> lk(_mu);
    return kind == 0 ? NextQuestionArgmaxLocked(err, iQuiz) : NextQuestionSampledLocked(err, iQuiz, rnd);
  }
  // Nobody else is inside a quiz-level call (the usual case of the reference's wrappers: one quiz loop on one thread): straight to
  // the single-quiz path -- no request to queue, no batch context, no flight.  Racing with a client that arrives just now is
  // harmless: each is served by itself, under the engine's lock, as with combining switched off.
  if (_activeCallers.load(std::memory_order_relaxed) == 1 && _extCallers == nullptr && _mu.try_lock()) {
    std::lock_guard<EngineMutex> lk(_mu, std::adopt_lock);
    _flushedSinceSweep.store(0, std::memory_order_relaxed);
    return kind == 0 ? NextQuestionArgmaxLocked(err, iQuiz) : NextQuestionSampledLocked(err, iQuiz, rnd);
  }
  SelRequest r;
  r.iQuiz = iQuiz; r.kind = kind; r.rnd = rnd;
  bool lead;
  {
    std::lock_guard<std::mutex> lk(_combMu);
    _combQueue.push_back(&r);
    lead = !_leaderActive;
    if (lead) _leaderActive = true;
  }
  if (!lead) {
    // (a combined sweep takes a fraction of a millisecond, and a thread woken through the kernel arrives tens of
    //  microseconds after its neighbours; but dozens of spinning client threads eat the cores the process is allowed:
    //  a short spin, then sleep)
    int st = 0;
    const auto tw0 = std::chrono::steady_clock::now();
    for (int spins = 0; spins < 1500 && (st = r.state.load(std::memory_order_acquire)) == 0; spins++) _mm_pause();
    if (st == 0 && ClientsFitCpus()) {
      // Fewer clients than CPUs: sleep most of the expected wait (about as long as the last combined sweeps took), spin the rest --
      // woken through the kernel the clients of one sweep arrive tens of microseconds apart.  More clients than CPUs: the
      // condition variable only (spinning waiters would take the CPUs from the threads that have work).
      const int64_t expect = _sweepNsEwma.load(std::memory_order_relaxed);
      if (expect > 90000) {
        static thread_local bool slackSet = false;
        if (!slackSet) { prctl(PR_SET_TIMERSLACK, 2000UL, 0, 0, 0); slackSet = true; }
        const auto until = tw0 + std::chrono::nanoseconds(std::min<int64_t>(expect - 50000, 2000000));
        // (in naps of 40 us: the lead may be handed to this request meanwhile, and the next sweep waits for its leader)
        while ((st = r.state.load(std::memory_order_acquire)) == 0 && std::chrono::steady_clock::now() < until) {
          struct timespec ts{0, 40000};
          nanosleep(&ts, nullptr);
        }
      }
      for (int spins = 0; spins < 12000 && (st = r.state.load(std::memory_order_acquire)) == 0; spins++) _mm_pause();
    }
    while (st == 0) {
      FutexWait(&r.state, 0);   // (returns at once if the state is no longer 0)
      st = r.state.load(std::memory_order_acquire);
    }
    {
      const int64_t waited = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - tw0).count();
      const int64_t old = _sweepNsEwma.load(std::memory_order_relaxed);
      _sweepNsEwma.store(old == 0 ? waited : old + (waited - old) / 8, std::memory_order_relaxed);
    }
    if (st == 1) { err = r.err; return r.result; }
    if (st == 3) {   // the sweep has run: this quiz's priorities are on the host, the selection is this thread's own work
      const int64_t sel = SelectFromPriorities(&r);
      r.ctx->readers.fetch_sub(1, std::memory_order_release);
      err = r.err;
      return sel;
    }
    // (2: the leader before served its own batch and handed the lead to this, the oldest waiting request)
  }
  ServeQueue(&r);
  err = r.err;
  return r.result;
}

// One request of a combined sweep, after the sweep: this quiz's priority vector out of the batch's matrix, then the selector (as
// the single-quiz path's host_sampled form: SelectSampledHost; the argmax by the device's rule: maximum, lowest index on ties,
// NaN never wins), then NextQuestion's bookkeeping under the engine's lock.
int64_t HipEngine::SelectFromPriorities(SelRequest *r) {
  // No engine lock: the priorities are the sweep's, the asked / gap bits the leader's snapshot of the moment it launched the
  // sweep (what the kernel saw), the quiz object is held by inSelection, and the two things written -- the quiz's active
  // question, the asked-questions counter -- are this quiz's own or atomic.
  Quiz *q = r->quiz;
  const int64_t nQ = r->nQ;
  auto skip = [&](int64_t k) { return BitTest(r->unavailable, k); };
  std::vector<double> run((size_t)nQ);
  if (r->priTag != 0) {
    // (the quiz's flag said that every workgroup had reported, not that every one of its stores had landed: an entry is taken
    //  once it carries the launch's tag -- it almost always does by now)
    const volatile double *rec = r->pri;
    SpinWait w;
    for (int64_t k = 0; k < nQ; k++) {
      if (skip(k)) { run[(size_t)k] = 0.0; continue; }
      const volatile uint64_t *tagWord = reinterpret_cast<const volatile uint64_t *>(rec + 2 * k + 1);
      while (*tagWord != r->priTag)
        if (!w.Tick(std::chrono::seconds(30))) {
          r->err = HipErr(hipErrorNotReady, "priority vector hand-over (combined sweep)");
          q->inSelection.store(false, std::memory_order_release);
          return r->result = -1;
        }
      std::atomic_thread_fence(std::memory_order_acquire);
      run[(size_t)k] = rec[2 * k];
    }
  } else {
    for (int64_t k = 0; k < nQ; k++) run[(size_t)k] = skip(k) ? 0.0 : r->pri[(size_t)k * (size_t)r->priStride];
  }
  int64_t pick = -1;
  if (r->kind == 1) {
    pick = SelectSampledHostBits(run.data(), nQ, r->nSub, r->rnd, r->unavailable.data(), nullptr);
  } else {
    double best = 0;
    for (int64_t k = 0; k < nQ; k++) {
      if (skip(k)) continue;
      double p = run[(size_t)k];
      if (p != p) p = -HUGE_VAL;
      if (pick < 0 || p > best) { best = p; pick = k; }
    }
  }
  // reference PqaCore/CpuEngine.cpp:403-413 (FinishSelection, over the snapshot)
  if (pick >= 0 && skip(pick)) pick = FindNearestInPacks(pick, nQ, [&](int64_t p) { return ~Pack64(r->unavailable, p); });
  if (pick < 0) {
    r->err = Error::Make(ErrCode::QuestionsExhausted, "Found no unasked question that is not in a gap.");
    r->result = -1;
  } else {
    q->activeQuestion = _qFirst + pick;
    _nQuestionsAsked.fetch_add(1, std::memory_order_relaxed);
    r->result = q->activeQuestion;
  }
  q->inSelection.store(false, std::memory_order_release);
  return r->result;
}

// How many of `m` waiting requests a combined sweep should take.  The (quiz, chunk) sweep costs by its quiz slots -- 8, 16, 32 or
// groups of 64 (tools/midbatch_bench.py at 1000 x 5 x 1000: 60 / 107 / 192 / 362 us of kernel) -- so 20 requests cost what 32 do; with
// the device as the bottleneck of a busy server, a sweep of 16 now and the other 4 with the next one serve more clients per second.
int64_t HipEngine::PreferredCombinedBatch(int64_t m) const {
  if (_optBatchForm != 0 || _elem != 8 || !EvalMidBatchSupported(View())) return m;
  if (m <= 8) return m;
  if (m <= 10) return 8;
  if (m <= 16) return m;
  if (m <= 25) return 16;
  if (m <= 32) return m;
  if (m <= 51) return 32;
  const int64_t full = m / 64 * 64, rem = m % 64;
  return rem == 0 || rem >= 52 ? m : std::max<int64_t>(full, 32);
}

// The leader's turn: ONE batch -- everything posted so far, distinct quizzes, `own` among them (it is the oldest request).  The
// lead goes on to the oldest request still waiting (or is given up) as soon as the batch's sweep is LAUNCHED: the next leader
// gathers and launches the next sweep -- into the other of the two batch contexts -- while this one's runs, so that the device
// finds the next sweep queued when it finishes this one.
void HipEngine::ServeQueue(SelRequest *own) {
  // The clients whose RecordAnswers ran since the last combined sweep are on their way here (their ListTopTargets have just
  // returned): a leader that starts at once sweeps for the two or three that were quickest and makes the rest wait for a
  // second sweep.  So it waits -- microseconds -- until most of them have posted, or nobody new comes.
  // While the previous leader's sweep still runs there is no hurry at all: a sweep launched now only queues behind it, so the
  // requests that arrive until it is (nearly) done ride along for free.
  if (_optLingerUs > 0 && Concurrent()) {   // (alone in the engine: nobody to wait for)
    const int64_t expect = std::min<int64_t>(_flushedSinceSweep.load(std::memory_order_relaxed), _activeCallers.load(std::memory_order_relaxed) - 1);
    const BatchCtx &other = _ctx[_ctxNext ^ 1];
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
  // this batch's context: its previous sweep has been collected, and the clients that were selecting out of its priority
  // buffer -- they need no lock for that -- are done (normally long ago)
  BatchCtx &c = _ctx[_ctxNext];
  _ctxNext ^= 1;
  const auto tA = std::chrono::steady_clock::now();
  std::unique_lock<std::mutex> ctxLock(c.mu);
  while (c.readers.load(std::memory_order_acquire) != 0) _mm_pause();
  std::vector<SelRequest *> batch;
  {
    std::lock_guard<std::mutex> lk(_combMu);
    std::vector<SelRequest *> rest;
    for (SelRequest *r : _combQueue) {
      bool take = (int64_t)batch.size() < kMaxBatch;
      for (size_t i = 0; take && i < batch.size(); i++) take = batch[i]->iQuiz != r->iQuiz;   // a quiz once per sweep
      (take ? batch : rest).push_back(r);
    }
    // (the sweep's lanes come in groups: the newest requests beyond the last well-filled group wait for the next sweep -- it is
    //  launched right behind this one)
    const size_t keep = (size_t)PreferredCombinedBatch((int64_t)batch.size());
    if (keep < batch.size()) {
      rest.insert(rest.begin(), batch.begin() + (std::ptrdiff_t)keep, batch.end());
      batch.resize(keep);
    }
    _combQueue.swap(rest);
  }
  Flight f;
  f.tA = tA;
  LaunchBatch(c, batch, f);   // (under the engine's lock; what could not be launched has its error -- or its result, for a batch of one)
  {
    std::lock_guard<std::mutex> lk(_combMu);
    if (_combQueue.empty()) _leaderActive = false;
    else PublishState(&_combQueue.front()->state, 2);
  }
  const bool ownSelects = f.live.empty() ? false : CollectBatch(c, batch, f, own);
  ctxLock.unlock();
  for (SelRequest *r : batch)
    if (r != nullptr && r != own) PublishState(&r->state, 1);   // (r is its caller's again from here on)
  if (ownSelects) {
    SelectFromPriorities(own);
    c.readers.fetch_sub(1, std::memory_order_release);
  }
}

// Validate and launch (the caller holds the context; the engine's lock is taken and released here).  f.live: the requests whose
// sweep is in flight; every other request of `batch` has its result or error.
void HipEngine::LaunchBatch(BatchCtx &c, std::vector<SelRequest *> &batch, Flight &f) {
  if (batch.size() > 1 && !_mu.try_lock()) {   // (the engine is taken: its holder launches this sweep on its way out)
    PostedOp op;
    op.kind = 3; op.ctx = &c; op.batch = &batch; op.flight = &f;
    RunPosted(op);
    return;
  }
  std::unique_lock<EngineMutex> lk(_mu, std::defer_lock);
  if (batch.size() > 1) lk = std::unique_lock<EngineMutex>(_mu, std::adopt_lock);
  else lk.lock();
  LaunchBatchLocked(c, batch, f);
}

void HipEngine::LaunchBatchLocked(BatchCtx &c, std::vector<SelRequest *> &batch, Flight &f) {
  auto single = [&](SelRequest *r) {
    r->result = r->kind == 0 ? NextQuestionArgmaxLocked(r->err, r->iQuiz) : NextQuestionSampledLocked(r->err, r->iQuiz, r->rnd);
  };
  f.tB = std::chrono::steady_clock::now();
  if (batch.size() == 1) { _flushedSinceSweep.store(0, std::memory_order_relaxed); single(batch[0]); return; }
  auto failAll = [&](const Error &e) { for (SelRequest *r : batch) { r->err = e; r->result = -1; } };
  Error err = CheckRegular("compute next question");
  if (!err.ok()) { failAll(err); return; }
  hipSetDevice(_device);
  err = FlushUpdates();
  if (!err.ok()) { failAll(err); return; }
  std::vector<SelRequest *> live;
  std::vector<int64_t> ids;
  for (SelRequest *r : batch) {
    Error qe;
    if (UseQuiz(qe, r->iQuiz) == nullptr) { r->err = qe; r->result = -1; continue; }
    live.push_back(r);
    ids.push_back(r->iQuiz);
    f.anySampled = f.anySampled || r->kind == 1;
  }
  if (live.empty()) return;
  if (live.size() == 1 || (_optServer && ServerUsable()) || _optUseGraph) {   // (the resident sweep and graph replay serve one quiz at a time)
    for (SelRequest *r : live) single(r);
    return;
  }
  const int64_t n = (int64_t)live.size();
  f.tag = NextLaunchTag();
  std::vector<Quiz *> quizzes;
  err = BatchSweep(c, n, ids.data(), quizzes, false, f.tag, f.anySampled, &f.quizMinor, &f.tagged);
  if (!err.ok()) { for (SelRequest *r : live) { r->err = err; r->result = -1; } return; }
  const int64_t nSubtasks = _optEvalSubtasks ? _optEvalSubtasks : 8 * _optWorkers;  // reference PqaCore/CpuEngine.cpp:339
  for (int64_t i = 0; i < n; i++) {
    SelRequest *r = live[(size_t)i];
    Quiz *q = quizzes[(size_t)i];
    r->serial = q->serial;
    // what finishes the selection once the sweep has run (the client itself, from the priorities, if any request of the batch is
    // sampled; else the leader, from the kernel's choices) without the engine's lock: the quiz (held), the asked questions and
    // gaps as the sweep sees them
    r->quiz = q;
    q->inSelection.store(true, std::memory_order_relaxed);
    r->nQ = _Q;
    r->nSub = nSubtasks;
    r->unavailable.resize(_hQGap.size());
    for (size_t w = 0; w < r->unavailable.size(); w++) r->unavailable[w] = _hQGap[w] | q->hAsked[w];
  }
  f.Bp = c.lastBp;
  f.nQ = _Q;
  if (f.anySampled && !f.tagged) f.he = hipEventRecord(c.event, _stream);
  _combBatches++;
  _combRequests += (uint64_t)n;
  if ((uint64_t)n > _combMaxBatch) _combMaxBatch = (uint64_t)n;
  _lastCombined.store(n, std::memory_order_relaxed);
  _flushedSinceSweep.store(0, std::memory_order_relaxed);
  f.live.swap(live);
  c.inFlight.store(true, std::memory_order_relaxed);
  f.tC = std::chrono::steady_clock::now();
}

// Wait for the sweep and hand the results out -- the engine open to the other clients' calls meanwhile (RecordAnswer,
// ListTopTargets, StartQuiz ... and the next leader's launch).  Returns true if `own` is to select for itself.
bool HipEngine::CollectBatch(BatchCtx &c, std::vector<SelRequest *> &batch, Flight &f, SelRequest *own) {
  const int64_t n = (int64_t)f.live.size();
  Error err;
  hipError_t he = f.he;
  if (he == hipSuccess && f.anySampled && !f.tagged) he = hipEventSynchronize(c.event);
  if (he == hipSuccess && (!f.anySampled || f.tagged)) err = WaitBatchFlags(c, n, f.tag);
  if (err.ok() && f.tagged)
    for (int64_t i = 0; i < n && err.ok(); i++)
      if (c.h->out[i].index == -3) err = HipErr(hipErrorLaunchFailure, "combined selection (incomplete sweep)");
  c.inFlight.store(false, std::memory_order_relaxed);
  const auto tD = std::chrono::steady_clock::now();
  auto ns = [](auto a, auto b) { return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(b - a).count(); };
  if (he != hipSuccess) err = HipErr(he, "combined selection");
  if (!err.ok()) {
    for (SelRequest *r : f.live) {
      r->err = err;
      r->result = -1;
      if (r->quiz) r->quiz->inSelection.store(false, std::memory_order_release);
    }
    return false;
  }
  _combNs[0] += ns(f.tA, f.tB); _combNs[1] += ns(f.tB, f.tC); _combNs[2] += ns(f.tC, tD);
  if (f.anySampled) {
    // The priority vectors are on the host: every client selects for ITSELF (the O(Q) scalar Kahan steps of the reference's
    // selector, CpuEngine.cpp:362-400, run on as many cores as there are clients), the leader only for its own request.
    c.readers.fetch_add((int)n, std::memory_order_acq_rel);
    bool ownLive = false;
    for (int64_t i = 0; i < n; i++) {
      SelRequest *r = f.live[(size_t)i];
      r->pri = f.tagged ? c.hPri + 2 * (size_t)i * (size_t)f.nQ : f.quizMinor ? c.hPri + i : c.hPri + (size_t)i * (size_t)f.nQ;
      r->priStride = f.tagged ? 2 : f.quizMinor ? f.Bp : 1;
      r->priTag = f.tagged ? f.tag : 0;
      r->ctx = &c;
      if (r == own) { ownLive = true; continue; }
      for (SelRequest *&slot : batch) if (slot == r) slot = nullptr;   // (published here: not the caller's to publish again)
      PublishState(&r->state, 3);
    }
        _combNs[3] += ns(tD, std::chrono::steady_clock::now());
    return ownLive;
  }
  // The kernel's choices: finished here for every request, and without the engine's lock -- the quizzes are held (inSelection:
  // a ReleaseQuiz of one waits), what is written is each quiz's own or atomic.
  const auto tE = std::chrono::steady_clock::now();
  for (int64_t i = 0; i < n; i++) {
    SelRequest *r = f.live[(size_t)i];
    Quiz *q = r->quiz;
    int64_t pick = c.h->out[i].index;
    if (pick == -3) { r->err = HipErr(hipErrorLaunchFailure, "combined selection (incomplete sweep)"); r->result = -1; }
    else {
      CheckPriority(c.h->out[i].priority, pick);
      // reference PqaCore/CpuEngine.cpp:403-413 (FinishSelection, over the snapshot)
      if (pick >= 0 && BitTest(r->unavailable, pick)) pick = FindNearestInPacks(pick, r->nQ, [&](int64_t p) { return ~Pack64(r->unavailable, p); });
      if (pick < 0) {
        r->err = Error::Make(ErrCode::QuestionsExhausted, "Found no unasked question that is not in a gap.");
        r->result = -1;
      } else {
        q->activeQuestion = _qFirst + pick;
        _nQuestionsAsked.fetch_add(1, std::memory_order_relaxed);
        r->result = q->activeQuestion;
      }
    }
    q->inSelection.store(false, std::memory_order_release);
  }
  _combNs[3] += ns(tD, tE);
  _combNs[4] += ns(tE, std::chrono::steady_clock::now());
  return false;
}

Error HipEngine::EvalPriorities(int64_t iQuiz, double *pOut, int64_t n) {
  std::lock_guard<EngineMutex> lk(_mu);
  Error err = CheckRegular("compute next question");
  if (!err.ok()) return err;
  Quiz *q = UseQuiz(err, iQuiz);
  if (!q) return err;
  if (!pOut) return Error::Make(ErrCode::NullArgument, "Nullptr is passed in place of the priority buffer.");
  if (n != _Q) return Error::MakeP(ErrCode::IndexOutOfRange, RangeParams(n, _Q, _Q), "Priority buffer length must equal the local question count.");
  hipSetDevice(_device);
  StopServer();   // a launched sweep has no room beside the resident one and would wait for it to idle out
  err = LaunchSingleSweep(q, nullptr);
  if (!err.ok()) return err;
  HIP_TRY(hipMemcpyAsync(pOut, _dPriority, (size_t)_Q * sizeof(double), hipMemcpyDeviceToHost, _stream));
  HIP_TRY(hipStreamSynchronize(_stream));
  return Error();
}

}  // namespace pqa
