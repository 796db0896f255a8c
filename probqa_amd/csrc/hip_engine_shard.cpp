// hip_engine_shard.cpp -- what a sharded engine (sharded_engine.cpp: one knowledge base, its question axis split over several
// HipEngines of one process) asks of its shards beyond the public surface.  SURVEY 8(e): a question's priority depends on its own
// rows and the (replicated) posterior only, so a shard sweeps its questions by itself; what needs the other shards is the
// posterior update of an answered question that another shard holds -- every shard computes it, reading that question's two
// rows where they are.
#include "hip_engine_internal.h"

namespace pqa {

Error HipEngine::GetRowPointers(int64_t qGlobal, int64_t iAnswer, const void **ppA, const void **ppD) {
  std::lock_guard<EngineMutex> lk(_mu);
  if (!OwnsQuestion(qGlobal))
    return Error::MakeP(ErrCode::IndexOutOfRange, RangeParams(qGlobal, _qFirst, _qFirst + _Q - 1), "Question is not held by this shard.");
  if (iAnswer < 0 || iAnswer >= _K)
    return Error::MakeP(ErrCode::IndexOutOfRange, RangeParams(iAnswer, 0, _K - 1), "Answer index is not in KB range.");
  *ppA = CubeAt(qGlobal - _qFirst, iAnswer);
  *ppD = CubeAt(qGlobal - _qFirst, _K);
  return Error();
}

// ResumeQuiz from row POINTERS (prior_kernels.hip: resume_quiz_kernel): the rows of questions other shards hold are read in place
// over peer access.  stageRow[i] set: row i (living on device rowDevices[i]) is first copied into a scratch buffer on this device
// (hipMemcpyPeerAsync needs no peer access), for hosts whose devices cannot map each other's memory.
int64_t HipEngine::ResumeQuizRows(Error &err, int64_t nAnswered, const AQ *pAQs, const void *const *rows, const int *rowDevices, const char *stageRow) {
  std::lock_guard<EngineMutex> lk(_mu);
  if (rowDevices == nullptr || stageRow == nullptr || nAnswered <= 0) return CreateQuiz(err, nAnswered, pAQs, rows, nullptr, 0, nullptr);
  hipSetDevice(_device);
  const size_t rowBytes = (size_t)_ldT * (size_t)_elem;
  std::vector<const void *> local(rows, rows + 2 * nAnswered);
  size_t nStage = 0;
  for (int64_t i = 0; i < 2 * nAnswered; i++) nStage += stageRow[i] ? 1 : 0;
  char *stage = nullptr;
  if (nStage > 0) {
    hipError_t he = hipMalloc((void **)&stage, nStage * rowBytes);
    size_t at = 0;
    for (int64_t i = 0; he == hipSuccess && i < 2 * nAnswered; i++) {
      if (!stageRow[i]) continue;
      he = hipMemcpyPeerAsync(stage + at * rowBytes, _device, rows[i], rowDevices[i], rowBytes, _stream);
      local[(size_t)i] = stage + at * rowBytes;
      at++;
    }
    if (he != hipSuccess) { hipFree(stage); err = HipErr(he, "staging another device's rows for ResumeQuiz"); return -1; }
  }
  const int64_t id = CreateQuiz(err, nAnswered, pAQs, local.data(), nullptr, 0, nullptr);   // (synchronises the stream: the copies are done)
  if (stage) { hipStreamSynchronize(_stream); hipFree(stage); }
  return id;
}

// bit i of words[i / 32] set = LOCAL question i is asked in the quiz or a gap (bits past the local count set)
Error HipEngine::UnavailableWords(int64_t iQuiz, std::vector<uint32_t> &words) {
  std::lock_guard<EngineMutex> lk(_mu);
  Error err;
  Quiz *q = UseQuiz(err, iQuiz);
  if (!q) return err;
  words.resize(_hQGap.size());
  for (size_t w = 0; w < words.size(); w++) words[w] = _hQGap[w] | q->hAsked[w];
  return Error();
}

Error HipEngine::FlushDeferred() {
  std::lock_guard<EngineMutex> lk(_mu);
  hipSetDevice(_device);
  return FlushUpdates();
}

// The answers the sharded engine has gathered, on this shard: per answer the bookkeeping of CEQuiz::RecordAnswer (PqaCore/CEQuiz.h:
// 77-122: the answer joins the quiz's list, the question counts as asked) -- the sharded engine has validated quiz, question and
// answer -- and then ONE launch for all the posteriors (FlushUpdates: record_answer_batch_kernel).  A question another shard holds
// comes with its two rows: read in place, or (`stage`) copied into the quiz's own staging rows first, in stream order.
Error HipEngine::ApplyAnswers(int64_t n, const ShardAnswer *answers) {
  if (n <= 0) return Error();
  std::lock_guard<EngineMutex> lk(_mu);
  Error err = CheckRegular("record an answer");
  if (!err.ok()) return err;
  hipSetDevice(_device);
  ServerQuiesce();
  const size_t rowBytes = (size_t)_ldT * (size_t)_elem;
  for (int64_t i = 0; i < n; i++) {
    const ShardAnswer &a = answers[i];
    Quiz *q = UseQuiz(err, a.iQuiz);
    if (!q) return err;
    if (a.iAnswer < 0 || a.iAnswer >= _K || a.qGlobal < 0 || a.qGlobal >= _qTotal)
      return Error::MakeP(ErrCode::Internal, "quizId=" + std::to_string(a.iQuiz), "An answer reached a shard unvalidated.");
    if (q->updatePending) {   // (a second answer of a quiz whose first is still deferred: that one runs now)
      Error fe = FlushUpdates();
      if (!fe.ok()) return fe;
    }
    const bool local = OwnsQuestion(a.qGlobal);
    if (!local && (a.rowA == nullptr || a.rowD == nullptr))
      return Error::MakeP(ErrCode::Internal, "quizId=" + std::to_string(a.iQuiz), "Another shard's question came without its rows.");
    q->answers.push_back(AQ{a.qGlobal, a.iAnswer});
    q->activeQuestion = -1;
    q->priorVersion++;
    PendingUpdate u{q, local ? a.qGlobal - _qFirst : 0, a.iAnswer, nullptr, nullptr, a.list};
    if (local) {
      BitSet(q->hAsked, a.qGlobal - _qFirst, true);
    } else if (a.stage) {
      if (q->dRowStage == nullptr) HIP_TRY(hipMalloc(&q->dRowStage, 2 * rowBytes));
      HIP_TRY(hipMemcpyPeerAsync(q->dRowStage, _device, a.rowA, a.srcDevice, rowBytes, _stream));
      HIP_TRY(hipMemcpyPeerAsync(static_cast<char *>(q->dRowStage) + rowBytes, _device, a.rowD, a.srcDevice, rowBytes, _stream));
      u.rowA = q->dRowStage;
      u.rowD = static_cast<char *>(q->dRowStage) + rowBytes;
    } else {
      u.rowA = a.rowA;
      u.rowD = a.rowD;
    }
    _pendingUpdates.push_back(u);
    q->updatePending = true;
  }
  _pendingCount.store(_pendingUpdates.size(), std::memory_order_relaxed);
  return FlushUpdates();
}

// First half of a combined sweep (see hip_engine.h).  A quiz that does not exist fails the call: the sharded engine keeps the
// registries of all shards in step and has checked.
Error HipEngine::EnqueueCombined(int ctx, int64_t n, const int64_t *pQuizzes, bool hostPriorities, CombinedFlight *f, std::vector<uint32_t> *unavailable) {
  *f = CombinedFlight();
  if (n <= 0) return Error();
  if (ctx < 0 || ctx > 1 || n > kMaxBatch || pQuizzes == nullptr)
    return Error::MakeP(ErrCode::IndexOutOfRange, RangeParams(n, 0, kMaxBatch), "Batch size or context is out of range.");
  std::lock_guard<EngineMutex> lk(_mu);
  Error err = CheckRegular("compute next question");
  if (!err.ok()) return err;
  hipSetDevice(_device);
  err = FlushUpdates();
  if (!err.ok()) return err;
  BatchCtx &c = _ctx[ctx];
  f->tag = NextLaunchTag();
  f->hostPriorities = hostPriorities;
  std::vector<Quiz *> quizzes;
  err = BatchSweep(c, n, pQuizzes, quizzes, false, f->tag, hostPriorities, &f->quizMinor, &f->tagged);
  if (!err.ok()) return err;
  if (unavailable != nullptr) {
    const size_t words = _hQGap.size();
    unavailable->resize((size_t)n * words);
    for (int64_t i = 0; i < n; i++)
      for (size_t w = 0; w < words; w++) (*unavailable)[(size_t)i * words + w] = _hQGap[w] | quizzes[(size_t)i]->hAsked[w];
  }
  f->Bp = c.lastBp;
  f->nQ = _Q;
  f->n = n;
  if (hostPriorities && !f->tagged) f->he = hipEventRecord(c.event, _stream);
  _lastCombined.store(n, std::memory_order_relaxed);
  _flushedSinceSweep.store(0, std::memory_order_relaxed);
  MarkStreamBusy();
  return Error();
}

// Second half: no lock of the engine is taken -- the flags, the event and the host buffers belong to the batch context, which the
// caller holds until every reader of the priority views is done.
Error HipEngine::CollectCombined(int ctx, const CombinedFlight &f, CiHipSelection *winners, PriorityView *views) {
  if (f.n <= 0) return Error();
  BatchCtx &c = _ctx[ctx];
  hipSetDevice(_device);
  hipError_t he = f.he;
  Error err;
  const bool byEvent = f.hostPriorities && !f.tagged;
  if (he == hipSuccess && byEvent) he = hipEventSynchronize(c.event);
  if (he != hipSuccess) return HipErr(he, "combined selection");
  if (!byEvent) {
    err = WaitBatchFlags(c, f.n, f.tag);
    if (!err.ok()) return err;
    for (int64_t i = 0; i < f.n; i++)
      if (c.h->out[i].index == -3) return HipErr(hipErrorLaunchFailure, "combined selection (incomplete sweep)");
  }
  for (int64_t i = 0; i < f.n; i++) {
    if (winners != nullptr && !f.hostPriorities) {   // (a batch that hands priority vectors over is selected from those)
      CheckPriority(c.h->out[i].priority, c.h->out[i].index);
      winners[i]._priority = c.h->out[i].priority;
      winners[i]._iQuestion = c.h->out[i].index < 0 ? -1 : c.h->out[i].index + _qFirst;
    }
    if (views != nullptr && f.hostPriorities) {
      views[i].pri = f.tagged ? c.hPri + 2 * (size_t)i * (size_t)f.nQ : f.quizMinor ? c.hPri + i : c.hPri + (size_t)i * (size_t)f.nQ;
      views[i].stride = f.tagged ? 2 : f.quizMinor ? f.Bp : 1;
      views[i].tag = f.tagged ? f.tag : 0;
    }
  }
  return Error();
}

}  // namespace pqa
