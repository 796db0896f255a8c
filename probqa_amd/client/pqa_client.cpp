// pqa_client.cpp -- the reference's learner client, restated for measuring the engine under MANY CONCURRENT CLIENT THREADS.
//
// Reference: ProbQA/PqaClient/PqaClient.cpp:150-245 -- `hardware_concurrency` learner threads on ONE engine, each running quiz
// after quiz (StartQuiz, then NextQuestion / RecordAnswer / ListTopTargets until the guessed target is on top or the question
// budget is spent, then RecordQuizTarget and ReleaseQuiz); the only rate the reference publishes (301.2
// NextQuestion/s) is the sum over those threads.  The answer rule is the trainer's of PqaCoreTests/DichotomyTest.cpp:50-64.
//
// This file is a CLIENT of libPqaCore.so: it calls the reference's C ABI (include/PqaCInterop.h) and nothing else -- no engine
// internals, no HIP.  bench.py's `quiz_loop_threads` extra and tests/test_gpu_concurrent.py drive it through ctypes.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/PqaCInterop.h"

extern "C" {

typedef struct {
  int64_t nQuizzes;        // quizzes completed
  int64_t nQuestions;      // NextQuestion calls answered
  int64_t nGuessedOnTop;   // quizzes that ended with the guessed target on top
  int64_t nErrors;         // calls of the ABI that returned an error
  double seconds;          // wall time of the threaded region
  uint64_t transcriptHash; // order-independent digest of every (guess, question, answer, top target) of the run
} PqaClientStats;

static inline uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

// nThreads learner threads share nQuizzes quizzes (a common counter hands them out); quiz i guesses target
// mix64(seed + i) % nTargets.  train: bit 0 = RecordQuizTarget at the end of every quiz, as the reference's learner does; bit 1 = a
// client that does not look at the targets between its questions (no ListTopTargets: RecordAnswer is followed by NextQuestion
// directly, every quiz runs to maxQuestions).
__attribute__((visibility("default"))) int64_t PqaClient_RunLearners(void *pvEngine, int64_t nThreads, int64_t nQuizzes, int64_t maxQuestions,
                                                                     uint64_t seed, int64_t train, PqaClientStats *pStats) {
  if (!pvEngine || !pStats || nThreads < 1 || nQuizzes < 0 || maxQuestions < 1) return -1;
  CiEngineDimensions dims;
  if (!PqaEngine_CopyDims(pvEngine, &dims)) return -1;
  const int64_t Q = dims._nQuestions, T = dims._nTargets;
  const int64_t width = (32 * T) / 1000 > 1 ? (32 * T) / 1000 : 1;
  std::atomic<int64_t> nextQuiz{0}, questions{0}, onTop{0}, errors{0}, done{0};
  std::atomic<uint64_t> digest{0};
  std::atomic<int64_t> nsIn[6] = {{0}, {0}, {0}, {0}, {0}, {0}};   // StartQuiz, NextQuestion, RecordAnswer, ListTopTargets, RecordQuizTarget, ReleaseQuiz
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto spent = [&](int which, std::chrono::steady_clock::time_point t0) {
    nsIn[which].fetch_add(std::chrono::duration_cast<std::chrono::nanoseconds>(now() - t0).count(), std::memory_order_relaxed);
  };
  const bool verbose = std::getenv("PQA_CLIENT_VERBOSE") != nullptr;
  auto failed = [&](void *e, const char *what) {
    if (verbose && errors.load() < 5) {
      void *str = PqaError_ToString(e, 1);
      std::fprintf(stderr, "pqa_client: %s failed: %s\n", what, str ? (const char *)str : "?");
      if (str) CiReleaseString(str);
    }
    CiReleasePqaError(e);
    errors++;
  };
  auto learner = [&]() {
    for (;;) {
      const int64_t i = nextQuiz.fetch_add(1, std::memory_order_relaxed);
      if (i >= nQuizzes) return;
      const int64_t guess = (int64_t)(mix64(seed + (uint64_t)i) % (uint64_t)T);
      void *err = nullptr;
      auto t0 = now();
      const int64_t quiz = PqaEngine_StartQuiz(pvEngine, &err);
      spent(0, t0);
      if (err) { failed(err, "StartQuiz"); continue; }
      uint64_t h = mix64((uint64_t)guess + 0x9E3779B97F4A7C15ULL);
      bool top = false;
      for (int64_t j = 0; j < maxQuestions && !top; j++) {
        t0 = now();
        const int64_t q = PqaEngine_NextQuestion(pvEngine, &err, quiz);
        spent(1, t0);
        if (err) { failed(err, "NextQuestion"); err = nullptr; break; }
        questions.fetch_add(1, std::memory_order_relaxed);
        const int64_t x = q * T / Q;   // the target the question "asks about" on the synthetic binary-search cube
        const int64_t a = guess < x - width ? 0 : guess < x ? 1 : guess == x ? 2 : guess <= x + width ? 3 : 4;
        t0 = now();
        void *e1 = PqaEngine_RecordAnswer(pvEngine, quiz, a);
        spent(2, t0);
        if (e1) { failed(e1, "RecordAnswer"); break; }
        CiRatedTarget best;
        best._iTarget = -1;
        int64_t n = 0;
        if (!(train & 2)) {
          t0 = now();
          n = PqaEngine_ListTopTargets(pvEngine, &err, quiz, 1, &best);
          spent(3, t0);
          if (err) { failed(err, "ListTopTargets"); err = nullptr; break; }
        }
        h = mix64(h ^ mix64((uint64_t)q * 31 + (uint64_t)a) ^ (uint64_t)(n > 0 ? best._iTarget : -1));
        top = n > 0 && best._iTarget == guess;
      }
      if (top) onTop++;
      if (train & 1) {
        t0 = now();
        void *e2 = PqaEngine_RecordQuizTarget(pvEngine, quiz, guess, 1.0);
        spent(4, t0);
        if (e2) failed(e2, "RecordQuizTarget");
      }
      t0 = now();
      void *e3 = PqaEngine_ReleaseQuiz(pvEngine, quiz);
      spent(5, t0);
      if (e3) failed(e3, "ReleaseQuiz");
      digest.fetch_add(h, std::memory_order_relaxed);   // (a sum: the order in which the threads finish does not matter)
      done++;
    }
  };
  const auto t0 = std::chrono::steady_clock::now();
  if (nThreads == 1) {
    learner();
  } else {
    std::vector<std::thread> threads;
    for (int64_t t = 0; t < nThreads; t++) threads.emplace_back(learner);
    for (auto &t : threads) t.join();
  }
  pStats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  pStats->nQuizzes = done.load();
  pStats->nQuestions = questions.load();
  pStats->nGuessedOnTop = onTop.load();
  pStats->nErrors = errors.load();
  pStats->transcriptHash = digest.load();
  if (verbose) {
    const double nq = (double)std::max<int64_t>(1, questions.load()), nz = (double)std::max<int64_t>(1, done.load());
    std::fprintf(stderr, "pqa_client: %lld threads, us per call: NextQuestion %.1f  RecordAnswer %.1f  ListTopTargets %.1f | per quiz: StartQuiz %.1f  RecordQuizTarget %.1f  ReleaseQuiz %.1f\n",
                 (long long)nThreads, nsIn[1] / nq * 1e-3, nsIn[2] / nq * 1e-3, nsIn[3] / nq * 1e-3, nsIn[0] / nz * 1e-3, nsIn[4] / nz * 1e-3, nsIn[5] / nz * 1e-3);
  }
  return 0;
}

// The synchronous selection step from a NATIVE caller: n x PqaEngine_NextQuestion on one quiz (the engine's selector as its options
// say), timed here -- what bench.py's step costs without the Python wrapper around every call.  Returns the seconds, < 0 on an error;
// pLast: the last selected question.
__attribute__((visibility("default"))) double PqaClient_TimeSelections(void *pvEngine, int64_t iQuiz, int64_t nWarm, int64_t n, int64_t *pLast) {
  if (!pvEngine || n < 1) return -1.0;
  int64_t last = -1;
  for (int64_t i = 0; i < nWarm; i++) {
    void *e = nullptr;
    last = PqaEngine_NextQuestion(pvEngine, &e, iQuiz);
    if (e) { CiReleasePqaError(e); return -1.0; }
  }
  const auto t0 = std::chrono::steady_clock::now();
  for (int64_t i = 0; i < n; i++) {
    void *e = nullptr;
    last = PqaEngine_NextQuestion(pvEngine, &e, iQuiz);
    if (e) { CiReleasePqaError(e); return -1.0; }
  }
  const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (pLast) *pLast = last;
  return dt;
}

}  // extern "C"
