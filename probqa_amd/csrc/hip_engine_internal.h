// hip_engine_internal.h -- small helpers shared by the files that implement HipEngine (hip_engine.cpp: construction, options,
// registry; hip_engine_select.cpp: the selection paths; hip_engine_combine.cpp: concurrent clients; hip_engine_update.cpp:
// posterior updates, listings, training; hip_engine_shard.cpp: what a sharded engine asks of its shards).
#pragma once

#include <immintrin.h>
#include <linux/futex.h>
#include <sched.h>
#include <sys/prctl.h>
#include <sys/syscall.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <random>
#include <sstream>

#include "hip_engine.h"

namespace pqa {
namespace {

inline Error HipErr(hipError_t e, const char *what) {
  std::string msg = std::string("HIP failure in ") + what + ": " + hipGetErrorString(e);
  DefaultLogger::Log(DefaultLogger::Severity::Error, msg);
  return Error::MakeP(ErrCode::Internal, std::string("Internal error at hip_engine.cpp(") + what + ")", msg);
}
#define HIP_TRY(expr)                                   \
  do {                                                  \
    const hipError_t e_ = (expr);                       \
    if (e_ != hipSuccess) return HipErr(e_, #expr);     \
  } while (0)

inline std::string RangeParams(int64_t subj, int64_t lo, int64_t hi) {  // IndexOutOfRangeErrorParams::ToString
  return "subjIndex=" + std::to_string(subj) + " not in " + std::to_string(lo) + "..." + std::to_string(hi);
}

inline bool BitTest(const std::vector<uint32_t> &bits, int64_t i) { return (bits[i >> 5] >> (i & 31)) & 1u; }
inline void BitSet(std::vector<uint32_t> &bits, int64_t i, bool v) {
  if (v) bits[i >> 5] |= 1u << (i & 31); else bits[i >> 5] &= ~(1u << (i & 31));
}
inline size_t BitWords(int64_t nBits) { return (size_t)((nBits + 63) / 64) * 2 + 2; }  // whole 64-bit packs + slack
inline uint64_t Pack64(const std::vector<uint32_t> &bits, int64_t iPack) {
  return (uint64_t)bits[2 * iPack] | ((uint64_t)bits[2 * iPack + 1] << 32);
}

inline uint64_t SplitMix64(uint64_t &x) {
  uint64_t z = (x += 0x9E3779B97F4A7C15ULL);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

// A waiting client sleeps on ITS OWN request's state word and is woken alone (futex): with one condition variable for all
// requests every published batch woke every sleeper, most of them only to find their own request unserved and sleep again.
inline void FutexWait(std::atomic<int> *word, int expected) {
  syscall(SYS_futex, reinterpret_cast<int *>(word), FUTEX_WAIT_PRIVATE, expected, nullptr, nullptr, 0);
}
inline void FutexWakeOne(std::atomic<int> *word) { syscall(SYS_futex, reinterpret_cast<int *>(word), FUTEX_WAKE_PRIVATE, 1, nullptr, nullptr, 0); }
// The new state, then the wake -- always: whether the owner sleeps cannot be asked once the state is stored (it may have seen it,
// returned and gone with its request), and a wake on a word nobody sleeps on only costs the call.
inline void PublishState(std::atomic<int> *word, int state) {
  word->store(state, std::memory_order_release);
  FutexWakeOne(word);
}
static_assert(sizeof(std::atomic<int>) == sizeof(int), "the state word is slept on as a futex");

// A wait for a word the GPU writes: a pure spin while the answer is a kernel's time away (the first ~50 us), then the core is
// offered to whoever else wants it between looks (sched_yield) -- a process whose clients outnumber its CPUs otherwise burns its
// allowance on waiting -- and the clock is read only every so often.  Tick() returns false once `limit` has passed.
struct SpinWait {
  uint64_t spins = 0;
  bool yielding = false;
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  // (by TIME, not by count: a pause is 15 - 40 ns on the hosts met so far, and a count that ran out just before a 20 us step's answer
  //  put a system call -- a microsecond -- into every selection)
  bool Tick(std::chrono::seconds limit) {
    ++spins;
    if (!yielding) {
      _mm_pause();
      if ((spins & 63) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(200)) yielding = true;
      return true;
    }
    if ((spins & 0xFF) == 0 && std::chrono::steady_clock::now() - t0 > limit) return false;
    sched_yield();
    return true;
  }
  bool Due() const { return (spins & 0xFFF) == 0; }   // (for the occasional look at something else while waiting)
};


}  // namespace
}  // namespace pqa
