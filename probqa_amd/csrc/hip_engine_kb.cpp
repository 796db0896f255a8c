// hip_engine_kb.cpp -- the parts of the engine around the hot path that change or persist the knowledge base:
// permanent<->compact id maps, quiz registry slots, .kb files, and the maintenance-mode operations.
// Reference: PqaCore/PermanentIdManager.cpp, PqaCore/BaseEngine.cpp:124-215,323-385,704-873,
// PqaCore/CpuEngine.cpp:468-658,664-688, PqaCore/PqaEngineBaseFactory.cpp:44-83.  Host bookkeeping is the reference's;
// the cube itself stays on the device and is edited there (kb_kernels.hip).
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "hip_engine.h"

namespace pqa {

namespace {

Error HipErr(hipError_t e, const char *what) {
  std::string msg = std::string("HIP failure in ") + what + ": " + hipGetErrorString(e);
  return Error::MakeP(ErrCode::Internal, std::string("Internal error at hip_engine_kb.cpp(") + what + ")", msg);
}
#define HIP_TRY(expr)                                   \
  do {                                                  \
    const hipError_t e_ = (expr);                       \
    if (e_ != hipSuccess) return HipErr(e_, #expr);     \
  } while (0)

inline bool BitTest(const std::vector<uint32_t> &bits, int64_t i) { return (bits[i >> 5] >> (i & 31)) & 1u; }
inline void BitSet(std::vector<uint32_t> &bits, int64_t i, bool v) {
  if (v) bits[i >> 5] |= 1u << (i & 31); else bits[i >> 5] &= ~(1u << (i & 31));
}
inline size_t BitWords(int64_t nBits) { return (size_t)((nBits + 63) / 64) * 2 + 2; }

Error FileErr(const char *path, const char *msg) {
  return Error::MakeP(ErrCode::FileOp, std::string("filePath=[") + path + "]", msg);
}

struct FileCloser {
  FILE *f;
  ~FileCloser() { if (f) std::fclose(f); }
};

Error WrongModeErr(const char *what) {
  return Error::Make(ErrCode::WrongMode, std::string("Can't perform maintenance-only mode operation - ") + what +
                                             " - because current mode is not maintenance (but regular/shutdown?).");
}

// PrecisionDefinition bitfield of reference PqaCore/Interface/PqaCommon.h:26-32 (type:4, mantissa:28, exponent:16, reserved:16)
uint64_t PackPrecision(uint64_t type, uint64_t mantissa, uint64_t exponent) {
  return (type & 0xF) | ((mantissa & 0xFFFFFFF) << 4) | ((exponent & 0xFFFF) << 32);
}

template <typename T>
hipError_t Upload(T **dst, const std::vector<T> &src, hipStream_t stream) {
  *dst = nullptr;
  if (src.empty()) return hipSuccess;
  hipError_t e = hipMalloc(dst, src.size() * sizeof(T));
  if (e != hipSuccess) return e;
  return hipMemcpyAsync(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice, stream);
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------------
// IdLedger: compact slot <-> permanent id (hip_engine.h).  What callers and files observe follows the reference's
// PermanentIdManager; see the class comment for the structure.
// ------------------------------------------------------------------------------------------------------------------
size_t IdLedger::LowerBound(int64_t permanent) const {
  size_t lo = 0, hi = _byPerm.size();
  while (lo < hi) {
    const size_t mid = lo + (hi - lo) / 2;
    if (_byPerm[mid].permanent < permanent) lo = mid + 1; else hi = mid;
  }
  return lo;
}

int64_t IdLedger::SlotOf(int64_t permanent) const {
  const size_t at = LowerBound(permanent);
  if (at == _byPerm.size() || _byPerm[at].permanent != permanent || !Live(_byPerm[at])) return kNone;
  return _byPerm[at].slot;
}

// Records that `slot` now carries `permanent` (the forward table is already written).  A leftover entry of the same id -- its
// slot was vacated earlier -- is taken over; otherwise the entry goes where the order puts it, which for a freshly issued id is
// the end.
void IdLedger::Enter(int64_t permanent, int64_t slot) {
  if (_byPerm.empty() || _byPerm.back().permanent < permanent) { _byPerm.push_back(Back{permanent, slot}); return; }
  const size_t at = LowerBound(permanent);
  if (at < _byPerm.size() && _byPerm[at].permanent == permanent) _byPerm[at].slot = slot;
  else _byPerm.insert(_byPerm.begin() + (ptrdiff_t)at, Back{permanent, slot});
}

void IdLedger::Rebuild() {
  _byPerm.clear();
  _live = 0;
  for (int64_t slot = 0; slot < (int64_t)_permOf.size(); slot++)
    if (_permOf[(size_t)slot] != kNone) { _byPerm.push_back(Back{_permOf[(size_t)slot], slot}); _live++; }
  std::sort(_byPerm.begin(), _byPerm.end(), [](const Back &x, const Back &y) { return x.permanent < y.permanent; });
}

bool IdLedger::Write(FILE *f, bool withoutSlots) const {
  const int64_t header[2] = {_issueNext, withoutSlots ? 0 : (int64_t)_permOf.size()};
  if (std::fwrite(header, sizeof(header), 1, f) != 1) return false;
  return header[1] == 0 || std::fwrite(_permOf.data(), sizeof(int64_t), (size_t)header[1], f) == (size_t)header[1];
}

bool IdLedger::Read(FILE *f) {
  int64_t header[2];
  if (std::fread(header, sizeof(header), 1, f) != 1 || header[1] < 0) return false;
  std::vector<int64_t> table((size_t)header[1]);
  if (header[1] > 0 && std::fread(table.data(), sizeof(int64_t), table.size(), f) != table.size()) return false;
  _issueNext = header[0];
  _permOf.swap(table);
  Rebuild();
  return true;
}

bool IdLedger::RaiseFloor(int64_t bound) {
  if (bound < _issueNext) return false;
  _issueNext = bound + 1;
  return true;
}

bool IdLedger::Vacate(int64_t slot) {
  if (!InRange(slot) || _permOf[(size_t)slot] == kNone) return false;
  _permOf[(size_t)slot] = kNone;     // its entry in _byPerm no longer agrees with the table: a leftover from here on
  _live--;
  if (_byPerm.size() > 64 && (int64_t)_byPerm.size() > 2 * _live) {
    size_t kept = 0;
    for (const Back &b : _byPerm) if (Live(b)) _byPerm[kept++] = b;
    _byPerm.resize(kept);
  }
  return true;
}

bool IdLedger::Reissue(int64_t slot) {
  if (!InRange(slot) || _permOf[(size_t)slot] != kNone) return false;   // (a slot that still holds an id is vacated first)
  _permOf[(size_t)slot] = _issueNext;
  Enter(_issueNext++, slot);
  _live++;
  return true;
}

bool IdLedger::Extend(int64_t nSlots) {
  if (nSlots < (int64_t)_permOf.size()) return false;
  _permOf.reserve((size_t)nSlots);
  while ((int64_t)_permOf.size() < nSlots) {
    Enter(_issueNext, (int64_t)_permOf.size());
    _permOf.push_back(_issueNext++);
    _live++;
  }
  return true;
}

// Compaction: the nSlots live slots move to 0 .. nSlots-1, slot i taking the permanent id slot from[i] held.  Checked as a whole
// before anything changes: exactly the live slots, each once.
bool IdLedger::Repack(int64_t nSlots, const int64_t *from) {
  if (nSlots != _live || nSlots > (int64_t)_permOf.size()) return false;
  std::vector<int64_t> packed((size_t)nSlots);
  std::vector<bool> taken(_permOf.size(), false);
  for (int64_t i = 0; i < nSlots; i++) {
    const int64_t src = from[i];
    if (!InRange(src) || _permOf[(size_t)src] == kNone || taken[(size_t)src]) return false;
    taken[(size_t)src] = true;
    packed[(size_t)i] = _permOf[(size_t)src];
  }
  _permOf.swap(packed);
  Rebuild();
  return true;
}

bool IdLedger::Rename(int64_t permanent, int64_t toPermanent) {
  if (toPermanent < 0 || toPermanent >= _issueNext) return false;   // only ids that can no longer be issued (the reference lets the
                                                                    // invalid id -1 through and corrupts its maps with it: refused here)
  if (SlotOf(toPermanent) != kNone) return false;       // in use
  const int64_t slot = SlotOf(permanent);
  if (slot == kNone) return false;
  _permOf[(size_t)slot] = toPermanent;                  // the old id's entry becomes a leftover
  Enter(toPermanent, slot);
  return true;
}

std::vector<int64_t> QuizzesToLetGo(const std::vector<QuizUsage> &quizzes, time_t now, int64_t maxCount, double maxAgeSec) {
  std::vector<int64_t> out;
  std::vector<QuizUsage> rest;
  for (const QuizUsage &u : quizzes) {
    if (difftime(now, u.lastUsage) > maxAgeSec) out.push_back(u.id); else rest.push_back(u);
  }
  if ((int64_t)rest.size() > maxCount) {
    const size_t surplus = rest.size() - (size_t)maxCount;
    // (stable: equal usage times -- the clock has one-second resolution -- stay in registry order)
    std::stable_sort(rest.begin(), rest.end(), [](const QuizUsage &x, const QuizUsage &y) { return x.lastUsage < y.lastUsage; });
    for (size_t i = 0; i < surplus; i++) out.push_back(rest[i].id);
  }
  return out;
}

// ------------------------------------------------------------------------------------------------------------------
// id maps and the quiz registry (reference PqaCore/BaseEngine.cpp:150-215, 780-802)
// ------------------------------------------------------------------------------------------------------------------
bool HipEngine::MapIds(int which, bool toPerm, int64_t count, int64_t *pIds) {
  std::lock_guard<EngineMutex> lk(_mu);
  StopServer();
  IdLedger &ids = which == 0 ? _questionIds : which == 1 ? _targetIds : _quizIds;
  for (int64_t i = 0; i < count; i++) pIds[i] = toPerm ? ids.PermanentOf(pIds[i]) : ids.SlotOf(pIds[i]);
  return true;
}
bool HipEngine::EnsurePermQuizGreater(int64_t bound) {
  std::lock_guard<EngineMutex> lk(_mu);
  StopServer();
  return _quizIds.RaiseFloor(bound);
}
bool HipEngine::RemapQuizPermId(int64_t srcPermId, int64_t destPermId) {
  std::lock_guard<EngineMutex> lk(_mu);
  StopServer();
  return _quizIds.Rename(srcPermId, destPermId);
}

int64_t HipEngine::AssignQuiz(Quiz *q) {
  int64_t id;
  if (!_quizGaps.empty()) {
    id = _quizGaps.back();
    _quizGaps.pop_back();
    _quizIds.Reissue(id);
  } else {
    id = (int64_t)_quizzes.size();
    _quizzes.push_back(nullptr);
    _quizIds.Extend((int64_t)_quizzes.size());
  }
  _quizzes[(size_t)id] = q;
  return id;
}

void HipEngine::UnassignQuiz(int64_t iQuiz) {
  _quizzes[(size_t)iQuiz] = nullptr;
  _quizGaps.push_back(iQuiz);
  _quizIds.Vacate(iQuiz);
}

Error HipEngine::ClearOldQuizzes(int64_t maxCount, double maxAgeSec) {  // behaviour: BaseEngine.cpp:814-873
  if (maxCount < 0)
    return Error::MakeP(ErrCode::NegativeCount, "count=" + std::to_string(maxCount),
                        "The number of quizzes to keep cannot be less than 0.");
  std::lock_guard<EngineMutex> lk(_mu);
  StopServer();
  if (_mode != Mode::Regular) return Error();  // quizzes are not expected to exist in maintenance / shutdown mode
  hipSetDevice(_device);
  hipStreamSynchronize(_stream);
  std::vector<QuizUsage> inUse;
  for (size_t slot = 0; slot < _quizzes.size(); slot++)
    if (_quizzes[slot]) inUse.push_back(QuizUsage{(int64_t)slot, _quizzes[slot]->lastUsage});
  for (int64_t id : QuizzesToLetGo(inUse, time(nullptr), maxCount, maxAgeSec)) {
    Quiz *q = _quizzes[(size_t)id];
    UnassignQuiz(id);
    DestroyQuiz(q);
  }
  return Error();
}

// ------------------------------------------------------------------------------------------------------------------
// the arrays of a .kb file, this engine's questions only: sequential I/O at the file's current position through a bounded
// host staging buffer.  The same code serves a whole-cube engine and every shard of a sharded one (the file orders its rows
// by question, so the shards' blocks follow each other).
// ------------------------------------------------------------------------------------------------------------------
// Two pinned staging buffers in turn: the file's read of one batch runs while the other batch's rows are on their way to the
// device (and a save's write while the next batch comes back), so a large knowledge base moves at the slower of the two rates, not
// at their sum (reference: PqaCore/BaseEngine.cpp:323-385 writes row by row; PqaCore/CudaPersistence.cpp:15-43 stages through one
// pageable buffer).  The mD rows of a batch are ONE strided copy (a row per question, (K + 1) ldT elements apart); a question's K sA
// rows are one.
Error HipEngine::IoRows(FILE *f, const char *filePath, bool mD, bool write) {   // sA rows [q][a] of T elements, or mD rows [q]
  hipSetDevice(_device);
  const size_t rowB = (size_t)_T * (size_t)_elem, ldB = (size_t)_ldT * (size_t)_elem;
  const int64_t rowsPerQ = mD ? 1 : _K;
  const int64_t batch = std::max<int64_t>(1, (int64_t)((64u << 20) / (rowB * (size_t)rowsPerQ)));
  const size_t bufBytes = (size_t)std::min(batch, _Q) * (size_t)rowsPerQ * rowB;
  struct Staging {
    char *buf[2] = {nullptr, nullptr};
    hipEvent_t done[2] = {nullptr, nullptr};
    bool pinned = false;
    std::vector<char> pageable[2];
    hipStream_t stream = nullptr;
    ~Staging() {
      (void)hipStreamSynchronize(stream);   // (an early return -- a failed read, a failed copy -- leaves copies in flight: not under buffers about to go)
      for (int i = 0; i < 2; i++) {
        if (pinned && buf[i]) hipHostFree(buf[i]);
        if (done[i]) hipEventDestroy(done[i]);
      }
    }
  } st;
  const int nBuf = _Q > batch ? 2 : 1;
  st.stream = _stream;
  st.pinned = true;
  for (int i = 0; i < nBuf && st.pinned; i++)
    if (hipHostMalloc((void **)&st.buf[i], bufBytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); st.pinned = false; }
  if (!st.pinned) {   // (no pinned memory to be had: pageable staging, the copies then synchronise by themselves)
    for (int i = 0; i < 2; i++) { if (st.buf[i]) hipHostFree(st.buf[i]); st.buf[i] = nullptr; }
    for (int i = 0; i < nBuf; i++) { st.pageable[i].resize(bufBytes); st.buf[i] = st.pageable[i].data(); }
  }
  for (int i = 0; i < nBuf; i++) HIP_TRY(hipEventCreateWithFlags(&st.done[i], hipEventDisableTiming));
  const char *what = mD ? "the target dimension of _mD weights." : "the target dimension of _sA weights.";
  auto copyBatch = [&](char *host, int64_t q0, int64_t nq) -> hipError_t {
    if (mD) {   // one row per question, (K + 1) ldT elements apart
      return write ? hipMemcpy2DAsync(host, rowB, CubeAt(q0, _K), ldB * (size_t)(_K + 1), rowB, (size_t)nq, hipMemcpyDeviceToHost, _stream)
                   : hipMemcpy2DAsync(CubeAt(q0, _K), ldB * (size_t)(_K + 1), host, rowB, rowB, (size_t)nq, hipMemcpyHostToDevice, _stream);
    }
    for (int64_t q = 0; q < nq; q++) {
      char *h = host + (size_t)q * (size_t)rowsPerQ * rowB;
      char *d = CubeAt(q0 + q, 0);
      const hipError_t he = write ? hipMemcpy2DAsync(h, rowB, d, ldB, rowB, (size_t)rowsPerQ, hipMemcpyDeviceToHost, _stream)
                                  : hipMemcpy2DAsync(d, ldB, h, rowB, rowB, (size_t)rowsPerQ, hipMemcpyHostToDevice, _stream);
      if (he != hipSuccess) return he;
    }
    return hipSuccess;
  };
  int64_t prevQ0 = -1, prevNq = 0;   // a save: the batch whose rows are on their way into the other buffer
  int turn = 0;
  for (int64_t q0 = 0; q0 < _Q; q0 += batch, turn ^= (nBuf - 1)) {
    const int64_t nq = std::min(batch, _Q - q0);
    const size_t nRows = (size_t)(nq * rowsPerQ);
    char *host = st.buf[turn];
    HIP_TRY(hipEventSynchronize(st.done[turn]));   // (whatever used this buffer two batches ago has finished; a fresh event is complete)
    if (!write) {
      if (std::fread(host, rowB, nRows, f) != nRows) { (void)hipStreamSynchronize(_stream); return FileErr(filePath, (std::string("Can't read ") + what).c_str()); }
      HIP_TRY(copyBatch(host, q0, nq));
      HIP_TRY(hipEventRecord(st.done[turn], _stream));
    } else {
      HIP_TRY(copyBatch(host, q0, nq));
      HIP_TRY(hipEventRecord(st.done[turn], _stream));
      if (prevQ0 >= 0) {   // while this batch comes back, the previous one goes to the file
        const int other = turn ^ (nBuf - 1);
        HIP_TRY(hipEventSynchronize(st.done[other]));
        if (std::fwrite(st.buf[other], rowB, (size_t)(prevNq * rowsPerQ), f) != (size_t)(prevNq * rowsPerQ)) { (void)hipStreamSynchronize(_stream); return FileErr(filePath, (std::string("Can't write ") + what).c_str()); }
      }
      if (nBuf == 1) {   // (a single batch: written right away)
        HIP_TRY(hipEventSynchronize(st.done[turn]));
        if (std::fwrite(host, rowB, nRows, f) != nRows) return FileErr(filePath, (std::string("Can't write ") + what).c_str());
      } else { prevQ0 = q0; prevNq = nq; }
    }
  }
  HIP_TRY(hipStreamSynchronize(_stream));
  if (write && nBuf == 2 && prevQ0 >= 0) {   // the last batch of a save
    const int last = turn ^ 1;
    if (std::fwrite(st.buf[last], rowB, (size_t)(prevNq * rowsPerQ), f) != (size_t)(prevNq * rowsPerQ)) return FileErr(filePath, (std::string("Can't write ") + what).c_str());
  }
  return Error();
}

// vB: fp64 on the device in both precisions; the file holds the engine's number type
Error HipEngine::IoVB(FILE *f, const char *filePath, bool write) {
  hipSetDevice(_device);
  std::vector<double> vb((size_t)_T);
  std::vector<float> vf(_elem == 4 ? (size_t)_T : 0);
  if (write) {
    HIP_TRY(hipMemcpyAsync(vb.data(), _dVB, (size_t)_T * sizeof(double), hipMemcpyDeviceToHost, _stream));
    HIP_TRY(hipStreamSynchronize(_stream));
    bool ok;
    if (_elem == 8) ok = std::fwrite(vb.data(), sizeof(double), (size_t)_T, f) == (size_t)_T;
    else {
      std::copy(vb.begin(), vb.end(), vf.begin());
      ok = std::fwrite(vf.data(), sizeof(float), (size_t)_T, f) == (size_t)_T;
    }
    return ok ? Error() : FileErr(filePath, "Can't write the _vB weights.");
  }
  if (_elem == 8) {
    if (std::fread(vb.data(), sizeof(double), (size_t)_T, f) != (size_t)_T) return FileErr(filePath, "Can't read the _vB weights.");
  } else {
    if (std::fread(vf.data(), sizeof(float), (size_t)_T, f) != (size_t)_T) return FileErr(filePath, "Can't read the _vB weights.");
    std::copy(vf.begin(), vf.end(), vb.begin());
  }
  return SetVBFromHost(vb.data());
}

Error HipEngine::SetVBFromHost(const double *vb) {
  hipSetDevice(_device);
  HIP_TRY(hipMemcpyAsync(_dVB, vb, (size_t)_T * sizeof(double), hipMemcpyHostToDevice, _stream));
  HIP_TRY(hipStreamSynchronize(_stream));
  return Error();
}

// ------------------------------------------------------------------------------------------------------------------
// .kb persistence (layout of reference PqaCore/BaseEngine.cpp:323-385 + PqaCore/CpuEngine.cpp:664-688):
//   PrecisionDefinition (8 B) | EngineDimensions {nAnswers, nQuestions, nTargets} | u64 nQuestionsAsked |
//   sA rows [q][a] of nTargets doubles | mD rows [q] | vB | question gaps (i64 n, n ids) | target gaps |
//   PermanentIdManager x3 (questions, targets, quizzes saved empty)
// ------------------------------------------------------------------------------------------------------------------
Error HipEngine::SaveKB(const char *filePath, bool doubleBuffer) {
  (void)doubleBuffer;  // the device copy already is the "second buffer": the file is written from a host snapshot
  if (!filePath) return Error::Make(ErrCode::NullArgument, "Nullptr is passed in place of KB file name.");
  std::lock_guard<EngineMutex> lk(_mu);
  StopServer();
  if (_qTotal != _Q) return Error::MakeP(ErrCode::NotImplemented, "Feature=SaveKB of a sharded engine", "Save the shards' owner instead.");
  FileCloser fc{std::fopen(filePath, "wb")};
  if (!fc.f)
    return Error::MakeP(ErrCode::CantOpenFile, std::string("filePath=[") + filePath + "]", "Can't open the file to write KB to.");
  hipSetDevice(_device);
  const uint64_t prec = PackPrecision(_precType, _precMantissa, _precExponent);   // Double | Float: the element type of the arrays below
  const int64_t dims[3] = {_K, _Q, _T};
  const uint64_t nAsked = _nQuestionsAsked.load(std::memory_order_acquire);
  if (std::fwrite(&prec, 8, 1, fc.f) != 1) return FileErr(filePath, "Can't write precision definition header.");
  if (std::fwrite(dims, sizeof(dims), 1, fc.f) != 1) return FileErr(filePath, "Can't write engine dimensions header.");
  if (std::fwrite(&nAsked, 8, 1, fc.f) != 1) return FileErr(filePath, "Can't write the number of questions asked.");
  Error e = IoRows(fc.f, filePath, false, true);      // sA rows [q][a]
  if (e.ok()) e = IoRows(fc.f, filePath, true, true);  // mD rows [q]
  if (e.ok()) e = IoVB(fc.f, filePath, true);
  if (!e.ok()) return e;
  auto writeGaps = [&](const std::vector<int64_t> &gaps) {
    const int64_t n = (int64_t)gaps.size();
    return std::fwrite(&n, 8, 1, fc.f) == 1 && std::fwrite(gaps.data(), 8, (size_t)n, fc.f) == (size_t)n;
  };
  if (!writeGaps(_questionGapList)) return FileErr(filePath, "Can't write the question gaps.");
  if (!writeGaps(_targetGapList)) return FileErr(filePath, "Can't write the target gaps.");
  if (!_questionIds.Write(fc.f)) return FileErr(filePath, "Can't write the question permanent-compact ID mappings.");
  if (!_targetIds.Write(fc.f)) return FileErr(filePath, "Can't write the target permanent-compact ID mappings.");
  if (!_quizIds.Write(fc.f, true)) return FileErr(filePath, "Can't write the quiz permanent-compact ID mappings.");
  if (std::fflush(fc.f) != 0) return FileErr(filePath, "Failed in hard flushing the KB.");
  FILE *f = fc.f;
  fc.f = nullptr;
  if (std::fclose(f) != 0) return FileErr(filePath, "Failed in closing the file.");
  return Error();
}

HipEngine *HipEngine::Load(Error &err, const char *filePath) {  // PqaEngineBaseFactory.cpp:44-83, CpuEngine.cpp:41-92
  if (!filePath) { err = Error::Make(ErrCode::NullArgument, "Nullptr is passed in place of KB file name."); return nullptr; }
  FileCloser fc{std::fopen(filePath, "rb")};
  if (!fc.f) {
    err = Error::MakeP(ErrCode::CantOpenFile, std::string("filePath=[") + filePath + "]", "Can't open the KB file to read.");
    return nullptr;
  }
  uint64_t prec = 0, nAsked = 0;
  int64_t dims[3];
  if (std::fread(&prec, 8, 1, fc.f) != 1) { err = FileErr(filePath, "Can't read precision definition header."); return nullptr; }
  if (std::fread(dims, sizeof(dims), 1, fc.f) != 1) { err = FileErr(filePath, "Can't read engine dimensions header."); return nullptr; }
  if (std::fread(&nAsked, 8, 1, fc.f) != 1) { err = FileErr(filePath, "Can't read the number of questions asked."); return nullptr; }
  CiEngineDefinition def;
  std::memset(&def, 0, sizeof(def));
  def._nAnswers = dims[0]; def._nQuestions = dims[1]; def._nTargets = dims[2];
  def._precType = (uint8_t)(prec & 0xF);
  def._precMantissa = (uint32_t)((prec >> 4) & 0xFFFFFFF);
  def._precExponent = (uint16_t)((prec >> 32) & 0xFFFF);
  def._initAmount = 1.0;  // not stored in the file; every count is overwritten below
  std::unique_ptr<HipEngine> eng(HipEngine::Create(err, def, nullptr));
  if (!eng) return nullptr;
  HipEngine &e = *eng;
  auto fail = [&](Error x) { err = std::move(x); return (HipEngine *)nullptr; };
  hipSetDevice(e._device);
  Error ioErr = e.IoRows(fc.f, filePath, false, false);
  if (ioErr.ok()) ioErr = e.IoRows(fc.f, filePath, true, false);
  if (ioErr.ok()) ioErr = e.IoVB(fc.f, filePath, false);
  if (!ioErr.ok()) return fail(std::move(ioErr));
  e._nQuestionsAsked.store(nAsked);
  auto readGaps = [&](std::vector<int64_t> &gaps, int64_t limit) {
    int64_t n;
    if (std::fread(&n, 8, 1, fc.f) != 1 || n < 0 || n > limit) return false;
    gaps.resize((size_t)n);
    if (std::fread(gaps.data(), 8, (size_t)n, fc.f) != (size_t)n) return false;
    for (int64_t g : gaps) if (g < 0 || g >= limit) return false;
    return true;
  };
  if (!readGaps(e._questionGapList, e._Q)) return fail(FileErr(filePath, "Can't read the question gaps."));
  if (!readGaps(e._targetGapList, e._T)) return fail(FileErr(filePath, "Can't read the target gaps."));
  for (int64_t g : e._questionGapList) BitSet(e._hQGap, g, true);
  for (int64_t g : e._targetGapList) BitSet(e._hTGap, g, true);
  e._nTargetGaps = (int64_t)e._targetGapList.size();
  if (!e._questionIds.Read(fc.f)) return fail(FileErr(filePath, "Can't read the question permanent-compact ID mapping."));
  if (!e._targetIds.Read(fc.f)) return fail(FileErr(filePath, "Can't read the target permanent-compact ID mapping."));
  if (!e._quizIds.Read(fc.f)) return fail(FileErr(filePath, "Can't read the quizzes permanent-compact ID mapping."));
  Error ue = e.UploadGaps();
  if (!ue.ok()) return fail(std::move(ue));
  err = Error();
  return eng.release();
}

// ------------------------------------------------------------------------------------------------------------------
// maintenance-mode operations
// ------------------------------------------------------------------------------------------------------------------
namespace {
// device allocation that is freed unless released: every buffer of a resize exists before the first one is committed
template <typename T>
struct DevBuf {
  T *p = nullptr;
  ~DevBuf() { if (p) hipFree(p); }
  hipError_t Alloc(size_t bytes) { return hipMalloc(reinterpret_cast<void **>(&p), bytes); }
  T *Release() { T *r = p; p = nullptr; return r; }
};
}  // namespace

// Grow the knowledge base to newQ questions and newT targets.  All-or-nothing: every new buffer is allocated and filled
// before the engine's members change, so a failure (out of memory while the old and the new cube coexist) leaves the engine
// as it was.
Error HipEngine::ReallocKB(int64_t newQ, int64_t newT) {
  const int64_t newLdT = std::max(_ldT, RoundLdT(newT, _elem));
  const int64_t newCap = std::max(_capQ, newQ);
  const bool regrow = newLdT != _ldT || newCap != _capQ;
  DevBuf<char> cube;
  DevBuf<double> vB, priority, runLength, poleScratch;
  DevBuf<int64_t> exps;
  DevBuf<uint32_t> tgapDev, qgapDev;
  // bitmaps: keep the old bits, new positions are not gaps, everything past the size is
  std::vector<uint32_t> tg(BitWords(newLdT), 0), qg(BitWords(newQ), 0);
  for (int64_t t = 0; t < _T; t++) if (BitTest(_hTGap, t)) BitSet(tg, t, true);
  for (int64_t t = newT; t < (int64_t)tg.size() * 32; t++) BitSet(tg, t, true);
  for (int64_t q = 0; q < _Q; q++) if (BitTest(_hQGap, q)) BitSet(qg, q, true);
  for (int64_t q = newQ; q < (int64_t)qg.size() * 32; q++) BitSet(qg, q, true);
  HIP_TRY(tgapDev.Alloc(tg.size() * sizeof(uint32_t)));
  HIP_TRY(qgapDev.Alloc(qg.size() * sizeof(uint32_t)));
  if (regrow) {
    const size_t el = (size_t)_elem;
    HIP_TRY(cube.Alloc((size_t)newCap * (size_t)(_K + 1) * (size_t)newLdT * el));
    HIP_TRY(vB.Alloc((size_t)newLdT * sizeof(double)));
    if (newLdT != _ldT) HIP_TRY(exps.Alloc((size_t)newLdT * sizeof(int64_t)));
    if (newCap != _capQ) {
      HIP_TRY(priority.Alloc((size_t)newCap * sizeof(double)));
      HIP_TRY(runLength.Alloc((size_t)newCap * sizeof(double)));
      HIP_TRY(poleScratch.Alloc(PoleScratchBytes(newCap)));
      HIP_TRY(hipMemsetAsync(poleScratch.p, 0, PoleScratchBytes(newCap), _stream));
    }
    // old rows keep their content; new padding columns get A = 0, D = 1 from the fill of new questions / a plain fill
    HIP_TRY(LaunchFillFresh(cube.p, _elem, vB.p, _K, newCap, 0, newLdT, 0.0, _stream));  // T = 0: every column is "padding"
    HIP_TRY(hipMemcpy2DAsync(cube.p, (size_t)newLdT * el, _dCube, (size_t)_ldT * el, (size_t)_T * el,
                             (size_t)_Q * (size_t)(_K + 1), hipMemcpyDeviceToDevice, _stream));
    HIP_TRY(hipMemcpyAsync(vB.p, _dVB, (size_t)_T * sizeof(double), hipMemcpyDeviceToDevice, _stream));
    HIP_TRY(hipStreamSynchronize(_stream));
  }
  // ---- commit (nothing below can fail except the bitmap upload, which leaves host and device views consistent in shape)
  if (regrow) {
    hipFree(_dCube); hipFree(_dVB);
    _dCube = cube.Release(); _dVB = vB.Release();
    if (exps.p) { hipFree(_dExps); _dExps = exps.Release(); }
    if (priority.p) {
      hipFree(_dPriority); hipFree(_dRunLength); hipFree(_dPoleScratch);
      _dPriority = priority.Release(); _dRunLength = runLength.Release(); _dPoleScratch = poleScratch.Release();
    }
    _ldT = newLdT;
    _capQ = newCap;
  }
  _hTGap.swap(tg);
  _hQGap.swap(qg);
  hipFree(_dTGap); hipFree(_dQGap);
  _dTGap = tgapDev.Release(); _dQGap = qgapDev.Release();
  _Q = newQ;
  _qTotal = newQ;
  _T = newT;
  return UploadGaps();
}

Error HipEngine::AddQsTs(int64_t nQuestions, CiAddQorTParam *pAqps, int64_t nTargets, CiAddQorTParam *pAtps) {
  std::lock_guard<EngineMutex> lk(_mu);
  StopServer();
  if (_mode != Mode::Maintenance) return WrongModeErr("add questions/targets");
  if (nQuestions < 0 || nTargets < 0)
    return Error::MakeP(ErrCode::NegativeCount, "count=" + std::to_string(std::min(nQuestions, nTargets)), "Counts must be non-negative.");
  if ((nQuestions > 0 && !pAqps) || (nTargets > 0 && !pAtps)) return Error::Make(ErrCode::NullArgument, "Nullptr parameters array.");
  if (_qFirst != 0 || _qTotal != _Q) return Error::MakeP(ErrCode::NotImplemented, "Feature=AddQsTs on a shard", "Not on a sharded engine.");
  hipSetDevice(_device);
  // CpuEngine::AddQsTsSpec, reference PqaCore/CpuEngine.cpp:468-575.  The ids are worked out first and committed -- gap
  // lists, permanent ids, bitmaps, the caller's _index fields -- only after the resize and the fills have succeeded.
  const int64_t nQReuse = std::min<int64_t>(nQuestions, (int64_t)_questionGapList.size()), nQNew = nQuestions - nQReuse;
  const int64_t nTReuse = std::min<int64_t>(nTargets, (int64_t)_targetGapList.size()), nTNew = nTargets - nTReuse;
  const int64_t nQOld = _Q, nTOld = _T;
  std::vector<int64_t> qIds, tIds;
  std::vector<double> qInit, tInit;
  for (int64_t i = 0; i < nQReuse; i++) qIds.push_back(_questionGapList[_questionGapList.size() - 1 - (size_t)i]);   // :476-482 gaps are reused LIFO
  for (int64_t i = 0; i < nTReuse; i++) tIds.push_back(_targetGapList[_targetGapList.size() - 1 - (size_t)i]);       // :488-493
  for (int64_t i = 0; i < nQNew; i++) qIds.push_back(nQOld + i);   // :500
  // NOTE reference :516,:523,:529 index the target parameters with nQReuse + j; the evident intent nTReuse + j is used
  for (int64_t j = 0; j < nTNew; j++) tIds.push_back(nTOld + j);    // :531
  for (int64_t i = 0; i < nQuestions; i++) qInit.push_back(pAqps[i]._initAmount);
  for (int64_t j = 0; j < nTargets; j++) tInit.push_back(pAtps[j]._initAmount);
  // whole questions first, then target columns over the questions not (re)initialised just now
  std::vector<uint32_t> skip(BitWords(nQOld + nQNew), 0);
  for (int64_t i = 0; i < nQReuse; i++) BitSet(skip, qIds[(size_t)i], true);    // :558-560 only reused questions are skipped
  DevBuf<int64_t> dQ, dT;
  DevBuf<double> dQi, dTi;
  DevBuf<uint32_t> dSkip;
  HIP_TRY(Upload(&dQ.p, qIds, _stream));
  HIP_TRY(Upload(&dQi.p, qInit, _stream));
  HIP_TRY(Upload(&dT.p, tIds, _stream));
  HIP_TRY(Upload(&dTi.p, tInit, _stream));
  HIP_TRY(Upload(&dSkip.p, skip, _stream));
  Error e = ReallocKB(nQOld + nQNew, nTOld + nTNew);   // all-or-nothing; the reused ids are still flagged as gaps
  if (!e.ok()) return e;
  // new target columns apply to every old question (:512-527); reused target columns skip reused questions (:553-567).
  // New questions are filled over ALL columns with their own amount (:497-510), so they are filled last.
  hipError_t he = hipSuccess;
  if (nTReuse > 0) he = LaunchFillTargets(_dCube, _elem, _dVB, _K, _ldT, nQOld, dSkip.p, dT.p, dTi.p, nTReuse, _stream);
  if (he == hipSuccess && nTNew > 0) he = LaunchFillTargets(_dCube, _elem, _dVB, _K, _ldT, nQOld, nullptr, dT.p + nTReuse, dTi.p + nTReuse, nTNew, _stream);
  if (he == hipSuccess && nQuestions > 0) he = LaunchFillQuestions(_dCube, _elem, _K, _T, _ldT, dQ.p, dQi.p, nQuestions, _stream);
  if (he == hipSuccess) he = hipStreamSynchronize(_stream);
  if (he != hipSuccess) {   // (a failed launch: the device is gone) keep the id maps the size of the grown KB
    _questionIds.Extend(_Q);
    _targetIds.Extend(_T);
    return HipErr(he, "AddQsTs");
  }
  // ---- commit
  for (int64_t i = 0; i < nQReuse; i++) {
    const int64_t curQ = qIds[(size_t)i];
    _questionGapList.pop_back();
    BitSet(_hQGap, curQ, false);
    _questionIds.Reissue(curQ);
  }
  for (int64_t i = 0; i < nTReuse; i++) {
    const int64_t curT = tIds[(size_t)i];
    _targetGapList.pop_back();
    BitSet(_hTGap, curT, false);
    _nTargetGaps--;
    _targetIds.Reissue(curT);
  }
  _questionIds.Extend(_Q);                           // :541-542
  _targetIds.Extend(_T);
  for (int64_t i = 0; i < nQuestions; i++) pAqps[i]._index = qIds[(size_t)i];
  for (int64_t j = 0; j < nTargets; j++) pAtps[j]._index = tIds[(size_t)j];
  return UploadGaps();
}

Error HipEngine::AdoptRows(const std::vector<const void *> &srcBlocks, int64_t ldTsrc, const std::vector<int64_t> &colMap, const double *srcVB) {
  std::lock_guard<EngineMutex> lk(_mu);
  StopServer();
  if ((int64_t)srcBlocks.size() != _Q || (int64_t)colMap.size() != _T)
    return Error::Make(ErrCode::Internal, "AdoptRows: the maps do not have this shard's dimensions.");
  hipSetDevice(_device);
  const void **dSrc = nullptr;
  int64_t *dMap = nullptr;
  hipError_t he = Upload(&dSrc, srcBlocks, _stream);
  if (he == hipSuccess) he = Upload(&dMap, colMap, _stream);
  if (he == hipSuccess) he = LaunchAdoptRows(_dCube, _elem, _dVB, _K, _Q, _T, _ldT, dSrc, ldTsrc, srcVB, dMap, _stream);
  if (he == hipSuccess) he = hipStreamSynchronize(_stream);
  hipFree(dSrc);
  hipFree(dMap);
  if (he != hipSuccess) return HipErr(he, "AdoptRows");
  return Error();
}

Error HipEngine::ApplyFills(const std::vector<int64_t> &tIds, const std::vector<double> &tInit, const std::vector<int64_t> &qLocalIds,
                            const std::vector<double> &qInit) {
  std::lock_guard<EngineMutex> lk(_mu);
  StopServer();
  hipSetDevice(_device);
  DevBuf<int64_t> dQ, dT;
  DevBuf<double> dQi, dTi;
  HIP_TRY(Upload(&dQ.p, qLocalIds, _stream));
  HIP_TRY(Upload(&dQi.p, qInit, _stream));
  HIP_TRY(Upload(&dT.p, tIds, _stream));
  HIP_TRY(Upload(&dTi.p, tInit, _stream));
  // target columns over every question, then whole questions over every column (the reference skips the re-initialised
  // questions in the first step only because the second overwrites them anyway, CpuEngine.cpp:558-560)
  hipError_t he = LaunchFillTargets(_dCube, _elem, _dVB, _K, _ldT, _Q, nullptr, dT.p, dTi.p, (int64_t)tIds.size(), _stream);
  if (he == hipSuccess) he = LaunchFillQuestions(_dCube, _elem, _K, _T, _ldT, dQ.p, dQi.p, (int64_t)qLocalIds.size(), _stream);
  if (he == hipSuccess) he = hipStreamSynchronize(_stream);
  if (he != hipSuccess) return HipErr(he, "ApplyFills");
  return Error();
}

// RemoveQuestions / RemoveTargets validate every id -- range, gaps, repeats within the call -- before the first one is removed:
// a failing call changes nothing, on the host or on the device.  (The reference removes id by id and stops at the first
// bad one, BaseEngine.cpp:722-765, leaving the earlier ones removed.)
Error HipEngine::RemoveQuestions(int64_t n, const int64_t *pQIds) {  // BaseEngine.cpp:722-743
  std::lock_guard<EngineMutex> lk(_mu);
  StopServer();
  if (_mode != Mode::Maintenance) return WrongModeErr("remove questions");
  if (n < 0) return Error::MakeP(ErrCode::NegativeCount, "count=" + std::to_string(n), "Counts must be non-negative.");
  if (n > 0 && !pQIds) return Error::Make(ErrCode::NullArgument, "Nullptr ids array.");
  if (_qFirst != 0 || _qTotal != _Q) return Error::MakeP(ErrCode::NotImplemented, "Feature=RemoveQuestions on a shard", "Not on a sharded engine.");
  std::vector<uint32_t> seen(BitWords(_Q), 0);
  for (int64_t i = 0; i < n; i++) {
    const int64_t iq = pQIds[i];
    if (iq < 0 || iq >= _Q || BitTest(_hQGap, iq) || BitTest(seen, iq))
      return Error::MakeP(ErrCode::AbsentId, "id=" + std::to_string(iq), "Question index is not in KB.");
    BitSet(seen, iq, true);
  }
  for (int64_t i = 0; i < n; i++) {
    const int64_t iq = pQIds[i];
    BitSet(_hQGap, iq, true);
    _questionGapList.push_back(iq);
    _questionIds.Vacate(iq);
  }
  hipSetDevice(_device);
  return UploadGaps();
}

Error HipEngine::RemoveTargets(int64_t n, const int64_t *pTIds) {  // BaseEngine.cpp:745-765
  std::lock_guard<EngineMutex> lk(_mu);
  StopServer();
  if (_mode != Mode::Maintenance) return WrongModeErr("remove targets");
  if (n < 0) return Error::MakeP(ErrCode::NegativeCount, "count=" + std::to_string(n), "Counts must be non-negative.");
  if (n > 0 && !pTIds) return Error::Make(ErrCode::NullArgument, "Nullptr ids array.");
  std::vector<uint32_t> seen(BitWords(_T), 0);
  for (int64_t i = 0; i < n; i++) {
    const int64_t it = pTIds[i];
    if (it < 0 || it >= _T || BitTest(_hTGap, it) || BitTest(seen, it))
      return Error::MakeP(ErrCode::AbsentId, "id=" + std::to_string(it), "Target index is not in KB (but rather at a gap).");
    BitSet(seen, it, true);
  }
  for (int64_t i = 0; i < n; i++) {
    const int64_t it = pTIds[i];
    BitSet(_hTGap, it, true);
    _targetGapList.push_back(it);
    _nTargetGaps++;
    _targetIds.Vacate(it);
  }
  hipSetDevice(_device);
  return UploadGaps();
}

Error HipEngine::Compact(int64_t *pnQuestions, const int64_t **ppOldQuestions, int64_t *pnTargets,
                         const int64_t **ppOldTargets) {  // CpuEngine::CompactSpec, CpuEngine.cpp:577-658
  std::lock_guard<EngineMutex> lk(_mu);
  StopServer();
  if (_mode != Mode::Maintenance) return WrongModeErr("compact the KB");
  if (!pnQuestions || !ppOldQuestions || !pnTargets || !ppOldTargets) return Error::Make(ErrCode::NullArgument, "Nullptr output.");
  hipSetDevice(_device);
  const int64_t nQ = _Q - (int64_t)_questionGapList.size(), nT = _T - (int64_t)_targetGapList.size();
  int64_t *oldQ = (int64_t *)std::malloc(sizeof(int64_t) * (size_t)std::max<int64_t>(nQ, 1));
  int64_t *oldT = (int64_t *)std::malloc(sizeof(int64_t) * (size_t)std::max<int64_t>(nT, 1));
  // questions: a gap in the kept prefix takes the LAST surviving question (:586-601)
  {
    int64_t iFirst = 0, iLast = _Q - 1;
    for (; iFirst <= iLast; iFirst++) {
      if (!BitTest(_hQGap, iFirst)) { oldQ[iFirst] = iFirst; continue; }
      while (BitTest(_hQGap, iLast) && iLast > iFirst) iLast--;
      if (iFirst == iLast) break;
      oldQ[iFirst] = iLast;
      const hipError_t ce = hipMemcpyAsync(CubeAt(iFirst), CubeAt(iLast), (size_t)(_K + 1) * (size_t)_ldT * (size_t)_elem,
                                           hipMemcpyDeviceToDevice, _stream);
      if (ce != hipSuccess) { std::free(oldQ); std::free(oldT); return HipErr(ce, "Compact (question move)"); }
      iLast--;
    }
  }
  // targets: gaps of the kept prefix (ascending) take the survivors of the dropped tail (ascending) -- the pairing the
  // reference produces when no gap lies in the tail (:604-618); with tail gaps the reference's move table is
  // under-filled, here the pairing simply continues.
  std::vector<int64_t> moves;
  {
    std::vector<int64_t> dst, src;
    for (int64_t t = 0; t < nT; t++) if (BitTest(_hTGap, t)) dst.push_back(t); else oldT[t] = t;
    for (int64_t t = nT; t < _T; t++) if (!BitTest(_hTGap, t)) src.push_back(t);
    for (size_t i = 0; i < dst.size(); i++) { oldT[dst[i]] = src[i]; moves.push_back(src[i]); moves.push_back(dst[i]); }
  }
  int64_t *dMoves = nullptr;
  hipError_t he = Upload(&dMoves, moves, _stream);
  if (he == hipSuccess) he = LaunchMoveTargets(_dCube, _elem, _dVB, _K, _ldT, nQ, dMoves, (int64_t)moves.size() / 2, _stream);
  if (he == hipSuccess) he = hipStreamSynchronize(_stream);
  hipFree(dMoves);
  if (he != hipSuccess) { std::free(oldQ); std::free(oldT); return HipErr(he, "Compact"); }
  _questionIds.Repack(nQ, oldQ);
  _targetIds.Repack(nT, oldT);
  _questionGapList.clear();
  _targetGapList.clear();
  _nTargetGaps = 0;
  // shrink the logical dimensions; the allocation (capacity, ldT) stays, padding is re-flagged as gap
  _hTGap.assign(BitWords(_ldT), 0);
  _hQGap.assign(BitWords(nQ), 0);
  for (int64_t t = nT; t < (int64_t)_hTGap.size() * 32; t++) BitSet(_hTGap, t, true);
  for (int64_t q = nQ; q < (int64_t)_hQGap.size() * 32; q++) BitSet(_hQGap, q, true);
  hipFree(_dQGap); _dQGap = nullptr;
  HIP_TRY(hipMalloc(&_dQGap, _hQGap.size() * sizeof(uint32_t)));
  _Q = nQ; _qTotal = nQ; _T = nT;
  *pnQuestions = nQ; *pnTargets = nT;
  *ppOldQuestions = oldQ; *ppOldTargets = oldT;
  return UploadGaps();
}

}  // namespace pqa
