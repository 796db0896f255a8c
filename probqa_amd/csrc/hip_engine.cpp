// hip_engine.cpp -- see hip_engine.h.  Reference files cited are under /root/reference/ProbQA.
#include "hip_engine_internal.h"

namespace pqa {

// ------------------------------------------------------------------------------------------------------------------
// errors (reference PqaCore/PqaErrors.cpp:13-62, :128-143)
// ------------------------------------------------------------------------------------------------------------------
const char *ErrCodeName(ErrCode c) {
  switch (c) {
    case ErrCode::None: return "Success";
    case ErrCode::NotImplemented: return "Not implemented";
    case ErrCode::SRException: return "SRException";
    case ErrCode::StdException: return "std::exception";
    case ErrCode::InsufficientEngineDimensions: return "Insufficient engine dimensions";
    case ErrCode::MaintenanceModeChangeInProgress: return "Maintenance mode change is in progress";
    case ErrCode::MaintenanceModeAlreadyThis: return "Maintenance mode is already this";
    case ErrCode::ObjectShutDown: return "Object is shut(ting) down";
    case ErrCode::IndexOutOfRange: return "Index is out of range";
    case ErrCode::Aggregate: return "Aggregate error";
    case ErrCode::NegativeCount: return "The count is negative";
    case ErrCode::NonPositiveAmount: return "The amount is not positive";
    case ErrCode::AbsentId: return "The ID is absent from KB";
    case ErrCode::WrongMode: return "An attempt to execute an operation in a wrong mode";
    case ErrCode::UnhandledCase: return "Unhandled case";
    case ErrCode::I64Underflow: return "Underflow of a 64-bit integer";
    case ErrCode::QuestionsExhausted: return "Engine has run out of questions";
    case ErrCode::NoQuizActiveQuestion: return "No active question in the quiz";
    case ErrCode::CantOpenFile: return "Cannot open file";
    case ErrCode::FileOp: return "File operation failed";
    case ErrCode::QuizzesActive: return "There are still active quizzes";
    case ErrCode::NullArgument: return "Expected non-null argument";
    default: return nullptr;
  }
}

std::string Error::ToString(bool withParams) const {
  std::string s = "[";
  const char *name = ErrCodeName(code);
  if (name) s += name; else s += "Unhandled" + std::to_string((int64_t)code);
  s += "] message=[" + message;
  if (!withParams) return s + "]";
  s += "] [";
  s += hasParams ? params : std::string("nullptr");
  return s + "]";
}

// ------------------------------------------------------------------------------------------------------------------
// default logger (reference SRPlatform/SRDefaultLogger.cpp:47-83; file naming of SRPlatform/SRLoggerFactory + FileLogger:
// <baseName>_<UTC date-time>_<pid>.log is what this build uses)
// ------------------------------------------------------------------------------------------------------------------
namespace {
std::mutex gLogMu;
FILE *gLogFile = nullptr;
}  // namespace

std::string DefaultLogger::Init(const char *baseName) {
  std::lock_guard<std::mutex> lk(gLogMu);
  if (gLogFile != nullptr)
    return "Default logger seems already initialized by the moment of calling DefaultLoggerImpl::Init().";
  if (baseName == nullptr || *baseName == 0) return "Nullptr or empty string is passed in place of the log file base name.";
  char stamp[64];
  const time_t now = time(nullptr);
  struct tm tmv;
  gmtime_r(&now, &tmv);
  std::strftime(stamp, sizeof(stamp), "%Y-%m-%d_%H-%M-%S", &tmv);
  const std::string path = std::string(baseName) + "_" + stamp + "_" + std::to_string((long long)getpid()) + ".log";
  FILE *f = std::fopen(path.c_str(), "a");
  if (f == nullptr) return "Can't open the log file " + path + ".";
  gLogFile = f;
  return std::string();
}

void DefaultLogger::Log(Severity sev, const std::string &message) {
  static const char *names[] = {"None", "Info", "Warning", "Error", "Critical"};
  std::lock_guard<std::mutex> lk(gLogMu);
  FILE *f = gLogFile ? gLogFile : stderr;
  char stamp[64];
  const time_t now = time(nullptr);
  struct tm tmv;
  gmtime_r(&now, &tmv);
  std::strftime(stamp, sizeof(stamp), "%Y-%m-%d %H:%M:%S", &tmv);
  std::fprintf(f, "%s [%s] %s\n", stamp, names[(int)sev <= 4 ? (int)sev : 0], message.c_str());
  std::fflush(f);
}

namespace {
std::once_flag gTableOnce[64];
}  // namespace

// ------------------------------------------------------------------------------------------------------------------
// construction
// ------------------------------------------------------------------------------------------------------------------
HipEngine *HipEngine::Create(Error &err, const CiEngineDefinition &def, const CiHipShard *shard) {
  std::unique_ptr<HipEngine> eng(new HipEngine());
  err = eng->Init(def, shard);
  if (!err.ok()) return nullptr;
  return eng.release();
}

Error HipEngine::Init(const CiEngineDefinition &def, const CiHipShard *shard) {
  _mu.owner = this;
  // reference PqaCore/PqaEngineBaseFactory.cpp:29-42: minimum dimensions
  const int64_t minA = 2, minQ = 1, minT = 2;
  if (def._nAnswers < minA || def._nQuestions < minQ || def._nTargets < minT) {
    std::ostringstream p;
    p << "[nAnswers=" << def._nAnswers << " of " << minA << "] [nQuestions=" << def._nQuestions << " of " << minQ
      << "] [nTargets=" << def._nTargets << " of " << minT << "]";
    return Error::MakeP(ErrCode::InsufficientEngineDimensions, p.str(), "Engine dimensions are too small.");
  }
  // TPqaPrecisionType (reference PqaCore/Interface/PqaCommon.h:17-24): Double is what the reference's CPU engine instantiates
  // (PqaEngineBaseFactory.cpp:19-27), Float what its GPU engine does (:124-142); both exist here.
  if (def._precType != 3 /*Double*/ && def._precType != 1 /*Float*/) {
    return Error::MakeP(ErrCode::NotImplemented, "Feature=HipEngine precision other than Double and Float",
                        "TPqaPrecisionType::Double and ::Float are instantiated (the reference: Double on the CPU, "
                        "PqaCore/PqaEngineBaseFactory.cpp:19-27, Float on the GPU, :124-142).");
  }
  _precType = def._precType;
  _elem = def._precType == 1 ? 4 : 8;
  int nDev = 0;
  if (hipGetDeviceCount(&nDev) != hipSuccess || nDev <= 0) {
    return Error::Make(ErrCode::NotInitialized,
                       "No HIP device is available: libPqaCore.so (MI355X build) has no CPU fallback.");
  }
  _K = def._nAnswers; _Q = def._nQuestions; _T = def._nTargets;
  _capQ = _Q;
  _precMantissa = def._precMantissa;
  _precExponent = def._precExponent;
  _initAmount = def._initAmount;
  _ldT = RoundLdT(_T, _elem);
  _qFirst = shard ? shard->_qFirst : 0;
  _qTotal = shard ? shard->_qTotal : _Q;
  if (_qFirst < 0 || _qFirst + _Q > _qTotal)
    return Error::MakeP(ErrCode::IndexOutOfRange, RangeParams(_qFirst + _Q, 0, _qTotal), "Shard exceeds the global question range.");
  if (shard && shard->_device >= 0) {
    if (shard->_device >= nDev)
      return Error::MakeP(ErrCode::IndexOutOfRange, RangeParams(shard->_device, 0, nDev - 1), "No such HIP device.");
    HIP_TRY(hipSetDevice(shard->_device));
  }
  HIP_TRY(hipGetDevice(&_device));
  {
    hipError_t tblErr = hipSuccess;
    std::call_once(gTableOnce[_device & 63], [&] {
      // SRVectMath::Initialize, reference SRPlatform/SRVectMath.cpp:30-44
      // per bucket: log2 of the midpoint, and 1/(2*midpoint) for the division-free quotient of log2hot (pqa_device.h)
      std::vector<double> tbl(2 * 1024);
      for (uint32_t i = 0; i < 1024; i++) {
        const uint64_t iZp = 0x3FF0000000000000ULL | ((uint64_t)i << 42) | (1ULL << 41);
        double zp;
        std::memcpy(&zp, &iZp, 8);
        tbl[2 * i] = std::log2(zp);
        tbl[2 * i + 1] = (double)(2.8853900817779268147198493620038L / (2.0L * (long double)zp));   // (2/ln 2)/(2m): pqa_device.h
      }
      tbl[0] *= 9.9999999999999927e-01;
      // The reference scales entry 0 so that Log2Hot(1) is (just) negative, -1.08e-19: lack = -sum invD^2 / log2(p)
      // relies on the sign, and SRVectMathTest.Log2Hot asserts it.  The device forms the quotient of log2hot by a series
      // instead of a division (pqa_device.h), which moves the value at 1 by a few 1e-18 -- across zero.  Re-seat entry
      // 0 on the device's own arithmetic (the same IEEE operations, replayed here): the largest table value for which
      // the device's Log2Hot(1) is negative.  Every other argument of bucket 0 moves by the same < 3e-18.
      {
        const double m = 1.0 + 0x1p-11, w = (1.0 - m) * tbl[1];
        double pc = std::fma(w, -0x1.55046a143789p-4, 0x1.47fd3ffac83b4p-3);
        pc = std::fma(w, pc, -0x1.62e42fefa39efp-2);
        pc = std::fma(w, pc, 1.0);
        auto at1 = [&](double y0) { return std::fma(w, pc, y0) + 0.0; };
        double y0 = tbl[0];
        while (at1(y0) >= 0) y0 = std::nextafter(y0, 0.0);
        while (at1(std::nextafter(y0, 1.0)) < 0) y0 = std::nextafter(y0, 1.0);
        tbl[0] = y0;
      }
      tblErr = UploadLog2Table(tbl.data());
      if (tblErr == hipSuccess) tblErr = UploadLog2TableBatch(tbl.data());
      if (tblErr == hipSuccess) tblErr = UploadLog2TableCluster(tbl.data());
    });
    HIP_TRY(tblErr);
  }
  HIP_TRY(hipStreamCreateWithFlags(&_ownStream, hipStreamNonBlocking));
  _stream = _ownStream;
  const size_t cubeElems = (size_t)_Q * (size_t)(_K + 1) * (size_t)_ldT;
  HIP_TRY(hipMalloc(&_dCube, cubeElems * (size_t)_elem));
  HIP_TRY(hipMalloc(&_dVB, (size_t)_ldT * sizeof(double)));
  HIP_TRY(hipMalloc(&_dPriority, (size_t)_Q * sizeof(double)));
  HIP_TRY(hipMalloc(&_dRunLength, (size_t)_Q * sizeof(double)));
  HIP_TRY(hipMalloc(&_dPoleScratch, (size_t)_Q * (size_t)(2 * _K + 2) * sizeof(double)));
  HIP_TRY(hipMalloc(&_dExps, (size_t)_ldT * sizeof(int64_t)));
  HIP_TRY(hipMalloc(&_dStatus, 2 * sizeof(int64_t)));
  HIP_TRY(hipMalloc(&_dNOut, sizeof(int64_t)));
  HIP_TRY(hipMalloc(&_dSel, sizeof(SelectResult)));
  HIP_TRY(hipMalloc(&_dSelScratch, kFusedMaxGrid * sizeof(SelectResult)));
  HIP_TRY(hipMemset(_dSelScratch, 0, kFusedMaxGrid * sizeof(SelectResult)));  // tag 0 is never used by a launch
  HIP_TRY(hipMalloc(&_dPriorScratch, (8 * kMaxWorkers + 2) * sizeof(double)));
  HIP_TRY(hipMemset(_dPriorScratch, 0, (8 * kMaxWorkers + 2) * sizeof(double)));   // (the arrival counter starts at 0; every launch leaves it there)
  HIP_TRY(hipHostMalloc(&_hPinned, sizeof(Pinned), hipHostMallocMapped | hipHostMallocCoherent));
  std::memset(_hPinned, 0, sizeof(Pinned));
  _hTGap.assign(BitWords(_ldT), 0);
  _hQGap.assign(BitWords(_Q), 0);
  for (int64_t t = _T; t < (int64_t)_hTGap.size() * 32; t++) BitSet(_hTGap, t, true);  // GapTracker.h:9-10
  for (int64_t q = _Q; q < (int64_t)_hQGap.size() * 32; q++) BitSet(_hQGap, q, true);
  HIP_TRY(hipMalloc(&_dTGap, _hTGap.size() * sizeof(uint32_t)));
  HIP_TRY(hipMalloc(&_dQGap, _hQGap.size() * sizeof(uint32_t)));
  Error e = UploadGaps();
  if (!e.ok()) return e;
  HIP_TRY(LaunchFillFresh(_dCube, _elem, _dVB, _K, _Q, _T, _ldT, _initAmount, _stream));
  HIP_TRY(hipStreamSynchronize(_stream));
  _questionIds.Extend(_Q);  // reference PqaCore/BaseCpuEngine.cpp:24-25
  _targetIds.Extend(_T);
  std::random_device rd;  // the reference seeds from RDRAND (SRPlatform/Interface/SRFastRandom.h:31-40)
  uint64_t seed = ((uint64_t)rd() << 32) ^ rd();
  _rng[0] = SplitMix64(seed);
  _rng[1] = SplitMix64(seed);
  ApplyEnvironment();
  return Error();
}

// The reference's wrappers can only call PqaEngineFactory_CreateCpuEngine / _LoadCpuEngine (ProbQA.py:131-145,
// PqaEngineFactory.cs:24-33) and know nothing of PqaHip_SetOption: what they cannot say in a call they say in the environment.
//   PQA_SELECT=sample|argmax   NextQuestion's selector (default sample: the reference's weighted draw, CpuEngine.cpp:362-400)
//   PQA_SERVER=0|1             argmax selections through the resident sweep kernel
//   PQA_BUG_COMPAT=0|1         ResumeQuiz as the reference binary (1, default) or as evidently intended (0)
//   PQA_WORKERS=n              emulated thread-pool size (summation order of the posterior updates, training buckets)
//   PQA_SEED=n                 seed of the selector's generator (the reference's cannot be seeded)
//   PQA_COMBINE=0|1            concurrent NextQuestion calls of different quizzes share one sweep, RecordAnswer's kernels are gathered (1, default)
//   PQA_POLE_FIX=0|1           launched sweeps re-evaluate rows at the pole of the lack term in the reference's order (1, default; eval_kernels.hip: pole_fix)
//   PQA_DEVICES=i[,j,...]      device ordinal(s): read by the factory (c_abi.cpp), which builds one shard per listed device
void HipEngine::ApplyEnvironment() {
  auto num = [](const char *name, int64_t lo, int64_t hi, int64_t &out) {
    const char *v = std::getenv(name);
    if (!v || !*v) return false;
    char *end = nullptr;
    const long long x = std::strtoll(v, &end, 10);
    if (end == v || *end != 0 || x < lo || x > hi) {
      std::fprintf(stderr, "PqaCore: ignoring %s=%s (expected an integer in %lld..%lld)\n", name, v, (long long)lo, (long long)hi);
      return false;
    }
    out = x;
    return true;
  };
  if (const char *v = std::getenv("PQA_SELECT")) {
    const std::string sel(v);
    if (sel == "argmax" || sel == "1") _optSelect = 1;
    else if (sel == "sample" || sel == "sampled" || sel == "0") _optSelect = 0;
    else if (!sel.empty()) std::fprintf(stderr, "PqaCore: ignoring PQA_SELECT=%s (expected sample or argmax)\n", v);
  }
  int64_t x = 0;
  if (num("PQA_SERVER", 0, 1, x)) _optServer = x;
  if (num("PQA_BUG_COMPAT", 0, 1, x)) _optBugCompat = x;
  if (num("PQA_SPECULATE", 0, 1, x)) _optSpeculate = x;
  if (num("PQA_COMBINE", 0, 1, x)) _optCombine = x;
  if (num("PQA_POLE_FIX", 0, 1, x)) _optPoleFix = x;
  if (num("PQA_WORKERS", 1, kMaxWorkers, x)) _optWorkers = x;
  if (num("PQA_SEED", INT64_MIN, INT64_MAX, x)) { uint64_t s = (uint64_t)x; _rng[0] = SplitMix64(s); _rng[1] = SplitMix64(s); }
}

HipEngine::~HipEngine() {
  hipSetDevice(_device);
  StopServer();
  if (_serverStream) hipStreamDestroy(_serverStream);
  if (_serverRequestInVram) hipFree((void *)_serverRequest);
  if (_hMailbox) hipHostFree(_hMailbox);
  hipFree(_dServerCtl);
  if (_stream) hipStreamSynchronize(_stream);
  for (Quiz *q : _quizzes) if (q) DestroyQuiz(q);
  hipFree(_dCube); hipFree(_dVB); hipFree(_dPriority); hipFree(_dRunLength); hipFree(_dPoleScratch); hipFree(_dExps); hipFree(_dStatus);
  hipFree(_dNOut); hipFree(_dSel); hipFree(_dSelScratch); hipFree(_dPriorScratch); hipFree(_dClusterScratch);
  for (BatchCtx &c : _ctx) {
    hipFree(c.dSlots); hipFree(c.dScratch); hipFree(c.dPriority); hipFree(c.dPT); hipFree(c.dAcc); hipFree(c.dRecs); hipFree(c.dPriT); hipFree(c.dRerank);
    if (c.hPri) hipHostFree(c.hPri);
    if (c.event) hipEventDestroy(c.event);
    if (c.h) hipHostFree(c.h);
  }
  DropQuizBufferPool();
  for (auto &g : _graphs) hipGraphExecDestroy(g.second.exec);
  hipFree(_dGraphScratch); hipFree(_dTagCell);
  for (QuizPinned *slab : _pinSlabs) hipHostFree(slab);
  hipFree(_dTGap); hipFree(_dQGap); hipFree(_dAqs); hipFree(_dTop);
  if (_hPinned) hipHostFree(_hPinned);
  if (_hHostPriority) hipHostFree(_hHostPriority);
  if (_ownStream) hipStreamDestroy(_ownStream);
}

Error HipEngine::UploadGaps() {
  StopServer();  // its launch arguments hold the old view
  _kbVersion++;  // every change of the KB's shape or gaps passes through here: captured graphs hold the old view
  HIP_TRY(hipMemcpyAsync(_dTGap, _hTGap.data(), _hTGap.size() * sizeof(uint32_t), hipMemcpyHostToDevice, _stream));
  HIP_TRY(hipMemcpyAsync(_dQGap, _hQGap.data(), _hQGap.size() * sizeof(uint32_t), hipMemcpyHostToDevice, _stream));
  HIP_TRY(hipStreamSynchronize(_stream));
  return Error();
}

KbView HipEngine::View() const {
  KbView v;
  v.cube = _dCube; v.elem = _elem; v.vB = _dVB; v.tgap = _dTGap; v.qgap = _dQGap;
  v.K = _K; v.Q = _Q; v.T = _T; v.ldT = _ldT;
  v.nValidTargets = _T - _nTargetGaps;
  v.smallLaunches = _optServer ? 1 : 0;
  v.priorScratch = _optLongRowForm ? _dPriorScratch : nullptr;
  v.maxGrid = (int)_optEvalMaxGrid;
  v.poleScratch = _optPoleFix ? _dPoleScratch : nullptr;
  return v;
}

// ------------------------------------------------------------------------------------------------------------------
// options
// ------------------------------------------------------------------------------------------------------------------
Error HipEngine::SetOption(const char *name, int64_t value) {
  std::lock_guard<EngineMutex> lk(_mu);
  const std::string n(name ? name : "");
  (void)FlushUpdates();   // (deferred updates run under the options they were recorded under)
  if (n == "select") { if (value != 0 && value != 1) goto bad; _optSelect = value; }
  else if (n == "combine") { _optCombine = value ? 1 : 0; }
  else if (n == "combine_spin") { _optCombineSpin = value ? 1 : 0; }
  else if (n == "pole_fix") { StopServer(); _optPoleFix = value ? 1 : 0; _kbVersion++; }   // 0: questions with a row at the pole of the lack term keep the sweep's own sums (eval_kernels.hip: pole_fix)
  else if (n == "long_row_form") { _optLongRowForm = value ? 1 : 0; }   // 0: the one-workgroup posterior kernels for rows beyond 16384 targets too
  else if (n == "fuse_update") { _optFuseUpdate = value ? 1 : 0; }   // RecordAnswer's posterior update inside the speculative sweep's launch
  else if (n == "post_always") { _optPostAlways = value ? 1 : 0; }   // test hook: RecordAnswer / ListTopTargets always as posted operations
  else if (n == "combine_linger_us") { if (value < 0 || value > 10000) goto bad; _optLingerUs = value; }
  else if (n == "workers") { if (value < 1 || value > kMaxWorkers) goto bad; _optWorkers = value; }
  else if (n == "eval_subtasks") { if (value < 0 || value > 8192) goto bad; _optEvalSubtasks = value; }
  else if (n == "eval_variant") { if (value < 0) goto bad; _optEvalVariant = value; }
  else if (n == "bug_compat") { _optBugCompat = value ? 1 : 0; }
  else if (n == "use_graph") { _optUseGraph = value ? 1 : 0; }
  else if (n == "top_cache") { if (value < 0 || value > 256) goto bad; _optTopCache = value; _topWantRecent = value; }
  else if (n == "server") { StopServer(); _optServer = value ? 1 : 0; }
  else if (n == "speculate") { DropSpeculation(); _optSpeculate = value ? 1 : 0; _specScore = 0; }
  else if (n == "eval_max_grid") { if (value < 0 || value > 65535) goto bad; StopServer(); _optEvalMaxGrid = value; _kbVersion++; }
  else if (n == "fused_sampled") { _optFusedSampled = value ? 1 : 0; }
  else if (n == "host_sampled") { _optHostSampled = value ? 1 : 0; }
  else if (n == "batch_min") { if (value < 0 || value > 257) goto bad; _optBatchMin = value; }
  else if (n == "rerank") { _optRerank = value ? 1 : 0; }
  else if (n == "batch_form") { if (value < 0 || value > 3) goto bad; _optBatchForm = value; }
  else if (n == "batch_qb") { if (value < 0 || value > 4) goto bad; _optBatchQb = value; }
  else if (n == "batch_tile") { if (value < 0 || value > 8192) goto bad; _optBatchTile = value; }
  else if (n == "server_vram_mailbox") { if (_serverStream) goto bad; _optServerVramMailbox = value ? 1 : 0; }
  else if (n == "server_idle_us") { if (value < 10 || value > 1000000) goto bad; StopServer(); _optServerIdleUs = value; }
  else if (n == "seed") { uint64_t s = (uint64_t)value; _rng[0] = SplitMix64(s); _rng[1] = SplitMix64(s); }
  else goto bad;
  return Error();
bad:
  return Error::Make(ErrCode::UnhandledCase, "Unknown option or value out of range: " + n);
}

int64_t HipEngine::GetOption(const char *name) const {
  const std::string n(name ? name : "");
  if (n == "select") return _optSelect;
  if (n == "workers") return _optWorkers;
  if (n == "eval_subtasks") return _optEvalSubtasks ? _optEvalSubtasks : 8 * _optWorkers;
  if (n == "eval_variant") return _optEvalVariant;
  if (n == "bug_compat") return _optBugCompat;
  if (n == "top_cache") return _optTopCache;
  if (n == "use_graph") return _optUseGraph;
  if (n == "server") return _optServer;
  if (n == "server_idle_us") return _optServerIdleUs;
  if (n == "fused_sampled") return _optFusedSampled;
  if (n == "host_sampled") return _optHostSampled;
  if (n == "speculate") return _optSpeculate;
  if (n == "combine") return _optCombine;
  if (n == "combine_linger_us") return _optLingerUs;
  if (n == "combine_spin") return _optCombineSpin;
  if (n == "post_always") return _optPostAlways;
  if (n == "fuse_update") return _optFuseUpdate;
  if (n == "fused_updates") return (int64_t)_fusedUpdates;           // RecordAnswers whose update ran inside the sweep's launch
  if (n == "long_row_form") return _optLongRowForm;
  if (n == "pole_fix") return _optPoleFix;
  if (n == "allowed_cpus") return AllowedCpus();
  if (n == "combined_batches") return (int64_t)_combBatches;        // sweeps that served more than one NextQuestion call ...
  if (n == "combined_requests") return (int64_t)_combRequests;      // ... the calls they served ...
  if (n == "combined_max_batch") return (int64_t)_combMaxBatch;     // ... and the largest of them
  if (n == "posted_ops") return (int64_t)_postedOps;                 // RecordAnswer / ListTopTargets calls that found the engine taken and were run by its holder ...
  if (n == "posted_drains") return (int64_t)_postedDrains;
  if (n == "train_batches") return (int64_t)_trainBatches;           // launches that ran posted RecordQuizTarget calls together ...
  if (n == "train_batch_calls") return (int64_t)_trainBatchCalls;    // ... this many of them           // ... in this many rounds
  if (n == "update_flushes") return (int64_t)_flushes;              // launches that ran deferred RecordAnswers ...
  if (n == "updates_flushed") return (int64_t)_flushedUpdates;      // ... the updates they ran ...
  if (n == "update_max_flush") return (int64_t)_maxFlush;           // ... and the most in one launch
  // where the combined sweeps' time went (ns, summed): waiting for the engine, launching, waiting for the device, waiting for
  // the engine again, selecting on the host
  if (n == "combined_ns_lock") return (int64_t)_combNs[0];
  if (n == "combined_ns_launch") return (int64_t)_combNs[1];
  if (n == "combined_ns_device") return (int64_t)_combNs[2];
  if (n == "combined_ns_relock") return (int64_t)_combNs[3];
  if (n == "combined_ns_select") return (int64_t)_combNs[4];
  if (n == "combined_ns_selmu") return (int64_t)_combNs[5];
  if (n == "combined_ns_readers") return (int64_t)_combNs[6];
  if (n == "spec_hits") return (int64_t)_specHits;         // speculative sweeps a NextQuestion used ...
  if (n == "spec_dropped") return (int64_t)_specDropped;   // ... and those nothing used
  if (n == "batch_min") return _optBatchMin;
  if (n == "rerank") return _optRerank;
  if (n == "batch_form") return _optBatchForm;
  if (n == "batch_tile") return _optBatchTile;
  if (n == "batch_qb") return _optBatchQb;
  if (n == "precision") return _precType;
  if (n == "server_vram_mailbox") return _serverStream ? (_serverRequestInVram ? 1 : 0) : _optServerVramMailbox;
  if (n == "debug_mailbox") return (int64_t)(uintptr_t)_hMailbox;
  if (n == "server_last_step_ns") {   // device-side duration of the newest finished step of the resident sweep (-1: none)
    if (!_hMailbox || _serverPosted == 0) return -1;
    volatile ServerMailbox *mb = _hMailbox;
    const auto t0 = std::chrono::steady_clock::now();
    while (mb->pad[1] != _serverPosted)   // written right after the answer
      if (mb->state == kServerExited || std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(100)) return -1;
    return (int64_t)mb->pad[0] * 10;      // 100 MHz ticks
  }
  if (n == "server_active") return (_optServer && ServerUsable()) ? 1 : 0;
  if (n == "ldT") return _ldT;
  if (n == "device") return _device;
  return -1;
}

const char *HipEngine::EvalKernelName() const { 
  if (UseClusterSweep()) return EvalClusterKernelName(View());
  if (_elem == 8) return EvalVariantName(View(), (int)_optEvalVariant);
  return _optEvalVariant != 99 ? EvalF32KernelName(View(), (int)_optEvalVariant) : "f32_stream";
}

uint64_t HipEngine::NextRandom() {  // xorshift128+, the generator family of SRPlatform/Interface/SRFastRandom.h:60-72
  uint64_t s1 = _rng[0];
  const uint64_t s0 = _rng[1];
  _rng[0] = s0;
  s1 ^= s1 << 23;
  _rng[1] = s1 ^ s0 ^ (s1 >> 18) ^ (s0 >> 5);
  return _rng[1] + s0;
}

// ------------------------------------------------------------------------------------------------------------------
// mode / quiz registry
// ------------------------------------------------------------------------------------------------------------------
Error HipEngine::CheckRegular(const char *what) const {
  if (_mode == Mode::Regular) return Error();
  return Error::Make(ErrCode::WrongMode, std::string("Can't perform regular-only mode operation (") + what +
                                             ") because current mode is not regular (but maintenance/shutdown?).");
}

Quiz *HipEngine::UseQuiz(Error &err, int64_t iQuiz) {
  const int64_t nQuizzes = (int64_t)_quizzes.size();
  if (iQuiz < 0 || iQuiz >= nQuizzes) {
    err = Error::MakeP(ErrCode::IndexOutOfRange, RangeParams(iQuiz, 0, nQuizzes - 1),
                       "Quiz index is not in quiz registry range.");
    return nullptr;
  }
  if (_quizzes[iQuiz] == nullptr) {
    err = Error::MakeP(ErrCode::AbsentId, "id=" + std::to_string(iQuiz),
                       "Quiz index is not in the registry (but rather at a gap).");
    return nullptr;
  }
  _quizzes[iQuiz]->lastUsage = time(nullptr);  // BaseQuiz::OnUsage (BaseEngine.cpp:417)
  return _quizzes[iQuiz];
}

QuizPinned *HipEngine::TakePin() {
  if (_pinFree.empty()) {
    QuizPinned *slab = nullptr;
    if (hipHostMalloc((void **)&slab, kPinSlab * sizeof(QuizPinned), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
      (void)hipGetLastError();
      return nullptr;
    }
    std::memset(slab, 0, kPinSlab * sizeof(QuizPinned));
    _pinSlabs.push_back(slab);
    for (int i = kPinSlab - 1; i >= 0; i--) _pinFree.push_back(slab + i);
  }
  QuizPinned *p = _pinFree.back();
  _pinFree.pop_back();
  return p;
}

void HipEngine::DestroyQuiz(Quiz *q) {
  ServerQuiesce();
  if (!q) return;
  while (q->inSelection.load(std::memory_order_acquire)) _mm_pause();   // (a NextQuestion of this quiz is selecting on another thread: a client's error, waited out)
  if (q->updatePending) (void)FlushUpdates();   // (its kernel works on the buffers that go back to the pool)
  if (q->pin != nullptr) {
    if (_pendingRecordFlag == &q->pin->topFlag) { _pendingRecordOp = 0; _pendingRecordFlag = nullptr; _mu.busy = true; }
    _pinFree.push_back(q->pin);
    q->pin = nullptr;
  }
  if (_spec.quiz == q) DropSpeculation();
  {
    auto it = _graphs.find(q);
    if (it != _graphs.end()) { hipGraphExecDestroy(it->second.exec); _graphs.erase(it); }
  }
  if (q->dRowStage) { hipFree(q->dRowStage); q->dRowStage = nullptr; }
  if (q->dPrior && q->dAsked && _quizBufferPool.size() < 4096)
    _quizBufferPool.push_back(QuizBuffers{q->dPrior, q->dAsked, _ldT, q->hAsked.size()});
  else {
    hipFree(q->dPrior);
    hipFree(q->dAsked);
  }
  delete q;
}

void HipEngine::DropQuizBufferPool() {
  for (const QuizBuffers &b : _quizBufferPool) {
    hipFree(b.dPrior);
    hipFree(b.dAsked);
  }
  _quizBufferPool.clear();
}

// rows / srcPrior: support for a knowledge base whose question axis is split over several engines (sharded_engine.cpp).
//   rows != nullptr: the 2 nAnswered row pointers of LaunchResumeQuiz, resolved by the owners of the answered questions;
//   srcPrior != nullptr: the posterior was computed by another engine -- copy it (after `ready`) instead of computing it.
int64_t HipEngine::CreateQuiz(Error &err, int64_t nAnswered, const AQ *pAQs, const void *const *rows, const double *srcPrior,
                              int srcDevice, hipEvent_t ready) {
  err = CheckRegular("Start/Resume quiz");
  if (!err.ok()) return -1;
  hipSetDevice(_device);
  std::unique_ptr<Quiz> quiz(new Quiz());
  auto fail = [&](Error e) {
    err = std::move(e);
    hipFree(quiz->dPrior);
    hipFree(quiz->dAsked);
    if (quiz->pin) _pinFree.push_back(quiz->pin);
    return (int64_t)-1;
  };
  quiz->serial = ++_quizSerial;
  quiz->pin = TakePin();
  if (quiz->pin == nullptr) return fail(HipErr(hipErrorOutOfMemory, "quiz result lines"));
  quiz->hAsked.assign(BitWords(_Q), 0);
  // validate the answered questions and set their "asked" bits (reference PqaCore/CpuEngine.cpp:216-233)
  bool allLocal = true;
  for (int64_t i = 0; i < nAnswered; i++) {
    const int64_t iq = pAQs[i].iQuestion, ia = pAQs[i].iAnswer;
    if (iq < 0 || iq >= _qTotal)
      return fail(Error::MakeP(ErrCode::IndexOutOfRange, RangeParams(iq, 0, _qTotal - 1), "Question index is not in KB range."));
    if (ia < 0 || ia >= _K)
      return fail(Error::MakeP(ErrCode::IndexOutOfRange, RangeParams(ia, 0, _K - 1), "Answer index is not in KB range."));
    if (iq >= _qFirst && iq < _qFirst + _Q) BitSet(quiz->hAsked, iq - _qFirst, true); else allLocal = false;
  }
  if (!allLocal && rows == nullptr && srcPrior == nullptr)
    return fail(Error::MakeP(ErrCode::NotImplemented, "Feature=ResumeQuiz across separately driven shards",
                             "An answered question belongs to another shard: its rows are not reachable from this engine alone "
                             "(PQA_DEVICES / the sharded engine of one process resolves them)."));
  hipError_t he = hipSuccess;
  while (!_quizBufferPool.empty() && quiz->dPrior == nullptr) {
    const QuizBuffers b = _quizBufferPool.back();
    _quizBufferPool.pop_back();
    if (b.ldT == _ldT && b.askedWords == quiz->hAsked.size()) {
      quiz->dPrior = b.dPrior;
      quiz->dAsked = b.dAsked;
    } else {  // the knowledge base changed shape since that quiz was released
      hipFree(b.dPrior);
      hipFree(b.dAsked);
    }
  }
  if (quiz->dPrior == nullptr) {
    he = hipMalloc(&quiz->dPrior, (size_t)_ldT * sizeof(double));
    if (he == hipSuccess) he = hipMalloc(&quiz->dAsked, quiz->hAsked.size() * sizeof(uint32_t));
  }
  const bool startClears = nAnswered == 0 && srcPrior == nullptr;   // StartQuiz: its kernel clears the bitmap itself
  if (he == hipSuccess && !startClears)   // (ResumeQuiz synchronises further down: the host source stays valid)
    he = nAnswered == 0 ? hipMemsetAsync(quiz->dAsked, 0, quiz->hAsked.size() * sizeof(uint32_t), _stream)
                        : hipMemcpyAsync(quiz->dAsked, quiz->hAsked.data(), quiz->hAsked.size() * sizeof(uint32_t),
                                         hipMemcpyHostToDevice, _stream);
  if (he != hipSuccess) return fail(HipErr(he, "quiz allocation"));
  const KbView kb = View();
  if (srcPrior != nullptr) {
    if (ready != nullptr) he = hipStreamWaitEvent(_stream, ready, 0);
    if (he == hipSuccess)
      he = hipMemcpyPeerAsync(quiz->dPrior, _device, srcPrior, srcDevice, (size_t)_ldT * sizeof(double), _stream);
    if (he != hipSuccess) return fail(HipErr(he, "adopting another shard's posterior"));
    for (int64_t i = 0; i < nAnswered; i++) quiz->answers.push_back(pAQs[i]);
  } else if (nAnswered == 0) {
    // CECreateQuizStart::UpdateLikelihoods, reference PqaCore/CECreateQuizOperation.cpp:22-53
    if (_startBatch != nullptr) {   // StartQuizBatch: one launch for all its quizzes
      _startBatch->prior[_startBatch->n] = quiz->dPrior;
      _startBatch->asked[_startBatch->n] = quiz->dAsked;
      _startBatch->n++;
    } else {
      he = LaunchStartQuiz(kb, quiz->dPrior, quiz->dAsked, (int64_t)quiz->hAsked.size(), _optWorkers, _stream);
      if (he != hipSuccess) return fail(HipErr(he, "LaunchStartQuiz"));
    }
    // no synchronisation: every reader of the prior or the bitmap is ordered behind these on the engine's stream
  } else {
    // CECreateQuizResume::UpdateLikelihoods, reference PqaCore/CECreateQuizOperation.cpp:55-83
    if (nAnswered > _aqCapacity) {
      hipFree(_dAqs);
      _dAqs = nullptr;
      _aqCapacity = std::max<int64_t>(nAnswered, 64);
      he = hipMalloc(&_dAqs, (size_t)_aqCapacity * 2 * sizeof(int64_t));
      if (he != hipSuccess) { _aqCapacity = 0; return fail(HipErr(he, "aq buffer")); }
    }
    std::vector<const void *> local(2 * (size_t)nAnswered);
    for (int64_t i = 0; i < nAnswered; i++) {
      if (rows != nullptr) { local[2 * i] = rows[2 * i]; local[2 * i + 1] = rows[2 * i + 1]; continue; }
      local[2 * i] = CubeAt(pAQs[i].iQuestion - _qFirst, pAQs[i].iAnswer);
      local[2 * i + 1] = CubeAt(pAQs[i].iQuestion - _qFirst, _K);
    }
    static_assert(sizeof(void *) == sizeof(int64_t), "the pointer list travels in the answered-question buffer");
    he = hipMemcpyAsync(_dAqs, local.data(), local.size() * sizeof(void *), hipMemcpyHostToDevice, _stream);
    if (he == hipSuccess)
      he = LaunchResumeQuiz(kb, quiz->dPrior, _dExps, reinterpret_cast<const void *const *>(_dAqs), nAnswered, _optWorkers,
                            (int)_optBugCompat, _dStatus, _stream);
    if (he == hipSuccess)
      he = hipMemcpyAsync(_hPinned->status, _dStatus, 2 * sizeof(int64_t), hipMemcpyDeviceToHost, _stream);
    if (he == hipSuccess) he = hipStreamSynchronize(_stream);
    if (he != hipSuccess) return fail(HipErr(he, "ResumeQuiz"));
    if (_hPinned->status[0] != 0) {  // reference PqaCore/CpuEngine.cpp:317-321
      const int64_t highBound = 1023 + 1023 - (int64_t)std::ceil(std::log2((double)_T)) - 2;
      const int64_t minAllowed = INT64_MIN + highBound + 1;
      return fail(Error::MakeP(ErrCode::I64Underflow,
                               "actual=" + std::to_string(_hPinned->status[1]) + ", minAllowed=" + std::to_string(minAllowed),
                               "Max exponent over the priors is too low. Are all the targets in gaps?"));
    }
    for (int64_t i = 0; i < nAnswered; i++) quiz->answers.push_back(pAQs[i]);
  }
  quiz->lastUsage = time(nullptr);
  return AssignQuiz(quiz.release());
}

int64_t HipEngine::StartQuiz(Error &err) {
  CallScope scope(_activeCallers);
  if (_optCombine && (_optPostAlways || !_mu.try_lock())) {   // (the engine is taken: the quizzes started meanwhile share ONE launch)
    PostedOp op;
    op.kind = 4;
    RunPosted(op);
    err = op.err;
    return op.result;
  }
  std::unique_lock<EngineMutex> lk(_mu, std::defer_lock);
  if (_optCombine) lk = std::unique_lock<EngineMutex>(_mu, std::adopt_lock);
  else lk.lock();
  return SpeculateFor(CreateQuiz(err, 0, nullptr, nullptr, nullptr, 0, nullptr));
}

// (what follows StartQuiz / ResumeQuiz is NextQuestion: its sweep goes out right behind the kernel that sets the priors)
int64_t HipEngine::SpeculateFor(int64_t iQuiz) {
  if (iQuiz >= 0 && (size_t)iQuiz < _quizzes.size() && _quizzes[(size_t)iQuiz] != nullptr) Speculate(_quizzes[(size_t)iQuiz]);
  return iQuiz;
}

int64_t HipEngine::ResumeQuiz(Error &err, int64_t nAnswered, const AQ *pAQs) {
  if (nAnswered < 0) {  // reference PqaCore/BaseEngine.cpp:388-392
    err = Error::MakeP(ErrCode::NegativeCount, "count=" + std::to_string(nAnswered), "|nAnswered| must be non-negative.");
    return -1;
  }
  if (nAnswered > 0 && pAQs == nullptr) {
    err = Error::Make(ErrCode::NullArgument, "Nullptr is passed in place of answered questions.");
    return -1;
  }
  CallScope scope(_activeCallers);
  std::lock_guard<EngineMutex> lk(_mu);
  return SpeculateFor(CreateQuiz(err, nAnswered, pAQs, nullptr, nullptr, 0, nullptr));  // nAnswered == 0 -> StartQuiz (BaseEngine.cpp:393-395)
}

Error HipEngine::ReleaseQuiz(int64_t iQuiz) {
  CallScope scope(_activeCallers);
  if (_optCombine && (_optPostAlways || !_mu.try_lock())) {
    PostedOp op;
    op.kind = 5; op.iQuiz = iQuiz;
    RunPosted(op);
    return op.err;
  }
  std::unique_lock<EngineMutex> lk(_mu, std::defer_lock);
  if (_optCombine) lk = std::unique_lock<EngineMutex>(_mu, std::adopt_lock);
  else lk.lock();
  return ReleaseQuizLocked(iQuiz, true);
}

Error HipEngine::ReleaseQuizLocked(int64_t iQuiz, bool mayWait) {
  Error err = CheckRegular("release quiz");
  if (!err.ok()) return err;
  Quiz *q = UseQuiz(err, iQuiz);
  if (!q) return err;
  // A NextQuestion of this quiz is selecting on another thread (a combined sweep's client, outside the lock): concurrent calls on
  // one quiz are the caller's error (IPqaEngine.h:44).  A direct call waits it out, briefly; a posted one cannot -- the thread
  // that runs the drain may be the very leader whose CollectBatch ends the selection -- and is refused.
  if (q->inSelection.load(std::memory_order_acquire)) {
    const auto t0 = std::chrono::steady_clock::now();
    while (mayWait && q->inSelection.load(std::memory_order_acquire) && std::chrono::steady_clock::now() - t0 < std::chrono::seconds(2)) _mm_pause();
    if (q->inSelection.load(std::memory_order_acquire))
      return Error::MakeP(ErrCode::Internal, "quizId=" + std::to_string(iQuiz),
                          "The quiz is inside a NextQuestion call on another thread: it cannot be released now (no concurrent calls on one quiz).");
  }
  hipSetDevice(_device);
  UnassignQuiz(iQuiz);
  DestroyQuiz(q);  // the buffers go to the pool; their next user is ordered behind pending work on the engine's stream
  return Error();
}

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
bool HipEngine::UseClusterSweep() const { return _optEvalVariant == 0 && _ldT > 16384 && EvalClusterSupported(View()); }

Error HipEngine::LaunchSingleSweep(Quiz *q, const FusedSelect *fused) {
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
    HIP_TRY(LaunchEvalQuestions(View(), q->dPrior, q->dAsked, 0, _Q, _dPriority, (int)_optEvalVariant, fused, _stream));
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
  if (_optServer && ServerUsable()) return ServerPost(q, (SelectResult *)pOut, (uint64_t *)pFlag, flagValue, _qFirst);
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
  const auto t0 = std::chrono::steady_clock::now();
  for (int64_t i = 0; i < _Q; i++) {
    if (BitTest(_hQGap, i) || BitTest(q->hAsked, i)) { _hostRun[(size_t)i] = 0.0; continue; }
    uint64_t spins = 0;
    while (rec[i].tag != tag)
      if ((++spins & 0xFFFF) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(30))
        return HipErr(hipErrorNotReady, "priority vector hand-over");
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
  const auto t0 = std::chrono::steady_clock::now();
  uint64_t spins = 0;
  while (*flag != value) {
    if ((++spins & 0xFFF) == 0) {
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

int64_t HipEngine::NextQuestionArgmax(Error &err, int64_t iQuiz) { return Combine(err, iQuiz, 0, 0); }

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
  if (_optServer && ServerUsable()) {
    // resident sweep: post the request, poll the answer -- no launch on the critical path
    const uint64_t value = kServerFlagBase | ++_opSeq;   // (its own range: see kGraphFlagBase)
    err = ServerPost(q, &_hPinned->sel, &_hPinned->seq, value, 0);
    if (err.ok()) err = ServerWait(&_hPinned->seq, value, "NextQuestionArgmax");
    if (!err.ok()) return -1;
    if (_hPinned->sel.index == -3) {
      err = HipErr(hipErrorLaunchFailure, "NextQuestionArgmax (incomplete sweep)");
      return -1;
    }
    CheckPriority(_hPinned->sel.priority, _hPinned->sel.index);
  return FinishSelection(err, q, _hPinned->sel.index);
  }
  // One launch; the last workgroup writes the winner and then a sequence number straight into host-coherent pinned
  // memory, which this thread polls: no D2H copy, no stream synchronisation on the critical path.
  uint64_t seq;
  if (TakeSpeculation(q, 1 << 1, &seq) == 0) {   // (else: RecordAnswer has launched this very sweep already)
    seq = NextLaunchTag();
    const FusedSelect fs{_dSelScratch, &_hPinned->sel, &_hPinned->seq, seq, 0, 0, seq, nullptr, 0, 0, nullptr, nullptr};
    StopServer();   // a launched sweep has no room beside the resident one and would wait for it to idle out
    err = LaunchSingleSweep(q, &fs);
    if (!err.ok()) return -1;
  }
  err = WaitFlag(&_hPinned->seq, seq, "NextQuestionArgmax");
  if (!err.ok()) return -1;
  std::atomic_thread_fence(std::memory_order_acquire);
  if (_hPinned->sel.index == -3) {  // the sweep's finisher gave up: some workgroup of the launch never reported
    err = HipErr(hipErrorLaunchFailure, "NextQuestionArgmax (incomplete sweep)");
    return -1;
  }
  CheckPriority(_hPinned->sel.priority, _hPinned->sel.index);
  return FinishSelection(err, q, _hPinned->sel.index);
}

// ------------------------------------------------------------------------------------------------------------------
// resident sweep (pqa_kernels.h: ServerMailbox; eval_kernels.hip: eval_server_f64)
// ------------------------------------------------------------------------------------------------------------------
bool HipEngine::ServerUsable() const { return _elem == 8 && EvalServerSupported(View(), (int)_optEvalVariant) && _Q > 0; }

void HipEngine::StopServer() {
  (void)FlushUpdates();   // whoever stops the resident sweep is about to read or change what the deferred updates read or write
  DropSpeculation();   // whatever ends the resident sweep's view of the engine (cube, gaps, stream, buffers) ends a speculative result's too
  if (!_serverLaunched) return;
  hipSetDevice(_device);
  _serverRequest[7] = 1;                 // `stop`
  std::atomic_thread_fence(std::memory_order_seq_cst);
  hipStreamSynchronize(_serverStream);   // bounded: the kernel polls `stop` and leaves, or has left already
  _serverRequest[7] = 0;
  std::atomic_thread_fence(std::memory_order_seq_cst);
  _serverLaunched = false;
}

void HipEngine::ServerQuiesce() {
  if (!_serverLaunched || _serverPosted == 0) return;
  volatile ServerMailbox *mb = _hMailbox;
  const auto t0 = std::chrono::steady_clock::now();
  uint64_t spins = 0;
  while (mb->done != _serverPosted && mb->state != kServerExited)
    if ((++spins & 0xFFF) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(30)) return;
}

Error HipEngine::ServerWait(volatile uint64_t *flag, uint64_t value, const char *what) {
  const auto t0 = std::chrono::steady_clock::now();
  uint64_t spins = 0;
  volatile ServerMailbox *mb = _hMailbox;
  while (*flag != value) {
    if ((++spins & 0xFFF) == 0) {
      if (mb->state == kServerExited && mb->taken != _serverPosted && *flag != value) return HipErr(hipErrorUnknown, what);
      if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(30)) return HipErr(hipErrorNotReady, what);
    }
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  return Error();
}

// Post one selection request for quiz `q`; the finisher writes {priority, index + outBase} to `out` and then flagValue to
// `flag` (host-coherent memory).  Starts the kernel if none is resident.
Error HipEngine::ServerPost(Quiz *q, SelectResult *out, uint64_t *flag, uint64_t flagValue, int64_t outBase) {
  { Error fe = FlushUpdates(); if (!fe.ok()) return fe; }   // (a deferred RecordAnswer of this quiz -- posterior and asked bit -- is what the request reads)
  // the resident kernel is not ordered behind the engine's stream: wait for what that stream still runs
  if (_pendingRecordOp != 0 && _pendingRecordFlag != nullptr && !_mu.wasBusy) {
    Error e = WaitFlag(_pendingRecordFlag, _pendingRecordOp, "ServerPost");
    if (!e.ok()) return e;
  } else if (_mu.wasBusy) {
    HIP_TRY(hipStreamSynchronize(_stream));
  }
  _pendingRecordOp = 0;
  _mu.busy = false;
  if (!_serverStream) {
    // A stream of its own PRIORITY, not just of its own: the runtime multiplexes streams of one priority over a few hardware
    // queues, and a posterior kernel whose packet sits behind the resident kernel's in the same queue waits until that
    // leaves (measured: 2 ms per quiz step, the idle time).  Queues are pooled per priority.
    int prLeast = 0, prGreatest = 0;
    HIP_TRY(hipDeviceGetStreamPriorityRange(&prLeast, &prGreatest));
    HIP_TRY(hipStreamCreateWithPriority(&_serverStream, hipStreamNonBlocking, prLeast));
    HIP_TRY(hipHostMalloc((void **)&_hMailbox, sizeof(ServerMailbox), hipHostMallocDefault));
    std::memset(_hMailbox, 0, sizeof(ServerMailbox));
    HIP_TRY(hipMalloc((void **)&_dServerCtl, sizeof(ServerCtl)));
    // The request line in device memory that the host can write (fine-grained allocation, mapped through the PCIe BAR):
    // the kernel's polls become local reads.  Where the platform does not map it, the mailbox's own first line is used.
    void *vram = nullptr;
    int largeBar = 0;
    if (_optServerVramMailbox && hipDeviceGetAttribute(&largeBar, hipDeviceAttributeIsLargeBar, _device) == hipSuccess && largeBar &&
        hipExtMallocWithFlags(&vram, 64, hipDeviceMallocFinegrained) == hipSuccess && vram != nullptr) {
      _serverRequest = (volatile uint64_t *)vram;   // large BAR: the device address is valid on the host as well
      _serverRequestInVram = true;
      for (int i = 0; i < 8; i++) _serverRequest[i] = 0;
      std::atomic_thread_fence(std::memory_order_seq_cst);
    } else {
      (void)hipGetLastError();
    }
    if (!_serverRequestInVram) _serverRequest = &_hMailbox->req;
  }
  if (_serverLaunched && (_serverKb != _kbVersion || _serverVariant != _optEvalVariant)) StopServer();
  volatile ServerMailbox *mb = _hMailbox;
  // the previous request's fields must have been read before they are overwritten
  if (_serverLaunched && _serverPosted != 0) {
    const auto t0 = std::chrono::steady_clock::now();
    uint64_t spins = 0;
    // (every workgroup reads the line itself when it is in device memory: then not before the step is done)
    while ((_serverRequestInVram ? mb->done : mb->taken) != _serverPosted && mb->state != kServerExited) {
      if ((++spins & 0xFFF) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(30))
        return HipErr(hipErrorNotReady, "ServerPost (previous request never taken)");
    }
  }
  const uint64_t prev = _serverReqSeq;
  const uint64_t seq = NextLaunchTag();
  volatile uint64_t *rq = _serverRequest;   // {req, prior, asked, out, flag, flagValue, outBase, stop}
  rq[1] = (uint64_t)(uintptr_t)q->dPrior;
  rq[2] = (uint64_t)(uintptr_t)q->dAsked;
  rq[3] = (uint64_t)(uintptr_t)out;
  rq[4] = (uint64_t)(uintptr_t)flag;
  rq[5] = flagValue;
  rq[6] = (uint64_t)outBase;
  std::atomic_thread_fence(std::memory_order_seq_cst);   // (also drains the write-combining buffer of a BAR mapping)
  rq[0] = seq;
  std::atomic_thread_fence(std::memory_order_seq_cst);
  _serverReqSeq = seq;
  _serverPosted = seq;
  if (_serverLaunched) {
    // Taken, or gone?  The kernel acknowledges a request as soon as it reads it (~2 us); a kernel that was leaving when the
    // request arrived ends in `exited` without the acknowledgement, and the request -- still in its line -- goes to a new
    // one.  (With the line in host memory "write mine, then read yours" on both sides would decide this without waiting:
    // PCIe keeps the kernel's read behind its write.  A line in device memory is written by the host with a posted write
    // that may still be in flight when the host looks at `state`, so the acknowledgement is what is relied on.)
    const auto t0 = std::chrono::steady_clock::now();
    uint64_t spins = 0;
    for (;;) {
      if (mb->taken == seq) return Error();
      if (mb->state == kServerExited) {
        std::atomic_thread_fence(std::memory_order_acquire);
        if (mb->taken == seq) return Error();
        break;
      }
      if ((++spins & 0xFFF) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(30))
        return HipErr(hipErrorNotReady, "ServerPost (request neither taken nor refused)");
    }
    _serverLaunched = false;   // it left without this request
  }
  HIP_TRY(hipStreamSynchronize(_serverStream));                       // the previous instance is gone entirely
  HIP_TRY(hipMemsetAsync(_dServerCtl, 0, sizeof(ServerCtl), _serverStream));
  mb->state = kServerRunning;
  std::atomic_thread_fence(std::memory_order_seq_cst);
  HIP_TRY(EnsureHostPriority());   // (a launch argument of the resident kernel: requests may ask for the priority vector)
  HIP_TRY(LaunchEvalServer(View(), 0, _Q, _dPriority, (int)_optEvalVariant, _dSelScratch, _hMailbox, (void *)_serverRequest, _serverRequestInVram, _dServerCtl, prev,
                           (uint64_t)_optServerIdleUs * 100, _hHostPriority, _serverStream));   // 100 MHz ticks
  _serverLaunched = true;
  _serverKb = _kbVersion;
  _serverVariant = _optEvalVariant;
  return Error();
}

// Argmax selections for several quizzes at once.  pOut[i] = the selected GLOBAL question of pQuizzes[i], or -1 when that quiz
// has run out of questions (not an error of the call).  Two forms:
//   * the row-sharing sweep (batch_kernels.hip; batches of at least `batch_min` quizzes, and every batch of a Float engine):
//     a lane is a quiz, the cube tile staged in LDS serves all quizzes of the batch -- the cube is read once per batch;
//   * grid.y = quiz over the single-quiz kernel (small batches of Double engines): one launch, but one cube read per quiz.
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
  const int64_t qb = _optBatchQb > 0 ? _optBatchQb : (_elem == 4 ? 4 : 2), wavesRowSharing = ((n + 63) / 64) * ((_Q + qb - 1) / qb);
  bool rowSharing = _elem == 4 || wantPriorities || (_optBatchMin > 0 ? n >= _optBatchMin : (n >= 32 && wavesRowSharing >= 1536));
  // ... and between the two, for a few dozen quizzes over short rows (a server's combined sweeps): a lane is a (quiz, chunk of the
  // row) -- batch_kernels.hip: eval_midbatch_kernel.  Option batch_form: 0 = by these rules, 1 grid.y = quiz, 2 row-sharing, 3 this one.
  // By the measured costs at 1000 x 5 x 1000 (tools/midbatch_bench.py): grid.y ~11.3 us per quiz + 25; this form 87 / 138 / 229 us for up
  // to 8 / 16 / 32 quizzes (its lanes come in 8, 16 or 32 quiz slots) and 6.2 us per slot of 64 beyond: it wins at 7 and 8 quizzes and from
  // 11 on, except 17 and 18.
  bool mid = EvalMidBatchSupported(View()) && ((_optBatchForm == 0 && !rowSharing && (n == 7 || n == 8 || (n >= 11 && n <= 16) || n >= 19)) || _optBatchForm == 3);
  if (_optBatchForm == 1 && _elem == 8 && !wantPriorities) { rowSharing = false; mid = false; }
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
  if (mid) {
    const KbView kb = View();
    BatchPlan plan{};
    HIP_TRY(LaunchEvalMidBatch(kb, c.dSlots, (int)n, &plan, nullptr, nullptr, nullptr, 0, tag, true, _stream));
    HIP_TRY(grow(&c.dPT, c.ptBytes, plan.ptBytes));
    HIP_TRY(grow((void **)&c.dRecs, c.recBytes, plan.recBytes));
    if (wantPriorities) HIP_TRY(grow((void **)&c.dPriT, c.priTBytes, (size_t)_Q * (size_t)plan.Bp * sizeof(double)));
    HIP_TRY(LaunchEvalMidBatch(kb, c.dSlots, (int)n, &plan, c.dPT, c.dRecs, wantPriorities ? c.dPriT : nullptr, 0, tag, false, _stream));
    c.lastBp = plan.Bp;
    return Error();
  }
  if (!rowSharing) {
    const FusedSelect fs{c.dScratch, nullptr, nullptr, tag, 0, kBatchGrid, tag, nullptr, tagged ? 1 : 0, 0, nullptr,
                         tagged ? reinterpret_cast<TaggedPriority *>(c.hPri) : nullptr};
    HIP_TRY(LaunchEvalQuestionsBatch(View(), c.dSlots, (int)n, 0, _Q, (int)_optEvalVariant, fs, _stream));
    if (hostPriorities && !tagged) return copyToHost(c.dPriority, (size_t)n * (size_t)_Q);
    return Error();
  }
  const KbView kb = View();
  BatchPlan plan{};
  plan.tileTargets = (int)_optBatchTile;
  plan.questionsPerBlock = (int)_optBatchQb;
  HIP_TRY(LaunchEvalBatch(kb, c.dSlots, (int)n, &plan, nullptr, nullptr, nullptr, nullptr, 0, tag, true, _stream));
  HIP_TRY(grow(&c.dPT, c.ptBytes, plan.ptBytes));
  HIP_TRY(grow((void **)&c.dAcc, c.accBytes, plan.accBytes));
  HIP_TRY(grow((void **)&c.dRecs, c.recBytes, plan.recBytes));
  // Float engines: the fp32 sweep nominates every quiz's best questions, fp64 decides among them (option "rerank", default on)
  const bool rerank = _elem == 4 && _optRerank != 0;
  if (wantPriorities || rerank) HIP_TRY(grow((void **)&c.dPriT, c.priTBytes, (size_t)_Q * (size_t)plan.Bp * sizeof(double)));
  if (rerank) HIP_TRY(grow(&c.dRerank, c.rerankBytes, BatchRerankScratchBytes()));
  HIP_TRY(LaunchEvalBatch(kb, c.dSlots, (int)n, &plan, c.dPT, c.dAcc, c.dRecs, (wantPriorities || rerank) ? c.dPriT : nullptr, 0, tag,
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
  const auto t0 = std::chrono::steady_clock::now();
  for (int64_t i = 0; i < n; i++) {
    volatile uint64_t *flag = &c.h->seq[i];
    uint64_t spins = 0;
    while (*flag != tag) {
      if ((++spins & 0xFFF) == 0) {
        if (hipStreamQuery(_stream) == hipSuccess && *flag != tag) {  // the kernel retired without publishing
          const hipError_t he = hipStreamSynchronize(_stream);
          if (he != hipSuccess || *flag != tag) return HipErr(he == hipSuccess ? hipErrorUnknown : he, "batched selection (result flag)");
        }
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(600))
          return HipErr(hipErrorNotReady, "batched selection (timeout)");
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
  if (_optServer && _optHostSampled && !_optFusedSampled && ServerUsable()) {
    // resident sweep: post the request with the hand-over mark, poll the flag, select on the host -- no launch on the path
    const uint64_t value = kServerFlagBase | ++_opSeq;   // (its own range: see kGraphFlagBase)
    err = ServerPost(q, &_hPinned->sel, &_hPinned->seq, value, (int64_t)kServerHandOver);
    if (err.ok()) err = ServerWait(&_hPinned->seq, value, "NextQuestionSampled");
    if (!err.ok()) return -1;
    if (_hPinned->sel.index == -3) { err = HipErr(hipErrorLaunchFailure, "NextQuestionSampled (incomplete sweep)"); return -1; }
    err = CollectHostPriority(_serverPosted, q);
    if (!err.ok()) return -1;
    const int64_t sel = SelectSampledHostBits(_hostRun.data(), _Q, nSub, rnd, _hQGap.data(), q->hAsked.data());
    return FinishSelection(err, q, sel);
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
    if (!speculated) {
      seq = NextLaunchTag();
      const FusedSelect fs{_dSelScratch, &_hPinned->sel, &_hPinned->seq, seq, 0, 0, seq, nullptr, 1, 0, nullptr, _hHostPriority};
      const hipError_t he = LaunchEvalQuestions(kb, q->dPrior, q->dAsked, 0, _Q, _dPriority, (int)_optEvalVariant, &fs, _stream);
      if (he != hipSuccess) { err = HipErr(he, "NextQuestionSampled"); return -1; }
    }
    err = WaitFlag(&_hPinned->seq, seq, "NextQuestionSampled");
    if (!err.ok()) return -1;
    if (_hPinned->sel.index == -3) { err = HipErr(hipErrorLaunchFailure, "NextQuestionSampled (incomplete sweep)"); return -1; }
    err = CollectHostPriority(seq, q);
    if (!err.ok()) return -1;
    const int64_t sel = SelectSampledHostBits(_hostRun.data(), _Q, nSub, rnd, _hQGap.data(), q->hAsked.data());
    return FinishSelection(err, q, sel);
  }
  if (took != 3 && _optFusedSampled && _elem == 8 && EvalVariantFusesSampled(kb, (int)_optEvalVariant, nSub)) {
    // ONE launch: the sweep's finisher workgroup runs the reference's selector once every workgroup has reported
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
    const auto t0 = std::chrono::steady_clock::now();
    for (int64_t k = 0; k < nQ; k++) {
      if (skip(k)) { run[(size_t)k] = 0.0; continue; }
      const volatile uint64_t *tagWord = reinterpret_cast<const volatile uint64_t *>(rec + 2 * k + 1);
      for (uint64_t spins = 0; *tagWord != r->priTag;)
        if ((++spins & 0xFFFF) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(30)) {
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
  if (!_optSpeculate || _optServer || _optUseGraph || _qTotal != _Q || _Q <= 0) return false;
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
  const FusedSelect fs{_dSelScratch, &_hPinned->sel, &_hPinned->seq, seq, 0, 0, seq, nullptr, kind == 2 ? 1 : 0, 0, nullptr,
                       kind == 2 ? _hHostPriority : nullptr};
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
  _spec.quiz = q; _spec.priorVersion = q->priorVersion; _spec.tag = seq; _spec.kind = kind;
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
                     _spec.variant == _optEvalVariant && _spec.stream == _stream && !_optServer && !_optUseGraph;
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

int64_t HipEngine::ListTopTargets(Error &err, int64_t iQuiz, int64_t maxCount, CiRatedTarget *pDest) {
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
  // large lists: sort on the host (the listing is O(T log T) on 8T bytes, not a cube operation)
  std::vector<double> pri((size_t)_T);
  hipError_t he = hipMemcpyAsync(pri.data(), q->dPrior, (size_t)_T * sizeof(double), hipMemcpyDeviceToHost, _stream);
  if (he == hipSuccess) he = hipStreamSynchronize(_stream);
  if (he != hipSuccess) { err = HipErr(he, "ListTopTargets"); return -1; }
  std::vector<int64_t> idx;
  idx.reserve((size_t)_T);
  for (int64_t t = 0; t < _T; t++) if (!BitTest(_hTGap, t)) idx.push_back(t);
  const int64_t n = std::min<int64_t>(want, (int64_t)idx.size());
  std::partial_sort(idx.begin(), idx.begin() + n, idx.end(),
                    [&](int64_t a, int64_t b) { return pri[a] > pri[b] || (pri[a] == pri[b] && a < b); });
  for (int64_t i = 0; i < n; i++) { pDest[i]._iTarget = idx[i]; pDest[i]._prob = pri[idx[i]]; }
  return n;
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

// ------------------------------------------------------------------------------------------------------------------
// misc
// ------------------------------------------------------------------------------------------------------------------
uint64_t HipEngine::GetTotalQuestionsAsked(Error &err) { err = Error(); return _nQuestionsAsked.load(std::memory_order_relaxed); }

void HipEngine::CopyDims(CiEngineDimensions *pDims) const {
  pDims->_nAnswers = _K;
  pDims->_nQuestions = _qTotal;
  pDims->_nTargets = _T;
}

Error HipEngine::StartMaintenance(bool forceQuizzes) {  // reference PqaCore/BaseEngine.cpp:640-690
  std::lock_guard<EngineMutex> lk(_mu);
  StopServer();
  if (_mode == Mode::Shutdown) return Error::MakeP(ErrCode::ObjectShutDown, "RejectedOperation=StartMaintenance", "Engine is shut down.");
  if (_mode == Mode::Maintenance) return Error::MakeP(ErrCode::MaintenanceModeAlreadyThis, "ActiveMode=#1", "Already in maintenance mode.");
  int64_t nActive = 0;
  for (Quiz *q : _quizzes) nActive += q ? 1 : 0;
  if (nActive > 0) {
    if (!forceQuizzes)
      return Error::MakeP(ErrCode::QuizzesActive, "nQuizzes=[" + std::to_string(nActive) + "]",
                          "Can't switch to maintenance mode while there are active quizzes.");
    hipSetDevice(_device);
    hipStreamSynchronize(_stream);
    for (size_t i = 0; i < _quizzes.size(); i++)
      if (_quizzes[i]) { Quiz *q = _quizzes[i]; UnassignQuiz((int64_t)i); DestroyQuiz(q); }
  }
  _mode = Mode::Maintenance;
  return Error();
}

Error HipEngine::FinishMaintenance() {  // reference PqaCore/BaseEngine.cpp:692-712
  std::lock_guard<EngineMutex> lk(_mu);
  if (_mode == Mode::Shutdown) return Error::MakeP(ErrCode::ObjectShutDown, "RejectedOperation=FinishMaintenance", "Engine is shut down.");
  if (_mode == Mode::Regular) return Error::MakeP(ErrCode::MaintenanceModeAlreadyThis, "ActiveMode=#0", "Already in regular mode.");
  _mode = Mode::Regular;
  return Error();
}

Error HipEngine::Shutdown(const char *saveFilePath) {
  if (saveFilePath && *saveFilePath) {  // reference PqaCore/BaseEngine.cpp:270-300: save, then shut down
    Error e = SaveKB(saveFilePath, false);
    if (!e.ok()) return e;
  }
  std::lock_guard<EngineMutex> lk(_mu);
  StopServer();
  if (_mode == Mode::Shutdown) return Error::MakeP(ErrCode::ObjectShutDown, "RejectedOperation=Shutdown", "Engine is already shut down.");
  hipSetDevice(_device);
  hipStreamSynchronize(_stream);
  _mode = Mode::Shutdown;
  return Error();
}

Error HipEngine::SetStream(hipStream_t s) {
  std::lock_guard<EngineMutex> lk(_mu);
  StopServer();
  hipSetDevice(_device);
  HIP_TRY(hipStreamSynchronize(_stream));
  _stream = s ? s : _ownStream;
  return Error();
}

// Everything this engine has put on the device has finished when this returns -- the engine's stream AND the resident sweep
// kernel, which is asked to leave (it is started again by the next selection).  What a caller does before a device-wide
// synchronisation of its own: without it, hipDeviceSynchronize waits for the resident kernel's idle exit (server_idle_us).
Error HipEngine::Synchronize() {
  std::lock_guard<EngineMutex> lk(_mu);
  hipSetDevice(_device);
  StopServer();
  HIP_TRY(hipStreamSynchronize(_stream));
  _mu.busy = false;
  _pendingRecordOp = 0;
  return Error();
}

// The same guarantee without sending the resident kernel away: the engine's stream is drained and the resident sweep has finished
// the step it was given (it stays, polling for the next request).  What brackets a timed region of synchronous selections: every
// one of them has returned its result, nothing of theirs is left on the device -- and the next selection finds the kernel it would
// have found, not a relaunch.
Error HipEngine::Quiesce() {
  std::lock_guard<EngineMutex> lk(_mu);
  hipSetDevice(_device);
  { Error fe = FlushUpdates(); if (!fe.ok()) return fe; }
  HIP_TRY(hipStreamSynchronize(_stream));
  ServerQuiesce();
  _mu.busy = false;
  _pendingRecordOp = 0;
  return Error();
}

// ------------------------------------------------------------------------------------------------------------------
// bulk KB transfer / synthetic KB / gaps
// ------------------------------------------------------------------------------------------------------------------
Error HipEngine::SetKB(const double *pA, const double *pD, const double *pB) {
  std::lock_guard<EngineMutex> lk(_mu);
  StopServer();
  if (!pA || !pD || !pB) return Error::Make(ErrCode::NullArgument, "Nullptr is passed in place of a KB array.");
  hipSetDevice(_device);
  const size_t el = (size_t)_elem, rowB = (size_t)_T * el, ldB = (size_t)_ldT * el;
  // Float engines: the fp64 arrays are rounded to fp32 on the way in (a question's rows at a time through a staging buffer)
  std::vector<float> stage;
  if (_elem == 4) stage.resize((size_t)(_K + 1) * (size_t)_T);
  for (int64_t q = 0; q < _Q; q++) {
    const void *srcA = pA + (size_t)q * _K * _T, *srcD = pD + (size_t)q * _T;
    if (_elem == 4) {
      HIP_TRY(hipStreamSynchronize(_stream));   // the staging buffer is reused
      for (size_t i = 0; i < (size_t)_K * (size_t)_T; i++) stage[i] = (float)pA[(size_t)q * _K * _T + i];
      for (size_t i = 0; i < (size_t)_T; i++) stage[(size_t)_K * _T + i] = (float)pD[(size_t)q * _T + i];
      srcA = stage.data();
      srcD = stage.data() + (size_t)_K * _T;
    }
    HIP_TRY(hipMemcpy2DAsync(CubeAt(q), ldB, srcA, rowB, rowB, (size_t)_K, hipMemcpyHostToDevice, _stream));
    HIP_TRY(hipMemcpyAsync(CubeAt(q, _K), srcD, rowB, hipMemcpyHostToDevice, _stream));
  }
  std::vector<double> vb(pB, pB + _T);
  if (_elem == 4) for (double &b : vb) b = (double)(float)b;   // vB is kept as fp64 words holding the engine's number type
  HIP_TRY(hipMemcpyAsync(_dVB, vb.data(), (size_t)_T * sizeof(double), hipMemcpyHostToDevice, _stream));
  HIP_TRY(hipStreamSynchronize(_stream));
  return Error();
}

Error HipEngine::GetKB(double *pA, double *pD, double *pB) {
  std::lock_guard<EngineMutex> lk(_mu);
  hipSetDevice(_device);
  const size_t el = (size_t)_elem, rowB = (size_t)_T * el, ldB = (size_t)_ldT * el;
  std::vector<float> stage;
  if (_elem == 4) stage.resize((size_t)(_K + 1) * (size_t)_T);
  for (int64_t q = 0; q < _Q; q++) {
    if (_elem == 8) {
      if (pA) HIP_TRY(hipMemcpy2DAsync(pA + (size_t)q * _K * _T, rowB, CubeAt(q), ldB, rowB, (size_t)_K, hipMemcpyDeviceToHost, _stream));
      if (pD) HIP_TRY(hipMemcpyAsync(pD + (size_t)q * _T, CubeAt(q, _K), rowB, hipMemcpyDeviceToHost, _stream));
    } else {
      HIP_TRY(hipMemcpy2DAsync(stage.data(), rowB, CubeAt(q), ldB, rowB, (size_t)(_K + 1), hipMemcpyDeviceToHost, _stream));
      HIP_TRY(hipStreamSynchronize(_stream));
      if (pA) for (size_t i = 0; i < (size_t)_K * (size_t)_T; i++) pA[(size_t)q * _K * _T + i] = (double)stage[i];
      if (pD) for (size_t i = 0; i < (size_t)_T; i++) pD[(size_t)q * _T + i] = (double)stage[(size_t)_K * _T + i];
    }
  }
  if (pB) HIP_TRY(hipMemcpyAsync(pB, _dVB, (size_t)_T * sizeof(double), hipMemcpyDeviceToHost, _stream));
  HIP_TRY(hipStreamSynchronize(_stream));
  return Error();
}

Error HipEngine::FillSynthetic(double nTrain, double noiseAmp, uint64_t seed) {
  std::lock_guard<EngineMutex> lk(_mu);
  StopServer();
  hipSetDevice(_device);
  HIP_TRY(LaunchFillSynthetic(_dCube, _elem, _dVB, _K, _Q, _T, _ldT, _qFirst, _qTotal, _initAmount, nTrain, noiseAmp, seed, _stream));
  HIP_TRY(hipStreamSynchronize(_stream));
  return Error();
}

Error HipEngine::SetTargetGaps(int64_t n, const int64_t *ids) {
  std::lock_guard<EngineMutex> lk(_mu);
  for (int64_t i = 0; i < n; i++)
    if (ids[i] < 0 || ids[i] >= _T) return Error::MakeP(ErrCode::IndexOutOfRange, RangeParams(ids[i], 0, _T - 1), "Target index is not in KB range.");
  for (int64_t i = 0; i < n; i++)
    if (!BitTest(_hTGap, ids[i])) {
      BitSet(_hTGap, ids[i], true);
      _nTargetGaps++;
      _targetGapList.push_back(ids[i]);
      _targetIds.Vacate(ids[i]);
    }
  hipSetDevice(_device);
  return UploadGaps();
}

Error HipEngine::SetQuestionGaps(int64_t n, const int64_t *ids) {
  std::lock_guard<EngineMutex> lk(_mu);
  for (int64_t i = 0; i < n; i++)
    if (ids[i] < 0 || ids[i] >= _qTotal) return Error::MakeP(ErrCode::IndexOutOfRange, RangeParams(ids[i], 0, _qTotal - 1), "Question index is not in KB range.");
  for (int64_t i = 0; i < n; i++)
    if (ids[i] >= _qFirst && ids[i] < _qFirst + _Q && !BitTest(_hQGap, ids[i] - _qFirst)) {
      BitSet(_hQGap, ids[i] - _qFirst, true);
      _questionGapList.push_back(ids[i] - _qFirst);
      _questionIds.Vacate(ids[i] - _qFirst);
    }
  hipSetDevice(_device);
  return UploadGaps();
}

}  // namespace pqa
