// hip_engine.cpp -- see hip_engine.h.  Reference files cited are under /root/reference/ProbQA.
// HipEngine by path: this file -- errors, the logger, construction, options, the quiz registry, maintenance mode, the test hooks
// that set and read the knowledge base; hip_engine_select.cpp -- one caller's selections (single-quiz sweeps, the selectors, the
// batch ABI's sweeps, graph replay); hip_engine_server.cpp -- the resident sweep kernel; hip_engine_combine.cpp -- concurrent
// callers (posted operations, combined sweeps); hip_engine_update.cpp -- RecordAnswer and the deferred posterior updates,
// speculation, ListTopTargets, training; hip_engine_kb.cpp -- the .kb file, AddQsTs / RemoveQuestions / Compact, the id ledger;
// hip_engine_shard.cpp -- what a sharded engine asks of its shards.
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
      if (tblErr == hipSuccess) tblErr = UploadLog2TablePole(tbl.data());
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
  // the sums of the questions that pass a sweep's pole watch, and behind them their list (pole_kernels.hip; every launch leaves it empty)
  HIP_TRY(hipMalloc(&_dPoleScratch, PoleScratchBytes(_Q)));
  HIP_TRY(hipMemset(_dPoleScratch, 0, PoleScratchBytes(_Q)));
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
    hipFree(c.dSlots); hipFree(c.dScratch); hipFree(c.dPriority); hipFree(c.dPT); hipFree(c.dAcc); hipFree(c.dRecs); hipFree(c.dPriT); hipFree(c.dRerank); hipFree(c.dPole);
    if (c.hPri) hipHostFree(c.hPri);
    if (c.event) hipEventDestroy(c.event);
    if (c.h) hipHostFree(c.h);
  }
  DropQuizBufferPool();
  for (auto &g : _graphs) hipGraphExecDestroy(g.second.exec);
  hipFree(_dGraphScratch); hipFree(_dTagCell);
  for (QuizPinned *slab : _pinSlabs) hipHostFree(slab);
  hipFree(_dTGap); hipFree(_dQGap); hipFree(_dAqs); hipFree(_dTopScratch[0]); hipFree(_dTopScratch[1]);
  if (_hTopBatch) hipHostFree(_hTopBatch);
  hipFree(_dTopExact);
  if (_evSweep[0]) { hipEventDestroy(_evSweep[0]); hipEventDestroy(_evSweep[1]); }
  if (_hPinned) hipHostFree(_hPinned);
  if (_hHostPriority) hipHostFree(_hHostPriority);
  if (_ownStream) hipStreamDestroy(_ownStream);
}

Error HipEngine::UploadGaps() {
  StopServer();  // its launch arguments hold the old view
  _kbVersion++;  // every change of the KB's shape or gaps passes through here: captured graphs hold the old view
  for (Quiz *q : _quizzes) if (q) q->topOp = 0;   // ... and a listing made ahead of ListTopTargets names targets by the old gaps
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
  v.clusterForm = (int)_optClusterForm;
  v.clusterShape = (int)_optClusterShape;
  v.clusterFrom = ClusterFrom();
  v.poleScratch = _optPoleFix ? _dPoleScratch : nullptr;
  v.poleNoFollow = _optPoleFollow ? 0 : 1;
  v.poleGate = _optPoleFix && _optPoleGate ? 1 : 0;
  v.poleList = _optPoleFix ? reinterpret_cast<PoleHeader *>(_dPoleScratch + (size_t)_capQ * (size_t)(2 * _K + 2)) : nullptr;
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
  else if (n == "pole_fix") { StopServer(); (void)SettlePoleList(); _optPoleFix = value ? 1 : 0; _kbVersion++; }   // (settled while the list is still in view)   // 0: questions with a row at the pole of the lack term keep the sweep's own sums (pole_kernels.hip)
  else if (n == "late_eager") { if (value < 0 || value > 1000000) goto bad; _optLateEager = value; }
  else if (n == "top_exact") { _optTopExact = value ? 1 : 0; }   // 0: equal probabilities always by ascending target (the fast listing alone)
  else if (n == "time_sweeps") { _optTimeSweeps = value ? 1 : 0; }   // measurement hook (hip_engine_select.cpp: LaunchSingleSweep)
  else if (n == "pole_gate") { StopServer(); (void)SettlePoleList(); _optPoleGate = value ? 1 : 0; }   // 0: a fused argmax's fix redoes every listed question
  else if (n == "pole_lazy") { StopServer(); (void)SettlePoleList(); _optPoleLazy = value ? 1 : 0; }
  else if (n == "pole_follow") { StopServer(); (void)SettlePoleList(); _optPoleFollow = value ? 1 : 0; }   // measurement hook: 0 = the watching sweeps without the fix launched behind them (KbView::poleNoFollow)
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
  else if (n == "cluster_shape") { if (value < 0 || value > 2) goto bad; _optClusterShape = value; }
  else if (n == "cluster_form") { if (value < 0 || value > 2) goto bad; _optClusterForm = value; }
  else if (n == "cluster_from") { if (value < 1024 || value > 16384) goto bad; StopServer(); _optClusterFrom = value; }   // rows longer than this take the cluster sweep
  else if (n == "batch_tail") { _optBatchTail = value ? 1 : 0; }
  else if (n == "batch_groups") { if (value < 0 || value > 8) goto bad; _optBatchGroups = value; }
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
  if (n == "late_eager") return _optLateEager;
  if (n == "pole_lazy") return _optPoleLazy;
  if (n == "pole_gate") return _optPoleGate;
  if (n == "time_sweeps") return _optTimeSweeps;
  if (n == "last_sweep_ns") {   // the newest timed sweep's launch, dispatch to retirement (-1: none)
    if (!_sweepTimed || !_evSweep[1]) return -1;
    float ms = 0;
    if (hipEventSynchronize(_evSweep[1]) != hipSuccess || hipEventElapsedTime(&ms, _evSweep[0], _evSweep[1]) != hipSuccess) { (void)hipGetLastError(); return -1; }
    return (int64_t)(ms * 1e6);
  }
  if (n == "top_exact") return _optTopExact;
  if (n == "top_exact_listings") return _topExactListings;
  if (n == "pole_follow") return _optPoleFollow;
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
  if (n == "batch_groups") return _optBatchGroups;
  if (n == "batch_tail") return _optBatchTail;
  if (n == "cluster_form") return _optClusterForm;
  if (n == "cluster_from") return _optClusterFrom;
  if (n == "cluster_shape") return _optClusterShape;
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
  if (q->updatePending) {   // (its kernel works on the buffers that go back to the pool)
    (void)FlushUpdates();
    // a launch that failed has put the updates back (hip_engine_update.cpp: requeue): this quiz's entries name buffers that are
    // about to be freed or pooled -- they go, whatever happens to the others
    if (q->updatePending) {
      _pendingUpdates.erase(std::remove_if(_pendingUpdates.begin(), _pendingUpdates.end(), [q](const PendingUpdate &u) { return u.q == q; }),
                            _pendingUpdates.end());
      _pendingCount.store(_pendingUpdates.size(), std::memory_order_relaxed);
      q->updatePending = false;
    }
  }
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
  StopServer();   // (drops a speculative sweep ...)
  hipSetDevice(_device);
  { Error se = SettlePoleList(); if (!se.ok()) return se; }   // (... whose listed rows are emptied on the stream that wrote them, before it is left)
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
