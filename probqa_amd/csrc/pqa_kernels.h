// pqa_kernels.h -- launch wrappers for the CDNA4 (gfx950) kernels of the ProbQA question-evaluation hot path.
// Host code (hip_engine.cpp) only sees these plain functions; everything runs on the stream it is given.
//
// Device layout of the knowledge base (one allocation, "cube"):
//   cube[q][r][t],  q in [0,Q), r in [0,K] , t in [0,ldT)      -- fp64
//     r <  K : sA[q][r][t]   squared answer counts        (reference: PqaCore/CpuEngine.decl.h:31-37 `_sA`)
//     r == K : mD[q][t]      sum over answers of sA       (reference: `_mD`)
//   ldT = T rounded up to 16 doubles (128 B) so every row starts on a cache line and 16-byte lane loads never straddle
//   rows.  Padding columns t in [T,ldT) hold A=0, D=1 and are flagged as gaps in the target-gap bitmap.
//   vB[t] (ldT doubles) is separate.  Bitmaps are uint32 words, LSB-first, bits past the size set (gap) as in
//   PqaCore/GapTracker.h:9-15; the "asked" bitmap of a quiz has bits past Q clear.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

namespace pqa {

// What a launch wrapper has asked the runtime about one kernel instantiation, PER DEVICE: the opt-in to more than 64 KiB of
// dynamic LDS (hipFuncSetAttribute) belongs to the device that was current when it was made, and one process drives
// several devices (sharded_engine.cpp).  One function-local static per instantiation; engines on different threads may
// race for a slot -- they would store the same value.
struct LaunchCache {
  static constexpr int kDevices = 64;
  std::atomic<uint64_t> slot[kDevices];   // (LDS bytes << 16) | (workgroups per CU + 1); 0 = nothing asked yet
  std::atomic<int> numCUs[kDevices];
  static int Device() { int d = 0; return hipGetDevice(&d) == hipSuccess ? (d & (kDevices - 1)) : 0; }
  bool Get(int dev, size_t shmem, int *perCU) const {
    const uint64_t v = slot[dev].load(std::memory_order_acquire);
    if (v == 0 || (v >> 16) != (uint64_t)shmem) return false;
    *perCU = (int)(v & 0xFFFF) - 1;
    return true;
  }
  void Put(int dev, size_t shmem, int perCU) { slot[dev].store(((uint64_t)shmem << 16) | (uint64_t)(perCU + 1), std::memory_order_release); }
  int NumCUs(int dev) {
    int n = numCUs[dev].load(std::memory_order_relaxed);
    if (n == 0) {
      int d = 0;
      n = (hipGetDevice(&d) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d) == hipSuccess && n > 0) ? n : 256;
      numCUs[dev].store(n, std::memory_order_relaxed);
    }
    return n;
  }
};

struct KbView {
  const void *cube;       // [Q][K+1][ldT], elements of `elem` bytes: double (Double engines) or float (Float engines)
  int elem;               // 8 | 4
  const double *vB;       // [ldT]
  const uint32_t *tgap;   // target gap bits, ldT bits (+ slack), bits >= T set
  const uint32_t *qgap;   // question gap bits, bits >= Q set
  int64_t K, Q, T, ldT;
  int64_t nValidTargets;  // T - #target gaps (PqaCore/CpuEngine.cpp:352)
  int smallLaunches;      // the engine runs the resident sweep: posterior kernels over <= 1024 targets use 256 threads
  double *priorScratch;   // 8 * kMaxWorkers + 2 doubles (device): the subtasks' sums of the long-row posterior kernels (prior_kernels.hip); may be null
  int maxGrid;            // test hook (engine option "eval_max_grid"): cap the workgroups of a sweep, so that a small cube makes
                          // every workgroup stream dozens of questions; 0 = no cap
  int clusterForm;        // long rows (cluster_kernels.hip): 0 = default, 1 = question by question, 2 = pass 1 a question ahead (option cluster_form)
  int clusterShape;       // ... the shape of the form that runs ahead: 0 = default, 1 = 512 threads x 1 unit, 2 = 256 x 2 (option cluster_shape)
  int64_t clusterFrom;    // rows LONGER than this many elements take the cluster sweep (engine option cluster_from)
  double *poleScratch;    // Q x (2 K + 2) doubles (device): the sums of questions with a row at the pole of the lack term, between the
                          // sweep and the fix launched behind it (pole_kernels.hip); may be null (then such questions keep the sweep's own sums)
  struct PoleHeader *poleList;   // ... and the list of those questions: PoleListBytes(Q) bytes, zeroed once (every launch leaves it empty)
  int poleNoFollow;       // measurement hook (engine option "pole_follow" = 0): the watching sweep WITHOUT the launch behind it -- for timing the sweep
                          // kernel by itself in a quiz state that lists nothing; anything listed would stay listed
  int poleGate;           // engine option "pole_gate" (default 1): where only the ARGMAX leaves the engine (a fused single-quiz argmax), the fix
                          // redoes only the listed questions that can still win (pole_kernels.hip: pole_bounds_kernel) -- the register-shape
                          // sweeps then track, per listed question, how close to 1 its largest posterior element can be
};

// ---- questions with a row at the pole of the lack term: listed by the sweeps, redone in the reference's order behind them
// (pole_kernels.hip).  A list is a header and `capacity` entries; a sweep appends at most one entry per question (and quiz).
struct PoleHeader { uint32_t count, arrived; unsigned long long floorBits; };   // floorBits: see PoleFix::gate
struct PoleEntry {
  uint32_t q;             // position in the priority vector (question qFirst + q of the cube)
  uint32_t rowMask;       // the answer rows that passed the sweep's watch, a bit each (0: not known -- every row is redone)
  uint32_t b;             // the quiz (batched sweeps; 0 otherwise)
  uint32_t gap;           // float bits.  From a sweep that tracks it (KbView::poleGate): a lower bound of 1 - p for the largest posterior
                          // element of the question's listed rows (0: not known); then, from pole_bounds_kernel: how far the fix can move
                          // the question's priority, relative (+inf: not known)
};
static_assert(sizeof(PoleHeader) == 16, "header and entries are 16 bytes each");
size_t PoleListBytes(int64_t capacity);
hipError_t UploadLog2TablePole(const double *hostTable);

struct SelectResult {     // 16 bytes, written by the select kernels
  double priority;        // argmax: winning priority; sampled: grand total of priorities
  int64_t index;          // selected question (global index) or -1
};

// Upload the Log2Hot table (1024 x {log2 midpoint, 1/(2 midpoint)}, built by the host); once per process and device.
hipError_t UploadLog2Table(const double *hostTable);
// out[i] = the device log2hot(x[i]) (device pointers); test hook for the function the sweep applies per element.
hipError_t LaunchLog2HotArray(const double *x, double *out, int64_t n, hipStream_t stream);

// ---- a1: priority sweep.  priority[q - qFirst] for q in [qFirst,qLimit); 0 for gap / asked questions.
// Returns hipSuccess or the launch error.  `variant`: 0 = auto, otherwise forces a kernel shape (tests/bench).
// `fused` (optional): let the sweep's last workgroup also pick the argmax, so that a selection is one launch.
constexpr int kFusedMaxGrid = 4096;     // workgroups of a fused launch (one record each)
struct alignas(16) TaggedPriority { double priority; uint64_t tag; };

struct FusedSelect {
  SelectResult *scratch;  // records (16-byte aligned): the workgroups' winners, tagged per launch; kFusedMaxGrid of them
                          // for a single sweep, scratchStride per quiz for a batch
  SelectResult *out;      // device or host-coherent memory; index = position in priority[] + outBase
  uint64_t *seq;          // optional host-coherent flag, set to flagValue after `out` is visible
  uint64_t seqValue;      // launch tag: its low 32 bits must differ from those of the previous fused launch on `scratch`
  int64_t outBase;
  int64_t scratchStride;  // batch only: records per quiz in `scratch` (the launch uses at most that many workgroups)
  uint64_t flagValue;     // what *seq receives (the engine's own callers pass the launch tag)
  // Launches replayed from a HIP graph have constant arguments: with tagCell != nullptr both the launch tag and the flag
  // value are read from this device word instead, and the finisher advances it for the next replay.
  uint64_t *tagCell;
  // The reference's sampled selector instead of the argmax, in the same launch: once the finisher has seen every
  // workgroup's record, workgroup 0 runs the selection (select_sampled_wg_impl, pqa_device.h) over the priority vector and
  // writes {sum of priorities, selected position} to `out`.  sampleSubtasks > 0 switches it on; priorities, run lengths and
  // totals are staged in the LDS that held the Log2Hot table (EvalVariantFusesSampled says whether they fit).
  int64_t sampleSubtasks;
  uint64_t sampleRnd;
  double *runLength;
  // ... or, with hostPriority != nullptr (and sampleSubtasks > 0): the host runs the selector (O(Q) scalar Kahan steps,
  // microseconds) instead of a second kernel whose dispatch alone costs more.  Every workgroup stores the priorities of its own
  // questions into this host-coherent array as 16-byte records {priority, launch tag} (one store each, nothing waited for); the
  // finisher raises the flag once it has seen every workgroup's record; the host takes an entry when it carries the launch's tag
  // (it almost always does by then; a straggling store is waited for there, not by two thousand waves on the device).
  TaggedPriority *hostPriority;
  // A synchronous caller that waits on `seq` right behind its launch may ask for the fix of pole_kernels.hip only when the sweep
  // has listed something (lazyFix != 0; argmax and hostPriority forms): nothing is launched behind the sweep, and a finisher
  // that saw suspects publishes index -4 instead of leaving the publication to the fix; the caller then launches
  // LaunchEvalPoleFixup (same arguments, a new flagValue) and waits again.  A fresh quiz's selections so cost one launch, not two.
  int64_t lazyFix;
};
// One quiz of a batched sweep (blockIdx.y selects it): everything that differs between the quizzes of one launch.
struct QuizSlot {
  const double *prior;
  const uint32_t *asked;
  double *priority;       // this quiz's priority vector
  SelectResult *out;      // host-coherent record of this quiz's winner
  uint64_t *seq;          // host-coherent flag of this quiz
  TaggedPriority *hostPriority;   // optional (grid.y = quiz launches with FusedSelect::sampleSubtasks > 0): this quiz's priorities go to
                                  // the host as tagged records, as FusedSelect::hostPriority delivers a single quiz's
};
hipError_t LaunchEvalQuestions(const KbView &kb, const double *prior, const uint32_t *asked, int64_t qFirst,
                               int64_t qLimit, double *priority, int variant, const FusedSelect *fused,
                               hipStream_t stream);
// the fix behind a sweep launched with FusedSelect::lazyFix whose answer was -4 (the sweep's own arguments; fused.flagValue: what the caller waits for now)
hipError_t LaunchEvalPoleFixup(const KbView &kb, const double *prior, const uint32_t *asked, double *priority, const FusedSelect &fused,
                               hipStream_t stream);
// The fix behind a watching sweep.  A record of sums is `sumsStride` doubles: W_k [K] at wOff, W_k sqrt(V_k) (secondIsWV) or V_k [K]
// at vOff, sum l log2 p at hOff, the lack sum at lOff; record of entry e: e itself (bySlot) or its question.  priority (or priorityT
// [q][Bp], or the quizzes' own vectors: slots) receives the corrected priorities -- all null: the records are corrected in place
// and the caller's epilogue follows.  fs: the sweep's own fused selection; when the list is not empty the sweep's finisher has left
// its publication to this launch.
struct PoleFix {
  const double *cube;
  const uint32_t *tgap, *qgap, *asked;
  const double *prior;            // single quiz
  const QuizSlot *slots;          // batched sweeps: entry.b selects the quiz
  int nSlots;                     // ... of a grid.y = quiz launch (fs set): this launch publishes every quiz's result when it has work
  PoleHeader *list;
  uint32_t *maskDense;            // optional [nQ]: the rows per question where several workgroups watch one question (cleared here)
  uint32_t *dirty;                // optional [quizzes]: set to 1 for every quiz with a corrected priority (batched sweeps: the pick reads it)
  double *sums;
  int64_t sumsStride;
  int bySlot, wOff, vOff, hOff, lOff, secondIsWV;
  double *priority, *priorityT;
  int Bp;
  TaggedPriority *hostPriority;   // optional: the corrected priorities also go to the host as {priority, hostTag} records
  uint64_t hostTag;
  int64_t K, T, ldT, qFirst, nQ, capacity;
  double vCompTail;
  FusedSelect fs;
  int rows, waveLds;              // (set by the launcher)
  // gate: only the argmax leaves the engine (fs names a fused argmax of ONE quiz, `priority` holds the sweep's vector, the entries carry
  // the sweep's gaps): a kernel ahead of the fix bounds, per listed question, how far the fix can move its priority, and the fix skips
  // the questions whose upper bound stays below the best lower bound (PoleHeader::floorBits) -- they cannot be the maximum.
  int gate;
};
hipError_t LaunchPoleFixup(const PoleFix &fix, hipStream_t stream);
// The same sweep for nSlots quizzes in one launch (grid.y = quiz): `slots` is a DEVICE array; fused->scratch holds
// nSlots * fused->scratchStride records; fused->out / seq are ignored (each slot has its own).
// pole (optional): EvalBatchPoleBytes(kb, nSlots) bytes of device memory, the first kBatchPoleClear cleared once after allocation -- the
// batch's suspect list and records (pole_kernels.hip); without it the quizzes of the launch keep the sweep's own sums at the pole.
size_t EvalBatchPoleBytes(const KbView &kb, int nSlots);
hipError_t LaunchEvalQuestionsBatch(const KbView &kb, const QuizSlot *slots, int nSlots, int64_t qFirst, int64_t qLimit,
                                    int variant, const FusedSelect &fused, hipStream_t stream, void *pole = nullptr);
// ---- many quizzes per sweep, one cube read per batch (batch_kernels.hip): lane = quiz, the cube tile staged in LDS is shared by
// all quizzes of the batch.  Double and Float engines.  nSlots <= 256.
constexpr int kBatchMaxGrid = 2048;
struct BatchRecord { double priority; int64_t index; };   // a workgroup's best question of one quiz (index < 0: none)
struct BatchPlan {
  int tileTargets;        // in: targets per LDS tile (0 = default)
  int questionsPerBlock;  // in: questions staged together (0 = default: 4 fp32, 2 fp64; else the largest built shape <= this)
  int splitTail;          // in: 1 = the questions of the last, partial round of the persistent grid go to a second launch with fewer questions per group (LaunchEvalBatch)
  int questionGroups;     // in: question groups side by side in a workgroup of a small batch (0 = automatic; see eval_batch_kernel)
  int grid, Bp;           // out: workgroups of the sweep; quizzes rounded up to whole waves
  size_t ptBytes, accBytes, recBytes;   // out: sizes of the scratch buffers PT / acc / recs the caller provides
  size_t poleBytes;       // out: ... and of `pole` (0: the sweep does not watch for rows at the pole of the lack term -- pole_kernels.hip)
  void *pole;             // in: poleBytes of device memory whose first kBatchPoleClear bytes were cleared once after allocation; the
                          //     caller then also provides priorityT (the fix corrects the priority matrix, the pick reads it)
};
constexpr size_t kBatchPoleClear = 1024 + 16;
// queryOnly: only fill `plan`.  Otherwise: transposed masked priors -> PT, the sweep, and every quiz's winner {priority,
// local index + outBase} to its slot's `out`, then flagValue to its `seq` (host-coherent).  priorityT (optional):
// [Q][plan->Bp] priorities, quiz-minor.
hipError_t LaunchEvalBatch(const KbView &kb, const QuizSlot *slots, int nSlots, BatchPlan *plan, void *PT, double *acc,
                           BatchRecord *recs, double *priorityT, int64_t outBase, uint64_t flagValue, bool queryOnly, hipStream_t stream,
                           bool skipPick = false);   // skipPick: the caller picks (LaunchBatchRerank)
// The sweep for a few dozen quizzes (batch_kernels.hip: eval_midbatch_kernel; Double engines, K == 5, short rows): a lane is a
// (quiz, chunk of the row).  plan / PT / recs / priorityT / flags as LaunchEvalBatch; slots with hostPriority get tagged records.
bool EvalMidBatchSupported(const KbView &kb);
hipError_t LaunchEvalMidBatch(const KbView &kb, const QuizSlot *slots, int nSlots, BatchPlan *plan, void *PT, BatchRecord *recs,
                              double *priorityT, int64_t outBase, uint64_t flagValue, bool queryOnly, hipStream_t stream);
// skipPick (LaunchEvalBatch's flagValue == 0 is not used for this: see the argument): Float engines' batched ARGMAX -- the fp32
// sweep nominates, fp64 decides (eval_kernels.hip: batch_rerank_kernel).  priorityT: the sweep's [Q][Bp] matrix; scratch:
// BatchRerankScratchBytes() of device memory.  Writes every quiz's winner and flag like LaunchEvalBatch's own pick.
size_t BatchRerankScratchBytes();
hipError_t LaunchBatchRerank(const KbView &kb, const QuizSlot *slots, int nSlots, int Bp, const double *priorityT, void *scratch,
                             int64_t outBase, uint64_t flagValue, hipStream_t stream);
// Single-quiz sweep of a Float engine: priority[q] for every local question (0 for gap / asked).
hipError_t LaunchEvalQuestionsF32(const KbView &kb, const double *prior, const uint32_t *asked, double *priority, hipStream_t stream);
// ... with the question's rows held in registers (eval_f32_kernels.hip): rows of up to 16384 targets, up to 16 answers
bool EvalF32RegisterShape(const KbView &kb, int variant);
const char *EvalF32KernelName(const KbView &kb, int variant);
hipError_t LaunchEvalQuestionsF32Reg(const KbView &kb, const double *prior, const uint32_t *asked, double *priority, int variant,
                                     hipStream_t stream);
hipError_t UploadLog2TableBatch(const double *hostTable);
// The single-quiz sweep for rows beyond the register shapes (ldT > 16384), fp32 and fp64: a question split over a cluster of
// workgroups (cluster_kernels.hip).  `scratch`: EvalClusterScratchBytes(kb) bytes of device memory (exchange + totals).
hipError_t UploadLog2TableCluster(const double *hostTable);
bool EvalClusterSupported(const KbView &kb);
const char *EvalClusterKernelName(const KbView &kb);
size_t EvalClusterScratchBytes(const KbView &kb);
hipError_t LaunchEvalCluster(const KbView &kb, const double *prior, const uint32_t *asked, double *priority, void *scratch, hipStream_t stream);
const char *EvalVariantName(const KbView &kb, int variant);
bool EvalVariantFusesSampled(const KbView &kb, int variant, int64_t nSubtasks);   // the launch can run the sampled selector itself
bool EvalVariantHasFinisherWorkgroup(const KbView &kb, int variant);              // ... or hand the priority vector to the host (hostPriority)

// ---- resident sweep ("server"): ONE launch serves many selections.  The host posts a request in pinned memory; workgroup
// 0 sees it, hands it to the other workgroups through a device word, everybody sweeps, the finisher answers straight into
// host-coherent memory.  What a step saves is the launch + dispatch + ramp of one kernel (~8 us of a 26 us selection at
// 1000 x 5 x 1000).  The kernel leaves by itself after idleTicks (100 MHz) without a request, or when asked to.
struct ServerMailbox {            // host-coherent (pinned) memory, 128 bytes
  // host -> device; the request fields are written before `req`
  uint64_t req;                   // sequence number of the newest request (never 0 / ~0)
  const double *prior;            // the quiz
  const uint32_t *asked;
  SelectResult *out;              // where the finisher writes {priority, index + outBase}
  uint64_t *flag;                 // ... and then flagValue
  uint64_t flagValue;
  int64_t outBase;
  uint64_t stop;                  // non-zero: leave now
  // device -> host
  uint64_t state;                 // kServerRunning / kServerExiting / kServerExited
  uint64_t taken;                 // newest request the kernel has started on
  uint64_t done;                  // newest request whose step has finished (its result was published before)
  uint64_t pad[5];
};
constexpr uint64_t kServerRunning = 1, kServerExiting = 2, kServerExited = 3;
struct ServerCtl {                // device memory, one line; zeroed before each launch: workgroup 0 -> the other workgroups
  uint64_t go;                    // newest request (the line's fields belong to it); ~0: leave
  const double *prior;
  const uint32_t *asked;
  SelectResult *out;
  uint64_t *flag;
  uint64_t flagValue;
  int64_t outBase;
  uint64_t pad;
};
// Only the short-row register shapes have a resident form (that is where a launch is a large part of a selection):
// returns hipErrorNotSupported otherwise.  `scratch`: kFusedMaxGrid records.  lastSeq: the kernel serves requests != lastSeq.
// requestLine: the 64-byte line the host writes requests to -- the mailbox's own first line, or a line of host-visible device
// memory (everyonePolls: every workgroup watches it; the host then waits for `done`, not `taken`, before the next request).
// hostPriority (optional, host-coherent, qLimit - qFirst doubles): a request whose outBase carries kServerHandOver is answered
// with the priority vector itself (for the host's sampled selector) instead of the argmax.
constexpr uint64_t kServerHandOver = 1ull << 62;
// watch: the resident sweep watches for rows at the pole of the lack term (pole_kernels.hip) and answers
// a step that found one with index -4 -- the caller then takes the launched path, behind which the fix can run; a request whose outBase
// carries kServerNoWatch is answered as if nothing had been found.
constexpr uint64_t kServerNoWatch = 1ull << 61;
hipError_t LaunchEvalServer(const KbView &kb, int64_t qFirst, int64_t qLimit, double *priority, int variant,
                            SelectResult *scratch, ServerMailbox *mailbox, void *requestLine, bool everyonePolls,
                            ServerCtl *ctl, uint64_t lastSeq, uint64_t idleTicks, TaggedPriority *hostPriority, hipStream_t stream,
                            bool watch = false);
bool EvalServerSupported(const KbView &kb, int variant);


struct RatedTargetDev { int64_t iTarget; double prob; };  // == CiRatedTarget
// RecordAnswer's posterior update in the prologue of the sweep that follows it (eval_kernels.hip: eval_questions_f64_upd): one
// launch does what LaunchRecordAnswer + LaunchEvalQuestions do (same posterior bits, same priorities), and the sweep does not wait
// for a posterior kernel.  `fused` must name a selection (scratch != nullptr); the listing arguments as LaunchRecordAnswer's.
bool EvalFusesUpdate(const KbView &kb, int variant, int64_t nWorkers);
hipError_t LaunchEvalQuestionsWithUpdate(const KbView &kb, double *prior, uint32_t *asked, double *priority, int variant, const FusedSelect &fused,
                                         int64_t iQuestion, int64_t iAnswer, int64_t nWorkers, RatedTargetDev *topOut, int64_t *topN,
                                         uint64_t *topFlag, uint64_t topFlagValue, int64_t topCount, hipStream_t stream);

// ---- selectors over priority[0..n) (questions qFirst..qFirst+n of the bitmaps); the reported index is
// (position in priority[]) + outBase
hipError_t LaunchSelectArgmax(const double *priority, const uint32_t *qgap, const uint32_t *asked, int64_t qFirst,
                              int64_t n, int64_t outBase, SelectResult *out, uint64_t *flag, uint64_t flagValue,
                              hipStream_t stream);   // flag (optional, host-coherent): receives flagValue after `out`
// Reference selector (PqaCore/CpuEngine.cpp:362-400): per-subtask Kahan run lengths, grand totals, upper_bound.
// runLength: scratch of n doubles. rnd: the 64-bit random number the reference would draw.
hipError_t LaunchSelectSampled(const double *priority, const uint32_t *qgap, const uint32_t *asked, int64_t qFirst,
                               int64_t n, int64_t nSubtasks, uint64_t rnd, double *runLength, SelectResult *out,
                               uint64_t *flag, uint64_t flagValue, hipStream_t stream);

// ---- prior updates (single workgroup, O(T)); nWorkers = emulated CPU worker count that fixes the summation order.
// The subtasks' partial sums live in (8 nWorkers + 1) doubles of LDS, within the 64 KiB a launch gets without opting in.
constexpr int64_t kMaxWorkers = 1000;
hipError_t LaunchStartQuiz(const KbView &kb, double *prior, uint32_t *asked, int64_t askedWords, int64_t nWorkers, hipStream_t stream);
// topOut (optional, host-coherent with topN / topFlag): also list the new posterior's topCount best targets, then store
// topFlagValue to *topFlag.
// rowA / rowD (optional): the rows of an answered question that ANOTHER shard of the question axis holds (sharded_engine.cpp), read
// where they are; iQuestion is then not used and no bit of `asked` is set.
hipError_t LaunchRecordAnswer(const KbView &kb, double *prior, uint32_t *asked, int64_t iQuestion, int64_t iAnswer,
                              int64_t nWorkers, RatedTargetDev *topOut, int64_t *topN, uint64_t *topFlag,
                              uint64_t topFlagValue, int64_t topCount, hipStream_t stream, const void *rowA = nullptr,
                              const void *rowD = nullptr);
// Several quizzes' RecordAnswer in ONE launch (grid.x = update; the same workgroup code and summation order per quiz as
// LaunchRecordAnswer, so every posterior is bit-identical to the one-by-one result): up to kRecordInline updates travel in the
// kernel's arguments.  CERecordAnswerSubtaskMul.cpp:15-42 per quiz; the reference runs concurrent quizzes' updates side by side.
constexpr int kRecordInline = 256;
struct RecordSlot {               // 56 bytes: 256 of them are 14 KB of kernel arguments
  double *prior;
  uint32_t *asked;
  void *pin;                      // the quiz's host-coherent lines {RatedTargetDev top[kQuizTopDev]; int64 nOut; uint64 topFlag} (optional: no listing without)
  int32_t iQuestion, iAnswer;     // local question
  uint64_t topFlagValue;
  const void *rowA, *rowD;        // non-null: another shard's question -- its rows where they are (then iQuestion is unused)
};
constexpr int kQuizTopDev = 32;   // == kQuizTop (hip_engine.h)
struct RecordBatchInline {
  int32_t n, topCount;
  RecordSlot s[kRecordInline];
};
hipError_t LaunchRecordAnswerBatch(const KbView &kb, const RecordBatchInline &batch, int64_t nWorkers, hipStream_t stream);
// Several quizzes' StartQuiz in ONE launch (grid.x = quiz; every workgroup runs CESetPriorsSubtaskSum exactly as LaunchStartQuiz).
constexpr int kStartInline = 256;
struct StartBatchInline {
  int32_t n;
  int64_t askedWords;
  double *prior[kStartInline];
  uint32_t *asked[kStartInline];
};
hipError_t LaunchStartQuizBatch(const KbView &kb, const StartBatchInline &batch, int64_t nWorkers, hipStream_t stream);
// rows: device array of 2 nAnswered row pointers, {sA[q_i][a_i], mD[q_i]} per answered question (rows of kb.elem-byte elements,
// ldT long; they may live on another device of the process).  exps: scratch of ldT int64.  status: device int64[2]
// {error code (0 / 16 = I64Underflow), fullMax}.  bugCompat reproduces PqaCore/CEUpdatePriorsSubtaskMul.cpp:53.
hipError_t LaunchResumeQuiz(const KbView &kb, double *prior, int64_t *exps, const void *const *rows, int64_t nAnswered,
                            int64_t nWorkers, int bugCompat, int64_t *status, hipStream_t stream);

// ---- KB construction / mutation
hipError_t LaunchFillFresh(void *cube, int elem, double *vB, int64_t K, int64_t Q, int64_t T, int64_t ldT, double initAmount,
                           hipStream_t stream);
// Deterministic synthetic "binary-search trained + hash noise" cube (see probqa_amd/synth.py for the definition).
hipError_t LaunchFillSynthetic(void *cube, int elem, double *vB, int64_t K, int64_t Q, int64_t T, int64_t ldT, int64_t qOffset,
                               int64_t qTotal, double initAmount, double nTrain, double noiseAmp, uint64_t seed,
                               hipStream_t stream);
// Train / RecordQuizTarget (PqaCore/CETrainOperation.cpp:15-83).  steps: device array in execution order, grouped by question;
// chain c = steps[chainStart[c] .. chainStart[c + 1]) all on one question (chainStart: nChains + 1 entries).  Always adds
// `amount` to vB[iTarget], also with no steps.
struct TrainStep { int64_t kind, q, a1, a2; };   // kind 1 | 2 | 3, see kb_kernels.hip; q local
hipError_t LaunchTrainSteps(void *cube, int elem, double *vB, int64_t K, int64_t ldT, const TrainStep *steps,
                            const int64_t *chainStart, int64_t nChains, int64_t iTarget, double amount, hipStream_t stream);
// Up to kTrainInlineSteps steps travel in the kernel's arguments (no staging copy, nothing for the host to wait for).
constexpr int kTrainInlineSteps = 48;
struct TrainStepsInline {
  int64_t nChains;
  int64_t chainStart[kTrainInlineSteps + 1];
  TrainStep steps[kTrainInlineSteps];
};
hipError_t LaunchTrainStepsInline(void *cube, int elem, double *vB, int64_t K, int64_t ldT, const TrainStepsInline &in,
                                  int64_t iTarget, double amount, hipStream_t stream);
// Several training calls -- each with its own target and amount, DIFFERENT targets (their cells are disjoint then) -- in one launch:
// workgroup c runs call c's chains exactly as train_steps_inline_kernel would.  What a server's clients' RecordQuizTarget calls
// become when they arrive together (hip_engine_combine.cpp: DrainPosted).
constexpr int kTrainBatchCalls = 24, kTrainBatchSteps = 192;
struct TrainBatchStep { int32_t q; uint8_t kind, a1, a2, pad; };
struct TrainBatchCall { int64_t iTarget; double amount; int32_t firstChain, nChains; };
struct TrainBatchInline {
  int32_t nCalls, nChainsTotal, nSteps, pad;
  TrainBatchCall calls[kTrainBatchCalls];
  uint16_t chainStart[kTrainBatchSteps + kTrainBatchCalls + 8];   // per call: its chains' starts and one end, indices into steps[]
  TrainBatchStep steps[kTrainBatchSteps];
};
hipError_t LaunchTrainBatchInline(void *cube, int elem, double *vB, int64_t K, int64_t ldT, const TrainBatchInline &in, hipStream_t stream);
// Maintenance (PqaCore/CpuEngine.cpp:468-658): (re)initialise whole questions / whole target columns; compact the target
// axis with (src,dst) column moves.  qs/ts/inits/moves are device arrays.
hipError_t LaunchFillQuestions(void *cube, int elem, int64_t K, int64_t T, int64_t ldT, const int64_t *qs, const double *inits,
                               int64_t n, hipStream_t stream);
hipError_t LaunchFillTargets(void *cube, int elem, double *vB, int64_t K, int64_t ldT, int64_t nQ, const uint32_t *skipQ,
                             const int64_t *ts, const double *inits, int64_t n, hipStream_t stream);
// src: device array of nQ block pointers (question q's K + 1 rows, ldTs apart; nullptr: nothing to take), colMap: device array of
// Tn old column indices (-1: a new column).  vB likewise.
hipError_t LaunchAdoptRows(void *dst, int elem, double *dstVB, int64_t K, int64_t nQ, int64_t Tn, int64_t ldTn, const void *const *src,
                           int64_t ldTs, const double *srcVB, const int64_t *colMap, hipStream_t stream);
hipError_t LaunchMoveTargets(void *cube, int elem, double *vB, int64_t K, int64_t ldT, int64_t nQ, const int64_t *moves,
                             int64_t n, hipStream_t stream);
// ListTopTargets (PqaCore/CEListTopTargetsAlgorithm.cpp): top maxCount (prob,target) pairs, descending, gaps skipped.
// (flag != nullptr: out / nOut / flag are host-coherent; the kernel stores flagValue to *flag after its results)
hipError_t LaunchTopTargets(const KbView &kb, const double *prior, int64_t maxCount, RatedTargetDev *out,
                            int64_t *nOut, uint64_t *flag, uint64_t flagValue, hipStream_t stream);
// ... over rows of any length and for up to kTopBatchQuizzes quizzes per launch (maxCount <= 256): a list per wave of 1024 targets,
// then merges; out[quiz][maxCount] records (unused ones {-1, -1}) and nOut[quiz], device or host-coherent; `flag` only with ONE quiz.
// The two scratch buffers hold nQuizzes * TopBatchScratchRecords(T, maxCount) records each.
constexpr int kTopBatchQuizzes = 256;
constexpr int64_t kTopChunkTargets = 4096, kTopMergeCapacity = 16384;
struct TopBatchPriors { const double *prior[kTopBatchQuizzes]; };
int64_t TopBatchScratchRecords(int64_t T, int64_t maxCount);
// ... and in the reference's order among EQUAL probabilities (its per-worker heaps and head heap, reproduced step for step; nWorkers:
// the emulated thread count): what ListTopTargets returns where the listing above shows a tie.  `scratch`: TopExactScratchBytes bytes.
size_t TopExactScratchBytes(int64_t T, int64_t nWorkers, int64_t maxCount, int64_t nQuizzes);
hipError_t LaunchTopTargetsExact(const KbView &kb, const TopBatchPriors &priors, int64_t nQuizzes, int64_t nWorkers, int64_t maxCount, void *scratch,
                                 RatedTargetDev *out, int64_t *nOut, uint64_t *flag, uint64_t flagValue, hipStream_t stream);
hipError_t LaunchTopTargetsBatch(const KbView &kb, const TopBatchPriors &priors, int64_t nQuizzes, int64_t maxCount, RatedTargetDev *scratchA,
                                 RatedTargetDev *scratchB, RatedTargetDev *out, int64_t *nOut, uint64_t *flag, uint64_t flagValue,
                                 hipStream_t stream);

}  // namespace pqa
