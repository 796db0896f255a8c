/*
 * PqaHipExt.h -- additive exports of libPqaCore.so (MI355X build).  Nothing here exists in the reference; unchanged
 * wrappers never need it.  It exposes (a) the deterministic outputs of the hot path, which the reference hides
 * behind a random draw (PqaCore/CpuEngine.cpp:379), so that parity can be checked; (b) bulk KB transfer for tests and
 * benchmarks; (c) stream-ordered (no host sync) entry points and question-axis sharding for multi-GPU hosts.
 * Same conventions as PqaCInterop.h: void* returns are NULL or a PqaError*.
 */
#ifndef PQA_HIP_EXT_H
#define PQA_HIP_EXT_H

#include "PqaCInterop.h"

#pragma pack(push, 8)
typedef struct {
  int64_t _qFirst;   /* first GLOBAL question index held by this engine */
  int64_t _qTotal;   /* global number of questions (== _nQuestions of the definition when unsharded) */
  int32_t _device;   /* HIP device ordinal, -1 = current device */
  int32_t _reserved;
} CiHipShard;

typedef struct {     /* result of a stream-ordered selection; lives in device or pinned host memory */
  double _priority;
  int64_t _iQuestion; /* GLOBAL question index, -1 if no eligible question in this shard */
} CiHipSelection;
#pragma pack(pop)

#ifdef __cplusplus
extern "C" {
#endif

/* Same as PqaEngineFactory_CreateCpuEngine (which creates the HIP engine too); explicit name for new callers. */
PQACORE_API void *PqaEngineFactory_CreateHipEngine(void *pvFactory, void **ppError, const CiEngineDefinition *pEngDef);
/* Same as PqaEngineFactory_LoadCpuEngine (which loads into the HIP engine too); explicit name for new callers. */
PQACORE_API void *PqaEngineFactory_LoadHipEngine(void *pvFactory, void **ppError, const char *filePath, uint64_t memPoolMaxBytes);
/* Engine over questions [_qFirst, _qFirst + pEngDef->_nQuestions) of a KB with _qTotal questions.  Question ids in
 * every call on such an engine are GLOBAL ids. */
PQACORE_API void *PqaEngineFactory_CreateHipEngineSharded(void *pvFactory, void **ppError,
                                                          const CiEngineDefinition *pEngDef, const CiHipShard *pShard);

/* ---- options: "select" (0 = sampled like the reference [default], 1 = argmax), "workers" (emulated CPU worker count
 * fixing the summation order of the prior updates, default 16), "eval_subtasks" (question subtasks of the sampled
 * selector, default 8*workers as PqaCore/CpuEngine.cpp:339), "eval_variant" (0 = auto), "bug_compat" (reproduce
 * PqaCore/CEUpdatePriorsSubtaskMul.cpp:53), "seed" (selector RNG seed), "use_graph" (argmax NextQuestion replays a per-quiz HIP graph
 * instead of launching the sweep), "top_cache" (how many of the new posterior's best
 * targets RecordAnswer's kernel lists at most ahead of the ListTopTargets call that follows it -- it lists as many as ListTopTargets
 * has been asked for lately; default 10, 0 = none), "server"
 * (argmax selections are served by a resident kernel instead of one launch each -- rows up to 1024 targets; default 0),
 * "server_idle_us" (that kernel leaves after this long without a request, default 500: what a device-wide synchronisation of the host waits at most -- PqaHip_Synchronize asks it to leave at once), "server_vram_mailbox" (requests
 * are written to host-visible device memory where the platform maps it, default 1; set before the first selection).
 * "host_sampled" (the sampled NextQuestion as one launch whose finisher hands the priority vector to the host, which runs the
 * reference's selector itself; default 1 -- 0: sweep + selector kernel), "fused_sampled" (the selector inside the sweep's launch;
 * default 0: measured slower),
 * "speculate" (StartQuiz / ResumeQuiz / RecordAnswer launch the sweep of the NextQuestion that normally follows them, which then only
 * waits for its result; same questions either way; default 1, also PQA_SPECULATE; read-only "spec_hits" / "spec_dropped" count the
 * speculative sweeps that were used / dropped),
 * "fuse_update" (RecordAnswer's posterior update runs inside the launch of that speculative sweep where its shape allows it -- rows
 * of up to 1024 targets -- instead of in a kernel of its own ahead of it; same bits; default 1; read-only "fused_updates"),
 * "combine" (concurrent client threads: their NextQuestion calls share batched sweeps, their RecordAnswer / StartQuiz /
 * RecordQuizTarget calls share launches, and a call that finds the engine taken posts its operation to the thread inside instead
 * of queueing on the lock; default 1, also PQA_COMBINE; "combine_linger_us": how long a leader / a ListTopTargets waits for the
 * other clients' requests, default 20; read-only "combined_batches", "combined_requests", "combined_max_batch", "update_flushes",
 * "updates_flushed", "update_max_flush", "posted_ops", "posted_drains", "train_batches", "train_batch_calls"),
 * "long_row_form" (StartQuiz / RecordAnswer over rows beyond 16384 targets as one workgroup per subtask of the reference's sum
 * plus a division launch; default 1), "post_always" (test hook: the posted form of the quiz-level calls even when the engine is free),
 * "eval_max_grid" (test hook: cap the workgroups of a sweep so that each streams many questions; 0 = no cap).
 * "batch_min" (PqaEngine_NextQuestionArgmaxBatch: batches of at least this many quizzes take the row-sharing sweep, which
 * reads the cube once per batch; default 0 = decided by how many waves the batch gives that sweep; Float engines always take it), "batch_tile" (targets per LDS tile of that sweep, 0 = default),
 * "batch_groups" (that sweep for batches of up to 128 quizzes: question groups side by side in a workgroup, so that the lanes a small
 * batch leaves over take further questions; 0 = as many as leave every CU a workgroup [default], else at most this many),
 * "batch_tail" (that sweep's last, partial round of question blocks as a second launch of a shape with fewer questions per group, where
 * exactly one full round precedes it; default 1),
 * "pole_fix" (every fp64 sweep watches, row by row, for a posterior element that holds nearly all -- or a quarter, while the answer
 * hardly moves the posterior -- of the row; such questions are listed and a kernel launched behind the sweep re-evaluates them in the
 * reference's own summation order -- SRAccumVectDbl256.h:40-46, :62-92 -- so that late quiz states stay within 1e-9 of the
 * reference's priorities on every kernel form; a resident sweep that finds such a row hands the quiz to the launched path; default 1,
 * also PQA_POLE_FIX), "top_exact" (ListTopTargets where probabilities tie among the listed targets or at the list's end: 1 [default] = the
 * reference's order among equals -- CEListTopTargetsAlgorithm::RunHeapifyBased's per-worker heaps and head heap reproduced on the device for
 * the emulated worker count --, 0 = by ascending target; read-only "top_exact_listings" counts the listings that took the heaps), "pole_gate" (where only the selected question leaves the engine -- NextQuestion with the argmax selector, one quiz -- that kernel
 * redoes only the listed questions whose priority can still be the maximum: per listed question the sweep hands over how close to 1 its
 * largest posterior element can be, a kernel ahead of the fix bounds how far the fix can move the priority, and questions whose upper bound
 * stays below the best lower bound keep the sweep's value; PqaEngine_EvalPriorities and the sampled selector always get every listed
 * question redone; default 1), "pole_lazy" (a synchronous single-quiz selection launches that kernel only when its sweep has listed something
 * -- one launch per selection of a fresh quiz instead of two; default 1; 0 = behind every sweep), "late_eager" (after this many
 * selections of a quiz in a row that needed the fix, RecordAnswer's speculative sweep has it launched right behind it again: it runs
 * while the client is elsewhere; default 3 -- long quizzes in late states +4 %, the learner loop unchanged), "pole_follow" (measurement hook: 0 = the watching sweep WITHOUT the launch behind it, for timing the sweep
 * kernel by itself in a quiz state that lists nothing; default 1),
 * "cluster_from" (rows of more than this many elements -- 1024..16384, default 10240: what the register shapes hold without spilling -- take the
 * cluster sweep, a question over a cluster of workgroups), "cluster_form" (that sweep: 0 = default, 1 = question by question, 2 = pass 1 a question
 * ahead of the exchange), "cluster_shape" (threads x 16-byte units per thread of the form that runs ahead: 0 = default, 1 = 512 x 1, 2 = 256 x 2;
 * two workgroups per CU both; 256 x 2 is built for questions of two to five answers, other answer counts take 512 x 1),
 * Read-only: "server_last_step_ns" (device-side duration of the newest finished step of the resident sweep: request in hand
 * to answer published, from the kernel's own 100 MHz clock; -1 if there is none), "precision" (TPqaPrecisionType of the engine: 1 = Float, 3 = Double), "server_active",
 * "ldT", "device". */
PQACORE_API void *PqaHip_SetOption(void *pvEngine, const char *name, int64_t value);
PQACORE_API int64_t PqaHip_GetOption(void *pvEngine, const char *name);
PQACORE_API const char *PqaHip_EvalKernelName(void *pvEngine);

/* ---- bulk KB transfer: dense host arrays without padding, A[q][k][t], D[q][t], B[t] (local questions only) */
PQACORE_API void *PqaHip_SetKB(void *pvEngine, const double *pA, const double *pD, const double *pB);
PQACORE_API void *PqaHip_GetKB(void *pvEngine, double *pA, double *pD, double *pB);
/* Fill the device cube with the deterministic synthetic KB of probqa_amd/synth.py (no host transfer). */
PQACORE_API void *PqaHip_FillSynthetic(void *pvEngine, double nTrain, double noiseAmp, uint64_t seed);
/* Mark targets / (global) questions as gaps without going through maintenance mode (tests of gap handling). */
PQACORE_API void *PqaHip_SetTargetGaps(void *pvEngine, int64_t n, const int64_t *pTargets);
PQACORE_API void *PqaHip_SetQuestionGaps(void *pvEngine, int64_t n, const int64_t *pQuestions);

/* ---- deterministic outputs of the hot path */
/* priority[i] of local question i (0 for gap / asked), i < n == local question count. */
PQACORE_API void *PqaEngine_EvalPriorities(void *pvEngine, const int64_t iQuiz, double *pOut, const int64_t n);
/* NextQuestion with the argmax selector / with the reference's selector driven by the given 64-bit random number. */
PQACORE_API int64_t PqaEngine_NextQuestionArgmax(void *pvEngine, void **ppError, const int64_t iQuiz);
PQACORE_API int64_t PqaEngine_NextQuestionSampled(void *pvEngine, void **ppError, const int64_t iQuiz,
                                                  const uint64_t rnd);
/* NextQuestion (argmax selector) for nQuizzes <= 256 distinct quizzes with ONE launch: pQuestions[i] = the selected
 * question of pQuizzes[i], -1 if that quiz has no question left.  What a server with many quizzes in flight calls instead
 * of nQuizzes x PqaEngine_NextQuestion (reference PqaCore/CpuEngine.cpp:337-415 serves them one sweep at a time). */
PQACORE_API void *PqaEngine_NextQuestionArgmaxBatch(void *pvEngine, const int64_t nQuizzes, const int64_t *pQuizzes,
                                                    int64_t *pQuestions);
/* RecordAnswer for nQuizzes quizzes (each with an active question) in one call and ONE launch -- every posterior bit-identical to
 * PqaEngine_RecordAnswer's (reference PqaCore/CERecordAnswerSubtaskMul.cpp:15-42 per quiz) -- and StartQuiz for nQuizzes new
 * quizzes likewise (pQuizzes receives their ids; all or none).  What a server with many quizzes in flight calls beside
 * PqaEngine_NextQuestionArgmaxBatch; concurrent PqaEngine_RecordAnswer calls of different client threads are gathered into the
 * same batched launch by the engine itself (option "combine"). */
PQACORE_API void *PqaEngine_RecordAnswerBatch(void *pvEngine, const int64_t nQuizzes, const int64_t *pQuizzes, const int64_t *pAnswers);
PQACORE_API void *PqaEngine_StartQuizBatch(void *pvEngine, const int64_t nQuizzes, int64_t *pQuizzes);
/* ListTopTargets for nQuizzes quizzes (any number; 256 per launch sequence) without copying a posterior to the host: pDest[i * maxCount + j],
 * j < pCounts[i], is the listing PqaEngine_ListTopTargets(pQuizzes[i], maxCount) returns -- descending probability, gaps and
 * probabilities <= 0 dropped (reference PqaCore/CEHeapifyPriorsSubtaskMake.cpp:42-52), equal probabilities in the order the reference's
 * per-worker heaps leave them (option "workers" = its thread count; option "top_exact" 0: by ascending target instead).
 * Rows of any length: 4096-target chunks list their own best maxCount on the device and merge there; what crosses to the host is
 * nQuizzes x maxCount records (the reference's GPU engine copies all nTargets posteriors per quiz: PqaCore/CudaEngine.cpp:251-289). */
PQACORE_API void *PqaEngine_ListTopTargetsBatch(void *pvEngine, const int64_t nQuizzes, const int64_t *pQuizzes, const int64_t maxCount,
                                                CiRatedTarget *pDest, int64_t *pCounts);
/* The priority vectors of nQuizzes <= 256 distinct quizzes from ONE sweep that reads the cube once for the whole batch
 * (batch_kernels.hip): pOut[i * nLocalQuestions + q] = priority of local question q for pQuizzes[i], 0 for gap / asked
 * questions.  The deterministic output behind PqaEngine_NextQuestionArgmaxBatch's row-sharing form. */
/* This engine's (shard's) winners of a batch, without NextQuestion's bookkeeping: pOut[i] = {priority, GLOBAL question index or
 * -1} of pQuizzes[i].  A host that shards the question axis gathers these from the shards, picks per quiz (maximum priority,
 * lowest index on ties) and calls PqaEngine_SetActiveQuestion on every shard. */
PQACORE_API void *PqaHip_SelectArgmaxBatch(void *pvEngine, const int64_t nQuizzes, const int64_t *pQuizzes, CiHipSelection *pOut);
PQACORE_API void *PqaEngine_EvalPrioritiesBatch(void *pvEngine, const int64_t nQuizzes, const int64_t *pQuizzes, double *pOut);
/* pOut[i] = the device's Log2Hot(pIn[i]) (host buffers): the function the sweep applies to every posterior element
 * (replaces SRVectMath::Log2Hot, reference SRPlatform/Interface/SRVectMath.h:87-135), exposed so that it can be held to
 * the reference's own SRVectMathTest.Log2Hot criteria (SRPlatformTests/SRVectMathTest.cpp:45-103). */
PQACORE_API void *PqaHip_Log2Hot(void *pvEngine, const double *pIn, double *pOut, const int64_t n);
/* Current target probabilities of a quiz (n == nTargets). */
PQACORE_API void *PqaHip_GetPriors(void *pvEngine, const int64_t iQuiz, double *pOut, const int64_t n);

/* ---- stream-ordered entry points (no host synchronisation inside) */
PQACORE_API void *PqaHip_GetStream(void *pvEngine);                 /* hipStream_t */
PQACORE_API void *PqaHip_SetStream(void *pvEngine, void *hipStream); /* run on the caller's stream, NULL = own */
PQACORE_API void *PqaHip_Synchronize(void *pvEngine);
/* Everything the engine has put on the device has finished, and its resident sweep kernel -- if one is serving the selections
   (option "server") -- STAYS, idle: the bracket of a timed region of synchronous calls (PqaHip_Synchronize sends the kernel away,
   for a caller about to synchronise the whole device). */
PQACORE_API void *PqaHip_Quiesce(void *pvEngine);
/* Enqueue sweep + local argmax; the 16-byte CiHipSelection is written to pOut (device or pinned host pointer). */
PQACORE_API void *PqaHip_EnqueueSelectArgmax(void *pvEngine, const int64_t iQuiz, void *pOut);
/* ---- exchange of the shards' winners through host memory shared by the ranks of a node (probqa_amd/dist.py).
 * The 16-byte message per rank is latency-bound: the sweep's finisher writes {priority, GLOBAL index} to pOut and then
 * flagValue to pFlag -- device-visible addresses of registered host memory -- and every rank's host picks the winner as
 * soon as all flags carry the step's value.  No collective launch, no copy, no stream synchronisation. */
PQACORE_API void *PqaHip_EnqueueSelectArgmaxFlag(void *pvEngine, const int64_t iQuiz, void *pOut, void *pFlag,
                                                 const uint64_t flagValue);
/* ... or through ONE RCCL collective on the engine's stream, for a process-per-GPU host that owns a communicator (pNcclComm: its
 * ncclComm_t; world: its size): the shards' 16-byte winners are all-gathered and every rank returns the same exact pick (max
 * priority, lowest GLOBAL index on ties, -1 if none).  librccl.so is loaded at the first call; libPqaCore.so does not link it.
 * (probqa_amd/dist.py does the same through torch.distributed, where the communicator is PyTorch's.) */
PQACORE_API void *PqaHip_SelectArgmaxRccl(void *pvEngine, const int64_t iQuiz, void *pNcclComm, const int64_t world, double *pPriority,
                                          int64_t *pIndex);
PQACORE_API void *PqaHip_HostRegister(void *pHost, const int64_t nBytes, void **ppDevice);
PQACORE_API void *PqaHip_HostUnregister(void *pHost);
/* Slots of strideBytes each, starting with {double priority; int64 index; uint64 flag}: wait until all `world` flags
 * equal flagValue, then the exact global pick (max priority, lowest index on ties, NaN never wins, -1 if none). */
PQACORE_API void *PqaHip_PickWhenAll(const void *pSlots, const int64_t world, const int64_t strideBytes,
                                     const uint64_t flagValue, const double timeoutSec, double *pPriority, int64_t *pIndex);
/* The two calls above as one step: this shard's selection goes to slot `rank` (pSlotsDev = the device-visible address of
 * pSlots), then the pick over all `world` slots. */
PQACORE_API void *PqaHip_SelectThroughSlots(void *pvEngine, const int64_t iQuiz, const void *pSlots, void *pSlotsDev,
                                            const int64_t rank, const int64_t world, const int64_t strideBytes,
                                            const uint64_t flagValue, const double timeoutSec, double *pPriority,
                                            int64_t *pIndex);
/* Enqueue only the sweep (dominant kernel), for kernel timing. */
PQACORE_API void *PqaHip_EnqueueEval(void *pvEngine, const int64_t iQuiz);
/* Device pointer of the quiz's prior vector (ldT doubles, *pLdT receives ldT) for collectives between shards.  The engine
 * updates it in stream order (PqaEngine_RecordAnswer returns once its kernel is enqueued): read it on the engine's stream
 * (PqaHip_GetStream / PqaHip_SetStream) or after PqaHip_Synchronize. */
PQACORE_API void *PqaHip_GetPriorDevicePtr(void *pvEngine, const int64_t iQuiz, void **ppDev, int64_t *pLdT);
/* RecordAnswer on a shard that does not own the active question: bookkeeping only; the owner's prior is expected to be
 * broadcast into PqaHip_GetPriorDevicePtr's buffer by the caller. */
PQACORE_API void *PqaHip_RecordAnswerRemote(void *pvEngine, const int64_t iQuiz, const int64_t iAnswer);

/* Host bookkeeping of the engine that needs no device, driven by a small script so that it is testable where there is no GPU.
   what = "id_ledger": pIn is a sequence of operations on one fresh compact<->permanent id map (reference behaviour:
   PqaCore/PermanentIdManager.cpp), each {op, a, b}: 0 permanent id of slot a; 1 slot of permanent id a; 2 raise the issue floor
   to a; 3 vacate slot a; 4 reissue slot a; 5 extend to a slots; 6 rename permanent a to b; 7 repack to a slots, the b = a source
   slots following inline; 8 write to a temporary file, read back into a second map, continue on that one; 9 the number of slots that hold an id.  One result per
   operation into pOut (ids, or 0 / 1 for the boolean ones).
   what = "let_go": pIn = {now, maxCount, maxAgeSec, n, then n x {quiz id, last usage}}; pOut receives the count and then the ids
   ClearOldQuizzes would release, in release order (reference behaviour: PqaCore/BaseEngine.cpp:814-873).
   Returns the number of results written, or -1 for a malformed script / too small an output. */
PQACORE_API int64_t PqaHip_HostLogicProbe(const char *what, const int64_t *pIn, const int64_t nIn, int64_t *pOut, const int64_t nOut);

#ifdef __cplusplus
}
#endif
#endif
