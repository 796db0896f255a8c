// hip_engine_server.cpp -- HipEngine: the resident sweep kernel (option server): one launch that serves selection after selection
// through a mailbox in host memory (eval_kernels.hip: eval_server_f64).
#include "hip_engine_internal.h"

namespace pqa {
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
  SpinWait w;
  while (mb->done != _serverPosted && mb->state != kServerExited)
    if (!w.Tick(std::chrono::seconds(30))) return;
}

Error HipEngine::ServerWait(volatile uint64_t *flag, uint64_t value, const char *what) {
  SpinWait w;
  volatile ServerMailbox *mb = _hMailbox;
  while (*flag != value) {
    if (!w.Tick(std::chrono::seconds(30))) return HipErr(hipErrorNotReady, what);
    if (w.Due() && mb->state == kServerExited && mb->taken != _serverPosted && *flag != value) return HipErr(hipErrorUnknown, what);
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
    SpinWait w;
    // (every workgroup reads the line itself when it is in device memory: then not before the step is done)
    while ((_serverRequestInVram ? mb->done : mb->taken) != _serverPosted && mb->state != kServerExited)
      if (!w.Tick(std::chrono::seconds(30))) return HipErr(hipErrorNotReady, "ServerPost (previous request never taken)");
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
    SpinWait w;
    for (;;) {
      if (mb->taken == seq) return Error();
      if (mb->state == kServerExited) {
        std::atomic_thread_fence(std::memory_order_acquire);
        if (mb->taken == seq) return Error();
        break;
      }
      if (!w.Tick(std::chrono::seconds(30))) return HipErr(hipErrorNotReady, "ServerPost (request neither taken nor refused)");
    }
    _serverLaunched = false;   // it left without this request
  }
  HIP_TRY(hipStreamSynchronize(_serverStream));                       // the previous instance is gone entirely
  HIP_TRY(hipMemsetAsync(_dServerCtl, 0, sizeof(ServerCtl), _serverStream));
  mb->state = kServerRunning;
  std::atomic_thread_fence(std::memory_order_seq_cst);
  HIP_TRY(EnsureHostPriority());   // (a launch argument of the resident kernel: requests may ask for the priority vector)
  HIP_TRY(LaunchEvalServer(View(), 0, _Q, _dPriority, (int)_optEvalVariant, _dSelScratch, _hMailbox, (void *)_serverRequest, _serverRequestInVram, _dServerCtl, prev,
                           (uint64_t)_optServerIdleUs * 100, _hHostPriority, _serverStream,
                           _optPoleFix != 0));   // 100 MHz ticks
  _serverLaunched = true;
  _serverKb = _kbVersion;
  _serverVariant = _optEvalVariant;
  return Error();
}

}  // namespace pqa
