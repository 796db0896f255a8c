// server_rt.hip -- round-trip floor of a resident ("server") kernel on gfx950: the host posts a sequence number in pinned
// memory, workgroup 0 sees it and broadcasts it through a device word, every workgroup answers with a tagged 16-byte
// record, workgroup 0 collects the records and writes the step's result + flag back to pinned memory.  No sweep in
// between: what is measured is the protocol that would replace launch + dispatch of one kernel per selection.
// Every spin is bounded; the kernel exits by itself after `idleUs` without a request.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/server_rt tools/server_rt.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
#include <algorithm>

struct Mailbox {            // pinned host memory
  volatile uint64_t req;    // host -> device: sequence number of the newest request
  volatile uint64_t pad0[7];
  volatile uint64_t done;   // device -> host: sequence number of the newest finished request
  volatile uint64_t result;
  volatile uint64_t state;  // 1 = running, 2 = exited
  volatile uint64_t pad1[5];
};

struct Ctl {                // device memory
  uint64_t go;              // broadcast word
  uint64_t pad[7];
};

typedef unsigned int u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint64_t now_ticks() { return wall_clock64(); }  // 100 MHz

__device__ __forceinline__ uint64_t uniform64(uint64_t x) {   // tells the compiler the value is wave-uniform
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)x), hi = __builtin_amdgcn_readfirstlane((uint32_t)(x >> 32));
  return ((uint64_t)hi << 32) | lo;
}

// Control flow is kept WAVE-UNIFORM on purpose: with `if (threadIdx.x == 0) { ...store... }` inside the resident loop the
// structurizer parks lane 0 behind the loop exit of its wave while the other lanes go on to the next iteration's barrier --
// a deadlock.  So whole waves poll (same address: one transaction) and whole waves store (same value, same address).
__global__ __launch_bounds__(256) void server(Mailbox *mb, Ctl *ctl, u4 *rec, uint64_t first, uint64_t idleTicks, int pollHostFromAll) {
  __shared__ uint64_t sGo;
  uint64_t last = first;
  const unsigned slot = blockIdx.x, nSlots = gridDim.x;
  const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x / 64), lane = threadIdx.x % 64;
  if (slot == 0 && wave == 0) __hip_atomic_store(&mb->state, 7ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  for (;;) {
    if (wave == 0) {
      uint64_t go = last;
      const uint64_t t0 = now_ticks();
      if (slot == 0) {
        for (;;) {
          const uint64_t r = uniform64(__hip_atomic_load(&mb->req, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
          if (r != last) { go = r; break; }
          if (now_ticks() - t0 > idleTicks) { go = ~0ull; break; }
        }
        __hip_atomic_store(&ctl->go, go, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        for (;;) {
          go = uniform64(__hip_atomic_load(&ctl->go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
          if (go != last) break;
          if (now_ticks() - t0 > 4 * idleTicks) { go = ~0ull; break; }   // safety net: workgroup 0 died
          __builtin_amdgcn_s_sleep(1);
        }
      }
      sGo = go;
    }
    __syncthreads();
    const uint64_t go = uniform64(sGo);
    __syncthreads();
    if (go == ~0ull) break;
    if (wave == 0) {
      // "sweep": publish a tagged record (all lanes: same value, same address)
      const u4 v = {slot, 0u, (unsigned)go, (unsigned)(go >> 32)};
      asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(rec + slot), "v"(v) : "memory");
      if (slot == 0) {
        // finisher: all records carry this step's tag?
        uint64_t sum = 0;
        const uint64_t t0 = now_ticks();
        bool ok = false;
        for (;;) {
          bool mine = true;
          sum = 0;
          for (unsigned i = lane; i < nSlots; i += 64) {
            u4 v2;
            asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v2) : "v"(rec + i) : "memory");
            if ((((uint64_t)v2.w << 32) | v2.z) != go) mine = false;
            sum += v2.x;
          }
          ok = __all(mine);
          if (ok || now_ticks() - t0 > 4 * idleTicks) break;
        }
        for (int m = 32; m >= 1; m >>= 1) sum += __shfl_xor(sum, m, 64);
        mb->result = ok ? sum : ~0ull;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        __hip_atomic_store(&mb->done, go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    last = go;
  }
  if (slot == 0 && wave == 0) __hip_atomic_store(&mb->state, 2ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void tiny(uint64_t *p) { if (threadIdx.x == 0) p[0] += 1; }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

int main(int argc, char **argv) {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const int grid = argc > 1 ? atoi(argv[1]) : 768;
  const int steps = argc > 2 ? atoi(argv[2]) : 20000;
  Mailbox *mb;
  Mailbox *mbDev;
  const bool vram = argc > 3 && atoi(argv[3]) != 0;   // mailbox in host-visible device memory instead of pinned host memory
  if (vram) {
    CK(hipExtMallocWithFlags((void **)&mb, sizeof(Mailbox), hipDeviceMallocFinegrained));
    mbDev = mb;
    printf("mailbox in fine-grained device memory at %p\n", (void *)mb);
  } else {
    CK(hipHostMalloc(&mb, sizeof(Mailbox), hipHostMallocCoherent));
    CK(hipHostGetDevicePointer((void **)&mbDev, mb, 0));
  }
  Ctl *ctl;
  u4 *rec;
  CK(hipMalloc(&ctl, sizeof(Ctl)));
  CK(hipMalloc(&rec, sizeof(u4) * grid));
  CK(hipMemset(ctl, 0, sizeof(Ctl)));
  CK(hipMemset(rec, 0xff, sizeof(u4) * grid));
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipStream_t s2;
  CK(hipStreamCreate(&s2));
  uint64_t *cnt;
  CK(hipMalloc(&cnt, 8));
  CK(hipMemsetAsync(cnt, 0, 8, s2));
  CK(hipStreamSynchronize(s2));
  mb->req = 0; mb->done = 0; mb->state = 1; for (int i = 0; i < 7; i++) mb->pad0[i] = 0;
  CK(hipDeviceSynchronize());
  const uint64_t idleTicks = 100 * 2000;  // 2 ms
  hipLaunchKernelGGL(server, dim3(grid), dim3(256), 0, s, mbDev, ctl, rec, 0ull, idleTicks, 0);
  CK(hipGetLastError());
  std::vector<double> lat;
  const uint64_t expect = (uint64_t)grid * (grid - 1) / 2;
  int bad = 0;
  for (int i = 1; i <= steps; i++) {
    const auto t0 = std::chrono::steady_clock::now();
    mb->req = (uint64_t)i;
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
    while (mb->done != (uint64_t)i) {
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 2.0) {
        printf("timeout at step %d (state %llu done %llu result %llu)\n", i, (unsigned long long)mb->state,
               (unsigned long long)mb->done, (unsigned long long)mb->result);
        return 3;
      }
    }
    const auto t1 = std::chrono::steady_clock::now();
    if (mb->result != expect) bad++;
    if (i > 1000) lat.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
  }
  // does a small kernel on another stream run while the resident one idles (it must, for the engine's posterior kernels)?
  {
    for (int rep = 0; rep < 5; rep++) {
      mb->req = (uint64_t)(steps + 1 + rep);   // keeps the resident kernel alive
      __atomic_thread_fence(__ATOMIC_SEQ_CST);
      const auto t0 = std::chrono::steady_clock::now();
      hipLaunchKernelGGL(tiny, dim3(1), dim3(256), 0, s2, cnt);
      CK(hipStreamSynchronize(s2));
      const auto t1 = std::chrono::steady_clock::now();
      printf("  small kernel beside the resident one: %.1f us (resident state %llu)\n",
             std::chrono::duration<double, std::micro>(t1 - t0).count(), (unsigned long long)mb->state);
    }
  }
  CK(hipStreamSynchronize(s));   // the kernel exits by itself after the idle time
  std::sort(lat.begin(), lat.end());
  printf("grid=%d steps=%d bad=%d state=%llu round trip us: p10 %.2f p50 %.2f p90 %.2f mean %.2f\n", grid, steps, bad,
         (unsigned long long)mb->state, lat[lat.size() / 10], lat[lat.size() / 2], lat[lat.size() * 9 / 10],
         [&] { double a = 0; for (double x : lat) a += x; return a / lat.size(); }());
  // for comparison: empty-kernel launch + synchronise
  return bad ? 1 : 0;
}
