// pole_kernels.hip -- questions with a row at the pole of the lack term, re-evaluated the reference's way BEHIND the sweep (gfx950).
//
// lack = -sum invD^2 / log2(p) (reference PqaCore/CEEvalQsSubtaskConsider.cpp:117) has a pole at p -> 1.  A quiz a handful of
// consistent answers deep -- where real quizzes end -- has its posterior on one target, p = 1 - 1e-7 ... 1 - 1e-16: there the last
// place of p = l * (1 / W_k), i.e. the ORDER in which W_k was summed, moves log2 p by 1.6e-16 absolute and the priority by
// 1e-16 / (1 - p) relative -- the ninth digit at 1 - 1e-7, the first at 1 - 1e-16 -- and the velocity term (p - prior)^2 (:119-127)
// of that element, a difference of two numbers next to 1, follows with a third of that.  The sweeps sum W_k as a GPU sums (lane
// partials and butterflies); the reference sums four serial Kahan lanes down the row (SRPlatform/Interface/SRAccumVectDbl256.h:
// 40-46) and folds them with PreciseSum (:62-92): T / 4 DEPENDENT steps per row, which no sweep can afford for every row
//  -- but only the rows AT the pole need it.
//
// So every sweep only WATCHES (a compare per row; eval_kernels.hip, cluster_kernels.hip, batch_kernels.hip): a question with a row
// whose largest posterior element is within 2^-10 of 1 (pole_device.h: kNearOneHi -- why there) leaves its sums in memory and an entry in the suspect list, and the sweep's finisher, seeing a non-empty list, leaves
// the publication of the result to this kernel, which is launched behind every watching sweep (an empty list: a few hundred
// threads read one word and leave).  Here a workgroup takes one suspect at a time:
//   * waves 1 - 3 form the likelihoods (A * invD) * prior of the listed rows again, chunk by chunk, into LDS -- bit for bit the
//     sweep's and the reference's (:72-82) -- and keep each row's largest element;
//   * wave 0 runs the rows' reference-order sums side by side, four lanes per row (the chunks double-buffered against the
//     stagers), PreciseSum folds the four;
//   * for the element within 2^-10 of 1: Log2Hot by the reference's own operation sequence (SRVectMath.h:87-135) on
//     p = l * (1 / W_k) with the reference-order W_k, and its entropy, lack and velocity terms replace what pass 2 had added;
//   * the epilogue (:134-207) again, the priority stored (and handed to the host where the sweep hands priorities over).
// The last workgroup to finish then does what the sweep's finisher left undone: the argmax / the reference's selector / the
// flag of the hand-over, and empties the list.
// In a late quiz EVERY question has such rows (the target's likelihood is all of W_k whatever the answer): the fix then re-reads
// the cube once and runs T / 4 dependent steps per question, a few questions per CU at a time -- about one more sweep's time,
// whatever the row length; the in-kernel fix of round 4 (rows of up to 4096 targets only, lists of 62 / 128 suspects) cost 4.5 - 6.5.
#include <cmath>

#include "eval_device.h"
#include "pole_device.h"
#include "pqa_device.h"
#include "pqa_kernels.h"

namespace pqa {

static __device__ double gLog2TableP[kLog2TableDoubles];   // this translation unit's copy of the Log2Hot table
static __device__ double gLog2Entry0RefP;                   // its entry 0 as the reference has it (eval_kernels.hip: gLog2Entry0Ref)

hipError_t UploadLog2TablePole(const double *hostTable) {
  const double entry0 = std::log2(1.0 + 0x1p-11) * 9.9999999999999927e-01;   // SRVectMath.cpp:31,42
  const hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(gLog2Entry0RefP), &entry0, sizeof(double));
  if (e != hipSuccess) return e;
  return hipMemcpyToSymbol(HIP_SYMBOL(gLog2TableP), hostTable, kLog2TableDoubles * sizeof(double));
}

namespace {

constexpr int kFixThreads = 256;
constexpr int kFixStagers = kFixThreads - kWave;   // waves 1 - 3
constexpr int kFixChunk = 2 * kFixStagers;         // targets of a row per LDS chunk: one 16-byte pair per stager
constexpr int kFixRows = 8;                        // rows of a question side by side (four chain lanes each, in wave 0)
constexpr int kFixRed = 24;                        // doubles of reduction scratch per row

// log2hot (pqa_device.h) with the table in global memory: the same operations on the same table entries, so the same bits -- what
// pass 2 of the sweep took for an element.  (This kernel's LDS is the rows' chunks; one lane per row needs the function.)
__device__ __forceinline__ double log2hot_global(double x, const double *__restrict__ tbl) {
  const uint32_t lo = (uint32_t)d2u(x);
  const uint32_t hi = (uint32_t)(d2u(x) >> 32);
  const double de = u2d(0x4330000000000000ULL | (uint64_t)(hi >> 20)) - 4503599627371519.0;
  const uint32_t idx = (hi >> 10) & 1023u;
  const double y = tbl[2 * idx], c1 = tbl[2 * idx + 1];
  const uint32_t mhi = (hi & 0x800ffc00u) | 0x3ff00200u;
  const uint32_t zhi = (hi & 0x800fffffu) | 0x3ff00000u;
  const double m = u2d((uint64_t)mhi << 32);
  const double z = u2d(((uint64_t)zhi << 32) | lo);
  const double w = (z - m) * c1;
  double c = fma(w, -0x1.55046a143789p-4, 0x1.47fd3ffac83b4p-3);
  c = fma(w, c, -0x1.62e42fefa39efp-2);
  c = fma(w, c, 1.0);
  const double log2z = fma(w, c, y);
  return log2z + de;
}

struct Best {
  double p;
  int64_t i;
};
__device__ __forceinline__ void best_merge(Best &b, double op, int64_t oi) {
  if (oi >= 0 && (b.i < 0 || op > b.p || (op == b.p && oi < b.i))) { b.p = op; b.i = oi; }
}

__global__ __launch_bounds__(kFixThreads) void pole_fixup_kernel(PoleFix a) {
  extern __shared__ double smem[];
  const uint32_t n = a.list->count;   // (written by the sweep: a kernel boundary ago)
  if (n == 0) return;
  const int tid = threadIdx.x, lane = tid % kWave, wave = tid / kWave;
  const int64_t K = a.K, ldT = a.ldT, nT = 4 * ((a.T + 3) >> 2);
  const int R = a.rows;                                        // rows side by side: min(K, kFixRows)
  double *buf = smem;                                          // [2][R][kFixChunk]
  double *red = buf + 2 * (size_t)R * kFixChunk;               // [R][kFixRed]
  double *misc = red + (size_t)R * kFixRed;                    // dH, dL of the question; the last-workgroup flag
  const PoleEntry *entries = reinterpret_cast<const PoleEntry *>(a.list + 1);
  const int nChunks = (int)((nT + kFixChunk - 1) / kFixChunk);
  for (uint32_t e = blockIdx.x; e < n; e += gridDim.x) {
    const PoleEntry en = entries[e];
    const int64_t qLocal = en.q;                               // position in the priority vector; the cube's question is qFirst + it
    const double *prior = a.slots != nullptr ? a.slots[en.b].prior : a.prior;
    uint32_t rowMask = en.rowMask;
    if (a.maskDense != nullptr) rowMask = a.maskDense[qLocal];
    if (K > 31 || rowMask == 0) rowMask = K >= 32 ? 0xFFFFFFFFu : (1u << K) - 1u;   // (not known, or dozens of answers: every row)
    double *rec = a.sums + (size_t)(a.bySlot ? (int64_t)e : qLocal) * a.sumsStride;
    double *recW = rec + a.wOff, *recV = rec + a.vOff;
    const double *qBase = a.cube + (a.qFirst + qLocal) * (K + 1) * ldT;
    const double2 *rowD2 = reinterpret_cast<const double2 *>(qBase + K * ldT);
    const double2 *prior2 = reinterpret_cast<const double2 *>(prior);
    if (tid == 0) { misc[0] = 0.0; misc[1] = 0.0; }
    for (int64_t kBase = 0; kBase < K; kBase += 32) {          // (more than 32 answers: every row, 32 at a time)
      uint32_t rest = K > 31 ? (K - kBase >= 32 ? 0xFFFFFFFFu : (1u << (K - kBase)) - 1u) : rowMask;
      while (rest != 0) {
        // ---- a batch of up to R listed rows
        int rowOf[kFixRows], nb = 0;
#pragma unroll
        for (int r = 0; r < kFixRows; r++) {
          rowOf[r] = 0;
          if (r < R && rest != 0) { rowOf[r] = (int)kBase + __builtin_ctz(rest); rest &= rest - 1; nb = r + 1; }
        }
        double mx[kFixRows], mxId[kFixRows];
        int mxT[kFixRows];
#pragma unroll
        for (int r = 0; r < kFixRows; r++) { mx[r] = 0.0; mxId[r] = 0.0; mxT[r] = 0; }
        double sum = 0.0, corr = 0.0;                          // (wave 0, lane 4 r + c: lane c of row r's accumulator, SRAccumVectDbl256.h:40-46)
        __syncthreads();                                       // (the previous batch's red[] has been read)
        for (int c = 0; c <= nChunks; c++) {
          if (wave != 0) {
            if (c < nChunks) {
              // ---- chunk c of the rows' likelihoods (:72-82, as pass 1 forms them), a pair of targets per thread
              const int s = tid - kWave;
              const int64_t t0 = (int64_t)c * kFixChunk + 2 * s;
              if (t0 < nT) {
                const double2 dv = rowD2[t0 >> 1], pv = prior2[t0 >> 1];
                const uint32_t gw = a.tgap[t0 >> 5] >> (t0 & 31);
                double2 av[kFixRows];
#pragma unroll
                for (int r = 0; r < kFixRows; r++)
                  if (r < nb) av[r] = reinterpret_cast<const double2 *>(qBase + (int64_t)rowOf[r] * ldT)[t0 >> 1];
                const bool g0 = gw & 1u, g1 = gw & 2u;
                const double id0 = g0 ? 0.0 : div_nr(1.0, dv.x), id1 = g1 ? 0.0 : div_nr(1.0, dv.y);   // :74
                const double p0 = g0 ? 0.0 : pv.x, p1 = g1 ? 0.0 : pv.y;                                  // :103
                double *dst = buf + ((size_t)(c & 1) * R) * kFixChunk + 2 * s;
#pragma unroll
                for (int r = 0; r < kFixRows; r++)
                  if (r < nb) {
                    const double l0 = (av[r].x * id0) * p0, l1 = (av[r].y * id1) * p1;                    // :81-82
                    dst[(size_t)r * kFixChunk] = l0;
                    dst[(size_t)r * kFixChunk + 1] = l1;
                    if (l0 > mx[r]) { mx[r] = l0; mxId[r] = id0; mxT[r] = (int)t0; }
                    if (l1 > mx[r]) { mx[r] = l1; mxId[r] = id1; mxT[r] = (int)t0 + 1; }
                  }
              }
            }
          } else if (c > 0 && lane < 4 * nb) {
            // ---- the reference-order sums over chunk c - 1: lane 4 r + cc takes the targets 4 j + cc of row r, in order
            const double *src = buf + ((size_t)((c - 1) & 1) * R + (lane >> 2)) * kFixChunk + (lane & 3);
            const int64_t left = nT / 4 - (int64_t)(c - 1) * (kFixChunk / 4);
            const int steps = (int)(left < kFixChunk / 4 ? left : kFixChunk / 4);
            int j = 0;
            for (; j + 8 <= steps; j += 8) {                   // (eight elements requested at once, added in order)
              double x[8];
#pragma unroll
              for (int u = 0; u < 8; u++) x[u] = src[4 * (j + u)];
#pragma unroll
              for (int u = 0; u < 8; u++) {
                const double y = x[u] - corr;
                const double t = sum + y;
                corr = (t - sum) - y;
                sum = t;
              }
            }
            for (; j < steps; j++) {
              const double y = src[4 * j] - corr;
              const double t = sum + y;
              corr = (t - sum) - y;
              sum = t;
            }
          }
          __syncthreads();
        }
        // ---- per row: the stagers' largest element, the chains' results
        if (wave != 0) {
#pragma unroll
          for (int r = 0; r < kFixRows; r++)
            if (r < nb) {
              const double wmx = wave_max_d(mx[r]);
              double *cw = red + (size_t)r * kFixRed + 3 * (wave - 1);
              if (lane == 0) { cw[0] = 0.0; cw[1] = 0.0; cw[2] = 0.0; }
              if (mx[r] == wmx && mx[r] > 0.0) { cw[0] = mx[r]; cw[1] = mxId[r]; cw[2] = (double)mxT[r]; }   // (behind lane 0's zeros; lanes that tie hold equal values: any one's index serves)
            }
        } else if (lane < 4 * nb) {
          double *out = red + (size_t)(lane >> 2) * kFixRed + 9;
          out[lane & 3] = sum;
          out[4 + (lane & 3)] = corr;
        }
        __syncthreads();
        if (tid < nb) {
          const int k = rowOf[0];                              // (rowOf[tid] without a dynamic register index)
          int kk = k;
#pragma unroll
          for (int r = 1; r < kFixRows; r++) kk = tid == r ? rowOf[r] : kk;
          double *rr = red + (size_t)tid * kFixRed;
          double cand = rr[0], candId = rr[1], candT = rr[2];
          for (int w = 1; w < 3; w++)
            if (rr[3 * w] > cand) { cand = rr[3 * w]; candId = rr[3 * w + 1]; candT = rr[3 * w + 2]; }
          const double Wx = precise_sum4(rr + 9, rr + 13);     // :88
          const double invWx = div_nr(1.0, Wx);                // :91
          const double pRef = cand * invWx;                    // :97
          double dH = 0.0, dL = 0.0;
          if (cand > 0.0 && (uint32_t)(d2u(pRef) >> 32) >= kNearOneHi) {
            const double Wf = recW[kk];                        // the sweep's W_k
            const double pFast = cand * div_nr(1.0, Wf);       // what pass 2 took for this element
            const double lFast = log2hot_global(pFast, gLog2TableP);
            const double lRef = log2hot_ref(pRef, gLog2TableP, gLog2Entry0RefP);   // :106
            dH = cand * lRef - cand * lFast;                   // :113-114 (weighted by W_k: eval_epilogue)
            const double id2 = candId * candId;
            dL = div_fast(id2, lRef) - div_fast(id2, lFast);   // :117 (pass 2's quotient was within 2^-48.8 of the second one)
            // :119-127 the element's velocity term: a difference of two numbers next to 1
            const double prh = prior[(int64_t)candT];
            const double dF = pFast - prh, dR = pRef - prh;
            const double vOld = a.secondIsWV ? [&] { const double sv = div_fast(recV[kk], Wf); return sv * sv; }() : recV[kk];
            double vNew = (vOld - dF * dF) + dR * dR;
            if (!(vNew > 0.0)) vNew = dR * dR;
            recV[kk] = a.secondIsWV ? Wx * sqrt(vNew) : vNew;  // :156-157
            recW[kk] = Wx;                                     // :90
          }
          rr[17] = dH;
          rr[18] = dL;
        }
        __syncthreads();
        if (tid == 0)
          for (int r = 0; r < nb; r++) { misc[0] += red[(size_t)r * kFixRed + 17]; misc[1] += red[(size_t)r * kFixRed + 18]; }
      }
    }
    if (tid == 0) {
      if (a.priority != nullptr || a.priorityT != nullptr) {
        const double pri = eval_epilogue(recW, -(rec[a.hOff] + misc[0]), recV, K, rec[a.lOff] + misc[1], a.vCompTail);   // :130-207
        double *dst = a.slots != nullptr && a.priorityT == nullptr ? a.slots[en.b].priority + qLocal
                      : a.priorityT != nullptr ? a.priorityT + (size_t)qLocal * a.Bp + en.b : a.priority + qLocal;
        __hip_atomic_store(dst, pri, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        TaggedPriority *hp = a.slots != nullptr ? a.slots[en.b].hostPriority : a.hostPriority;
        if (hp != nullptr) {
          typedef unsigned int u4 __attribute__((ext_vector_type(4)));
          const uint64_t w0 = d2u(pri), w1 = a.hostTag;
          const u4 x = {(unsigned)w0, (unsigned)(w0 >> 32), (unsigned)w1, (unsigned)(w1 >> 32)};
          asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(hp + qLocal), "v"(x) : "memory");
        }
      } else {                                                 // (the caller's epilogue kernel follows: cluster_kernels.hip)
        rec[a.hOff] += misc[0];
        rec[a.lOff] += misc[1];
      }
      if (a.maskDense != nullptr) a.maskDense[qLocal] = 0;
      if (a.dirty != nullptr) a.dirty[en.b] = 1u;
    }
    __syncthreads();
  }
  // ---- the last workgroup to get here publishes what the sweep's finisher left to this kernel, and empties the list
  __threadfence_system();                                      // (the corrected priorities, the host's records among them)
  __syncthreads();
  if (tid == 0) {
    const uint32_t arrived = atomicAdd(&a.list->arrived, 1u);
    misc[2] = arrived == gridDim.x - 1 ? 1.0 : 0.0;
  }
  __syncthreads();
  if (misc[2] == 0.0) return;
  __threadfence();
  if (a.fs.scratch != nullptr && a.slots == nullptr) {
    const int64_t nQ = a.nQ;
    const bool sampled = a.fs.sampleSubtasks > 0;
    uint64_t seqValue = a.fs.seqValue, flagValue = a.fs.flagValue;
    if (a.fs.tagCell != nullptr) seqValue = flagValue = __hip_atomic_load(a.fs.tagCell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    double outP = 0.0;
    int64_t outI = 0;
    if (!sampled) {
      // the argmax over every evaluated question (maximum priority, lowest index on ties, NaN never wins: eval_kernels.hip)
      Best b{0.0, -1};
      for (int64_t j = tid; j < nQ; j += kFixThreads) {
        const int64_t q = a.qFirst + j;
        if (bit_test(a.qgap, q) || bit_test(a.asked, q)) continue;
        double p = __hip_atomic_load(a.priority + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (p != p) p = -__builtin_huge_val();
        best_merge(b, p, j);
      }
      for (int m = kWave / 2; m >= 1; m >>= 1) {
        const double op = __shfl_xor(b.p, m, kWave);
        const int64_t oi = __shfl_xor(b.i, m, kWave);
        best_merge(b, op, oi);
      }
      Best *wb = reinterpret_cast<Best *>(buf);
      if (lane == 0) wb[wave] = b;
      __syncthreads();
      if (tid == 0) {
        for (int w = 1; w < kFixThreads / kWave; w++) best_merge(b, wb[w].p, wb[w].i);
        outP = b.i < 0 ? 0.0 : b.p;
        outI = b.i < 0 ? -1 : b.i + a.fs.outBase;
      }
    } else if (a.fs.hostPriority == nullptr) {
      // the reference's selector over the corrected vector (the sweep's own workgroup 0 would have run it)
      const SampledPick r = select_sampled_wg_lds<true>(a.priority, a.qgap, a.asked, a.qFirst, nQ, a.fs.sampleSubtasks, a.fs.sampleRnd, buf);
      outP = r.priority;
      outI = r.index + a.fs.outBase;
    }
    if (tid == 0) {
      a.fs.out->priority = outP;
      a.fs.out->index = outI;
      if (a.fs.seq != nullptr) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");          // system scope: the record (and the handed-over priorities) before the flag
        __hip_atomic_store(a.fs.seq, flagValue, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      if (a.fs.tagCell != nullptr) {
        uint64_t next = seqValue + 1;
        if ((uint32_t)next == 0) next++;
        __hip_atomic_store(a.fs.tagCell, next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  if (tid == 0) {
    a.list->arrived = 0;
    __hip_atomic_store(&a.list->count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

}  // namespace

size_t PoleListBytes(int64_t capacity) { return sizeof(PoleHeader) + (size_t)capacity * sizeof(PoleEntry); }

// Launched behind every sweep that watches (the caller set `fix` up for the sweep's own shape).  selectLdsDoubles: LDS the
// reference's selector needs when this kernel has to run it (0: not asked for).
hipError_t LaunchPoleFixup(const PoleFix &fix, hipStream_t stream) {
  if (fix.list == nullptr || fix.sums == nullptr) return hipErrorInvalidValue;
  PoleFix a = fix;
  a.rows = (int)(a.K < kFixRows ? a.K : kFixRows);
  size_t shmem = ((size_t)2 * a.rows * kFixChunk + (size_t)a.rows * kFixRed + 8) * sizeof(double);
  if (a.fs.scratch != nullptr && a.fs.sampleSubtasks > 0 && a.fs.hostPriority == nullptr) {
    const size_t need = (size_t)select_sampled_lds_doubles(a.nQ, a.fs.sampleSubtasks) * sizeof(double);
    if (need > shmem) shmem = need;
  }
  if (shmem > 160 * 1024) return hipErrorInvalidValue;
  static LaunchCache cache;
  const int dev = LaunchCache::Device();
  int perCU = 0;
  if (!cache.Get(dev, shmem, &perCU)) {
    if (shmem > 64 * 1024) {
      const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(pole_fixup_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
      if (e != hipSuccess) return e;
    }
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, pole_fixup_kernel, kFixThreads, shmem) != hipSuccess || perCU < 1) perCU = 1;
    if (perCU > 4) perCU = 4;
    cache.Put(dev, shmem, perCU);
  }
  int64_t grid = (int64_t)cache.NumCUs(dev) * perCU;
  if (fix.capacity > 0 && grid > fix.capacity) grid = fix.capacity;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(pole_fixup_kernel, dim3((unsigned)grid), dim3(kFixThreads), shmem, stream, a);
  return hipGetLastError();
}

}  // namespace pqa
